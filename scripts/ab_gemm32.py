#!/usr/bin/env python3
"""Time calm_gemm on fp32 tensors over the Small-224 step's shapes (activation x weight, data and weight gradients,
per-image and per-head products).  A/B the two fp32 families on one box: CALM_GEMM_PIPE32=0 / 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()


def t_med(fn, n=8, warm=2):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


f = lambda *s: torch.randn(*s, device="cuda")
tot = 0.0
print("CALM_GEMM_PIPE32 =", os.environ.get("CALM_GEMM_PIPE32", "1"))
for M, N, K in ((57344, 672, 672), (57344, 1344, 672), (57344, 672, 1344), (45056, 528, 528), (45056, 1056, 528),
                (32768, 384, 384), (32768, 768, 384), (20480, 240, 240), (20480, 480, 240), (45056, 352, 176),
                (45056, 120, 264), (57344, 224, 448)):
    x, w, y = f(M, K), f(N, K), f(M, N)
    fwd = t_med(lambda: be.gemm(x, w, y, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1))
    dgr = t_med(lambda: be.gemm(y, w, x, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1))
    G = f(N, K)
    wgr = t_med(lambda: be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0)))
    fl = 2.0 * M * N * K / 1e9
    tot += fwd + dgr + wgr
    print(f"M={M:6d} N={N:5d} K={K:5d}: fwd {fwd:7.3f} ms {fl/fwd:6.1f} TF | dgrad {dgr:7.3f} ms {fl/dgr:6.1f} TF | wgrad {wgr:7.3f} ms {fl/wgr:6.1f} TF")
for (Ms, Ns, Ks, nb) in ((176, 528, 176, 256), (224, 672, 224, 256), (80, 240, 80, 256), (128, 384, 128, 256), (224, 112, 224, 1536),
                         (224, 224, 112, 1536), (176, 88, 176, 1536)):
    a_, b_, c_ = f(nb, Ms, Ks), f(nb, Ks, Ns), f(nb, Ms, Ns)
    t1 = t_med(lambda: be.gemm(a_, b_, c_, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(nb, 1), split_k=1))
    at = f(nb, Ks, Ms)
    t2 = t_med(lambda: be.gemm(at, b_, c_, Ms, Ns, Ks, (1, Ms, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(nb, 1), split_k=1))
    bt = f(nb, Ns, Ks)
    t3 = t_med(lambda: be.gemm(a_, bt, c_, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (Ks, 1, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(nb, 1), split_k=1))
    fl = 2.0 * Ms * Ns * Ks * nb / 1e9
    print(f"per-image {Ms}x{Ns}x{Ks} x{nb}: (1,0) {t1:7.3f} ms {fl/t1:6.1f} TF | (0,0) {t2:7.3f} ms {fl/t2:6.1f} TF | (1,1) {t3:7.3f} ms {fl/t3:6.1f} TF")
    tot += t1 + t2 + t3
print(f"sum {tot:.3f} ms")
