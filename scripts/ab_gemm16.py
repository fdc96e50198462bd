#!/usr/bin/env python3
"""Time calm_gemm on bf16 TENSORS (the bf16 pipeline's shapes: activation x weight, weight gradients, per-image
sequence-axis products).  A/B two builds on the same box via CALM_VIT_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")


def t_med(fn, n=12, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


g = lambda *s: torch.randn(*s, device="cuda").bfloat16()
f = lambda *s: torch.randn(*s, device="cuda")
tot = 0.0
print(os.environ.get("CALM_VIT_LIB", "default lib"))
for M, N, K in ((57344, 672, 672), (57344, 1344, 672), (57344, 672, 1344), (45056, 528, 528), (45056, 1056, 528),
                (32768, 384, 384), (32768, 768, 384), (20480, 240, 240), (20480, 480, 240), (45056, 352, 176)):
    x, w, y = g(M, K), g(N, K), g(M, N)
    fwd = t_med(lambda: be.gemm(x, w, y, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1))
    dgr = t_med(lambda: be.gemm(y, w, x, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1))
    G = f(N, K)
    wgr = t_med(lambda: be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0)))
    fl = 2.0 * M * N * K / 1e9
    tot += fwd + dgr + wgr
    print(f"M={M:6d} N={N:5d} K={K:5d}: fwd {fwd:7.3f} ms {fl/fwd:6.1f} TF | dgrad {dgr:7.3f} ms {fl/dgr:6.1f} TF | wgrad {wgr:7.3f} ms {fl/wgr:6.1f} TF")
for (Ms, Ns, Ks) in ((176, 528, 176), (224, 672, 224), (80, 240, 80), (128, 384, 128), (128, 80, 240)):
    a_, b_, c_ = g(256, Ms, Ks), g(256, Ks, Ns), g(256, Ms, Ns)
    t1 = t_med(lambda: be.gemm(a_, b_, c_, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(256, 1), split_k=1))
    at = g(256, Ks, Ms)
    t2 = t_med(lambda: be.gemm(at, b_, c_, Ms, Ns, Ks, (1, Ms, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(256, 1), split_k=1))
    fl = 2.0 * Ms * Ns * Ks * 256 / 1e9
    print(f"per-image {Ms}x{Ns}x{Ks} x256: (1,0) {t1:7.3f} ms {fl/t1:6.1f} TF | (0,0) {t2:7.3f} ms {fl/t2:6.1f} TF"); tot += t1 + t2
print(f"sum {tot:.3f} ms")
