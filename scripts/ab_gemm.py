#!/usr/bin/env python3
"""Time key calm_gemm shapes (median of N, HIP events).  A/B two builds on the SAME box:
   CALM_VIT_LIB=/path/libA.so python3 scripts/ab_gemm.py ; CALM_VIT_LIB=/path/libB.so python3 scripts/ab_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
calm.backend.set_matmul_precision(prec)


def t_med(fn, n=12, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


g = lambda *s: torch.randn(*s, device="cuda")
tot = 0.0
print(os.environ.get("CALM_VIT_LIB", "default lib"), prec)
for M, N, K in ((57344, 672, 672), (57344, 1344, 672), (57344, 672, 1344), (45056, 528, 528), (32768, 384, 384), (20480, 240, 240)):
    x, w, y = g(M, K), g(N, K), g(M, N)
    fwd = t_med(lambda: be.gemm(x, w, y, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1))
    dgr = t_med(lambda: be.gemm(y, w, x, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1))
    G = g(N, K)
    wgr = t_med(lambda: be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0)))
    fl = 2.0 * M * N * K / 1e9
    tot += fwd + dgr + wgr
    print(f"M={M:6d} N={N:5d} K={K:5d}: fwd {fwd:7.3f} ms {fl/fwd:6.1f} TF | dgrad {dgr:7.3f} ms {fl/dgr:6.1f} TF | wgrad {wgr:7.3f} ms {fl/wgr:6.1f} TF")
B, H, S, hd = 256, 6, 224, 112
D = H * hd
q, k, P = g(B, S, D), g(B, S, D), g(B, H, S, S)
t = t_med(lambda: be.gemm(q, k, P, S, S, hd, (D, 1, S * D, hd), (D, 1, S * D, hd), (S, H * S * S, S * S), batch=(B, H)))
print(f"batched logits 224x224x112 x1536: {t:7.3f} ms {2.0*S*S*hd*B*H/1e9/t:6.1f} TF")
tot += t
# epilogue-heavy short-K shapes
M, N, K = 45056, 352, 176
dy, w, aux, out = g(M, K), g(K, N), g(M, N), g(M, N)      # dgrad with GELU' epilogue: out = (dy . w) * gelu'(aux)
t = t_med(lambda: be.gemm(dy, w, out, M, N, K, (K, 1, 0, 0), (1, N, 0, 0), (N, 0, 0), act=2, aux=aux, split_k=1))
print(f"dgrad+GELU' {M}x{N}x{K}: {t:7.3f} ms {2.0*M*N*K/1e9/t:6.1f} TF"); tot += t
B2, S2, D2 = 256, 176, 528
dR, kk, dq = g(B2, S2, S2), g(B2, S2, D2), g(B2, S2, D2)   # dq += dR . K  (accumulate epilogue), per image
t = t_med(lambda: be.gemm(dR, kk, dq, S2, D2, S2, (S2, 1, S2 * S2, 0), (1, D2, S2 * D2, 0), (D2, S2 * D2, 0), batch=(B2, 1), accumulate=True))
print(f"accumulate {S2}x{D2}x{S2} x{B2}: {t:7.3f} ms {2.0*S2*D2*S2*B2/1e9/t:6.1f} TF"); tot += t
mask = g(B, S, S)
t = t_med(lambda: be.gemm(q, k, P, S, S, hd, (D, 1, S * D, hd), (D, 1, S * D, hd), (S, H * S * S, S * S), batch=(B, H), alpha=0.1, residual=mask, r=(S, S * S, 0)))
print(f"logits+mask residual 224x224x112 x1536: {t:7.3f} ms {2.0*S*S*hd*B*H/1e9/t:6.1f} TF"); tot += t
x, w, res, y = g(57344, 672), g(672, 672), g(57344, 672), g(57344, 672)
ls = g(672)
t = t_med(lambda: be.gemm(x, w, y, 57344, 672, 672, (672, 1, 0, 0), (672, 1, 0, 0), (672, 0, 0), col_scale=ls, residual=res, r=(672, 0, 0), split_k=1))
print(f"out_proj (ls+residual) 57344x672x672: {t:7.3f} ms {2.0*57344*672*672/1e9/t:6.1f} TF"); tot += t
# short-M sequence-axis products (per image): M = 40 / 176 rows in 128-row tiles
for (Ms, Ns, Ks) in ((40, 528, 176), (176, 528, 176), (176, 672, 224)):
    a_, b_, c_ = g(256, Ms, Ks), g(256, Ks, Ns), g(256, Ms, Ns)
    t = t_med(lambda: be.gemm(a_, b_, c_, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(256, 1), split_k=1))
    print(f"per-image {Ms}x{Ns}x{Ks} x256: {t:7.3f} ms {2.0*Ms*Ns*Ks*256/1e9/t:6.1f} TF"); tot += t
print(f"sum {tot:.3f} ms")
