#!/usr/bin/env python3
"""Time split-K weight-gradient launches of calm_gemm (dW = dY^T X) over the model's output sizes, small to large.
   A/B two builds on the SAME box with CALM_VIT_LIB=..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
calm.backend.set_matmul_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")


def t_med(fn, n=20, warm=5):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


g = lambda *s: torch.randn(*s, device="cuda")
print(os.environ.get("CALM_VIT_LIB", "default lib"))
tot = 0.0
for M, N, K in ((80, 160, 20480), (160, 80, 20480), (240, 240, 20480), (240, 480, 20480), (480, 240, 20480),
                (264, 120, 45056), (192, 120, 32768), (176, 352, 45056), (352, 176, 45056), (256, 128, 32768),
                (384, 384, 32768), (384, 768, 32768), (528, 528, 45056), (528, 1056, 45056), (672, 672, 57344),
                (1344, 672, 57344)):
    dy, x, G = g(K, M), g(K, N), g(M, N)
    t = t_med(lambda: be.gemm(dy, x, G, M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0)))
    tot += t
    print(f"wgrad {M:5d}x{N:5d} K={K:6d}: {t*1e3:8.1f} us {2.0*M*N*K/1e9/t:6.1f} TF")
print(f"sum {tot:.3f} ms")
