#!/usr/bin/env python3
"""Small-output weight gradients (k-split through atomics / the workspace): time against the number of k-slices."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")
g = lambda *s: torch.randn(*s, device="cuda").bfloat16()
f = lambda *s: torch.randn(*s, device="cuda")


def t_b2b(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for (M, N, K) in ((240, 480, 20480), (480, 240, 20480), (240, 240, 20480), (264, 240, 45056), (176, 352, 45056), (352, 176, 45056), (192, 240, 32768),
                  (256, 128, 32768), (80, 160, 20480), (160, 80, 20480), (384, 768, 32768), (528, 528, 45056)):
    dy, xx, G = g(K, M), g(K, N), f(M, N)
    row = []
    for sk in (0, 16, 32, 64, 128, 256):
        try:
            t = t_b2b(lambda: be.gemm(dy, xx, G, M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0), accumulate=True, split_k=sk))
            row.append(f"{sk}:{t:5.1f}")
        except Exception as e:
            row.append(f"{sk}: err")
    print(f"wgrad {M}x{N}x{K}: " + "  ".join(row) + "  us   (%.0f TF at default)" % (2.0 * M * N * K / 1e6 / float(row[0].split(':')[1])))
