#!/usr/bin/env python3
"""Batch-reduced weight gradients of the sequence-axis linears (G[M,N] = sum_b A_b^T B_b, tiny M x N, 256 images): time
against the number of k-slices that combine through atomics on the one output tile."""
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


for (M, N, K, nb) in ((128, 80, 240, 256), (176, 80, 240, 256), (80, 176, 528, 256), (224, 176, 528, 256), (80, 80, 240, 256), (80, 224, 672, 256)):
    A, B, C = g(nb, M, K), g(nb, N, K), f(M, N)
    row = []
    for sk in (0, 8, 16, 32, 64, 128, 256):
        t = t_b2b(lambda: be.gemm(A, B, C, M, N, K, (K, 1, M * K, 0), (K, 1, N * K, 0), (N, 0, 0), batch=(nb, 1), reduce_batch=True, split_k=sk))
        row.append(f"{sk}:{t:5.1f}")
    print(f"{M}x{N}x{K} x{nb} reduce: " + "  ".join(row) + "  us")
