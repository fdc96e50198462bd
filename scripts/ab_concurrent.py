#!/usr/bin/env python3
"""Does running the weight-gradient GEMM of a linear on a second HIP stream beside its input-gradient GEMM pay?
(Round 4 experiment: sum of kernel durations == step time, i.e. everything runs serially; the pipelined bf16 family
holds 128 KiB of LDS per workgroup — two such launches cannot share a CU, the 128x128-tile kernels can.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
calm.backend.set_matmul_precision(prec)
side = torch.cuda.Stream()


def t_med(fn, n=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


cast = (lambda t: t.bfloat16()) if prec == "bf16" else (lambda t: t)
g = lambda *s: cast(torch.randn(*s, device="cuda"))
f = lambda *s: torch.randn(*s, device="cuda")
print(prec)
for M, N, K in ((57344, 672, 672), (45056, 528, 528), (45056, 1056, 528), (32768, 384, 384), (32768, 768, 384),
                (20480, 240, 240), (20480, 480, 240), (20480, 160, 80)):
    x, w, y, dx = g(M, K), g(N, K), g(M, N), g(M, K)
    G = f(N, K)
    dgrad = lambda: be.gemm(y, w, dx, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1)
    wgrad = lambda: be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0))

    def serial():
        dgrad(); wgrad()

    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        cur.wait_stream(side)

    td, tw, ts, tf = t_med(dgrad), t_med(wgrad), t_med(serial), t_med(forked)
    print(f"M={M:6d} N={N:5d} K={K:5d}: dgrad {td*1e3:7.1f} us  wgrad {tw*1e3:7.1f} us  serial {ts*1e3:7.1f}  two streams {tf*1e3:7.1f}  ({ts/tf:.2f}x)")
