#!/usr/bin/env python3
"""Does running dgrad and wgrad of one Linear on two HIP streams fill each other's tail rounds?
Total time of (dgrad; wgrad) on one stream vs the pair issued on two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
g = lambda *s: torch.randn(*s, device="cuda")
side = torch.cuda.Stream()


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in ((57344, 672, 672), (45056, 528, 528), (45056, 1056, 528), (32768, 384, 384), (20480, 240, 240), (20480, 480, 240)):
    x, w, y, dx, G = g(M, K), g(N, K), g(M, N), g(M, K), g(N, K)
    dgrad = lambda: be.gemm(y, w, dx, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1)
    wgrad = lambda: be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0))

    def serial():
        dgrad(); wgrad()

    def forked():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        torch.cuda.current_stream().wait_stream(side)

    a, b = timeit(serial), timeit(forked)
    print(f"M={M} N={N} K={K}: one stream {a:.3f} ms, two streams {b:.3f} ms ({100*(a-b)/a:+.1f}%)")
