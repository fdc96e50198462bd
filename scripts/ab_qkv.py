#!/usr/bin/env python3
"""What would grouping the q/k/v projections of a self-attention block into one launch buy?
3 GEMMs (+2 gradient adds in backward) vs one launch over the concatenated problem."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
g = lambda *s: torch.randn(*s, device="cuda")


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, D in ((57344, 672), (45056, 528), (32768, 384), (20480, 240)):
    x, w, w3 = g(M, D), [g(D, D) for _ in range(3)], g(3 * D, D)
    y, y3 = [g(M, D) for _ in range(3)], g(M, 3 * D)
    dx, tmp = g(M, D), [g(M, D) for _ in range(3)]
    G3, G = g(3 * D, D), [g(D, D) for _ in range(3)]
    fwd3 = lambda: [be.gemm(x, w[i], y[i], M, D, D, (D, 1, 0, 0), (D, 1, 0, 0), (D, 0, 0), split_k=1) for i in range(3)]
    fwd1 = lambda: be.gemm(x, w3, y3, M, 3 * D, D, (D, 1, 0, 0), (D, 1, 0, 0), (3 * D, 0, 0), split_k=1)

    def dg3():
        for i in range(3):
            be.gemm(y[i], w[i], tmp[i], M, D, D, (D, 1, 0, 0), (1, D, 0, 0), (D, 0, 0), split_k=1)
        torch.add(tmp[0], tmp[1], out=dx); dx.add_(tmp[2])
    dg1 = lambda: be.gemm(y3, w3, dx, M, D, 3 * D, (3 * D, 1, 0, 0), (1, D, 0, 0), (D, 0, 0), split_k=1)
    wg3 = lambda: [be.gemm(y[i], x, G[i], D, D, M, (1, D, 0, 0), (1, D, 0, 0), (D, 0, 0)) for i in range(3)]
    wg1 = lambda: be.gemm(y3, x, G3, 3 * D, D, M, (1, 3 * D, 0, 0), (1, D, 0, 0), (D, 0, 0))
    r = [timeit(f) for f in (fwd3, fwd1, dg3, dg1, wg3, wg1)]
    print(f"M={M} D={D}: fwd 3x {r[0]:.3f} -> 1x {r[1]:.3f} | dgrad 3x+adds {r[2]:.3f} -> 1x {r[3]:.3f} | wgrad 3x {r[4]:.3f} -> 1x {r[5]:.3f}"
          f" | saved {sum(r[0::2]) - sum(r[1::2]):.3f} ms")
