#!/usr/bin/env python3
"""dgrad || wgrad of one Linear inside a hipGraph: serial capture vs a fork onto a second captured stream (graph edges
cost no host events).  Several Linear backward pairs per graph, replayed 20 times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
calm.backend.set_matmul_precision(prec)
cast = (lambda t: t.bfloat16()) if prec == "bf16" else (lambda t: t)
g = lambda *s: cast(torch.randn(*s, device="cuda"))
f = lambda *s: torch.randn(*s, device="cuda")
side = torch.cuda.Stream()
print(prec)
for M, N, K in ((57344, 672, 672), (45056, 528, 528), (32768, 384, 384), (20480, 240, 240), (20480, 480, 240), (20480, 160, 80)):
    sets = []
    for _ in range(4):
        x, w, y, dx, G = g(M, K), g(N, K), g(M, N), g(M, K), f(N, K)
        sets.append((x, w, y, dx, G))
    # workspace tensors allocated inside be.gemm would be allocated during capture from the graph pool: fine
    def dgrad(s):
        x, w, y, dx, G = s
        be.gemm(y, w, dx, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1)
    def wgrad(s):
        x, w, y, dx, G = s
        be.gemm(y, x, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0), accumulate=True)
    def serial():
        for s in sets:
            dgrad(s); wgrad(s)
    def forked():
        cur = torch.cuda.current_stream()
        for s in sets:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                wgrad(s)
            dgrad(s)
            cur.wait_stream(side)
    res = {}
    for name, fn in (("serial", serial), ("forked", forked)):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        gr.replay(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); gr.replay(); b.record()
        torch.cuda.synchronize()
        res[name] = sorted(a.elapsed_time(b) for a, b in ev)[10] / len(sets)
    print(f"M={M:6d} N={N:5d} K={K:5d}: per pair serial {res['serial']*1e3:7.1f} us  forked {res['forked']*1e3:7.1f} us  ({res['serial']/res['forked']:.2f}x)")
