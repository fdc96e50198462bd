#!/usr/bin/env python3
"""The small per-image / latent-branch products of the Base-224 bf16 step that sit below 120 TFLOP/s in the per-shape
table (9 ms per step over 290 launches): per-launch time with events around each launch (as bench.py measures them) and
back-to-back (100 launches between two events) — how much of the in-step figure is the launch itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")
g = lambda *s: torch.randn(*s, device="cuda").bfloat16()
f = lambda *s: torch.randn(*s, device="cuda")


def both(fn, n=100):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * t[len(t) // 2], 1e3 * e0.elapsed_time(e1) / n


x = f(1024)
print("torch add_ tiny: per-launch events %.1f us, back-to-back %.1f us" % both(lambda: x.add_(1.0)))
for (M, N, K, nb, akc, bkc, red) in ((80, 240, 80, 256, 0, 0, 0), (80, 240, 80, 256, 1, 0, 0), (128, 80, 240, 256, 1, 1, 1), (176, 80, 240, 256, 1, 1, 1),
                                     (80, 176, 528, 256, 1, 1, 1), (176, 240, 80, 256, 1, 0, 0), (128, 384, 128, 256, 0, 0, 0)):
    A = g(nb, M, K) if akc else g(nb, K, M)
    B = g(nb, N, K) if bkc else g(nb, K, N)
    a = (K, 1, M * K, 0) if akc else (1, M, M * K, 0)
    b = (K, 1, N * K, 0) if bkc else (1, N, N * K, 0)
    if red:
        C = f(M, N)
        fn = lambda: be.gemm(A, B, C, M, N, K, a, b, (N, 0, 0), batch=(nb, 1), reduce_batch=True, split_k=1)
    else:
        C = g(nb, M, N)
        fn = lambda: be.gemm(A, B, C, M, N, K, a, b, (N, M * N, 0), batch=(nb, 1), split_k=1)
    t1, t2 = both(fn)
    fl = 2.0 * M * N * K * nb / 1e6
    print(f"{M}x{N}x{K} x{nb} ({akc},{bkc}) reduce={red}: events {t1:6.1f} us ({fl/t1:6.1f} GF/us)   back-to-back {t2:6.1f} us")
for (M, N, K) in ((240, 480, 20480), (240, 240, 20480), (264, 240, 45056), (80, 160, 20480)):
    dy, xx, G = g(K, M), g(K, N), f(M, N)
    fn = lambda: be.gemm(dy, xx, G, M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0), accumulate=True)
    t1, t2 = both(fn)
    print(f"wgrad {M}x{N}x{K}: events {t1:6.1f} us   back-to-back {t2:6.1f} us")
