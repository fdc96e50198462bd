#!/usr/bin/env python3
"""Timing of the bf16 attention kernels (forward / backward) at the stage shapes of the BASELINE configs, bs as in the
bench.  CALM_VIT_LIB=<other build> A/Bs kernel variants (run both in one gpurun call: same box)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()


def t_med(fn, n=8, warm=2):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


SHAPES = [(256, 224, 12, 56), (256, 176, 12, 44), (256, 128, 12, 32), (256, 80, 12, 20),
          (256, 224, 6, 112), (256, 176, 6, 88), (32, 384, 12, 96), (32, 288, 12, 72), (128, 200, 6, 100)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
print(f"lib: {os.environ.get('CALM_VIT_LIB', 'default')}")
for B, S, H, hd in SHAPES:
    D = H * hd
    bf = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).bfloat16()
    q, k, v, dout = bf(B, S, D), bf(B, S, D), bf(B, S, D, sc=1.0), bf(B, S, D, sc=1.0)
    w1, w2 = bf(2 * S, S, sc=S ** -0.5), bf(S, 2 * S, sc=(2 * S) ** -0.5)
    b1, b2 = torch.randn(2 * S, device="cuda") * 0.1, torch.randn(S, device="cuda") * 0.1
    s1, s2 = torch.tensor([1.3], device="cuda"), torch.tensor([0.8], device="cuda")
    e = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device="cuda")
    out, R, hp, hg, Mk, MkT = e(B, S, D), e(B, S, S), e(B * S, 2 * S), e(B * S, 2 * S), e(B, S, S), e(B, S, S)
    lse, delta = torch.empty(B, H, S, device="cuda"), torch.empty(B, H, S, device="cuda")
    dq, dk, dv, dM = e(B, S, D), e(B, S, D), e(B, S, D), e(B * S, S)
    tf = t_med(lambda: be.attn16_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd))
    tb = t_med(lambda: be.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta, dq, dk, dv, dM, B, S, H, hd))
    ff = B * (6.0 * S * S * D + 8.0 * S ** 3) / 1e12
    fb = B * (14.0 * S * S * D) / 1e12            # 7 products of 2 S^2 D (two of them recomputed in each pass)
    print(f"B{B} S{S} H{H} hd{hd}: fwd {1e3 * tf:7.1f} us ({ff / tf * 1e3:6.1f} TF)   bwd {1e3 * tb:7.1f} us ({fb / tb * 1e3:6.1f} TF)")
