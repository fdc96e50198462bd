#!/usr/bin/env python3
"""Same-process timing: fused attention backward core (calm_attention_bwd) vs the composition it replaces."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()


def t_med(fn, n=10, warm=2):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


for B, S, H, hd in ((256, 224, 6, 112), (256, 176, 6, 88), (256, 128, 6, 64), (256, 80, 6, 40), (256, 224, 12, 56)):
    D = H * hd
    g = lambda *s: torch.randn(*s, device="cuda") * 0.3
    q, k, v, dout = g(B, S, D), g(B, S, D), g(B, S, D), g(B, S, D)
    P = torch.softmax(g(B, H, S, S), dim=-1)
    dS, dq, dk, dv, dM = torch.empty_like(P), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B, S, S, device="cuda")
    scale = 1 / math.sqrt(hd)
    pb = (H * S * S, S * S)

    def composite():
        be.gemm(dout, v, dS, S, S, hd, (D, 1, S * D, hd), (D, 1, S * D, hd), (S,) + pb, batch=(B, H))
        be.gemm(P, dout, dv, S, hd, S, (1, S) + pb, (1, D, S * D, hd), (D, S * D, hd), batch=(B, H))
        be.softmax_bwd(P, dS, B * H * S, S)
        be.sum_heads(dS, dM, B, H, S * S)
        be.gemm(dS, k, dq, S, hd, S, (S, 1) + pb, (1, D, S * D, hd), (D, S * D, hd), batch=(B, H), alpha=scale)
        be.gemm(dS, q, dk, S, hd, S, (1, S) + pb, (1, D, S * D, hd), (D, S * D, hd), batch=(B, H), alpha=scale)

    tc = t_med(composite)
    tf = t_med(lambda: be.attn_bwd(q, k, v, dout, P, dS, dq, dk, dv, dM, B, S, S, H, hd))
    fl = 4 * 2.0 * S * S * hd * H * B / 1e9
    print(f"B={B} S={S} H={H} hd={hd}: composite {tc:7.3f} ms ({fl/tc:5.1f} TF)   fused {tf:7.3f} ms ({fl/tf:5.1f} TF)")
