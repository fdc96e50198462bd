#!/usr/bin/env python3
"""Time calm_rope_fwd / _bwd at the stage sizes of Base-224 (bs=256), fp32 and bf16 tensors.  A/B via CALM_VIT_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
def t_med(fn, n=12, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]
print(os.environ.get("CALM_VIT_LIB", "default lib"))
for dt in (torch.float32, torch.bfloat16):
    for B, S, H, dc, dr in ((256, 224, 12, 0, 56), (256, 224, 12, 28, 28), (256, 176, 12, 0, 44), (256, 176, 12, 22, 22), (256, 128, 12, 0, 32), (256, 80, 12, 0, 20), (256, 80, 12, 10, 10)):
        esz = 4 if dt == torch.float32 else 2
        content = torch.randn(B, S, H * dc, device="cuda").to(dt) if dc else None
        xr = torch.randn(B, S, H * dr, device="cuda").to(dt)
        inv = torch.rand(dr // 2, device="cuda") + 0.01
        table = torch.empty(2 * S * (dr // 2), device="cuda")
        out = torch.empty(B, S, H * (dc + dr), device="cuda", dtype=dt)
        g = torch.randn(B, S, H * (dc + dr), device="cuda").to(dt)
        d_c = torch.empty_like(content) if dc else None
        d_x, d_f = torch.empty_like(xr), torch.zeros(dr // 2, device="cuda")
        tf = t_med(lambda: be.rope_fwd(content, xr, inv, table, out, B, S, H, dc, dr))
        tb = t_med(lambda: be.rope_bwd(g, xr, table, d_c, d_x, d_f, B, S, H, dc, dr))
        n = B * S * H * (dc + dr)
        print(f"{str(dt):15s} B{B} S{S} H{H} dc{dc} dr{dr}: fwd {1e3*tf:6.1f} us ({2*n*esz/tf/1e9:5.2f} TB/s)  bwd {1e3*tb:6.1f} us ({(2*n+B*S*H*dr)*esz/tb/1e9:5.2f} TB/s)")
