#!/usr/bin/env python3
"""Time calm_layernorm_fwd / _bwd at the stage sizes (bs=256), fp32 and bf16 gradient / output.  A/B via CALM_VIT_LIB."""
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
for rows, D in ((57344, 672), (45056, 528), (32768, 384), (20480, 240)):
    for g16 in (False, True):
        x = torch.randn(rows, D, device="cuda"); w = torch.randn(D, device="cuda")
        y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16 if g16 else torch.float32)
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        dy = torch.randn(rows, D, device="cuda").to(y.dtype); skip = torch.randn(rows, D, device="cuda")
        dx, dw = torch.empty_like(x), torch.zeros(D, device="cuda")
        tf = t_med(lambda: be.layernorm_fwd(x, w, y, mean, rstd, rows, D, 1e-6))
        tb = t_med(lambda: be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, rows, D))
        ts = t_med(lambda: be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, rows, D, dx_add=skip))
        e = y.element_size(); n = rows * D
        print(f"rows {rows} D {D} {'bf16' if g16 else 'fp32'}: fwd {1e3*tf:6.1f} us ({n*(4+e)/tf/1e9:4.2f} TB/s) bwd {1e3*tb:6.1f} us ({n*(8+e)/tb/1e9:4.2f} TB/s) bwd+skip {1e3*ts:6.1f} us ({n*(12+e)/ts/1e9:4.2f} TB/s)")
