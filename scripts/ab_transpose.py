#!/usr/bin/env python3
"""Time calm_grid_transpose (the [B,S,S,3] grid transpose between the row and the column VMLA of a Block) at the four
stage sizes of Base-224 / Small-224, bs=256; A/B two builds via CALM_VIT_LIB.  Bytes moved: 2 x B x S x S x 12."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()
print(os.environ.get("CALM_VIT_LIB", "default lib"))
for S in (224, 176, 128, 80, 36):
    B = 256
    x = torch.randn(B, S, 3 * S, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        be.grid_transpose(x, y, B, S)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); be.grid_transpose(x, y, B, S); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[10]
    ref = x.view(B, S, S, 3).transpose(1, 2).reshape(B, S, 3 * S)
    print(f"S={S:4d}: {t*1e3:7.1f} us  {2*x.numel()*4/t/1e6:7.1f} GB/s  exact={bool(torch.equal(y, ref))}")
