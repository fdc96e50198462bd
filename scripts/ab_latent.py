#!/usr/bin/env python3
"""Time calm_latent_fwd / _bwd at Base-224's and Small-224's latent sizes (bs=256); A/B via CALM_VIT_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
print(os.environ.get("CALM_VIT_LIB", "default lib"))
for rows, mvh in ((256 * 80, 240), (256 * 40, 120), (128 * 160, 480)):
    mv, noise = torch.randn(rows, 2 * mvh, device="cuda"), torch.randn(rows, mvh, device="cuda")
    z, std, kl = torch.empty(rows, mvh, device="cuda"), torch.empty(rows, mvh, device="cuda"), torch.zeros((), device="cuda")
    def f():
        kl.zero_(); be.latent_fwd(mv, noise, z, std, kl, rows, mvh)
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[10]
    sd = torch.nn.functional.softplus(mv[:, mvh:].double()) + 1e-6
    ref_kl = (1 + 2 * sd.log() - mv[:, :mvh].double() ** 2 - sd ** 2).sum()
    print(f"rows {rows} mvh {mvh}: {t*1e3:6.1f} us ({rows*mvh*20/t/1e6:6.1f} GB/s)  std err {float((std.double()-sd).abs().max()/sd.abs().max()):.2e} "
          f"std rel err {float(((std.double()-sd).abs()/sd).max()):.2e}  kl err {abs(float(kl)-float(ref_kl))/abs(float(ref_kl)):.2e}")
