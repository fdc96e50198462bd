#!/usr/bin/env python3
"""Section timings of attn16_fwd3_core_kernel from an ATT16_STAMP3 build (CALM_VIT_LIB=ab/libcalmvit_stamp3.so): every
workgroup writes its stamps over its lse row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
B, S, H, hd = 256, 224, 12, 56
D = H * hd
bf = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).bfloat16()
q, k, v = bf(B, S, D), bf(B, S, D), bf(B, S, D, sc=1.0)
w1, w2 = bf(2 * S, S, sc=S ** -0.5), bf(S, 2 * S, sc=(2 * S) ** -0.5)
b1, b2 = torch.randn(2 * S, device="cuda") * 0.1, torch.randn(S, device="cuda") * 0.1
s1, s2 = torch.tensor([1.3], device="cuda"), torch.tensor([0.8], device="cuda")
e = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device="cuda")
out, R, hp, hg, Mk, MkT = e(B, S, D), e(B, S, S), e(B * S, 2 * S), e(B * S, 2 * S), e(B, S, S), e(B, S, S)
lse = torch.empty(B, H, S, device="cuda")
for _ in range(3):
    be.attn16_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd)
torch.cuda.synchronize()
st = lse[:, :, :9].reshape(-1, 9).cpu()
order = st[:, 8].argsort()
st = st[order]
names = ["rt100MHz", "issued", "staged", "qk_done", "softmax_done", "pv_done", "end"]  # qk/softmax/pv: first tile PAIR of wave 0
import statistics
print("workgroups", st.shape[0])
for lo, hi, tag in ((0, 512, "first round (ids 0..511)"), (1024, 2048, "middle"), (2560, 3072, "last")):
    seg = st[lo:hi]
    med = [float(seg[:, i].median()) for i in range(7)]
    print(f"{tag:28s} " + " ".join(f"{n}={m:9.0f}" for n, m in zip(names, med)))
    print(f"{'':28s} stage_wait={med[2]-med[1]:8.0f} qk={med[3]-med[2]:8.0f} softmax={med[4]-med[3]:8.0f} pv={med[5]-med[4]:8.0f} "
          f"tile1={med[5]-med[2]:8.0f} all_tiles={med[6]-med[2]:8.0f} cycles/100MHz-tick={med[6]/max(med[0],1):6.2f}")
