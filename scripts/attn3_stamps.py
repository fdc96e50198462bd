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
rows = []  # first item of workgroup wg = image wg, head 0
for wg in range(256):                                  # first item of workgroup wg (grid 256, B % 8 == 0): xcd = wg & 7, it0 = wg >> 3
    pass
    rows.append(lse[wg, 0, :9].cpu())
st = torch.stack(rows)
names = ["rt100MHz", "prologue", "qk_done", "softmax_done", "pv_done", "item_done", "end"]
print("workgroups with stamps", st.shape[0])
med = [float(st[:, i].median()) for i in range(7)]
items = float(st[:, 7].median())
print(" ".join(f"{n}={m:9.0f}" for n, m in zip(names, med)))
print(f"first item: qk={med[2]-med[1]:8.0f} softmax={med[3]-med[2]:8.0f} pv={med[4]-med[3]:8.0f} barrier_wait={med[5]-med[4]:8.0f} item={med[5]-med[1]:8.0f}; "
      f"items/wg={items:.0f} avg item={(med[6]-med[1])/items:8.0f} cycles; cycles per 100MHz tick={med[6]/max(med[0],1):6.2f}")
