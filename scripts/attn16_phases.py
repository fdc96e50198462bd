#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of attn16_fwd_kernel (library built with -DATT16_STAMP, loaded through CALM_VIT_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
for B, S, H, hd in [(256, 224, 12, 56), (256, 176, 12, 44), (256, 128, 12, 32), (256, 80, 12, 20)]:
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
    ph = lse[:, 0, :12].double().mean(0).tolist()
    if os.environ.get("CALM_ATTN16_V2", "1") != "0" and S >= 16:
        print(f"   v2 wave 0: wait+barrier {ph[3]:8.0f} {ph[4]:8.0f} {ph[5]:8.0f} | DMA issue {ph[6]:8.0f} {ph[7]:8.0f} {ph[8]:8.0f} | compute {ph[9]:8.0f} {ph[10]:8.0f} {ph[11]:8.0f}")
    print(f"S{S} hd{hd}: phase1 (R = Q K^T) {ph[0]:9.0f}  phase2 (mask MLP) {ph[1]:9.0f}  phase3 (heads) {ph[2]:9.0f} cycles per workgroup")
