import math, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch
import calm_vit_dte_amd as calm
from helpers import rel_err
from test_fullsize_gpu import g
be = calm.backend.get_backend()
B, S, H, hd = 256, 224, 12, 56
D = H * hd
b16 = lambda t: t.bfloat16()
q, k, v = b16(g(B, S, D, seed=1, scale=0.3)), b16(g(B, S, D, seed=2, scale=0.3)), b16(g(B, S, D, seed=3))
w1, b1 = b16(g(2 * S, S, seed=4, scale=S ** -0.5)), g(2 * S, seed=5, scale=0.1)
w2, b2 = b16(g(S, 2 * S, seed=6, scale=(2 * S) ** -0.5)), g(S, seed=7, scale=0.1)
s1, s2 = torch.tensor([0.9], device="cuda"), torch.tensor([1.2], device="cuda")
e = lambda *s: torch.empty(*s, device="cuda", dtype=torch.bfloat16)
outs = lambda n: (e(n, S, D), e(n, S, S), e(n, S, 2 * S), e(n, S, 2 * S), e(n, S, S), e(n, S, S), torch.empty(n, H, S, device="cuda"))
out, R, hp, hg, Mk, MkT, lse = outs(B)
be.attn16_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd)
n = 3
Rref = torch.einsum("bid,bjd->bij", q[:n].float(), k[:n].float())
print("R", rel_err(R[:n].float(), Rref), "limit", 2.0 ** -7)
mask = torch.nn.functional.gelu(R[:n].float() @ w1.float().T / 0.9 + b1) @ w2.float().T / 1.2 + b2
print("Mk", rel_err(Mk[:n].float(), mask), "limit 3e-2")
qh, kh, vh = (t[:n].float().view(n, S, H, hd).transpose(1, 2) for t in (q, k, v))
logits = qh @ kh.transpose(-1, -2) / math.sqrt(hd) + Mk[:n].float()[:, None]
ref = torch.softmax(logits, dim=-1) @ vh
print("out", rel_err(out[:n].float(), ref.transpose(1, 2).reshape(n, S, D)), "limit 3e-2")
print("lse", rel_err(lse[:n], torch.logsumexp(logits, dim=-1)), "limit 2e-3")
ones = torch.ones_like(v); o1 = outs(B)
be.attn16_fwd(q, k, ones, w1, b1, s1, w2, b2, s2, *o1, B, S, H, hd)
print("V=1", float((o1[0].float() - 1.0).abs().max()), "limit", 2.0 ** -7)
dout = b16(g(B, S, D, seed=8))
dq, dk, dv, dM = e(B, S, D), e(B, S, D), e(B, S, D), e(B * S, S)
delta = torch.empty(B, H, S, device="cuda")
be.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta, dq, dk, dv, dM, B, S, H, hd)
dMf = dM.float().view(B, S, S)
print("dM rowsum", float(dMf.sum(-1).abs().max() / dMf.abs().sum(-1).max()), "limit 2e-2")
P = torch.softmax(logits, dim=-1)
dvr = (P.transpose(-1, -2) @ dout[:n].float().view(n, S, H, hd).transpose(1, 2)).transpose(1, 2).reshape(n, S, D)
print("dv", rel_err(dv[:n].float(), dvr), "limit 3e-2")
