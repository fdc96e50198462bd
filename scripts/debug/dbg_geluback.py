import os, sys
sys.path.insert(0, os.getcwd())
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=gen, device="cuda") * scale


M, K, N = 57344, 672, 1344
b16 = lambda t: t.bfloat16()
x, w1, dy = b16(g(M, K, seed=1)), b16(g(N, K, seed=2, scale=K ** -0.5)), b16(g(M, N, seed=3))
bias, sigma = g(N, seed=4, scale=0.1), torch.tensor([1.3], device="cuda")
nanb = lambda *s: torch.full(s, float("nan"), device="cuda", dtype=torch.bfloat16)
hp, hg = nanb(M, N), nanb(M, N)
be.gemm(x, w1, hg, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sigma, bias=bias, act=1, C_pre=hp, split_k=1)
z = (x.float() @ w1.float().T) / 1.3 + bias
print("hp err", float((hp.float() - z).abs().max() / z.abs().max()), "hg err",
      float((hg.float() - torch.nn.functional.gelu(z)).abs().max() / z.abs().max()), "nan:", int(torch.isnan(hp.float()).sum()), int(torch.isnan(hg.float()).sum()))
m2 = 3000
hp2, hg2 = nanb(m2, N), nanb(m2, N)
be.gemm(x[:m2], w1, hg2, m2, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sigma, bias=bias, act=1, C_pre=hp2, split_k=1)
print("slice equal:", bool(torch.equal(hp2, hp[:m2]) and torch.equal(hg2, hg[:m2])))
w2 = b16(g(N, N, seed=5, scale=N ** -0.5))
zz = hp.float().requires_grad_(True)
torch.nn.functional.gelu(zz).backward((dy.float() @ w2.float()) / 1.3)
ref = zz.grad
first = None
for it in range(4):
    for mode in (1, 0):
        be.gemm_set_option(be.GEMM_OPT_PIPE, mode)
        dz = nanb(M, N)
        be.gemm(dy, w2, dz, M, N, N, (N, 1, 0, 0), (1, N, 0, 0), (N, 0, 0), inv_scale=sigma, act=2, aux=hp, split_k=1)
        o = dz.float()
        d = (o - ref).abs()
        d[torch.isnan(d)] = 1e9
        i = int(d.argmax()); r, c = divmod(i, N)
        msg = f"iter {it} {'pipe' if mode else 'old '} rel_err {float(d.max() / ref.abs().max()):.5f} at {(r, c)} got {float(o[r, c]):.5f} ref {float(ref[r, c]):.5f} nan {int(torch.isnan(o).sum())}"
        if first is None:
            first = o.clone()
        else:
            dd = (o - first).abs(); dd[torch.isnan(dd)] = 1e9
            nb = int((dd > 0).sum())
            msg += f" | vs first run: {nb} differing"
            if nb:
                idx = (dd > 0).nonzero()
                rows, cols = idx[:, 0], idx[:, 1]
                msg += f" rows {int(rows.min())}..{int(rows.max())} ({rows.unique().numel()}) cols {int(cols.min())}..{int(cols.max())} ({cols.unique().numel()}) maxdiff {float(dd.max()):.4f}"
        print(msg)
