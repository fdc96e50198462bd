#!/usr/bin/env python3
"""Where do the HIP bf16 path and the CPU emulation of its rounding points part?  Runs one fixture block on both
backends in bf16 mode, records every operator's outputs in call order and prints the first / largest differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import calm_vit_dte_amd as calm
import weights as W
from emulated_backend import EmulatedBackend
from helpers import BLOCK_FIXTURES, block_fixture_params, load_golden, rel_err

ops = calm.ops
FNS = [getattr(ops, n) for n in dir(ops) if isinstance(getattr(ops, n), type) and issubclass(getattr(ops, n), torch.autograd.Function)
       and getattr(ops, n) is not torch.autograd.Function]
trace = []
orig = {}
def wrap(cls):
    o = cls.apply
    orig[cls] = o
    def w(*a, **k):
        out = o(*a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        for i, t in enumerate(outs):
            if torch.is_tensor(t):
                trace.append((f"{cls.__name__}[{i}]", t.detach().float().cpu().clone()))
        return out
    cls.apply = w
for c in FNS:
    wrap(c)

def run(name, dev, prec):
    vt = calm.Vi_Tools_CNN_less_V2
    g = load_golden("block_" + name); kw = BLOCK_FIXTURES[name]
    shapes, P = block_fixture_params(name, g)
    blk = vt.VMLA_Block(mlp_dim=2 * kw["dim2"], force_reduce=False, **kw)
    blk.load_state_dict({k: v.clone() for k, v in P.items()})
    blk = blk.to(dev).train()
    S, D1 = kw["seq_length"], kw["dim1"]
    xq = torch.from_numpy(W.make_input((1, S, D1), 5, "xq")).to(dev).requires_grad_(True)
    xkv = torch.from_numpy(W.make_input((1, S, D1), 6, "xkv")).to(dev).requires_grad_(True) if kw["is_cross"] else None
    sm = vt.ResidualStateManager(mode="sum")
    calm.backend.set_matmul_precision(prec)
    calm.ops.set_noise_override(W.NoiseStream(9))
    trace.clear()
    y = blk(xq, input_kv=xkv, state_manager=sm, mask=True)
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 8, "gy")).to(dev)
    ((y * gy).sum() + 0.5 * sm.get_kl_loss()).backward()
    calm.ops.set_noise_override(None)
    fw = list(trace)
    return fw, xq.grad.detach().cpu()

name = sys.argv[1] if len(sys.argv) > 1 else "A_hd56"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
th, dxh = run(name, "cuda", prec)
with calm.backend.use_backend(EmulatedBackend()):
    te, dxe = run(name, "cpu", prec)
print(name, prec, "ops", len(th), len(te))
for (n1, a), (n2, b) in zip(th, te):
    print(f"{n1:32s} {tuple(a.shape)!s:22s} rel {rel_err(a, b):.3e}")
print("dxq rel", rel_err(dxh, dxe))
