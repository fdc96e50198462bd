#!/usr/bin/env python3
"""Context measurement (NOT the product, NOT the bench contract): the CPU oracle's pure-torch arithmetic run on
the MI355X through stock PyTorch-ROCm eager kernels (rocBLAS/hipBLASLt GEMMs, ATen elementwise), same synthetic
batch, same step (fwd + soft-target CE + bwd + clip + AdamW).  Answers "what would the reference's eager code
get on this GPU".  usage: eager_gpu_baseline.py [--batch 256] [--autocast]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
import weights as W
from oracle import calm_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--autocast", action="store_true", help="bf16 autocast, as the reference trainer (cls:84)")
a = ap.parse_args()
cfg = O.ViTConfig(heads=6, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=120, seq_len_step=16,
                  seq_len_reduce=40, out_features=1000, force_reduce=False, generate=False)
dev = "cuda"
P = {k: torch.from_numpy(v).to(dev) for k, v in W.make_params(O.vit_param_shapes(cfg), 1234).items()}
leaves = [P[k].requires_grad_(True) for k in P if not O.is_buffer(k)]
opt = torch.optim.AdamW(leaves, lr=3.1e-3, weight_decay=0.02, betas=(0.9, 0.98), fused=True)
g = np.random.default_rng(0)
x = torch.from_numpy(g.standard_normal((a.batch, 3, 224, 224)).astype(np.float32)).to(dev)
y = torch.nn.functional.one_hot(torch.from_numpy(g.integers(0, 1000, a.batch)), 1000).float().to(dev)
# the oracle builds its RoPE position vector on the CPU; patch arange for the GPU run
_arange = torch.arange
torch.arange = lambda *s, **k: _arange(*s, **{**k, "device": k.get("device", dev)})


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.autocast):
        out, _ = O.vit_forward(P, cfg, x, True)
        loss = torch.nn.functional.cross_entropy(out.float(), y)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(leaves, 1.0)
    opt.step(); opt.zero_grad()
    return loss


for _ in range(a.warmup):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"torch-eager on MI355X ({'bf16 autocast' if a.autocast else 'fp32'}), bs={a.batch}: {1e3*dt:.1f} ms/step, "
      f"{a.batch/dt:.1f} images/s, loss {float(loss):.3f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
