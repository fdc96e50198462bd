#!/usr/bin/env python3
"""Which host call sites issue the stock-torch kernels left on the training step (copies, fills, adds, muls)?
Runs one Small-224 step at a small batch under torch.profiler with Python stacks and prints, per aten op, the
innermost frames inside this repo.  Diagnostic tooling only."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import calm_vit_dte_amd as calm  # noqa: E402
from importlib import import_module  # noqa: E402

trainer = import_module("calm_vit_dte_amd.trainer")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "small224"]
dev = torch.device("cuda", 0)
model = bench.build_model(calm, wl["kw"], dev).train()
x, y = bench.synthetic_batch(4, wl["kw"]["seq_length"], wl["kw"]["out_features"], 0, dev)
opt = trainer.FusedClipAdamW(model)
step = trainer.TrainStep(model, opt, None)
for _ in range(2):
    step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, y)
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::clone", "aten::zeros", "aten::fill_", "aten::zero_", "aten::add", "aten::add_",
         "aten::mul", "aten::mul_", "aten::contiguous", "aten::zeros_like", "aten::empty_like")
agg = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in WATCH:
        frames = [f for f in (ev.stack or []) if "calm-vit-dte_amd" in f or "calm_vit" in f or "bench.py" in f]
        key = " <- ".join(frames[:3]) if frames else "(autograd engine / no repo frame): " + " | ".join((ev.stack or [])[:2])
        agg[ev.name][key] += 1
for name, c in agg.items():
    print(f"== {name}: {sum(c.values())}")
    for k, n in c.most_common(12):
        print(f"   {n:5d}  {k}")
