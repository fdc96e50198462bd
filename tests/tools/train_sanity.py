#!/usr/bin/env python3
"""Sanity run (not a test): N training steps of Small-224 on one fixed synthetic batch with the fused optimizer-side
step; prints the loss trajectory and the gradient norm (must fall / stay finite)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from importlib import import_module
import torch
import bench
import calm_vit_dte_amd as calm

trainer = import_module("calm_vit_dte_amd.trainer")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.WORKLOADS["small224"]
dev = torch.device("cuda", 0)
m = bench.build_model(calm, wl["kw"], dev).train()
x, y = bench.synthetic_batch(bs, 224, 1000, 0, dev)
opt = trainer.FusedClipAdamW(m, lr=1e-3)
step = trainer.TrainStep(m, opt, None)
torch.manual_seed(0)
for i in range(steps):
    loss, _ = step(x, y)
    if i % 5 == 0 or i == steps - 1:
        print(f"step {i:3d}  loss {float(loss):8.4f}  grad-norm {float(opt.stats[0]):9.4f}  found_inf {float(opt.stats[1]):.0f}", flush=True)
opt.close()
