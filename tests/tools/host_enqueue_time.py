#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step (Small-224 bs=256) compared with the GPU time of the step?
(margin against becoming launch-bound, e.g. with 8 ranks sharing the host)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from importlib import import_module
import torch
import bench
import calm_vit_dte_amd as calm

trainer = import_module("calm_vit_dte_amd.trainer")
wl = bench.WORKLOADS["small224"]
dev = torch.device("cuda", 0)
m = bench.build_model(calm, wl["kw"], dev).train()
x, y = bench.synthetic_batch(256, 224, 1000, 0, dev)
opt = trainer.FusedClipAdamW(m)
step = trainer.TrainStep(m, opt, None)
for _ in range(3):
    step(x, y)
torch.cuda.synchronize()
n = 4
t0 = time.perf_counter()
for _ in range(n):
    step(x, y)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3*t_enq/n:.1f} ms/step, wall {1e3*t_all/n:.1f} ms/step (enqueue runs ahead of the GPU by the difference)")
# enqueue cost with an idle GPU queue: tiny batch
xs, ys = x[:2].contiguous(), y[:2].contiguous()
for _ in range(2):
    step(xs, ys)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step(xs, ys)
torch.cuda.synchronize()
print(f"bs=2 step (launch-bound): {1e3*(time.perf_counter()-t0)/n:.1f} ms/step = host cost of issuing one step")
opt.close()
