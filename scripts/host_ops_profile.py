#!/usr/bin/env python3
"""Which ATen ops (outside the library's kernels) run per training step, and how many launches they cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import importlib.util, torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import calm_vit_dte_amd as calm
from importlib import import_module
trainer = import_module("calm_vit_dte_amd.trainer")
wl = bench.WORKLOADS["small224"]
dev = torch.device("cuda", 0)
model = bench.build_model(calm, wl["kw"], dev).train()
opt = trainer.make_optimizer(model)
step = trainer.TrainStep(model, opt, None)
x, y = bench.synthetic_batch(32, 224, 1000, 0, dev)
for _ in range(2):
    step(x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(x, y)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_cpu_time_total, getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[1])
print(f"{'op':60s} {'calls':>6s} {'self_cpu_ms':>11s} {'self_gpu_ms':>11s}")
for k, n, c, g in rows[:45]:
    print(f"{k[:60]:60s} {n:6d} {c/1e3:11.2f} {g/1e3:11.2f}")
