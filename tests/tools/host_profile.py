#!/usr/bin/env python3
"""Host cost of issuing one training step (launch-bound run at bs=2) for a workload / mode, and where it goes
(cProfile, top functions by own time).  usage: host_profile.py [workload] [--autocast]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from importlib import import_module
import torch
import bench
import calm_vit_dte_amd as calm

trainer = import_module("calm_vit_dte_amd.trainer")
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "base224"
autocast = "--autocast" in sys.argv
wl = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
m = bench.build_model(calm, wl["kw"], dev).train()
S = wl["kw"]["seq_length"]
x, y = bench.synthetic_batch(2, S, wl["kw"]["out_features"], 0, dev)
opt = trainer.FusedClipAdamW(m)
step = trainer.TrainStep(m, opt, None, scaler=torch.amp.GradScaler("cuda") if autocast else None,
                         autocast_dtype=torch.bfloat16 if autocast else None)
for _ in range(3):
    step(x, y)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    step(x, y)
torch.cuda.synchronize()
print(f"{name} autocast={autocast} bs=2 (launch-bound): {1e3*(time.perf_counter()-t0)/n:.1f} ms/step = host cost of issuing one step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step(x, y)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
opt.close()
