#!/usr/bin/env python3
"""Where the host time of one eager training step goes: cProfile over a few steps of the Base-224 autocast step (the GPU
runs behind; one synchronisation per step outside the profiled call would hide nothing, so none is made inside)."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import calm_vit_dte_amd as calm
from importlib import import_module
trainer = import_module("calm_vit_dte_amd.trainer")

wl = bench.WORKLOADS["base224"]
dev = torch.device("cuda:0")
calm.backend.set_matmul_precision("fp32")
model = bench.build_model(calm, wl["kw"], dev).train()
x, y = bench.synthetic_batch(wl["batch"], wl["kw"]["seq_length"], wl["kw"]["out_features"], seed=0, device=dev)
opt = trainer.FusedClipAdamW(model)
step = trainer.TrainStep(model, opt, None, scaler=torch.amp.GradScaler("cuda"), autocast_dtype=torch.bfloat16)
for _ in range(3):
    step(x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 4
pr.enable()
for _ in range(N):
    step(x, y)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("tottime")
st.print_stats(28)
print(out.getvalue()[:6000])
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
print(out.getvalue()[:5000])
