#!/usr/bin/env python3
"""Sanity soak: N optimizer steps on one fixed synthetic batch (overfitting it): the loss must stay finite and go down,
in fp32 and under autocast(bfloat16) + GradScaler.  usage: soak_train.py [workload] [steps] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from importlib import import_module
import torch
import bench
import calm_vit_dte_amd as calm

trainer = import_module("calm_vit_dte_amd.trainer")
name = sys.argv[1] if len(sys.argv) > 1 else "base224"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
wl = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
S, C = wl["kw"]["seq_length"], wl["kw"]["out_features"]
for autocast in (False, True):
    m = bench.build_model(calm, wl["kw"], dev).train()
    x, y = bench.synthetic_batch(batch, S, C, 0, dev)
    opt = trainer.FusedClipAdamW(m, lr=3e-4)
    step = trainer.TrainStep(m, opt, None, scaler=torch.amp.GradScaler("cuda") if autocast else None,
                             autocast_dtype=torch.bfloat16 if autocast else None)
    losses = []
    for i in range(steps):
        loss, _ = step(x, y)
        if i % 5 == 0 or i == steps - 1:
            losses.append(float(loss))
    print(f"{name} autocast={autocast} bs={batch}: loss " + " ".join(f"{l:.3f}" for l in losses))
    assert all(l == l and abs(l) < 1e4 for l in losses), "loss not finite"
    assert losses[-1] < losses[0], "loss did not go down on a fixed batch"
    opt.close()
    del m, opt, step
    torch.cuda.empty_cache()
print("soak ok")
