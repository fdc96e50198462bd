#!/usr/bin/env python3
"""Which host-side call sites issue the small aten fill / copy / add launches of one training step?
(torch profiler with Python stacks on the Nano-48 model: the host code path is the same for every configuration)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from collections import Counter
from importlib import import_module
import torch
from torch.profiler import profile, ProfilerActivity
import calm_vit_dte_amd as calm
from helpers import CONFIGS, load_golden
from test_host_logic_cpu import build_model

trainer = import_module("calm_vit_dte_amd.trainer")
name = "nano48_cls"
g = load_golden(name)
cfg = CONFIGS[name]
x = torch.randn(4, 3, 48, 48, device="cuda")
y = torch.nn.functional.one_hot(torch.tensor([1, 2, 3, 4]), cfg.out_features).float().cuda()
m = build_model(name, g, "cuda").train()
opt = trainer.FusedClipAdamW(m)
step = trainer.TrainStep(m, opt, None)
for _ in range(2):
    step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(x, y)
torch.cuda.synchronize()
watch = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous")
c = Counter()
for e in prof.events():
    if e.name in watch:
        st = [s for s in (e.stack or []) if "calm-vit-dte_amd" in s]
        c[(e.name, st[0].split("calm-vit-dte_amd/")[-1] if st else "<torch internal / autograd>")] += 1
for (n, s), k in sorted(c.items(), key=lambda t: -t[1])[:45]:
    print(f"{k:5d} {n:16s} {s[:110]}")
