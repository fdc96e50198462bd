#!/usr/bin/env python3
"""Does a hipGraph capture of the training step survive with the bucketed RCCL all-reduces inside?  (VERDICT r3 #4b.)
Run in a process of its own: round 3's attempt died with a segfault inside capture_end.  World of one on cuda:0, the
reducer forced on; prints CAPTURE_OK <max rel diff vs eager> or dies with the runtime's own message."""
import faulthandler, os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for pth in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, pth)
faulthandler.enable()
import numpy as np
import torch
import torch.distributed as dist
import calm_vit_dte_amd as calm
from importlib import import_module
from helpers import CONFIGS, load_golden, rel_err
from test_host_logic_cpu import build_model
trainer = import_module("calm_vit_dte_amd.trainer")

with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
name = "tiny32_cls"
g = load_golden(name); cfg = CONFIGS[name]
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.standard_normal((8, 3, cfg.seq_length, cfg.seq_length)).astype(np.float32)).cuda()
y = torch.zeros(8, cfg.out_features).cuda(); y[torch.arange(8), torch.arange(8) % cfg.out_features] = 1.0
res = []
for graphed in (False, True):
    m = build_model(name, g, "cuda").train()
    opt = trainer.FusedClipAdamW(m)
    red = trainer.BucketedGradReducer(m, bucket_mb=1, tail_mb=1, force=True)
    assert red.enabled and len(red.buckets) >= 2
    if graphed:
        print("capturing ...", flush=True)
        step = trainer.GraphedTrainStep(m, opt, x, y, warmup=2, reducer=red)
        print("captured", flush=True)
    else:
        step = trainer.TrainStep(m, opt, red)
        step(x, y); step(x, y)
    losses = [float(step(x, y)[0]) for _ in range(3)]
    torch.cuda.synchronize()
    res.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    opt.close()
(l0, s0), (l1, s1) = res
worst = max(rel_err(s1[k].float(), s0[k].float()) for k in s0)
print("CAPTURE_OK", worst, l0, l1, flush=True)
# orderly teardown: the captured graph holds RCCL kernels of this communicator — release it (and everything that
# references it) BEFORE the process group goes away; one of four runs of the first version of this script, which
# destroyed the group with the graph still alive, ended in SIGABRT after its work was done
del step, red, opt, m
import gc
gc.collect()
torch.cuda.synchronize()
dist.destroy_process_group()
