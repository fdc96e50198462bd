"""N>1 path on CPU: world_size-2 gloo run of the package's data-parallel training step
(init_distributed / sync_module_states / BucketedGradReducer / TrainStep) on the Tiny-32 model —
BASELINE.json configs[0], the reference's CPU/gloo plumbing case (distributed_trainer_cls.py:46,51).
The model's kernels are the torch emulation of the C-ABI (no GPU here); what is under test is the
sharding, the bucketed mean all-reduce and the step semantics: two ranks with half the batch each
must end bit-identical to each other and equal to one process with the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup_path():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_steps(rank, world, x, y, steps, bucket_mb, stock_ddp=False, fused_optim=False):
    from importlib import import_module
    import calm_vit_dte_amd as calm
    from emulated_backend import EmulatedBackend
    from test_host_logic_cpu import build_model
    trainer = import_module("calm_vit_dte_amd.trainer")
    torch.manual_seed(100 + rank)                        # different init per rank: sync must fix it
    m = build_model("tiny32_cls", None).train()
    if rank != 0:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    with calm.backend.use_backend(EmulatedBackend()):
        if stock_ddp:
            # exactly what the reference's train() does (distributed_trainer_cls.py:55)
            from torch.nn.parallel import DistributedDataParallel as DDP
            wrapped = DDP(m)
            opt = trainer.make_optimizer(m)
            step = trainer.TrainStep(wrapped, opt, None)
            step.params = [p for p in m.parameters() if p.requires_grad]
        else:
            trainer.sync_module_states(m)
            # fused_optim: gradients of the spectral-normed weights stay un-corrected through the all-reduce and are
            # corrected inside the optimizer-side step (mean all-reduce commutes with the linear correction)
            opt = trainer.FusedClipAdamW(m) if fused_optim else trainer.make_optimizer(m)
            red = trainer.BucketedGradReducer(m, bucket_mb=bucket_mb) if world > 1 else None
            step = trainer.TrainStep(m, opt, red)
        n = x.shape[0] // world
        xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
        try:
            losses = [float(step(xs, ys)[0]) for _ in range(steps)]
        finally:
            if fused_optim:
                opt.close()
    return m, losses


def _worker(rank, world, port, x, y, steps, outdir, stock_ddp=False, fused_optim=False):
    _setup_path()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")
    r, lr, w = trainer.init_distributed(use_gpu=False)
    assert (r, w) == (rank, world)
    m, losses = _run_steps(rank, world, x, y, steps, bucket_mb=1, stock_ddp=stock_ddp, fused_optim=fused_optim)   # 1 MiB buckets -> several
    torch.save({"sd": m.state_dict(), "losses": losses}, os.path.join(outdir, f"rank{rank}.pt"))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("stock_ddp,fused_optim", [(False, False), (True, False), (False, True)],
                         ids=["bucketed_reducer", "torch_DDP_wrapper", "bucketed_reducer+fused_optimizer_step"])
def test_two_rank_gloo_training_matches_single_process(tmp_path, stock_ddp, fused_optim):
    _setup_path()
    import numpy as np
    g = np.random.default_rng(0)
    x = torch.from_numpy(g.standard_normal((8, 3, 32, 32)).astype(np.float32))          # bs=8, CIFAR-shaped
    y = torch.nn.functional.one_hot(torch.from_numpy(g.integers(0, 10, 8)), 10).float() * 0.9 + 0.01
    steps = 2
    mp.spawn(_worker, args=(2, _free_port(), x, y, steps, str(tmp_path), stock_ddp, fused_optim), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"ranks diverged at {k}"
    torch.set_num_threads(2)
    m, losses = _run_steps(0, 1, x, y, steps, bucket_mb=1)
    mean_loss = [(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])]
    assert all(abs(a - b) < 1e-4 * max(1.0, abs(b)) for a, b in zip(mean_loss, losses))
    sd = m.state_dict()
    worst = max(float((sd[k] - r0["sd"][k]).abs().max()) for k in sd)
    assert worst < 5e-4, worst                     # AdamW's 1/sqrt(v) amplifies fp32 reduction-order noise
