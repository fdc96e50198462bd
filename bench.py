#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the CALM-ViT path on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one training iteration of distributed_trainer_cls.py:79-96 on one synthetic batch that is
already resident in HBM: forward, soft-target cross-entropy, backward (+ gradient all-reduce when N>1),
clip_grad_norm_(1.0), AdamW, zero_grad.  Workload at N=1: BASELINE.json configs[1] — CALM-ViT-Small,
224x224, bs=256/GPU, fp32 (SURVEY.md 8(d) variant table).  Weak scaling: every rank keeps bs/GPU.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline":     the dominant kernel (the fp32 MFMA GEMM family, calm_gemm): algorithmic FLOPs of the
                  calls of `prof_steps` steps / their summed durations measured with HIP events on the
                  launch stream, against the dense fp32-matrix peak (157.3 TFLOP/s).
  "cpu_baseline": the CPU oracle (oracle/calm_oracle.py, a port; the reference's Python cannot travel)
                  timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # kwargs of ViT(...); fwd+bwd GFLOP/img measured on the reference
    "small224": dict(kw=dict(heads=6, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=120,
                             seq_len_step=16, seq_len_reduce=40, out_features=1000), gflop_img=46.063, batch=256),
    "base224": dict(kw=dict(heads=12, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=240,
                            seq_len_step=16, seq_len_reduce=80, out_features=1000), gflop_img=48.359, batch=256),
    # bs=128: BASELINE config #4 names no batch; 128 images fill the chip (384 workgroups of the attention kernels) and
    # use 45 of the 288 GB (round 1 ran bs=32: `--batch 32` reproduces it)
    "base384": dict(kw=dict(heads=12, seq_length=384, in_features=1152, dim_step=48, mean_var_hidden=240,
                            seq_len_step=16, seq_len_reduce=80, out_features=1000), gflop_img=356.425, batch=128),
    "large224": dict(kw=dict(heads=6, seq_length=224, in_features=672, dim_step=24, mean_var_hidden=480,
                             seq_len_step=8, seq_len_reduce=160, out_features=1000), gflop_img=92.167, batch=128),
    "nano48": dict(kw=dict(heads=3, seq_length=48, in_features=144, dim_step=12, mean_var_hidden=24,
                           seq_len_step=4, seq_len_reduce=16, out_features=10), gflop_img=0.500, batch=64),
}
PEAK_FP32_MATRIX_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_DENSE_TFLOPS = 2500.0
# per precision: (dtype label, matrix-pipe peak for the ALGORITHMIC flops, kernel label)
PRECISION_INFO = {
    "fp32": ("f32", PEAK_FP32_MATRIX_TFLOPS, "gemm_f32_kernel (calm_gemm, v_mfma_f32_32x32x2_f32)"),
    "bf16": ("bf16 (bf16 GEMM / attention tensors, f32 accumulate, f32 residual stream and parameters)",
             PEAK_BF16_DENSE_TFLOPS, "gemm_bf16p_kernel (pipelined persistent family, v_mfma_f32_16x16x32_bf16) + gemm_bf16w / gemm_bf16c_kernel "
             "(calm_gemm)"),
    "fp8": ("bf16 + fp8 (e4m3 / e5m2 operands of the Linear forward and input-gradient GEMMs, f32 accumulate; bf16 "
            "pipeline elsewhere)", PEAK_BF16_DENSE_TFLOPS,
            "gemm_fp8w_kernel + gemm_bf16w/c_kernel (calm_gemm; non-scaled fp8 MFMAs run at the bf16 rate)"),
    "bf16x3": ("f32 via bf16x3 split", PEAK_BF16_DENSE_TFLOPS / 3.0,
               "gemm_bf16c_kernel<3> (calm_gemm, 3 x v_mfma_f32_32x32x16_bf16 per product)"),
}
WEIGHT_SEED = 1234


def synthetic_batch(batch, S, classes, seed, device):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.standard_normal((batch, 3, S, S)).astype(np.float32))
    # MixUp-style soft labels: lam * onehot(a) + (1-lam) * onehot(b), lam ~ Beta(0.8, 0.8) (cls:58-61)
    a, b = g.integers(0, classes, batch), g.integers(0, classes, batch)
    lam = g.beta(0.8, 0.8)
    y = np.zeros((batch, classes), dtype=np.float32)
    y[np.arange(batch), a] += lam
    y[np.arange(batch), b] += 1.0 - lam
    return x.to(device), torch.from_numpy(y).to(device)


def build_model(calm, kw, device):
    W = calm.synthetic_weights
    m = calm.ViT(torch.device("cpu"), type=8, force_reduce=False, generate=False, **kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_params(shapes, WEIGHT_SEED).items()})
    return m.to(device)


class GemmProfiler:
    """Wraps backend.gemm with HIP events on the launch stream (torch's current stream, which is the
    stream every kernel of the library is enqueued on)."""

    def __init__(self, be):
        self.be = be
        self.orig = be.gemm
        self.records = []

    def __enter__(self):
        def gemm(A, B, C, M, N, K, a, b, c, batch=(1, 1), **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig(A, B, C, M, N, K, a, b, c, batch=batch, **kw)
            e1.record()
            self.records.append((2.0 * M * N * K * batch[0] * batch[1], e0, e1,
                                 (M, N, K, batch[0] * batch[1], int(a[1] == 1), int(b[1] == 1),
                                  int(bool(kw.get("reduce_batch"))))))
        self.be.gemm = gemm
        return self

    def __exit__(self, *exc):
        self.be.gemm = self.orig
        return False

    def report(self, path, steps):
        """Per-shape table (M,N,K,batch,A k-contig,B k-contig,reduce): launches, ms/step, TFLOP/s."""
        torch.cuda.synchronize()
        agg = {}
        for fl, e0, e1, key in self.records:
            t = agg.setdefault(key, [0, 0.0, 0.0])
            t[0] += 1
            t[1] += e0.elapsed_time(e1)
            t[2] += fl
        with open(path, "w") as f:
            f.write("M,N,K,batch,a_kc,b_kc,reduce,launches_per_step,ms_per_step,tflops\n")
            for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(",".join(map(str, key)) + f",{n // steps},{ms / steps:.3f},{fl / (ms * 1e-3) / 1e12:.2f}\n")

    def summary(self):
        torch.cuda.synchronize()
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return flops, ms, len(self.records)


class AttentionProfiler:
    """HIP events around every fused-attention forward launch (fp32: calm_attention_fwd, bf16: calm_attention16_fwd);
    algorithmic FLOPs per launch = B * (6 S^2 D + 8 S^3) (raw QK^T 2 S^2 D + mask MLP 8 S^3 + QK^T and
    PV 4 S^2 D)."""

    def __init__(self, be):
        self.be, self.records, self.bwd = be, [], []
        self.orig = {n: getattr(be, n) for n in ("attn_fwd", "attn16_fwd", "attn16_bwd")}

    def __enter__(self):
        def wrap(name, dims):
            fn = self.orig[name]

            def timed(*a):
                B, S, H, hd = dims(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*a)
                e1.record()
                self.records.append((float(B) * (6.0 * S * S * H * hd + 8.0 * S ** 3), e0, e1, (name, S, H, hd)))
            return timed
        self.be.attn_fwd = wrap("attn_fwd", lambda a: (a[15], a[16], a[18], a[19]))          # (..., B, Sq, Skv, H, hd)
        self.be.attn16_fwd = wrap("attn16_fwd", lambda a: (a[16], a[17], a[18], a[19]))      # (..., B, S, H, hd)
        bwd_fn = self.orig["attn16_bwd"]

        def timed_bwd(*a):                     # the flash-style bf16 backward pair: 14 S^2 D FLOP per image (7 products, two recomputed)
            B, S, H, hd = a[13], a[14], a[15], a[16]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bwd_fn(*a)
            e1.record()
            self.bwd.append((float(B) * 14.0 * S * S * H * hd, e0, e1, (S, H, hd)))
        self.be.attn16_bwd = timed_bwd
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.be, n, f)
        return False

    def summary(self):
        torch.cuda.synchronize()
        if not self.records:
            return None
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        big = max(self.records, key=lambda r: r[0])
        same = [r for r in self.records if r[3] == big[3]]
        out = {"flops": flops, "ms": ms, "n": len(self.records), "largest": big[3],
               "largest_us": 1e3 * sum(r[1].elapsed_time(r[2]) for r in same) / len(same),
               "largest_tflops": big[0] / (1e-3 * sum(r[1].elapsed_time(r[2]) for r in same) / len(same)) / 1e12}
        if self.bwd:
            bb = max(self.bwd, key=lambda r: r[0])
            sb = [r for r in self.bwd if r[3] == bb[3]]
            us = 1e3 * sum(r[1].elapsed_time(r[2]) for r in sb) / len(sb)
            out["bwd"] = {"ms": sum(r[1].elapsed_time(r[2]) for r in self.bwd), "n": len(self.bwd), "largest": bb[3],
                          "largest_us": us, "largest_tflops": bb[0] / (us * 1e-6) / 1e12}
        return out


# kernel-name prefixes of the GEMM families in the rocprofv3 summaries (tests/test_host_logic_cpu.py checks that the
# committed summaries still have rows under them: a renamed kernel or namespace silently turned `traffic` into null)
GEMM_PMC_PREFIX = {"fp32": "calm_gemm_detail::gemm_f32", "bf16": "calm_gemm_detail::gemm_bf16"}
ATTN_PMC_PREFIX = {"fp32": "attn_fwd_kernel", "bf16": "attn16_fwd"}       # attn16_fwd2_kernel (pipelined) and attn16_fwd_kernel


def pmc_rows(kernel_prefix):
    """(rows, source, stale) of the newest committed rocprofv3 PMC summary for kernels starting with `kernel_prefix`.
    The summary carries a stamp (scripts/pmc_summary.py: sha256 of the kernel sources it was measured with); `stale` is
    True when a source of calm-vit-dte_amd/csrc has changed since — the counters are then not reported."""
    import csv
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.csv")))       # names sort by round
    for f in reversed(files):                              # newest summary that has this kernel family
        # (rocprofv3 leaves names with __bf16 template arguments mangled: _ZN16calm_gemm_detail17gemm_bf16w_kernelIDF16b...)
        parts = kernel_prefix.split("::")
        rows = [r for r in csv.DictReader(open(f)) if r["kernel"].startswith(kernel_prefix) or
                (r["kernel"].startswith("_ZN") and all(x in r["kernel"] for x in parts))]
        if not rows:
            continue
        stale = True                                       # a summary without a stamp cannot be vouched for
        if os.path.exists(f + ".stamp.json"):
            stamp = json.load(open(f + ".stamp.json"))["sources_sha256"]
            now = {os.path.basename(x): hashlib.sha256(open(x, "rb").read()).hexdigest()
                   for x in glob.glob(os.path.join(ROOT, "calm-vit-dte_amd", "csrc", "*"))}
            stale = any(now.get(k) != v for k, v in stamp.items()) or set(now) != set(stamp)
        return rows, os.path.relpath(f, ROOT), stale
    return [], None, None


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*pmc_summary.csv: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes,
    scripts/gpu_pmc.sh) — the counters cannot be read from inside this process."""
    rows, source, stale = pmc_rows(kernel_prefix)
    n = rd = wr = 0.0
    for r in rows:
        n += float(r["dispatches"])
        rd += float(r["hbm_read_bytes(2xFETCH)"])
        wr += float(r["hbm_write_bytes"])
    if not n or stale:                       # no summary / kernel sources changed since it was taken: report nothing
        return None
    return {"hbm_bytes_per_launch": round((rd + wr) / n), "read": round(rd / n), "write": round(wr / n),
            "source": source, "stale": stale}


def host_cpu_info():
    """(cpu model, physical cores inside this process's affinity mask, logical CPUs in the mask) from /proc/cpuinfo."""
    aff = sorted(os.sched_getaffinity(0))
    model, cores, cur = "unknown", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (t.strip() for t in line.split(":", 1))
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in aff:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    model = cur.get("model name", model)
                cur = {}
        if cur and int(cur.get("processor", -1)) in aff:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
            model = cur.get("model name", model)
    except OSError:
        pass
    return model, max(1, len(cores) or len(aff)), len(aff)


def cpu_quota():
    """CPUs this container may actually use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: the
    affinity mask of a GPU box lists every CPU of the host, but the box's share is a quota (16 for one GPU), and more
    threads than that are time-sliced — 128 threads measured 8x SLOWER than 16 on such a box."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, int(round(q / per)))
    except (OSError, ValueError):
        return None


def cpu_baseline(wl, budget_s=30.0):
    """CPU oracle (port of the reference path) fwd+bwd+AdamW on a bounded sample of the workload, as BASELINE.md
    section 3 prescribes: fp32, one thread per PHYSICAL core of the affinity mask (never more: oversubscription cost 20x
    in the survey), bs = 8, 2 warm-up + >= 5 timed steps, median."""
    from oracle import calm_oracle as O
    import calm_vit_dte_amd as calm
    W = calm.synthetic_weights
    cpu_model, physical, logical = host_cpu_info()
    quota = cpu_quota()
    threads = physical if quota is None else max(1, min(physical, quota))
    torch.set_num_threads(threads)
    cfg = O.ViTConfig(force_reduce=False, generate=False, **wl["kw"])
    P = {k: torch.from_numpy(v) for k, v in W.make_params(O.vit_param_shapes(cfg), WEIGHT_SEED).items()}
    leaves = []
    for k in P:
        if not O.is_buffer(k):
            P[k].requires_grad_(True)
            leaves.append(P[k])
    opt = torch.optim.AdamW(leaves, lr=3.1e-3, weight_decay=0.02, betas=(0.9, 0.98))
    bs = 8 if cfg.seq_length >= 128 else 32
    x, y = synthetic_batch(bs, cfg.seq_length, cfg.out_features, 0, "cpu")

    def step():
        out, _ = O.vit_forward(P, cfg, x, True)
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 1.0)
        opt.step()
        opt.zero_grad()

    step()
    step()                                           # 2 warm-up steps (timing only)
    times, t_start = [], time.perf_counter()
    while len(times) < 5 or (len(times) < 16 and (time.perf_counter() - t_start) < budget_s):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(bs / med, 3), "unit": "images/sec", "cores": threads, "physical_cores": physical,
            "affinity": logical, "cpu_quota": quota, "cpu_model": cpu_model, "kind": "port",
            "sample": f"oracle fwd+bwd+AdamW fp32, bs={bs}, {len(times)} timed steps after 2 warm-up, median; threads = "
                      f"min(physical cores in the affinity mask = {physical}, cgroup CPU quota = {quota})"}


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: BEFORE anything touches the GPU, start N fresh ranks
    under torch.distributed.run (env:// rendezvous on 127.0.0.1, one process per GPU), relay rank 0's JSON line and
    return the launcher's exit status.  (Never re-exec a process that has initialised the GPU.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="small224", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3", "fp8"],
                    help="matrix pipe of the GEMMs (tensors stay fp32): exact fp32 MFMA (default, config #2), "
                         "bf16 operands (autocast arithmetic, configs #3-5), the fp32-accurate bf16x3 split, or fp8: "
                         "the bf16 pipeline with fp8 Linear forward / input-gradient GEMMs (config #5)")
    ap.add_argument("--prof-steps", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="capture the whole step (with the RCCL all-reduces when N > 1) into a hipGraph")
    ap.add_argument("--torch-optim", action="store_true",
                    help="clip_grad_norm_ + torch.optim.AdamW(fused) instead of the library's 3-launch optimizer-side "
                         "step (calm_optim_step, which also folds in the spectral-norm gradient correction)")
    ap.add_argument("--autocast", action="store_true",
                    help="the reference trainer's call pattern (cls:84-95): forward under torch.autocast(bfloat16), "
                         "loss scaled by a torch GradScaler — the GEMMs then run on the bf16 pipe as with --precision bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the Base-224 autocast measurement that the default (small224 fp32) run attaches as 'secondary'")
    ap.add_argument("--gemm-report", default="", help="write a per-shape GEMM table (csv) from the profiled steps")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import calm_vit_dte_amd as calm
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")

    rank, local_rank, world = trainer.init_distributed(use_gpu=True)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev_index = int(os.environ.get("CALM_LOCAL_DEVICE", local_rank))   # rehearsal: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    def measure(workload, precision, autocast, steps, warmup, prof_steps, gemm_report=""):
        """Build the workload's model, time `steps` training steps after `warmup`, then profile `prof_steps` more."""
        wl = WORKLOADS[workload]
        batch = (args.batch if workload == args.workload else 0) or wl["batch"]
        S, classes = wl["kw"]["seq_length"], wl["kw"]["out_features"]
        # under autocast the region selects the bf16 pipeline itself; a global 'fp8' adds the fp8 Linear products to it
        calm.backend.set_matmul_precision(("fp8" if precision == "fp8" else "fp32") if autocast else precision)
        selfcheck = None
        if autocast or precision in ("bf16", "fp8"):
            # canary for the default bf16 GEMM family on THIS box (un-timed): a box on which the
            # pipelined family disagrees with the 256x128 one runs the bench on the latter and says so in the line
            sc = calm.backend.get_backend().selfcheck_bf16_gemm(on_mismatch="fallback")
            selfcheck = {"ok": sc["ok"], "n_beyond_one_ulp": sum(c["n_bad"] for c in sc["cases"]),
                         "n_one_ulp": sum(c["n_diff"] for c in sc["cases"]),
                         "max_diff": max(c["max_diff"] for c in sc["cases"]),
                         "family_used": "pipelined persistent (gemm_bf16p)" if sc["ok"] else "256x128 / 128-row (fallback)"}
            torch.cuda.empty_cache()
        model = build_model(calm, wl["kw"], device).train()
        trainer.sync_module_states(model)
        x, y = synthetic_batch(batch, S, classes, seed=rank, device=device)     # resident in HBM before timing
        if args.graph:
            opt = trainer.make_optimizer(model, capturable=True) if args.torch_optim else trainer.FusedClipAdamW(model)
            if autocast and precision != "fp8":
                precision = "bf16"
            # N > 1: the bucketed RCCL all-reduces are captured with the step (round 4: capture-compatible collectives,
            # trainer.BucketedGradReducer._launch; verified in a world of one, tests/test_trainer_gpu.py)
            reducer = trainer.BucketedGradReducer(model) if world > 1 else None
            step = trainer.GraphedTrainStep(model, opt, x, y, scaler=torch.amp.GradScaler("cuda") if autocast else None,
                                            autocast_dtype=torch.bfloat16 if autocast else None, reducer=reducer)
            prof_steps = 0                                 # events cannot be recorded inside a replayed graph
        else:
            opt = trainer.make_optimizer(model) if args.torch_optim else trainer.FusedClipAdamW(model)
            reducer = trainer.BucketedGradReducer(model) if world > 1 else None
            if autocast:
                if precision != "fp8":
                    precision = "bf16"                      # what the autocast region selects; labels the JSON line
                step = trainer.TrainStep(model, opt, reducer, scaler=torch.amp.GradScaler("cuda"),
                                         autocast_dtype=torch.bfloat16)
            else:
                step = trainer.TrainStep(model, opt, reducer)

        def sync():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            step(x, y)
        sync()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(steps):
            h0 = time.perf_counter()
            loss, _ = step(x, y)
            host += time.perf_counter() - h0              # host time to ENQUEUE the step (no synchronisation inside)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res = {"workload": workload, "precision": precision, "batch": batch, "S": S, "wl": wl,
               "ms_per_step": 1e3 * dt / steps, "value": world * batch * steps / dt, "loss": float(loss),
               "host_enqueue_ms": 1e3 * host / steps, "selfcheck": selfcheck,
               "roofline": None, "attention": None}

        # dominant-kernel roofline: HIP events around every calm_gemm launch of prof_steps extra steps
        if prof_steps > 0:
            # every rank runs the profiled steps (they contain the gradient all-reduce); only rank 0 records
            prof = GemmProfiler(calm.backend.get_backend()) if rank == 0 else None
            aprof = AttentionProfiler(calm.backend.get_backend()) if rank == 0 else None
            if prof is not None:
                prof.__enter__()
                aprof.__enter__()
            for _ in range(prof_steps):
                step(x, y)
            if prof is not None:
                aprof.__exit__()
                prof.__exit__()
                flops, ms, n = prof.summary()
                if gemm_report:
                    prof.report(gemm_report, prof_steps)
                achieved = flops / (ms * 1e-3) / 1e12
                _, peak, klabel = PRECISION_INFO[precision]
                roofline = {"bound": "mfma", "kernel": klabel,
                            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                            "frac": round(achieved / peak, 4),
                            "traffic": None,
                            "launches_per_step": n // prof_steps, "avg_launch_us": round(1e3 * ms / n, 2),
                            "gemm_ms_per_step": round(ms / prof_steps, 2),
                            "algorithmic_gflop_per_step": round(flops / prof_steps / 1e9, 1)}
                # HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate passes) from the committed summary
                detail = pmc_traffic(GEMM_PMC_PREFIX["fp32" if precision == "fp32" else "bf16"])
                if detail is not None:
                    roofline["traffic"] = detail["hbm_bytes_per_launch"]
                    roofline["traffic_detail"] = detail
                res["roofline"] = roofline
                # the axial-attention kernel (the other half of BASELINE.json's metric): live launch times of the fused
                # forward + the MFMA utilisation / HBM rate of the committed PMC summary (refused when stale)
                a = aprof.summary()
                if a is not None:
                    name, aS, aH, ahd = a["largest"]
                    kname = "attn16_fwd2_kernel / attn16_fwd_kernel" if name == "attn16_fwd" else "attn_fwd_kernel"
                    rows, source, stale = pmc_rows(ATTN_PMC_PREFIX["bf16" if name == "attn16_fwd" else "fp32"])
                    # the instantiation of the largest shape (attn16: <key-tile pairs NP, padded head dim>), else the busiest row
                    tag = f"<{(aS + 31) // 32}, {(ahd + 31) // 32 * 32}" if name == "attn16_fwd" else None
                    cand = [r for r in rows if tag and tag in r["kernel"]] or rows
                    best = max(cand, key=lambda r: float(r["gui_active_sum"])) if rows and not stale else None
                    res["attention"] = {
                        "kernel": f"{kname} (fused latent-mask attention forward, "
                                  f"{'v_mfma_f32_16x16x32_bf16' if name == 'attn16_fwd' else 'v_mfma_f32_16x16x4_f32'})",
                        "bound": "mfma", "launches_per_step": a["n"] // prof_steps,
                        "ms_per_step": round(a["ms"] / prof_steps, 3),
                        "achieved": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2), "peak": round(peak, 1),
                        "unit": "TFLOP/s", "frac": round(a["flops"] / (a["ms"] * 1e-3) / 1e12 / peak, 4),
                        "largest_shape": {"S": aS, "H": aH, "hd": ahd, "avg_launch_us": round(a["largest_us"], 1),
                                          "tflops": round(a["largest_tflops"], 2)},
                        "mfma_util_pmc": round(float(best["mfma_util"]), 4) if best else None,
                        "hbm_GBps_pmc": round(float(best["hbm_GBps"]), 1) if best else None,
                        "pmc_kernel": best["kernel"] if best else None, "pmc_source": source, "pmc_stale": stale}
                    if "bwd" in a:             # the backward pair of the same kernel family (bf16 pipeline), live timing
                        bw = a["bwd"]
                        res["attention"]["backward"] = {
                            "kernel": "attn16_bwd2_kernel query side + key side / attn16_bwd_q_kernel + attn16_bwd_kv_kernel",
                            "launches_per_step": bw["n"] // prof_steps, "ms_per_step": round(bw["ms"] / prof_steps, 3),
                            "largest_shape": {"S": bw["largest"][0], "H": bw["largest"][1], "hd": bw["largest"][2],
                                              "avg_launch_us": round(bw["largest_us"], 1),
                                              "tflops": round(bw["largest_tflops"], 2)}}
        if world > 1:
            dist.barrier()
        res["hbm_peak_gib"] = round(torch.cuda.max_memory_allocated(device) / 2**30, 1)
        if isinstance(opt, trainer.FusedClipAdamW):
            opt.close()
        del step, opt, model
        torch.cuda.empty_cache()
        return res

    main_res = measure(args.workload, args.precision, args.autocast, args.steps, args.warmup, args.prof_steps, args.gemm_report)
    args.precision = main_res["precision"]
    wl, batch, S = main_res["wl"], main_res["batch"], main_res["S"]
    value, ms_per_step, loss = main_res["value"], main_res["ms_per_step"], main_res["loss"]
    roofline, attention = main_res["roofline"], main_res["attention"]
    # the reference's own configuration (BASELINE configs[2]: Base-224 under autocast(bfloat16) + GradScaler, cls:84-95) in
    # the same run, so that the driver's record carries it beside the fp32 headline: >= 10 timed steps
    secondary = None
    if world == 1 and not args.no_secondary and args.workload == "small224" and not args.autocast and not args.graph:
        torch.cuda.reset_peak_memory_stats(device)
        r2 = measure("base224", "bf16", True, 10, 3, 1)
        secondary = {"workload": f"CALM-ViT base224 cls (the reference's model), 224x224x3 synthetic, bs={r2['batch']}/GPU, "
                                 "torch.autocast(bfloat16) + GradScaler, fwd+loss+bwd+clip+AdamW",
                     "metric": "training images/sec (224^2, bs=256/GPU)", "value": round(r2["value"], 2), "unit": "images/sec",
                     "steps": 10, "warmup": 3, "ms_per_step": round(r2["ms_per_step"], 3), "dtype": PRECISION_INFO["bf16"][0],
                     "model_tflops": round(r2["value"] * r2["wl"]["gflop_img"] / 1e3, 2), "hbm_peak_gib": r2["hbm_peak_gib"],
                     "host_enqueue_ms_per_step": round(r2["host_enqueue_ms"], 2), "gemm_family_selfcheck": r2["selfcheck"],
                     "roofline": r2["roofline"], "attention": r2["attention"]}

    if rank == 0:
        out = {
            "metric": "training images/sec (224^2, bs=256/GPU)", "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": PRECISION_INFO[args.precision][0],
            "data": "synthetic",
            "config": {"workload": f"CALM-ViT {args.workload} cls, {S}x{S}x3 synthetic, bs={batch}/GPU, "
                                   f"{args.precision} matmuls, fwd+loss+bwd+clip+AdamW; batch resident in HBM when the "
                                   "timed region starts (the trainer's H2D copy of the batch is excluded)",
                       "global_batch": world * batch,
                       "parallelism": f"dp{world}", "world": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": dist.get_backend() if dist.is_initialized() else "none (single process)",
                       "loss": loss, "hipgraph": bool(args.graph),
                       "optimizer": "torch clip_grad_norm_ + AdamW" if args.torch_optim
                       else "calm_optim_step (norm + clip + AdamW + spectral-norm grad correction, 3 launches)"},
            "model_tflops": round(value * wl["gflop_img"] / 1e3, 2),
            "hbm_peak_gib": main_res["hbm_peak_gib"],
            "host_enqueue_ms_per_step": round(main_res["host_enqueue_ms"], 2),
            "gemm_family_selfcheck": main_res["selfcheck"],
            "roofline": roofline,
            "attention": attention if roofline is not None else None,
        }
        if secondary is not None:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        import gc
        gc.collect()                      # captured graphs (--graph) hold RCCL kernels: gone before the communicator
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
