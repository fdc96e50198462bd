"""Tensor-level front of the C-ABI: turns torch tensors into raw device pointers and enqueues the
HIP kernels of libcalmvit_hip.so on torch's current HIP stream.  PyTorch is only the allocator
and stream provider here.

`get_backend()` returns the one product backend (HipBackend) and raises if the library cannot be
loaded; HipBackend rejects non-CUDA tensors.  `use_backend()` exists so that tests/ can check the
host-side autograd plumbing against a torch emulation of the C-ABI kept under tests/ — nothing
in the package ever installs another backend.
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import ACT_GELU, ACT_GELU_BWD, ACT_NONE  # noqa: F401  (re-exported)

SN_EPS = 1e-12

# calm_gemm_args.dtype (include/calm_vit.h): which matrix pipe the GEMMs use.  Tensors are fp32 in every mode.
GEMM_WORKSPACE = os.environ.get("CALM_GEMM_WORKSPACE", "1") != "0"      # A/B switch: 0 = split launches always use atomics
BF16_STORAGE = os.environ.get("CALM_BF16_STORAGE", "1") != "0"          # A/B switch: 0 = fp32 tensors in bf16 mode too
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp8": 1}     # calm_gemm_args.dtype; "fp8" = the bf16 pipeline with the
                                                                # forward / input-gradient Linear GEMMs on fp8 operands
FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)
_precision = "fp32"


def set_matmul_precision(name):
    """'fp32'   exact fp32 MFMA (default; BASELINE config #2, parity 1e-3 rel fp32 with margin 1e3)
    'bf16'   bf16 operands, fp32 accumulate (autocast(bfloat16) arithmetic; BASELINE configs #3-#5)
    'bf16x3' bf16 hi/lo split, 3 MFMA passes: fp32-level results (~1e-5) on the bf16 matrix pipe
    'fp8'    the bf16 pipeline with the forward and input-gradient products of the large Linear layers on per-tensor
             scaled OCP fp8 operands (e4m3 activations / weights, e5m2 gradients; BASELINE config #5); weight gradients,
             attention and everything else as in 'bf16'"""
    global _precision
    if name not in PRECISIONS:
        raise ValueError(f"unknown matmul precision {name!r}; choose from {sorted(PRECISIONS)}")
    _precision = name


def get_matmul_precision():
    return _precision


def effective_precision():
    """The pipe a GEMM issued now runs on.  Inside `torch.autocast("cuda", dtype=torch.bfloat16)` — how the reference
    trainer calls the model (distributed_trainer_cls.py:84-85) — Linear/matmul operands are rounded to bf16 with fp32
    accumulation, i.e. the 'bf16' mode; activations and parameters stay fp32 tensors.  Outside autocast: the mode
    chosen with set_matmul_precision."""
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt != torch.bfloat16:
            raise NotImplementedError(f"autocast dtype {dt} is not supported on this path (the reference uses bfloat16)")
        return "bf16"
    return "bf16" if _precision == "fp8" else _precision


def fp8_linears():
    """True when the Linear layers' forward / input-gradient GEMMs are to run on fp8 operands (set_matmul_precision('fp8');
    also inside autocast(bfloat16), which then selects the bf16 pipeline around them)."""
    return _precision == "fp8" and BF16_STORAGE


def _ptr(t, allow_none=False, bf16_ok=False, fp8_ok=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("required tensor is None")
    if not t.is_cuda:
        raise RuntimeError("CALM-ViT ops run only on the MI355X HIP path: got a CPU tensor "
                           "(there is no CPU fallback; move the model and inputs to 'cuda')")
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16) and not (fp8_ok and t.dtype in FP8_DTYPES):
        raise TypeError(f"fp32 tensor expected, got {t.dtype}")
    return t.data_ptr()


def _ptr16(t):
    if t is None or not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise TypeError("contiguous bf16 CUDA tensor expected")
    return t.data_ptr()


_ST_OF = {torch.bfloat16: _lib.ST_BF16, torch.float8_e4m3fn: _lib.ST_FP8_E4M3, torch.float8_e5m2: _lib.ST_FP8_E5M2}


def _st(t):
    """CALM_ST_* storage type of a tensor argument (None -> fp32)."""
    if t is None:
        return _lib.ST_F32
    return _ST_OF.get(t.dtype, _lib.ST_F32)


def bf16_pipeline():
    """True when activations that only GEMMs consume are to be kept as bf16 tensors: the bf16 matrix pipe is selected
    (autocast(bfloat16) or set_matmul_precision('bf16')) and the pipeline is not switched off (CALM_BF16_STORAGE=0:
    fp32 tensors, operands rounded while staged — the A/B switch)."""
    return BF16_STORAGE and effective_precision() == "bf16"


def act_dtype(last_dim):
    """Storage type for a GEMM-only activation whose contiguous extent is `last_dim` (bf16 rows are staged as 16-byte
    vectors of 8 elements)."""
    return torch.bfloat16 if bf16_pipeline() and last_dim % 8 == 0 else torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  The raw accessor (what torch's own generated code
    uses) costs 0.3 us; `torch.cuda.current_stream().cuda_stream` builds a Stream object per call — 9 us x 1 500 launches
    per step = 3.5 ms of the 58 ms a Base-224 step takes the host to issue (scripts/host_profile.py)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class OptimPlan:
    """Device table of calm_optim_tensor records for calm_optim_step; only the gradient pointers change per step.

    The pointers travel through a RING of pinned staging buffers, each guarded by an event recorded behind its
    host-to-device copy: an asynchronous copy from pinned memory reads the host bytes when the DMA runs, not when it is
    enqueued, and the host runs steps ahead of the GPU — a single buffer would be overwritten with the next step's
    pointers before the copy of this step's has executed.  A step whose gradients sit at the addresses of the previous
    upload (gradients kept in the all-reduce buckets, `BucketedGradReducer`) uploads nothing."""
    RING = 4

    def __init__(self, be, records):
        chunk = int(be.lib.calm_optim_chunk_elems())
        self.n = len(records)
        rec = np.zeros(self.n, dtype=np.dtype(_lib.OptimTensor))
        chunk_tensor = []
        for i, r in enumerate(records):
            p = r["param"]
            if not p.is_contiguous():
                raise TypeError("optimizer-side step expects contiguous parameters")
            e = rec[i]
            e["param"], e["exp_avg"], e["exp_avg_sq"] = _ptr(p), _ptr(r["exp_avg"]), _ptr(r["exp_avg_sq"])
            if r["exp_avg"].shape != p.shape or r["exp_avg_sq"].shape != p.shape:
                raise ValueError("optimizer state shape mismatch")
            e["numel"] = p.numel()
            if r["sn"] is not None:
                u, v, sigma, rows, cols = r["sn"]
                if rows * cols != p.numel() or p.numel() >= 2 ** 31:      # the kernels index (row, col) in 32 bits
                    raise ValueError("spectral-norm layer too large for the optimizer-side step")
                e["sn_u"], e["sn_v"], e["sn_sigma"], e["rows"], e["cols"] = _ptr(u), _ptr(v), _ptr(sigma), rows, cols
            e["chunk0"] = len(chunk_tensor)
            chunk_tensor += [i] * ((p.numel() + chunk - 1) // chunk)
        dev = records[0]["param"].device
        self.rec = rec
        self.n_chunks = len(chunk_tensor)
        self.chunk_dev = torch.tensor(chunk_tensor, dtype=torch.int32, device=dev)
        self.hosts = [torch.empty(rec.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.graph_host = torch.empty(rec.nbytes, dtype=torch.uint8).pin_memory()   # the table a captured step copies from
        self.events = [None] * self.RING
        self.slot = 0
        self.uploaded = None                    # gradient pointers of the table now (being) copied to table_dev
        self.table_dev = torch.empty(rec.nbytes, dtype=torch.uint8, device=dev)
        self.scratch = torch.empty(6 * self.n + 4 + 3 * self.n_chunks, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # advanced by the device, not on skipped steps
        self.param_ptrs = rec["param"].copy()

    def upload(self, grad_ptrs):
        """Stage this step's gradient pointers (numpy uint64 array) and enqueue their copy on the current stream."""
        if self.uploaded is not None and np.array_equal(self.uploaded, grad_ptrs):
            return
        if torch.cuda.is_current_stream_capturing():
            # hipGraph capture (GraphedTrainStep): the gradients live in the graph's private pool, so these addresses are
            # the ones every replay sees; the copy becomes a memcpy node out of a pinned buffer of its own that is never
            # rewritten (no event bookkeeping: a captured event cannot be synchronised on)
            self.rec["grad"] = grad_ptrs
            self.graph_host.numpy()[:] = self.rec.view(np.uint8).reshape(-1)       # (pinned at construction: no allocation here)
            self.table_dev.copy_(self.graph_host, non_blocking=True)
            self.uploaded = grad_ptrs.copy()
            return
        k = self.slot
        if self.events[k] is not None:
            self.events[k].synchronize()        # the copy out of this staging buffer (RING steps ago) has executed
        self.rec["grad"] = grad_ptrs
        self.hosts[k].numpy()[:] = self.rec.view(np.uint8).reshape(-1)
        self.table_dev.copy_(self.hosts[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        self.slot = (k + 1) % self.RING
        self.uploaded = grad_ptrs.copy()


class SnPlan:
    """Device-resident plan for the batched spectral-norm power iteration."""

    def __init__(self, info, blob_dev, scratch, key):
        self.info = info
        self.blob_dev = blob_dev
        self.scratch = scratch
        self.key = key


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._scratch_bufs = {}
        self._scratch_need = {}
        self.upcast_launches = 0        # calm_gemm launches re-run on fp32 copies (CALM_E_LAYOUT with bf16 tensors)

    def _partials(self, op, rows, cols, device):
        """Device scratch for the fixed-order cross-workgroup reduction of one call (calm_reduce_scratch_floats).  One
        buffer per (device, stream), reused by every call on that stream: the launches of one stream run in order, so a
        call's rows of partials have been consumed by its own reduction pass before the next call writes any."""
        need = self._scratch_need.get((op, rows, cols))
        if need is None:                                         # (a step has ~170 such calls over ~20 distinct shapes)
            need = self._scratch_need[(op, rows, cols)] = int(self.lib.calm_reduce_scratch_floats(op, rows, cols))
        key = (device.index, _stream())
        buf = self._scratch_bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=device)
            self._scratch_bufs[key] = buf
        return buf.data_ptr()

    # ---- GEMM -------------------------------------------------------------------------
    GEMM_OPT_PIPE, GEMM_OPT_PIPE32, GEMM_OPT_DETERMINISTIC = 0, 1, 2

    def gemm_set_option(self, option, value):
        """calm_gemm_set_option (ABI v6; v7: GEMM_OPT_DETERMINISTIC = k-split launches through the workspace and a
        fixed-order reduction instead of fp32 atomics); returns the previous value."""
        prev = self.lib.calm_gemm_set_option(int(option), int(value))
        if prev < 0:
            raise ValueError(f"calm_gemm_set_option({option}, {value}) -> {prev}")
        return prev

    def selfcheck_bf16_gemm(self, rows=57344, on_mismatch="raise"):
        """Canary for the default bf16 GEMM family: round 3 saw ONE box of the pool on which the
        pipelined persistent family (gemm_bf16p_kernel: 128 KiB of LDS, 256 VGPRs, LDS-DMA) returned a deterministic
        wrong GELU' input gradient at 57344 x 1344 x 1344 while the 256x128 family was right; the audit found no code
        defect.  Until the cause is pinned, a training job can ask the box itself: this runs the three
        operand-layout instantiations of the family at the bench's sizes — forward with fused bias + GELU (k-contiguous
        pair), the GELU' input gradient (k-contiguous x row-contiguous) and a per-image product (row-contiguous pair) —
        on both families and compares them (healthy boxes: identical up to a handful of results that sit on a bf16
        rounding boundary — one ulp; a fault: more than one ulp AND more than 2^-9 of the largest element).  on_mismatch: "raise" (RuntimeError with the pattern of the differing elements), or "fallback"
        (warn and switch this process to the 256x128 family — still the HIP path, 10-20 % slower GEMMs).
        Returns {"ok", "cases": [{name, n_diff, n_bad, max_diff, plan, first}]}.  ~0.8 GB of scratch tensors, a few ms."""
        import warnings
        dev = torch.device("cuda", torch.cuda.current_device())
        gen = torch.Generator(device=dev).manual_seed(20061)           # (on the device: 250 M normals take seconds on the host)
        mk = lambda *shape, scale=1.0: (torch.randn(*shape, generator=gen, device=dev) * scale).bfloat16()
        M, K, N = int(rows), 672, 1344
        x, w1, dy, w2 = mk(M, K), mk(N, K, scale=K ** -0.5), mk(M, N), mk(N, N, scale=N ** -0.5)
        bias = torch.randn(N, generator=gen, device=dev) * 0.1
        sigma = torch.tensor([1.3], device=dev)
        S, nb = 224, 64
        pa, pb = mk(nb, S, S), mk(nb, S, 3 * S, scale=S ** -0.5)
        hp = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        e16 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.bfloat16)
        cases = [
            ("forward bias+GELU (kk)", lambda out: ((x, w1, out, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0)),
                                                    dict(inv_scale=sigma, bias=bias, act=ACT_GELU, C_pre=hp, split_k=1)), (M, N)),
            ("GELU' input gradient (km)", lambda out: ((dy, w2, out, M, N, N, (N, 1, 0, 0), (1, N, 0, 0), (N, 0, 0)),
                                                       dict(inv_scale=sigma, act=ACT_GELU_BWD, aux=hp, split_k=1)), (M, N)),
            ("per-image product (mm)", lambda out: ((pa, pb, out, S, 3 * S, S, (1, S, S * S, 0), (1, 3 * S, S * 3 * S, 0),
                                                     (3 * S, S * 3 * S, 0)), dict(batch=(nb, 1), split_k=1)), (nb, S, 3 * S)),
        ]
        prev_prec = get_matmul_precision()
        set_matmul_precision("bf16")
        report, ok = [], True
        try:
            for name, build, shape in cases:
                got, alt = e16(*shape), e16(*shape)
                args, kw = build(got)
                plan = self.gemm_describe(*args, **kw)
                self.gemm(*args, **kw)
                prev = self.gemm_set_option(self.GEMM_OPT_PIPE, 0)
                try:
                    args2, kw2 = build(alt)
                    self.gemm(*args2, **kw2)
                finally:
                    self.gemm_set_option(self.GEMM_OPT_PIPE, prev)
                # the families accumulate k in different groupings, so a result that sits on a bf16 rounding boundary may
                # land on either neighbour (measured: ~700 of 77 M elements of the GELU case, one ulp each); a fault is
                # an element MORE than one bf16 ulp away AND off by more than 2^-9 of the largest element (the round-3
                # box: 1.2e-2 of it; one stale 16-byte operand chunk: ~2e-2)
                key = lambda t: torch.where(t.view(torch.int16) < 0, -(t.view(torch.int16).int() & 0x7FFF), t.view(torch.int16).int())
                ulps = (key(got) - key(alt)).abs()
                scale = float(alt.float().abs().max())
                err = (got.float() - alt.float()).abs()
                bad = (ulps > 1) & (err > 2.0 ** -9 * scale)
                n_diff, n_bad = int((ulps > 0).sum()), int(bad.sum())
                rec = {"name": name, "n_diff": n_diff, "n_bad": n_bad, "max_diff": float(err.max()) / max(scale, 1e-30),
                       "plan": plan, "first": []}
                if n_bad:
                    idx = bad.reshape(-1, shape[-1]).nonzero()[:8].tolist()
                    g2, a2 = got.reshape(-1, shape[-1]), alt.reshape(-1, shape[-1])
                    rec["first"] = [(r, c, float(g2[r, c]), float(a2[r, c])) for r, c in idx]
                    rec["rows"] = [int(bad.reshape(-1, shape[-1]).nonzero()[:, 0].min()), int(bad.reshape(-1, shape[-1]).nonzero()[:, 0].max())]
                    if plan["family"] == 3:
                        ok = False
                report.append(rec)
        finally:
            set_matmul_precision(prev_prec)
        res = {"ok": ok, "cases": report}
        if not ok:
            msg = ("the pipelined bf16 GEMM family disagrees with the 256x128 family on this box (see DESIGN.md section 2): " +
                   "; ".join(f"{r['name']}: {r['n_bad']} elements beyond one ulp (rows {r.get('rows')}), max {r['max_diff']:.2e} "
                             f"of the largest element, first (row, col, pipelined, 256x128) {r['first'][:3]}"
                             for r in report if r["n_bad"]))
            if on_mismatch == "fallback":
                self.gemm_set_option(self.GEMM_OPT_PIPE, 0)
                warnings.warn(msg + " -- continuing on the 256x128 family (calm_gemm_set_option(PIPE, 0))")
            else:
                raise RuntimeError(msg)
        return res

    def gemm_describe(self, *args, **kw):
        """calm_gemm_describe (ABI v7) for the launch `gemm(*args, **kw)` would make: a dict of the plan's fields
        (family, tile, tiles, k-slices, items, grid ...).  Nothing is enqueued."""
        g, _keep = self._gemm_args(*args, **kw)
        plan = _lib.GemmPlan()
        _lib.check(self.lib.calm_gemm_describe(C.byref(g), C.byref(plan)), "calm_gemm_describe")
        return {name: getattr(plan, name) for name, _ in _lib.GemmPlan._fields_}

    def gemm(self, A, B, Cout, M, N, K, a, b, c, batch=(1, 1), alpha=1.0, inv_scale=None, bias=None,
             col_scale=None, residual=None, r=(0, 0, 0), C_pre=None, aux=None, act=ACT_NONE,
             accumulate=False, reduce_batch=False, split_k=0, a_dq=None, b_dq=None):
        """a_dq / b_dq: device dequantisation factors of fp8 operands (quantize_fp8's state[1:2])."""
        g, (A, B, Cout) = self._gemm_args(A, B, Cout, M, N, K, a, b, c, batch, alpha, inv_scale, bias, col_scale, residual,
                                          r, C_pre, aux, act, accumulate, reduce_batch, split_k, a_dq, b_dq)
        ws = None
        if split_k != 1 and GEMM_WORKSPACE:
            # split launches with many k-slices per output combine them through a workspace instead of atomics; the
            # buffer comes from torch's caching allocator and is stream-ordered like every other tensor of the step
            need = self.lib.calm_gemm_workspace_bytes(C.byref(g))
            if need > 0:
                ws = torch.empty(need // 4, dtype=torch.float32, device=Cout.device)
                g.workspace, g.workspace_bytes = ws.data_ptr(), need
        rc = self.lib.calm_gemm(C.byref(g), _stream())
        fp8_operand = g.a_type >= _lib.ST_FP8_E4M3 or g.b_type >= _lib.ST_FP8_E4M3     # fp8 tensors have no fp32 re-run: report
        if rc == _lib.E_LAYOUT and not fp8_operand and (g.a_type or g.b_type or g.c_type or g.aux_type or g.r_type):
            # a bf16 tensor in a launch whose sizes / strides rule out 16-byte staging (the 10-class head of the fixture
            # models, a 36-token stage): rare and tiny — run it on fp32 copies through the generic kernels
            return self._gemm_upcast(A, B, Cout, M, N, K, a, b, c, batch, alpha, inv_scale, bias, col_scale, residual,
                                     r, C_pre, aux, act, accumulate, reduce_batch, split_k, g)
        _lib.check(rc, "calm_gemm")

    def _gemm_args(self, A, B, Cout, M, N, K, a, b, c, batch=(1, 1), alpha=1.0, inv_scale=None, bias=None,
                   col_scale=None, residual=None, r=(0, 0, 0), C_pre=None, aux=None, act=ACT_NONE,
                   accumulate=False, reduce_batch=False, split_k=0, a_dq=None, b_dq=None):
        """(calm_gemm_args, (A, B, C) base tensors) for a launch."""
        g = _lib.GemmArgs()
        # grouped form: A / B / Cout / inv_scale given as lists of separately allocated tensors, one per b0 entry
        groups = [len(t) for t in (A, B, Cout, inv_scale) if isinstance(t, (list, tuple))]
        if groups:
            n = groups[0]
            if any(k != n for k in groups) or batch != (n, 1):
                raise ValueError("grouped gemm: every list needs batch[0] entries and batch[1] must be 1")
            g.n_group = n
            for name, t in (("A_group", A), ("B_group", B), ("C_group", Cout), ("inv_scale_group", inv_scale)):
                if isinstance(t, (list, tuple)):
                    tab = getattr(g, name)
                    if name != "inv_scale_group" and len({ti.dtype for ti in t}) != 1:
                        raise TypeError("grouped gemm: the tensors of one operand must share a storage type")
                    for i, ti in enumerate(t):
                        tab[i] = _ptr(ti, True, bf16_ok=name != "inv_scale_group")
            A, B, Cout = (t[0] if isinstance(t, (list, tuple)) else t for t in (A, B, Cout))
            inv_scale = None
        g.A, g.B, g.C = _ptr(A, bf16_ok=True, fp8_ok=True), _ptr(B, bf16_ok=True, fp8_ok=True), _ptr(Cout, bf16_ok=True)
        g.a_dq, g.b_dq = _ptr(a_dq, True), _ptr(b_dq, True)
        g.a_type, g.b_type, g.c_type, g.aux_type, g.r_type = _st(A), _st(B), _st(Cout), _st(aux), _st(residual)
        if C_pre is not None and C_pre.dtype != Cout.dtype:
            raise TypeError("gemm: C_pre must have C's storage type")
        g.M, g.N, g.K = M, N, K
        g.batch0, g.batch1 = batch
        g.a_rs, g.a_cs, g.a_b0, g.a_b1 = a
        g.b_rs, g.b_cs, g.b_b0, g.b_b1 = b
        g.c_rs, g.c_b0, g.c_b1 = c
        g.alpha = alpha
        g.inv_scale = _ptr(inv_scale, True)
        g.bias = _ptr(bias, True)
        g.col_scale = _ptr(col_scale, True)
        g.residual = _ptr(residual, True, bf16_ok=True)
        g.r_rs, g.r_b0, g.r_b1 = r
        g.C_pre = _ptr(C_pre, True, bf16_ok=True)
        g.aux = _ptr(aux, True, bf16_ok=True)
        g.act = act
        g.accumulate = int(accumulate)
        g.reduce_batch = int(reduce_batch)
        g.split_k = split_k
        g.dtype = PRECISIONS[effective_precision()]
        return g, (A, B, Cout)

    def _gemm_upcast(self, A, B, Cout, M, N, K, a, b, c, batch, alpha, inv_scale, bias, col_scale, residual, r, C_pre,
                     aux, act, accumulate, reduce_batch, split_k, g):
        if g.n_group:
            raise RuntimeError("calm_gemm: grouped launch with bf16 tensors that cannot be staged (CALM_E_LAYOUT)")

        self.upcast_launches += 1

        def up(t):
            if t is None or t.dtype != torch.bfloat16:
                return t
            if not t.is_contiguous():
                raise RuntimeError("calm_gemm: non-contiguous bf16 tensor in a launch that cannot be vectorised")
            t32 = torch.empty(t.shape, dtype=torch.float32, device=t.device)
            _lib.check(self.lib.calm_cast_f32_one(t.data_ptr(), t32.data_ptr(), t.numel(), _stream()), "calm_cast_f32_one")
            return t32
        C32, P32 = up(Cout), up(C_pre)
        self.gemm(up(A), up(B), C32, M, N, K, a, b, c, batch=batch, alpha=alpha, inv_scale=inv_scale, bias=bias,
                  col_scale=col_scale, residual=up(residual), r=r, C_pre=P32, aux=up(aux), act=act, accumulate=accumulate,
                  reduce_batch=reduce_batch, split_k=split_k)
        if C32 is not Cout:
            self.cast_bf16(C32, Cout)
        if P32 is not C_pre:
            self.cast_bf16(P32, C_pre)

    # ---- device-side collate -------------------------------------------------------------
    def collate_mix(self, img_u8, flip, out, mode, lam, box, mean, std):
        """img_u8 [B,3,H,W] uint8, flip [B] uint8 or None, out [B,3,H,W] fp32; mode 0/1/2 = none/MixUp/CutMix."""
        if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or not img_u8.is_contiguous():
            raise TypeError("collate_mix expects a contiguous uint8 CUDA image batch")
        B, _, H, W = img_u8.shape
        cbox = (C.c_int32 * 4)(*box) if box is not None else None
        cm, cs = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        fp = flip.data_ptr() if flip is not None else None
        _lib.check(self.lib.calm_collate_mix(img_u8.data_ptr(), fp, _ptr(out), B, H, W, mode, float(lam), cbox, cm, cs,
                                             _stream()), "calm_collate_mix")

    def collate_crop_mix(self, img_u8, crop_yx, flip, out, mode, lam, box, mean, std, tokens=False):
        """img_u8 [B,3,Hs,Ws] uint8; crop_yx [B,2] int32 on the device (or None); out [B,3,H,W] or, tokens=True,
        [B,H,3W] fp32 (the row tokens of the first Block)."""
        if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or not img_u8.is_contiguous():
            raise TypeError("collate_crop_mix expects a contiguous uint8 CUDA image batch")
        B, _, Hs, Ws = img_u8.shape
        if tokens:
            H, W = out.shape[1], out.shape[2] // 3
        else:
            H, W = out.shape[2], out.shape[3]
        if crop_yx is not None and (crop_yx.dtype != torch.int32 or not crop_yx.is_cuda or not crop_yx.is_contiguous()):
            raise TypeError("collate_crop_mix: crop_yx must be a contiguous int32 CUDA tensor [B,2]")
        cbox = (C.c_int32 * 4)(*box) if box is not None else None
        cm, cs = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        _lib.check(self.lib.calm_collate_crop_mix(img_u8.data_ptr(), Hs, Ws, crop_yx.data_ptr() if crop_yx is not None else None,
                                                  flip.data_ptr() if flip is not None else None, _ptr(out), B, H, W,
                                                  int(tokens), mode, float(lam), cbox, cm, cs, _stream()),
                   "calm_collate_crop_mix")

    # ---- optimizer-side step ------------------------------------------------------------
    def optim_plan(self, records):
        """records: one dict per parameter — param, exp_avg, exp_avg_sq, sn (None or (u, v, sigma, rows, cols))."""
        return OptimPlan(self, records)

    def optim_step(self, plan, grads, hp, grad_scale, stats_out, lr_dev=None):
        """hp = (lr, beta1, beta2, eps, weight_decay, max_norm, step); stats_out[2] <- grad norm, found_inf.
        The step count used for the bias correction is plan.step_dev (device; not advanced on skipped steps).
        lr_dev: device scalar that overrides hp's lr (a captured step reads its learning rate from it)."""
        plan.upload(np.asarray([_ptr(g) for g in grads], dtype=np.uint64))
        h = _lib.OptimHparams(*hp)
        _lib.check(self.lib.calm_optim_step(plan.table_dev.data_ptr(), plan.n, plan.chunk_dev.data_ptr(), plan.n_chunks,
                                            _ptr(plan.scratch), C.byref(h), _ptr(grad_scale, True), _ptr(stats_out),
                                            plan.step_dev.data_ptr(), _ptr(lr_dev, True), _stream()), "calm_optim_step")

    # ---- LayerNorm --------------------------------------------------------------------
    def layernorm_fwd(self, x, w, y, mean, rstd, rows, D, eps):
        """y may be a bf16 tensor (bf16 pipeline: the output only feeds GEMMs)."""
        _lib.check(self.lib.calm_layernorm_fwd(_ptr(x), _ptr(w), _ptr(y, bf16_ok=True), _ptr(mean), _ptr(rstd), rows, D,
                                               eps, _st(y), _stream()), "calm_layernorm_fwd")

    def layernorm_bwd(self, dy, x, w, mean, rstd, dx, dw, rows, D, dx_add=None):
        _lib.check(self.lib.calm_layernorm_bwd(_ptr(dy, bf16_ok=True), _ptr(x), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dx),
                                               _ptr(dw), _ptr(dx_add, True), rows, D, _st(dy),
                                               self._partials(_lib.RED_LAYERNORM_BWD, rows, D, x.device), _stream()),
                   "calm_layernorm_bwd")

    # ---- bf16 weight copies --------------------------------------------------------------
    def cast_plan(self, pairs):
        """pairs: [(fp32 source tensor, bf16 destination tensor of the same numel)] -> plan for cast_run."""
        chunk = int(self.lib.calm_cast_chunk_elems())
        ent = np.zeros(len(pairs), dtype=np.dtype(_lib.CastEntry))
        chunk_entry = []
        for i, (src, dst) in enumerate(pairs):
            if dst.dtype != torch.bfloat16 or dst.numel() != src.numel() or not (src.is_contiguous() and dst.is_contiguous()):
                raise TypeError("cast_plan: contiguous fp32 source / bf16 destination of equal size expected")
            ent[i]["src"], ent[i]["dst"], ent[i]["numel"] = _ptr(src), _ptr(dst, bf16_ok=True), src.numel()
            ent[i]["chunk0"] = len(chunk_entry)
            chunk_entry += [i] * ((src.numel() + chunk - 1) // chunk)
        dev = pairs[0][0].device
        plan = SnPlan(None, torch.from_numpy(ent.view(np.uint8).reshape(-1).copy()).to(dev),
                      torch.tensor(chunk_entry, dtype=torch.int32, device=dev),
                      tuple(t.data_ptr() for pr in pairs for t in pr))
        plan.n_chunks = len(chunk_entry)
        return plan

    # ---- fp8 (BASELINE config #5) ------------------------------------------------------
    def quantize_fp8(self, x, q_dtype):
        """Per-tensor scaled fp8 copy of a contiguous fp32 / bf16 tensor: returns (q, dq) with dq a device scalar
        (amax / FP8_MAX) for calm_gemm's a_dq / b_dq."""
        if not x.is_contiguous():
            raise TypeError("quantize_fp8 expects a contiguous tensor")
        q = torch.empty(x.shape, dtype=q_dtype, device=x.device)
        state = torch.empty(2, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.calm_quantize_fp8(_ptr(x, bf16_ok=True), _st(x), x.numel(), q.data_ptr(), _st(q), _ptr(state),
                                              _stream()), "calm_quantize_fp8")
        return q, state[1:2]

    def transpose_u8(self, x):
        """[rows, cols] one-byte tensor -> its transpose [cols, rows] (contiguous)."""
        rows, cols = x.shape
        out = torch.empty(cols, rows, dtype=x.dtype, device=x.device)
        _lib.check(self.lib.calm_transpose_u8(x.data_ptr(), out.data_ptr(), rows, cols, _stream()), "calm_transpose_u8")
        return out

    def cast_bf16(self, src, dst):
        _lib.check(self.lib.calm_cast_bf16_one(_ptr(src), _ptr(dst, bf16_ok=True), src.numel(), _stream()),
                   "calm_cast_bf16_one")

    def cast_run(self, plan):
        _lib.check(self.lib.calm_cast_bf16(plan.blob_dev.data_ptr(), plan.scratch.data_ptr(), plan.n_chunks, _stream()),
                   "calm_cast_bf16")

    # ---- RoPE -------------------------------------------------------------------------
    def rope_fwd(self, content, xr, inv_freq, table, out, B, S, H, dc, dr):
        """content / xr / out: fp32 or bf16 tensors, independently."""
        _lib.check(self.lib.calm_rope_fwd(_ptr(content, True, bf16_ok=True), _ptr(xr, bf16_ok=True), _ptr(inv_freq),
                                          _ptr(table), _ptr(out, bf16_ok=True), B, S, H, dc, dr, _st(content), _st(xr),
                                          _st(out), _stream()), "calm_rope_fwd")

    def rope_bwd(self, d_out, xr, table, d_content, d_xr, d_inv_freq, B, S, H, dc, dr):
        _lib.check(self.lib.calm_rope_bwd(_ptr(d_out, bf16_ok=True), _ptr(xr, bf16_ok=True), _ptr(table),
                                          _ptr(d_content, True, bf16_ok=True), _ptr(d_xr, bf16_ok=True), _ptr(d_inv_freq),
                                          B, S, H, dc, dr, _st(d_out), _st(xr), _st(d_content), _st(d_xr),
                                          self._partials(_lib.RED_ROPE_BWD, B * S * H, dr, xr.device), _stream()),
                   "calm_rope_bwd")

    # ---- softmax ----------------------------------------------------------------------
    def softmax_fwd(self, x, rows, cols):
        _lib.check(self.lib.calm_softmax_fwd(_ptr(x), rows, cols, _stream()), "calm_softmax_fwd")

    def softmax_bwd(self, p, dp, rows, cols):
        _lib.check(self.lib.calm_softmax_bwd(_ptr(p), _ptr(dp), rows, cols, _stream()), "calm_softmax_bwd")

    def softmax_bwd_heads(self, p, dp, dm, B, H, Sq, cols):
        _lib.check(self.lib.calm_softmax_bwd_heads(_ptr(p), _ptr(dp), _ptr(dm), B, H, Sq, cols, _stream()),
                   "calm_softmax_bwd_heads")

    def sum_heads(self, dl, dm, B, H, per_head):
        _lib.check(self.lib.calm_sum_heads(_ptr(dl), _ptr(dm), B, H, per_head, _stream()), "calm_sum_heads")

    # ---- fused latent-mask attention ---------------------------------------------------
    def attn_fwd_supported(self, Sq, Skv, H, hd):
        return bool(self.lib.calm_attention_fwd_supported(Sq, Skv, H, hd))

    def attn_fwd(self, q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, P, B, Sq, Skv, H, hd):
        _lib.check(self.lib.calm_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(w1), _ptr(b1), _ptr(s1), _ptr(w2),
                                               _ptr(b2), _ptr(s2), _ptr(out), _ptr(R), _ptr(hp), _ptr(hg),
                                               _ptr(Mk), _ptr(P, True), B, Sq, Skv, H, hd, _stream()), "calm_attention_fwd")

    def attn_bwd_preferred(self, Sq, Skv, H, hd):
        return bool(self.lib.calm_attention_bwd_preferred(Sq, Skv, H, hd))

    def attn_bwd(self, q, k, v, dout, P, dS, dq, dk, dv, dM, B, Sq, Skv, H, hd):
        _lib.check(self.lib.calm_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(dout), _ptr(P), _ptr(dS), _ptr(dq),
                                               _ptr(dk), _ptr(dv), _ptr(dM), B, Sq, Skv, H, hd, _stream()),
                   "calm_attention_bwd")

    # ---- the same attention on the bf16 matrix pipe (bf16 pipeline) ----------------------
    def attn16_supported(self, S, H, hd):
        return bool(self.lib.calm_attention16_supported(S, H, hd))

    def attn16_fwd(self, q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd):
        h = lambda t: _ptr16(t)
        _lib.check(self.lib.calm_attention16_fwd(h(q), h(k), h(v), h(w1), _ptr(b1), _ptr(s1), h(w2), _ptr(b2), _ptr(s2),
                                                 h(out), h(R), h(hp), h(hg), h(Mk), h(MkT), _ptr(lse), B, S, H, hd,
                                                 _stream()), "calm_attention16_fwd")

    def attn16_bwd(self, q, k, v, out, dout, Mk, MkT, lse, delta, dq, dk, dv, dM, B, S, H, hd):
        h = lambda t: _ptr16(t)
        _lib.check(self.lib.calm_attention16_bwd(h(q), h(k), h(v), h(out), h(dout), h(Mk), h(MkT), _ptr(lse), _ptr(delta),
                                                 h(dq), h(dk), h(dv), h(dM), B, S, H, hd, _stream()),
                   "calm_attention16_bwd")

    # ---- latent -----------------------------------------------------------------------
    def latent_fwd(self, mv, noise, z, std, kl_sum, rows, mvh):
        _lib.check(self.lib.calm_latent_fwd(_ptr(mv), _ptr(noise, True), _ptr(z), _ptr(std), _ptr(kl_sum), rows,
                                            mvh, self._partials(_lib.RED_LATENT_FWD, rows, mvh, mv.device), _stream()),
                   "calm_latent_fwd")

    def latent_bwd(self, dz, d_kl_sum, mv, noise, std, dmv, rows, mvh):
        _lib.check(self.lib.calm_latent_bwd(_ptr(dz, True), _ptr(d_kl_sum, True), _ptr(mv), _ptr(noise, True),
                                            _ptr(std), _ptr(dmv), rows, mvh, _stream()), "calm_latent_bwd")

    # ---- spectral norm ----------------------------------------------------------------
    def sn_plan(self, layers):
        """layers: list of (w2d, u, v, sigma) tensors.  Returns an SnPlan bound to their addresses."""
        n = len(layers)
        arr = (_lib.SnLayer * n)()
        for i, (w, u, v, sg) in enumerate(layers):
            rows = w.shape[0]
            cols = w.numel() // rows
            arr[i].w, arr[i].u, arr[i].v, arr[i].sigma = _ptr(w), _ptr(u), _ptr(v), _ptr(sg)
            arr[i].rows, arr[i].cols = rows, cols
        info = _lib.SnPlanInfo()
        _lib.check(self.lib.calm_sn_plan(arr, n, None, C.byref(info)), "calm_sn_plan")
        blob = np.zeros(info.blob_bytes, dtype=np.uint8)
        _lib.check(self.lib.calm_sn_plan(arr, n, blob.ctypes.data, C.byref(info)), "calm_sn_plan")
        dev = layers[0][0].device
        blob_dev = torch.from_numpy(blob).to(dev)
        scratch = torch.empty(int(info.scratch_floats), dtype=torch.float32, device=dev)
        key = tuple(t.data_ptr() for layer in layers for t in layer)
        return SnPlan(info, blob_dev, scratch, key)

    def sn_power_iter(self, plan, training):
        _lib.check(self.lib.calm_sn_power_iter(plan.blob_dev.data_ptr(), C.byref(plan.info), int(training), SN_EPS,
                                               _ptr(plan.scratch), _stream()), "calm_sn_power_iter")

    def sn_weight_bwd(self, G, w, u, v, sigma, ls, dW, d_ls, rows, cols):
        scratch = torch.empty(rows + 2, dtype=torch.float32, device=G.device)
        _lib.check(self.lib.calm_sn_weight_bwd(_ptr(G), _ptr(w), _ptr(u), _ptr(v), _ptr(sigma), _ptr(ls, True),
                                               _ptr(dW), _ptr(d_ls, True), rows, cols, _ptr(scratch), _stream()),
                   "calm_sn_weight_bwd")

    # ---- tokenisation / conv ----------------------------------------------------------
    def image_to_rows(self, img, rows, B, S):
        _lib.check(self.lib.calm_image_to_rows(_ptr(img), _ptr(rows), B, S, _stream()), "calm_image_to_rows")

    def rows_to_image(self, rows, img, B, S):
        _lib.check(self.lib.calm_rows_to_image(_ptr(rows), _ptr(img), B, S, _stream()), "calm_rows_to_image")

    def grid_transpose(self, x, out, B, S):
        _lib.check(self.lib.calm_grid_transpose(_ptr(x), _ptr(out), B, S, _stream()), "calm_grid_transpose")

    def dwconv_fwd(self, x, w, inv_scale, bias, y, y_pre, act, B, S, Cch):
        _lib.check(self.lib.calm_dwconv3x3_fwd(_ptr(x), _ptr(w), _ptr(inv_scale, True), _ptr(bias, True), _ptr(y),
                                               _ptr(y_pre, True), act, B, S, Cch, _stream()), "calm_dwconv3x3_fwd")

    def dwconv_bwd(self, dz, x, w, inv_scale, dx, dw, db, B, S, Cch):
        _lib.check(self.lib.calm_dwconv3x3_bwd(_ptr(dz), _ptr(x), _ptr(w), _ptr(inv_scale, True), _ptr(dx),
                                               _ptr(dw), _ptr(db), B, S, Cch, _stream()), "calm_dwconv3x3_bwd")

    def cnn_fwd(self, x, w0, s0, b0, w2, s2, b2, w4, s4, b4, out, B, S, hidden, residual=True):
        """residual=False: the bare proj(x) (a caller of the reference invoking `block.proj` on its own)."""
        _lib.check(self.lib.calm_cnn_residual_fwd(_ptr(x), _ptr(w0), _ptr(s0), _ptr(b0), _ptr(w2), _ptr(s2),
                                                  _ptr(b2), _ptr(w4), _ptr(s4), _ptr(b4), _ptr(out), B, S, hidden,
                                                  int(residual), _stream()), "calm_cnn_residual_fwd")

    def cnn_bwd(self, dy, x, w0, s0, b0, w2, s2, b2, w4, s4, b4, dx, g0, gb0, g2, gb2, g4, gb4, B, S, hidden,
                residual=True):
        _lib.check(self.lib.calm_cnn_residual_bwd(_ptr(dy), _ptr(x), _ptr(w0), _ptr(s0), _ptr(b0), _ptr(w2),
                                                  _ptr(s2), _ptr(b2), _ptr(w4), _ptr(s4), _ptr(b4), _ptr(dx),
                                                  _ptr(g0), _ptr(gb0), _ptr(g2), _ptr(gb2), _ptr(g4), _ptr(gb4),
                                                  B, S, hidden, int(residual),
                                                  self._partials(_lib.RED_CNN_BWD, B, S, dy.device), _stream()),
                   "calm_cnn_residual_bwd")

    # ---- helpers ----------------------------------------------------------------------
    def add(self, a, b, out, n):
        _lib.check(self.lib.calm_add(_ptr(a), _ptr(b), _ptr(out), n, _stream()), "calm_add")

    def gelu_fwd(self, x, y, n):
        _lib.check(self.lib.calm_gelu_fwd(_ptr(x), _ptr(y), n, _stream()), "calm_gelu_fwd")

    def gelu_bwd(self, dy, z, dz, n):
        _lib.check(self.lib.calm_gelu_bwd(_ptr(dy), _ptr(z), _ptr(dz), n, _stream()), "calm_gelu_bwd")

    def colsum(self, x, out, rows, cols):
        _lib.check(self.lib.calm_colsum(_ptr(x, bf16_ok=True), _ptr(out), rows, cols, _st(x),
                                        self._partials(_lib.RED_COLSUM, rows, cols, x.device), _stream()), "calm_colsum")

    def row_scale(self, x, s, out, rows, cols):
        _lib.check(self.lib.calm_row_scale(_ptr(x), _ptr(s), _ptr(out, bf16_ok=True), rows, cols, _st(out), _stream()),
                   "calm_row_scale")

    def mean_seq_fwd(self, x, y, B, S, D):
        _lib.check(self.lib.calm_mean_seq_fwd(_ptr(x), _ptr(y), B, S, D, _stream()), "calm_mean_seq_fwd")

    def mean_seq_bwd(self, dy, dx, B, S, D):
        _lib.check(self.lib.calm_mean_seq_bwd(_ptr(dy), _ptr(dx), B, S, D, _stream()), "calm_mean_seq_bwd")


_backend = None


def get_backend():
    """The product backend.  Raises RuntimeError when libcalmvit_hip.so is not built/loadable."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


@contextlib.contextmanager
def use_backend(b):
    """TEST HOOK ONLY (tests/): run the host logic against another implementation of the C-ABI."""
    global _backend
    old = _backend
    _backend = b
    try:
        yield b
    finally:
        _backend = old
