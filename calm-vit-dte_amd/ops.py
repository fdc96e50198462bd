"""Autograd operators of the CALM-ViT path.  Every forward/backward is an explicit sequence of
C-ABI kernel launches (backend.py -> libcalmvit_hip.so); torch only allocates the tensors and
threads the autograd graph.  File:line citations are into /root/reference/CALM-ViT/.

Layout conventions: activations are contiguous fp32; token tensors are [B,S,D]; per-head tensors
stay in the [B,S,H*hd] layout of the projection output (heads are addressed through GEMM strides,
never materialised by a transpose).
"""
import math
import os
import threading

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .backend import ACT_GELU, ACT_GELU_BWD, ACT_NONE, act_dtype, bf16_pipeline, fp8_linears, get_backend
from .spectral_norm import W16_ATTR, W16_GEN_ATTR

# The reference calls the model under autocast(bfloat16) (distributed_trainer_cls.py:84): custom_fwd records the
# autocast state of the forward call, custom_bwd re-establishes it around backward (autograd runs backward outside the
# `with` block), and backend.effective_precision() maps that state to the GEMM pipe — forward and backward GEMMs of
# one call therefore always use the same arithmetic.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda")
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")

# grouped q/k/v launches (SNLinearGroupFn); CALM_GROUP_PROJECTIONS=0 restores one launch per projection (A/B switch)
GROUP_PROJECTIONS = os.environ.get("CALM_GROUP_PROJECTIONS", "1") != "0"
# skip-connection gradient summed inside the LayerNorm backward kernel (LayerNormSkipFn); =0: separate add (A/B switch)
FUSE_LN_SKIP = os.environ.get("CALM_FUSE_LN_SKIP", "1") != "0"

_noise_override = None


def set_noise_override(fn):
    """Tests inject the latent noise (`fn(like) -> tensor`); None restores torch.randn_like."""
    global _noise_override
    _noise_override = fn


def draw_noise(like):
    return _noise_override(like) if _noise_override is not None else torch.randn_like(like)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _wop(w):
    """The GEMM operand for weight `w`: in the bf16 pipeline the bf16 copy made at the start of this forward
    (spectral_norm._refresh_bf16_weights) — the product is the same as rounding the fp32 weight while it is staged —
    when its rows can be staged as 16-byte bf16 vectors (in_features % 8 == 0)."""
    if bf16_pipeline() and w.shape[-1] % 8 == 0:
        w16 = getattr(w, W16_ATTR, None)
        if w16 is not None and w16.shape == w.shape:
            return w16
    return w


def _wgen(*ws):
    """Generation of the bf16 copies of `ws` at forward time (None for weights without a copy)."""
    return tuple(getattr(w, W16_GEN_ATTR, None) for w in ws)


def _check_wgen(gen, *ws):
    """Backward must see the bf16 weight copies its forward used: they are rewritten in place by every forward."""
    if gen != _wgen(*ws):
        raise RuntimeError("one of the bf16 weight copies needed for gradient computation has been rewritten by a later "
                           "forward (the copies are refreshed in place at the start of every forward): run backward "
                           "before the next forward of the same model")


class _ZeroArena:
    """Small zero-initialised gradient accumulators (LayerNorm dw, bias column sums, RoPE d_inv_freq, the CNN tail's
    weight gradients: ~200 per step, each a few hundred floats) are handed out as slices of 1 MB chunks that one
    fill kernel zeroes, instead of one torch.zeros launch each.  Chunks are never reset or reused: a chunk lives as
    long as any slice of it (a .grad) does, so there is no aliasing between steps."""
    def __init__(self, chunk_floats=1 << 18):
        self.CHUNK = chunk_floats
        self.buf, self.off, self.lock = None, 0, threading.Lock()
        self.owner = None            # (device, stream) the current chunk was allocated on: the caching allocator ties a
                                     # block to its allocation stream, so slices are only handed out on that stream

    def take(self, shape, like):
        n = int(math.prod(shape))
        m = (n + 63) & ~63                                   # 256-byte granules
        capturing = like.is_cuda and torch.cuda.is_current_stream_capturing()
        if not ZERO_ARENA or m > self.CHUNK // 4 or like.dtype != torch.float32 or capturing:
            return torch.zeros(shape, dtype=like.dtype, device=like.device)
        owner = (like.device, torch.cuda.current_stream(like.device).cuda_stream if like.is_cuda else 0)
        with self.lock:
            if self.buf is None or self.owner != owner or self.off + m > self.CHUNK:
                self.buf = torch.zeros(self.CHUNK, dtype=torch.float32, device=like.device)
                self.off, self.owner = 0, owner
            v = self.buf[self.off:self.off + n].view(shape)
            self.off += m
        return v


ZERO_ARENA = os.environ.get("CALM_ZERO_ARENA", "1") != "0"      # A/B switch
_zeros = _ZeroArena().take
# weight-gradient outputs of the split-K launches (up to a few MB each, ~160 MB per step): slices of 64 MB zeroed
# chunks + accumulate=True, so that calm_gemm needs no memset per output
_zeros_big = _ZeroArena(1 << 24).take


# ---------------------------------------------------------------------------------------
# GEMM patterns (x2: [M,K], w: [N,K], dy2: [M,N]; all row-major contiguous)
# ---------------------------------------------------------------------------------------
def _lin_fwd(be, x2, w, sigma, out, bias=None, act=ACT_NONE, col_scale=None, residual=None, pre=None):
    M, K = x2.shape
    N = w.shape[0]
    be.gemm(x2, w, out, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sigma, bias=bias, act=act,
            col_scale=col_scale, residual=residual, r=(N, 0, 0), C_pre=pre, split_k=1)


def _lin_dgrad(be, dy2, w, sigma, dx, act=ACT_NONE, aux=None, residual=None, accumulate=False):
    M, N = dy2.shape
    K = w.shape[1]
    be.gemm(dy2, w, dx, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), inv_scale=sigma, act=act, aux=aux,
            residual=residual, r=(K, 0, 0), accumulate=accumulate, split_k=1)


def _lin_wgrad(be, dy2, x2, G):
    """split_k=0: the library may slice the long token reduction and combine with fp32 atomics (the
    only non-bitwise-reproducible launches of the path; forward and dgrad GEMMs never split)."""
    M, N = dy2.shape
    K = x2.shape[1]
    be.gemm(dy2, x2, G, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0), accumulate=True)     # G comes zeroed (_zeros_big)


# Attribute set (by trainer.FusedClipAdamW) on the weight_orig PARAMETER of a spectral-normed layer whose weight-gradient
# correction is applied later, by the fused optimizer-side step: its backward hands the gradient w.r.t. the NORMALISED
# weight to autograd unchanged.  The forward of every operator reads the mark from the parameter object it is given
# (`_deferred(w)`) and keeps it in ctx for its backward.
DEFER_ATTR = "_calm_sn_deferred"


def _deferred(w):
    return bool(getattr(w, DEFER_ATTR, False))


def _sn_wbwd(be, G, w, u, v, sigma, ls=None, defer=False):
    """G (grad wrt the effective weight, before LayerScale) -> (dW_orig, d_ls)."""
    if defer:
        if ls is not None:
            raise RuntimeError("a layer used with LayerScale cannot defer its spectral-norm gradient")
        return G.view_as(w), None
    rows = w.shape[0]
    cols = w.numel() // rows
    dW = torch.empty_like(w)
    d_ls = torch.empty_like(ls) if ls is not None else None
    be.sn_weight_bwd(G, w, u, v, sigma, ls, dW, d_ls, rows, cols)
    return dW, d_ls


E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2


def _fp8_ok(*dims):
    """fp8 Linear products (backend.fp8_linears, BASELINE config #5): every contraction / row extent a multiple of 16."""
    return fp8_linears() and all(d % 16 == 0 for d in dims)


def _lin_fwd8(be, x2, w, sigma, out, **epi):
    """_lin_fwd with both operands quantised to e4m3 (per-tensor scale, just in time)."""
    M, K = x2.shape
    N = w.shape[0]
    xq, dqx = be.quantize_fp8(x2, E4M3)
    wq, dqw = be.quantize_fp8(_c(w), E4M3)
    be.gemm(xq, wq, out, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sigma, r=(N, 0, 0), split_k=1,
            a_dq=dqx, b_dq=dqw, **epi)


def _lin_dgrad8(be, dy2, w_eff, sigma, dx, **epi):
    """dx = dy W / sigma with dy in e5m2 and a transposed e4m3 copy of the (effective) weight [N, K] -> [K, N]."""
    M, N = dy2.shape
    K = w_eff.shape[1]
    dyq, dqd = be.quantize_fp8(dy2, E5M2)
    wq, dqw = be.quantize_fp8(_c(w_eff), E4M3)
    wtq = be.transpose_u8(wq)
    be.gemm(dyq, wtq, dx, M, K, N, (N, 1, 0, 0), (N, 1, 0, 0), (K, 0, 0), inv_scale=sigma, r=(K, 0, 0), split_k=1,
            a_dq=dqd, b_dq=dqw, **epi)


CAST_GRADS = os.environ.get("CALM_CAST_GRADS", "1") != "0"      # A/B switch


def _gop(be, g2):
    """GEMM operand for a fp32 gradient [M, N] that the backward is about to read TWICE (weight gradient and input
    gradient): in the bf16 pipeline a bf16 copy made by one pass (6 bytes per element) — each of the two GEMMs then
    stages 2 bytes per element without converting.  Same products as rounding while staging."""
    if CAST_GRADS and bf16_pipeline() and g2.dtype == torch.float32 and g2.shape[-1] % 8 == 0 and g2.numel() >= (1 << 20):
        g16 = torch.empty(g2.shape, dtype=torch.bfloat16, device=g2.device)
        be.cast_bf16(g2, g16)
        return g16
    return g2


def _colsum(be, x2):
    out = _zeros((x2.shape[1],), x2 if x2.dtype == torch.float32 else x2.new_empty(0, dtype=torch.float32))
    be.colsum(x2, out, x2.shape[0], x2.shape[1])            # fp32 column sums of a fp32 or bf16 tensor
    return out


# ---------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """LayerNorm(D, eps=1e-6, bias=False) (Vi_Tools:131-132,197,494).  gemm_only: the output feeds nothing but GEMMs
    (ln_q / ln_kv / ln_2 of a block) — the bf16 pipeline then stores it as bf16, rounded once here instead of every
    time a GEMM stages a tile of it."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, eps, gemm_only=False):
        be = get_backend()
        x = _c(x)
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty(x.shape, dtype=act_dtype(D) if gemm_only else x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=x.dtype, device=x.device)
        rstd = torch.empty_like(mean)
        be.layernorm_fwd(x, w, y, mean, rstd, rows, D, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        be = get_backend()
        x, w, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        D = x.shape[-1]
        dx = torch.empty_like(x)
        dw = _zeros(w.shape, w)
        be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, x.numel() // D, D)
        return dx, dw, None, None


class LayerNormSkipFn(Function):
    """(LayerNorm(x), x): the norm together with the skip connection that bypasses it.  Every block feeds its input
    both to a LayerNorm and to a residual add (Vi_Tools:209-211 with 309, 310-315), so autograd would sum the two
    input gradients with one more elementwise pass; here the skip's gradient is added inside the LayerNorm backward
    kernel (dx_add of calm_layernorm_bwd).  Use the second output wherever the reference re-uses x."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, eps):
        be = get_backend()
        x = _c(x)
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty(x.shape, dtype=act_dtype(D), device=x.device)       # feeds GEMMs only (see LayerNormFn)
        mean = torch.empty(rows, dtype=x.dtype, device=x.device)
        rstd = torch.empty_like(mean)
        be.layernorm_fwd(x, w, y, mean, rstd, rows, D, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy, dskip):
        be = get_backend()
        x, w, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dskip, None, None
        dy = _c(dy)
        D = x.shape[-1]
        dx = torch.empty_like(x)
        dw = _zeros(w.shape, w)
        be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, x.numel() // D, D,
                         dx_add=_c(dskip).reshape(x.shape) if dskip is not None else None)
        return dx, dw, None


class SNLinearFn(Function):
    """y = act(x W_orig^T / sigma + bias) * ls + residual — a spectral-normed nn.Linear with the
    epilogues the block applies right after it (Vi_Tools:265-267, 276-277, 230-231, 300, 308)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, bias, ls, residual, u, v, sigma, act, out16=False):
        be = get_backend()
        x = _c(x)
        K = x.shape[-1]
        N = w.shape[0]
        x2 = x.reshape(-1, K)
        wop = _wop(w)
        # out16: the output feeds RoPE / the bf16 attention only (q, k, v projections) -> a bf16 tensor in the bf16 pipeline
        if out16 and (act != ACT_NONE or ls is not None or residual is not None):
            raise ValueError("a bf16 projection output takes no activation / LayerScale / residual epilogue")
        out = torch.empty(x.shape[:-1] + (N,), dtype=act_dtype(N) if out16 else torch.float32, device=x.device)
        pre = torch.empty_like(out) if act == ACT_GELU else None
        res2 = _c(residual).reshape(-1, N) if residual is not None else None
        ctx.fp8 = _fp8_ok(K, N) and x2.shape[0] >= 1024          # the large token-axis linears only
        if ctx.fp8:
            _lin_fwd8(be, x2, w, sigma, out.view(-1, N), bias=bias, act=act, col_scale=ls, residual=res2,
                      C_pre=pre.view(-1, N) if pre is not None else None)
        else:
            _lin_fwd(be, x2, wop, sigma, out.view(-1, N), bias=bias, act=act, col_scale=ls, residual=res2,
                     pre=pre.view(-1, N) if pre is not None else None)
        ctx.act = act
        ctx.wop = wop
        ctx.wgen = _wgen(w)
        ctx.defer = _deferred(w)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.save_for_backward(x2, w, ls, u, v, sigma, pre)
        ctx.xshape = x.shape
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        be = get_backend()
        x2, w, ls, u, v, sigma, pre = ctx.saved_tensors
        _check_wgen(ctx.wgen, w)
        N, K = w.shape
        dy2 = _c(dy).reshape(-1, N)
        if ctx.act == ACT_GELU:
            dz = torch.empty_like(dy2)
            be.gelu_bwd(dy2, pre.view(-1, N), dz, dy2.numel())
        else:
            dz = dy2
        dzg = _gop(be, dz) if ctx.needs_input_grad[0] else dz
        G = _zeros_big(w.shape, w)
        _lin_wgrad(be, dzg, x2, G)
        dW, d_ls = _sn_wbwd(be, G, w, u, v, sigma, ls, defer=ctx.defer)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)                       # a bf16 input (LayerNorm output) gets a bf16 gradient
            if ls is not None:
                wl = torch.empty_like(w if ctx.fp8 else ctx.wop)
                be.row_scale(w, ls, wl, N, K)
            else:
                wl = w if ctx.fp8 else ctx.wop
            if ctx.fp8:
                _lin_dgrad8(be, dz, wl, sigma, dx)
            else:
                _lin_dgrad(be, dzg, wl, sigma, dx)
            dx = dx.view(ctx.xshape)
        db = _colsum(be, dz) if ctx.has_bias else None
        dres = dy if ctx.has_res else None
        return dx, dW, db, d_ls, dres, None, None, None, None, None


class SNLinearGroupFn(Function):
    """y_g = x W_g^T / sigma_g for g = 0..n-1: the bias-free spectral-normed projections of a block that read the same
    activation — q/k/v of a plain self-attention block, k/v of the others (Vi_Tools:265-267) — as ONE grouped launch
    (more tiles per launch: fewer partly filled rounds), and their input gradient dx = sum_g dy_g W_g / sigma_g as one
    pass over the concatenated reduction instead of n GEMMs plus n-1 gradient additions.
    apply(x, out16, w_0, u_0, v_0, sigma_0, w_1, ...) -> (y_0, ..., y_{n-1});  out16: bf16 outputs in the bf16 pipeline"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, out16, *wuvs):
        be = get_backend()
        n = len(wuvs) // 4
        ws, sigmas = list(wuvs[0::4]), list(wuvs[3::4])
        x = _c(x)
        K = x.shape[-1]
        N = ws[0].shape[0]
        assert all(w.shape == (N, K) for w in ws)
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        wops = [_wop(w) for w in ws]
        odt = act_dtype(N) if out16 else torch.float32
        outs = [torch.empty(x.shape[:-1] + (N,), dtype=odt, device=x.device) for _ in range(n)]
        be.gemm(x2, wops, [o.view(-1, N) for o in outs], M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), batch=(n, 1),
                inv_scale=sigmas, split_k=1)
        ctx.n = n
        ctx.wops = wops
        ctx.wgen = _wgen(*ws)
        ctx.defer = [_deferred(w) for w in ws]
        ctx.xshape = x.shape
        ctx.save_for_backward(x2, *wuvs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, *dys):
        be = get_backend()
        x2, *wuvs = ctx.saved_tensors
        n = ctx.n
        ws, us, vs, sigmas = list(wuvs[0::4]), list(wuvs[1::4]), list(wuvs[2::4]), list(wuvs[3::4])
        _check_wgen(ctx.wgen, *ws)
        N, K = ws[0].shape
        M = x2.shape[0]
        dy2 = [_c(d).reshape(-1, N) for d in dys]
        if ctx.needs_input_grad[0]:
            dy2 = [_gop(be, d) for d in dy2]             # each is read by the weight-gradient and the input-gradient launch
        # weight gradients G_g = dy_g^T x: one grouped launch, every group split over its own k-slices
        Gs = [_zeros_big(w.shape, w) for w in ws]
        be.gemm(dy2, x2, Gs, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0), batch=(n, 1), accumulate=True)
        grads = []
        for g in range(n):
            grads += [_sn_wbwd(be, Gs[g], ws[g], us[g], vs[g], sigmas[g], defer=ctx.defer[g])[0], None, None, None]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            be.gemm(dy2, ctx.wops, dx, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), batch=(n, 1), inv_scale=sigmas,
                    reduce_batch=True, split_k=1)
            dx = dx.view(ctx.xshape)
        return (dx, None, *grads)


class MlpFn(Function):
    """out = (gelu(x W1^T/s1 + b1) W2^T/s2 + b2) * ls + residual  (Vi_Tools:199-205,310-315 block MLP;
    CALM_ViT_V2.py:49-53,76 cls head).  The GELU backward is fused into the dgrad GEMM epilogue."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, b1, w2, b2, ls, residual, u1, v1, s1, u2, v2, s2):
        be = get_backend()
        x = _c(x)
        K = x.shape[-1]
        Hd = w1.shape[0]
        N = w2.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        wop1, wop2 = _wop(w1), _wop(w2)
        # hidden state and pre-activation feed GEMMs / the GELU' epilogue only: bf16 tensors in the bf16 pipeline
        hp = torch.empty(M, Hd, dtype=act_dtype(Hd), device=x.device)
        hg = torch.empty_like(hp)
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        res2 = _c(residual).reshape(-1, N) if residual is not None else None
        ctx.fp8 = _fp8_ok(K, Hd, N)
        if ctx.fp8:                      # both products on fp8 operands; hidden state / pre-activation stay bf16 tensors
            _lin_fwd8(be, x2, w1, s1, hg, bias=b1, act=ACT_GELU, C_pre=hp)
            _lin_fwd8(be, hg, w2, s2, out.view(-1, N), bias=b2, col_scale=ls, residual=res2)
        else:
            _lin_fwd(be, x2, wop1, s1, hg, bias=b1, act=ACT_GELU, pre=hp)
            _lin_fwd(be, hg, wop2, s2, out.view(-1, N), bias=b2, col_scale=ls, residual=res2)
        ctx.wops = (wop1, wop2)
        ctx.wgen = _wgen(w1, w2)
        ctx.has_b1, ctx.has_b2, ctx.has_res = b1 is not None, b2 is not None, residual is not None
        ctx.defer = (_deferred(w1), _deferred(w2))
        ctx.xshape = x.shape
        ctx.save_for_backward(x2, hp, hg, w1, w2, ls, u1, v1, s1, u2, v2, s2)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        be = get_backend()
        x2, hp, hg, w1, w2, ls, u1, v1, s1, u2, v2, s2 = ctx.saved_tensors
        _check_wgen(ctx.wgen, w1, w2)
        N, Hd = w2.shape
        K = w1.shape[1]
        do2 = _c(dout).reshape(-1, N)
        do2g = _gop(be, do2)
        G2 = _zeros_big(w2.shape, w2)
        _lin_wgrad(be, do2g, hg, G2)
        dW2, d_ls = _sn_wbwd(be, G2, w2, u2, v2, s2, ls, defer=ctx.defer[1])
        db2 = _colsum(be, do2) if ctx.has_b2 else None
        wop1, wop2 = ctx.wops
        if ls is not None:
            w2l = torch.empty_like(w2 if ctx.fp8 else wop2)
            be.row_scale(w2, ls, w2l, N, Hd)
        else:
            w2l = w2 if ctx.fp8 else wop2
        dhp = torch.empty_like(hp)
        if ctx.fp8:
            _lin_dgrad8(be, do2, w2l, s2, dhp, act=ACT_GELU_BWD, aux=hp)
        else:
            _lin_dgrad(be, do2g, w2l, s2, dhp, act=ACT_GELU_BWD, aux=hp)
        G1 = _zeros_big(w1.shape, w1)
        _lin_wgrad(be, dhp, x2, G1)
        dW1, _ = _sn_wbwd(be, G1, w1, u1, v1, s1, defer=ctx.defer[0])
        db1 = _colsum(be, dhp) if ctx.has_b1 else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            if ctx.fp8:
                _lin_dgrad8(be, dhp, w1, s1, dx)
            else:
                _lin_dgrad(be, dhp, wop1, s1, dx)
            dx = dx.view(ctx.xshape)
        dres = dout if ctx.has_res else None
        return dx, dW1, db1, dW2, db2, d_ls, dres, None, None, None, None, None, None


class SeqLinearFn(Function):
    """Y[b] = W X[b] / sigma along the sequence axis — the reference's permute->Linear->permute
    (Vi_Tools:224-229,250-264,304-306) as a batched GEMM with a transposed operand (no copies)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, u, v, sigma):
        be = get_backend()
        x = _c(x)
        B, S, D = x.shape
        S2 = w.shape[0]
        wop = _wop(w)
        out = torch.empty(B, S2, D, dtype=torch.float32, device=x.device)
        be.gemm(wop, x, out, S2, D, S, (S, 1, 0, 0), (1, D, S * D, 0), (D, S2 * D, 0), batch=(B, 1), inv_scale=sigma)
        ctx.defer = _deferred(w)
        ctx.wop = wop
        ctx.wgen = _wgen(w)
        ctx.save_for_backward(x, w, u, v, sigma)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        be = get_backend()
        x, w, u, v, sigma = ctx.saved_tensors
        _check_wgen(ctx.wgen, w)
        B, S, D = x.shape
        S2 = w.shape[0]
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dX[b] = W^T dY[b] / sigma : A(m=s,k=s2) = W[s2,s]
            be.gemm(ctx.wop, dy, dx, S, D, S2, (1, S, 0, 0), (1, D, S2 * D, 0), (D, S * D, 0), batch=(B, 1),
                    inv_scale=sigma)
        # G[s2,s] = sum_b sum_d dY[b,s2,d] X[b,s,d]
        G = _zeros_big(w.shape, w)
        be.gemm(dy, x, G, S2, S, D, (D, 1, S2 * D, 0), (D, 1, S * D, 0), (S, 0, 0), batch=(B, 1), reduce_batch=True,
                accumulate=True)
        dW, _ = _sn_wbwd(be, G, w, u, v, sigma, defer=ctx.defer)
        return dx, dW, None, None, None


class RopeFn(Function):
    """out[...,h,:dc] = content, out[...,h,dc:] = rope(xr) with learned inv_freq (Vi_Tools:80-95,275-285)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, content, xr, inv_freq, H, out16=False):
        be = get_backend()
        xr = _c(xr)
        B, S, W = xr.shape
        dr = W // H
        dc = 0
        if content is not None:
            content = _c(content)
            dc = content.shape[-1] // H
        table = torch.empty(2 * S * (dr // 2), dtype=torch.float32, device=xr.device)
        # out16: q / k for the bf16 attention kernels; the inputs may be fp32 or bf16 projection outputs, independently
        out = torch.empty(B, S, H * (dc + dr), dtype=torch.bfloat16 if out16 else torch.float32, device=xr.device)
        be.rope_fwd(content, xr, inv_freq, table, out, B, S, H, dc, dr)
        ctx.dims = (B, S, H, dc, dr)
        ctx.content_dtype = content.dtype if content is not None else None
        ctx.save_for_backward(xr, table)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        be = get_backend()
        xr, table = ctx.saved_tensors
        B, S, H, dc, dr = ctx.dims
        dout = _c(dout)
        d_xr = torch.empty_like(xr)                          # gradients take the storage type of their tensors
        d_content = torch.empty(B, S, H * dc, dtype=ctx.content_dtype, device=xr.device) if dc > 0 else None
        d_if = _zeros((dr // 2,), table)
        be.rope_bwd(dout, xr, table, d_content, d_xr, d_if, B, S, H, dc, dr)
        return d_content, d_xr, d_if, None, None


class LatentMaskAttentionFn(Function):
    """softmax(Q_h K_h^T / sqrt(hd) + M) V_h with M = W2 gelu(W1 (sum_h Q_h K_h^T) + b1) + b2 applied
    along the key axis and shared by all heads (Vi_Tools:288-299).  q:[B,Sq,H*hd] k,v:[B,Skv,H*hd]."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, w1, b1, w2, b2, u1, v1, s1, u2, v2, s2, H):
        be = get_backend()
        q, k, v = _c(q), _c(k), _c(v)
        B, Sq, D = q.shape
        Skv = k.shape[1]
        hd = D // H
        dev, dt = q.device, q.dtype
        R = torch.empty(B, Sq, Skv, dtype=dt, device=dev)
        hp = torch.empty(B * Sq, w1.shape[0], dtype=dt, device=dev)
        hg = torch.empty_like(hp)
        P = torch.empty(B, H, Sq, Skv, dtype=dt, device=dev)
        out = torch.empty(B, Sq, D, dtype=dt, device=dev)
        if be.attn_fwd_supported(Sq, Skv, H, hd):
            # one fused kernel: mask produced and applied in-kernel, softmax by wave shuffles
            Mk = torch.empty(B * Sq, Skv, dtype=dt, device=dev)
            be.attn_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, P, B, Sq, Skv, H, hd)
        else:
            # shapes without a fused instantiation: the same arithmetic composed from GEMM + softmax kernels
            be.gemm(q, k, R, Sq, Skv, D, (D, 1, Sq * D, 0), (D, 1, Skv * D, 0), (Skv, Sq * Skv, 0), batch=(B, 1))
            R2 = R.view(B * Sq, Skv)
            _lin_fwd(be, R2, w1, s1, hg, bias=b1, act=ACT_GELU, pre=hp)
            Mk = torch.empty(B * Sq, Skv, dtype=dt, device=dev)
            _lin_fwd(be, hg, w2, s2, Mk, bias=b2)
            scale = 1.0 / math.sqrt(hd)
            be.gemm(q, k, P, Sq, Skv, hd, (D, 1, Sq * D, hd), (D, 1, Skv * D, hd), (Skv, H * Sq * Skv, Sq * Skv),
                    batch=(B, H), alpha=scale, residual=Mk, r=(Skv, Sq * Skv, 0))
            be.softmax_fwd(P, B * H * Sq, Skv)
            be.gemm(P, v, out, Sq, hd, Skv, (Skv, 1, H * Sq * Skv, Sq * Skv), (1, D, Skv * D, hd), (D, Sq * D, hd),
                    batch=(B, H))
        ctx.H = H
        ctx.defer = (_deferred(w1), _deferred(w2))
        ctx.save_for_backward(q, k, v, R, hp, hg, P, w1, w2, u1, v1, s1, u2, v2, s2)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        be = get_backend()
        q, k, v, R, hp, hg, P, w1, w2, u1, v1, s1, u2, v2, s2 = ctx.saved_tensors
        H = ctx.H
        B, Sq, D = q.shape
        Skv = k.shape[1]
        hd = D // H
        scale = 1.0 / math.sqrt(hd)
        dout = _c(dout)
        dev, dt = q.device, q.dtype
        pb = (H * Sq * Skv, Sq * Skv)
        dP = torch.empty_like(P)
        dv, dq, dk = torch.empty_like(v), torch.empty_like(q), torch.empty_like(k)
        dM = torch.empty(B * Sq, Skv, dtype=dt, device=dev)
        if be.attn_bwd_preferred(Sq, Skv, H, hd):
            # fused core: dP, softmax backward, head-sum of dS and the four per-head products in two launches
            # (taken where it is measured faster than the composition below: head dims <= 64)
            be.attn_bwd(q, k, v, dout, P, dP, dq, dk, dv, dM, B, Sq, Skv, H, hd)
        else:
            # dP = dO V^T ; dV = P^T dO
            be.gemm(dout, v, dP, Sq, Skv, hd, (D, 1, Sq * D, hd), (D, 1, Skv * D, hd), (Skv,) + pb, batch=(B, H))
            be.gemm(P, dout, dv, Skv, hd, Sq, (1, Skv) + pb, (1, D, Sq * D, hd), (D, Skv * D, hd), batch=(B, H))
            be.softmax_bwd_heads(P, dP, dM, B, H, Sq, Skv)         # dP now holds dL, dM its sum over the heads
            be.gemm(dP, k, dq, Sq, hd, Skv, (Skv, 1) + pb, (1, D, Skv * D, hd), (D, Sq * D, hd), batch=(B, H),
                    alpha=scale)
            be.gemm(dP, q, dk, Skv, hd, Sq, (1, Skv) + pb, (1, D, Sq * D, hd), (D, Skv * D, hd), batch=(B, H),
                    alpha=scale)
        # mask MLP backward
        G2 = _zeros_big(w2.shape, w2)
        _lin_wgrad(be, dM, hg, G2)
        dW2, _ = _sn_wbwd(be, G2, w2, u2, v2, s2, defer=ctx.defer[1])
        db2 = _colsum(be, dM)
        dhp = torch.empty_like(hp)
        _lin_dgrad(be, dM, w2, s2, dhp, act=ACT_GELU_BWD, aux=hp)
        R2 = R.view(B * Sq, Skv)
        G1 = _zeros_big(w1.shape, w1)
        _lin_wgrad(be, dhp, R2, G1)
        dW1, _ = _sn_wbwd(be, G1, w1, u1, v1, s1, defer=ctx.defer[0])
        db1 = _colsum(be, dhp)
        dR = torch.empty(B, Sq, Skv, dtype=dt, device=dev)
        _lin_dgrad(be, dhp, w1, s1, dR.view(B * Sq, Skv))
        # dQ_all += dR K_all ; dK_all += dR^T Q_all
        be.gemm(dR, k, dq, Sq, D, Skv, (Skv, 1, Sq * Skv, 0), (1, D, Skv * D, 0), (D, Sq * D, 0), batch=(B, 1),
                accumulate=True)
        be.gemm(dR, q, dk, Skv, D, Sq, (1, Skv, Sq * Skv, 0), (1, D, Sq * D, 0), (D, Skv * D, 0), batch=(B, 1),
                accumulate=True)
        return dq, dk, dv, dW1, db1, dW2, db2, None, None, None, None, None, None, None


class LatentMaskAttention16Fn(Function):
    """The same attention on the bf16 matrix pipe (bf16 pipeline; backend.attn16_supported): q, k, v, the output and
    every saved tensor are bf16, the probabilities are never stored (row log-sum-exp saved, P recomputed in backward).
    q, k, v: [B,S,H*hd] bf16."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, w1, b1, w2, b2, u1, v1, s1, u2, v2, s2, H):
        be = get_backend()
        q, k, v = _c(q), _c(k), _c(v)
        B, S, D = q.shape
        hd = D // H
        dev = q.device
        w1o, w2o = _wop(w1), _wop(w2)
        if w1o.dtype != torch.bfloat16 or w2o.dtype != torch.bfloat16:
            raise RuntimeError("bf16 attention needs the step's bf16 weight copies (spectral_norm._refresh_bf16_weights)")
        bf = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
        out, R, hp, hg, Mk, MkT = bf(B, S, D), bf(B, S, S), bf(B * S, 2 * S), bf(B * S, 2 * S), bf(B, S, S), bf(B, S, S)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        be.attn16_fwd(q, k, v, w1o, b1, s1, w2o, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd)
        ctx.H = H
        ctx.defer = (_deferred(w1), _deferred(w2))
        ctx.wops = (w1o, w2o)
        ctx.wgen = _wgen(w1, w2)
        ctx.save_for_backward(q, k, v, out, R, hp, hg, Mk, MkT, lse, w1, w2, u1, v1, s1, u2, v2, s2)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        be = get_backend()
        q, k, v, out, R, hp, hg, Mk, MkT, lse, w1, w2, u1, v1, s1, u2, v2, s2 = ctx.saved_tensors
        _check_wgen(ctx.wgen, w1, w2)
        w1o, w2o = ctx.wops
        H = ctx.H
        B, S, D = q.shape
        hd = D // H
        dev = q.device
        dout = _c(dout)
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dM = torch.empty(B * S, S, dtype=torch.bfloat16, device=dev)
        delta = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        be.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta, dq, dk, dv, dM, B, S, H, hd)
        # mask MLP backward: typed GEMMs on the bf16 tensors
        G2 = _zeros_big(w2.shape, w2)
        _lin_wgrad(be, dM, hg, G2)
        dW2, _ = _sn_wbwd(be, G2, w2, u2, v2, s2, defer=ctx.defer[1])
        db2 = _colsum(be, dM)
        dhp = torch.empty_like(hp)
        _lin_dgrad(be, dM, w2o, s2, dhp, act=ACT_GELU_BWD, aux=hp)
        R2 = R.view(B * S, S)
        G1 = _zeros_big(w1.shape, w1)
        _lin_wgrad(be, dhp, R2, G1)
        dW1, _ = _sn_wbwd(be, G1, w1, u1, v1, s1, defer=ctx.defer[0])
        db1 = _colsum(be, dhp)
        dR = torch.empty(B, S, S, dtype=torch.bfloat16, device=dev)
        _lin_dgrad(be, dhp, w1o, s1, dR.view(B * S, S))
        # dQ_all += dR K_all ; dK_all += dR^T Q_all  (accumulated into the bf16 gradients)
        be.gemm(dR, k, dq, S, D, S, (S, 1, S * S, 0), (1, D, S * D, 0), (D, S * D, 0), batch=(B, 1), accumulate=True)
        be.gemm(dR, q, dk, S, D, S, (1, S, S * S, 0), (1, D, S * D, 0), (D, S * D, 0), batch=(B, 1), accumulate=True)
        return dq, dk, dv, dW1, db1, dW2, db2, None, None, None, None, None, None, None


def use_attention16(S, H, hd):
    """The bf16 attention kernels serve this block: bf16 pipeline on, shape supported, token width a multiple of 8
    (q / k / v are also operands of the batched dR products)."""
    return bf16_pipeline() and (H * hd) % 8 == 0 and get_backend().attn16_supported(S, H, hd)


class LatentFn(Function):
    """(mean|raw) -> z = mean + eps*std, std = softplus(raw)+1e-6 (Vi_Tools:232-242) and this
    tensor's KL term -0.5*mean(1 + 2 log std - mean^2 - std^2) (Vi_Tools:24-25)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, mv, noise):
        be = get_backend()
        mv = _c(mv)
        mvh = mv.shape[-1] // 2
        rows = mv.numel() // (2 * mvh)
        z = torch.empty(mv.shape[:-1] + (mvh,), dtype=mv.dtype, device=mv.device)
        std = torch.empty_like(z)
        kl = torch.zeros((), dtype=mv.dtype, device=mv.device)
        if noise is not None:
            noise = _c(noise)
        be.latent_fwd(mv, noise, z, std, kl, rows, mvh)
        ctx.scale = -0.5 / z.numel()
        kl.mul_(ctx.scale)
        ctx.save_for_backward(mv, noise, std)
        ctx.mark_non_differentiable(std)
        return z, std, kl

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dz, _dstd, dkl):
        be = get_backend()
        mv, noise, std = ctx.saved_tensors
        mvh = mv.shape[-1] // 2
        rows = mv.numel() // (2 * mvh)
        dks = (dkl * ctx.scale).reshape(1).contiguous() if dkl is not None else None
        dz = _c(dz) if dz is not None else None
        dmv = torch.empty_like(mv)
        be.latent_bwd(dz, dks, mv, noise, std, dmv, rows, mvh)
        return dmv, None


class AddFn(Function):
    """Residual / U-net skip adds (Vi_Tools:309,315,403,513-522) and the latent running sum (43-44)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, a, b):
        be = get_backend()
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        be.add(a, b, out, a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class RowsToImageFn(Function):
    """[B,S,3S] -> [B,3,S,S]: the inverse tokenisation (the image a bare `proj(img)` call hands back)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, rows):
        be = get_backend()
        rows = _c(rows)
        B, S, W = rows.shape
        assert W == 3 * S, "token grid must be [B,S,3S]"
        img = torch.empty(B, 3, S, S, dtype=rows.dtype, device=rows.device)
        be.rows_to_image(rows, img, B, S)
        ctx.dims = (B, S)
        return img

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g):
        be = get_backend()
        B, S = ctx.dims
        g = _c(g)
        rows = torch.empty(B, S, 3 * S, dtype=g.dtype, device=g.device)
        be.image_to_rows(g, rows, B, S)
        return rows


class ImageToRowsFn(Function):
    """[B,3,S,S] -> [B,S,3S] (Vi_Tools:389-391)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, img):
        be = get_backend()
        img = _c(img)
        B, Cc, S, S2 = img.shape
        assert Cc == 3 and S == S2, "row tokenisation expects [B,3,S,S]"
        rows = torch.empty(B, S, 3 * S, dtype=img.dtype, device=img.device)
        be.image_to_rows(img, rows, B, S)
        ctx.dims = (B, S)
        return rows

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g):
        be = get_backend()
        B, S = ctx.dims
        g = _c(g)
        img = torch.empty(B, 3, S, S, dtype=g.dtype, device=g.device)
        be.rows_to_image(g, img, B, S)
        return img


class GridTransposeFn(Function):
    """rows <-> columns of the [B,S,S,3] token grid (Vi_Tools:394-395,397-398); self-inverse."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        be = get_backend()
        x = _c(x)
        B, S, W = x.shape
        assert W == 3 * S, "token grid must be [B,S,3S]"
        out = torch.empty_like(x)
        be.grid_transpose(x, out, B, S)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g):
        be = get_backend()
        g = _c(g)
        B, S, _ = g.shape
        out = torch.empty_like(g)
        be.grid_transpose(g, out, B, S)
        return out


class MeanSeqFn(Function):
    """AdaptiveAvgPool1d(1) over the sequence (CALM_ViT_V2.py:74-75)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        be = get_backend()
        x = _c(x)
        B, S, D = x.shape
        y = torch.empty(B, D, dtype=x.dtype, device=x.device)
        be.mean_seq_fwd(x, y, B, S, D)
        ctx.dims = (B, S, D)
        return y

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g):
        be = get_backend()
        B, S, D = ctx.dims
        g = _c(g)
        dx = torch.empty(B, S, D, dtype=g.dtype, device=g.device)
        be.mean_seq_bwd(g, dx, B, S, D)
        return dx


class CnnResidualFn(Function):
    """x + conv1x1(32->3)(gelu(dw3x3(gelu(conv1x1(3->32)(x))))) on the token grid as a channels-last
    image (Vi_Tools:378-385,400-403; CALM_ViT_V2.py:60-67,80-83): one fused kernel each way, hidden
    maps kept in LDS and recomputed in backward (only x is saved)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w0, b0, w2, b2, w4, b4, u0, v0, s0, u2, v2, s2, u4, v4, s4, residual=True):
        be = get_backend()
        x = _c(x)
        B, S, W = x.shape
        Ch = w0.shape[0]
        out = torch.empty_like(x)
        be.cnn_fwd(x, w0, s0, b0, w2, s2, b2, w4, s4, b4, out, B, S, Ch, residual=residual)
        ctx.dims = (B, S, Ch)
        ctx.residual = residual
        ctx.defer = (_deferred(w0), _deferred(w2), _deferred(w4))
        ctx.save_for_backward(x, w0, b0, w2, b2, w4, b4, u0, v0, s0, u2, v2, s2, u4, v4, s4)
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        be = get_backend()
        x, w0, b0, w2, b2, w4, b4, u0, v0, s0, u2, v2, s2, u4, v4, s4 = ctx.saved_tensors
        B, S, Ch = ctx.dims
        dy = _c(dy)
        dev, dt = dy.device, dy.dtype
        dx = torch.empty_like(dy)
        gall = _zeros((Ch * 3 + Ch + Ch * 9 + Ch + 3 * Ch + 3,), dy)
        G0, db0, G2, db2, G4, db4 = torch.split(gall, [Ch * 3, Ch, Ch * 9, Ch, 3 * Ch, 3])
        be.cnn_bwd(dy, x, w0, s0, b0, w2, s2, b2, w4, s4, b4, dx, G0, db0, G2, db2, G4, db4, B, S, Ch,
                   residual=ctx.residual)
        dW0, _ = _sn_wbwd(be, G0.view(Ch, 3), w0.view(Ch, 3), u0, v0, s0, defer=ctx.defer[0])
        dW2, _ = _sn_wbwd(be, G2.view(Ch, 9), w2.view(Ch, 9), u2, v2, s2, defer=ctx.defer[1])
        dW4, _ = _sn_wbwd(be, G4.view(3, Ch), w4.view(3, Ch), u4, v4, s4, defer=ctx.defer[2])
        return (dx, dW0.view_as(w0), db0, dW2.view_as(w2), db2, dW4.view_as(w4), db4,
                None, None, None, None, None, None, None, None, None, None)


class GeluFn(Function):
    """erf-GELU on its own (the GELU module of a Sequential called directly; everywhere on the path the activation is
    a GEMM epilogue or lives inside the fused CNN tail)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        be = get_backend()
        x = _c(x)
        y = torch.empty_like(x)
        be.gelu_fwd(x, y, x.numel())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        be = get_backend()
        x, = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        be.gelu_bwd(dy, x, dx, x.numel())
        return dx


# ---------------------------------------------------------------------------------------
# functional front-ends
# ---------------------------------------------------------------------------------------
def layer_norm(x, w, eps=1e-6):
    return LayerNormFn.apply(x, w, eps)


def add(a, b):
    return AddFn.apply(a, b)


def image_to_rows(img):
    return ImageToRowsFn.apply(img)


def grid_transpose(x):
    return GridTransposeFn.apply(x)


def mean_seq(x):
    return MeanSeqFn.apply(x)
