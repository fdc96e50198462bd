"""CPU oracle for the CALM-ViT cross-axial latent-masking attention path.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch, *functional* (no nn.Module)
restatement in plain PyTorch-CPU fp32 of the reference's algorithm.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product path (`calm-vit-dte_amd/`) never does and fails loudly without its HIP library.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference
(`/root/reference/CALM-ViT/{Vi_Tools_CNN_less_V2,CALM_ViT_V2}.py`) in the build container
and commits its outputs as fixtures; `tests/test_oracle_golden.py` checks this restatement
against them (<=1e-5 rel).  The reference has no tests/golden vectors of its own
(SURVEY.md section 4).

All parameters live in a flat dict keyed by the reference's state-dict names
(`autoencoder.encoder_blocks.0.encoder.q_proj.weight_orig`, ...).  Citations below are
file:line into /root/reference/CALM-ViT/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

Tensor = torch.Tensor
LN_EPS = 1e-6          # Vi_Tools_CNN_less_V2.py:115  partial(LayerNorm, eps=1e-6)
SN_EPS = 1e-12         # torch.nn.utils.spectral_norm default eps (Vi_Tools:5)
SOFTPLUS_FLOOR = 1e-6  # Vi_Tools:234-235


@dataclass(frozen=True)
class ViTConfig:
    """kwargs of CALM_ViT_V2.ViT.__init__ (CALM_ViT_V2.py:22-25)."""
    heads: int = 12
    seq_length: int = 256
    in_features: int = 768
    dim_step: int = 48
    mean_var_hidden: int = 192
    seq_len_step: int = 16
    seq_len_reduce: int = 128
    out_features: int = 1000
    force_reduce: bool = False
    generate: bool = True


# ----------------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------------
def gelu_erf(x: Tensor) -> Tensor:
    """GELU(approximate='none') (Vi_Tools:191,202,381,383; CALM_ViT_V2.py:51)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: Tensor, w: Tensor) -> Tensor:
    """LayerNorm(D, eps=1e-6, bias=False): biased variance (Vi_Tools:131-132,197,494)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * w


def _l2_normalize(x: Tensor) -> Tensor:
    return x / x.norm().clamp_min(SN_EPS)


def spectral_weight(P: Dict[str, Tensor], name: str, training: bool) -> Tensor:
    """Hook-based torch spectral_norm `compute_weight` (applied at Vi_Tools:137-205,380-384;
    CALM_ViT_V2.py:50-52,62-66): one power iteration in train mode, updating u,v IN PLACE and
    without grad, then sigma = u^T W v and W = W_orig / sigma (grad flows through sigma with u,v
    constant)."""
    w = P[name + ".weight_orig"]
    u = P[name + ".weight_u"]
    v = P[name + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(_l2_normalize(wm.t().mv(u)))
            u.copy_(_l2_normalize(wm.mv(v)))
        u = u.clone()
        v = v.clone()
    sigma = torch.dot(u, wm.mv(v))
    return w / sigma


def sn_linear(P, name: str, x: Tensor, training: bool) -> Tensor:
    w = spectral_weight(P, name, training)
    y = x @ w.t()
    b = P.get(name + ".bias")
    return y if b is None else y + b


def seq_linear(P, name: str, x: Tensor, training: bool) -> Tensor:
    """permute(0,2,1) -> Linear -> permute(0,2,1) (Vi_Tools:224-229,250-264,304-306):
    a left-multiplication along the sequence axis, Y[b] = W @ X[b]."""
    w = spectral_weight(P, name, training)
    return torch.matmul(w, x)


def rope(x: Tensor, inv_freq: Tensor) -> Tensor:
    """Learned-frequency NeoX-style RoPE (Vi_Tools:80-95). x: [B,H,S,d]."""
    s, d = x.shape[2], x.shape[3]
    t = torch.arange(s, dtype=torch.float32)
    fr = torch.outer(t, inv_freq)                     # [S, d/2]
    ang = torch.cat((fr, fr), dim=-1)                 # [S, d]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    rot = torch.cat((-x2, x1), dim=-1)
    return x * ang.cos() + rot * ang.sin()


def rope_init_inv_freq(dim: int, theta: float = 10000.0) -> Tensor:
    """Vi_Tools:62."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))


# ----------------------------------------------------------------------------------
# residual state manager (Vi_Tools:7-50)
# ----------------------------------------------------------------------------------
class LatentState:
    def __init__(self, smooth_factor: float = 2.0, momentum: float = 0.9, mode: str = "ema"):
        self.zq = None
        self.zkv = None
        self.kl = 0.0
        self.count = 0
        self.smooth_factor = smooth_factor
        self.momentum = momentum
        self.mode = mode

    @staticmethod
    def _kl(mean: Tensor, std: Tensor) -> Tensor:
        # Vi_Tools:24-25 ("var" there is the softplus output, used as a std-dev)
        return -0.5 * torch.mean(1 + 2 * torch.log(std) - mean.pow(2) - std.pow(2))

    def merge(self, zq, zkv, mean_q, std_q, mean_kv, std_kv):
        self.kl = self._kl(mean_q, std_q) + self._kl(mean_kv, std_kv) + self.kl
        if self.zq is None:
            self.zq, self.zkv, self.count = zq, zkv, 1
        elif self.mode not in ("sum", "sma"):
            self.count += 1
            if self.mode == "ema":
                self.momentum = self.smooth_factor / (self.count + 1)
            elif self.mode == "lp":
                self.momentum = self.count / (self.count + 1)
            m = self.momentum
            self.zq = m * zq + (1 - m) * self.zq
            self.zkv = m * zkv + (1 - m) * self.zkv
        else:
            self.count += 1
            self.zq = self.zq + zq
            self.zkv = self.zkv + zkv
            if self.mode == "sma":
                return self.zq / self.count, self.zkv / self.count
        return self.zq, self.zkv

    def kl_loss(self):
        return self.kl / self.count if self.count > 0 else 0.0


# ----------------------------------------------------------------------------------
# VMLA block (Vi_Tools:98-315)
# ----------------------------------------------------------------------------------
@dataclass(frozen=True)
class VMLAShape:
    heads: int
    dim1: int
    dim2: int
    mvh: int
    seq: int
    seq_reduce: int
    seq_new: int
    force_reduce: bool
    t_force_reduce: bool
    is_cross: bool

    @property
    def hd_half(self):            # head_dim_content == head_dim_rope (Vi_Tools:123-124)
        return self.dim2 // self.heads // 2

    @property
    def hd(self):
        return 2 * self.hd_half

    @property
    def t_reduce(self):           # Vi_Tools:128
        return self.seq_new != self.seq or self.t_force_reduce

    @property
    def reduce(self):             # Vi_Tools:129
        return self.dim1 != self.dim2 or self.force_reduce


def attention_with_latent_mask(P, pre: str, q: Tensor, k: Tensor, v: Tensor, training: bool) -> Tensor:
    """Vi_Tools:288-299.  q,k,v: [B,H,S,hd].  The additive mask is the 2-layer MLP along the KEY
    axis of the all-head sum of the raw (unscaled, post-RoPE) logits; softmax uses 1/sqrt(hd)."""
    b, h, sq, hd = q.shape
    skv = k.shape[2]
    q_all = q.transpose(1, 2).reshape(b, sq, h * hd)
    k_all = k.transpose(1, 2).reshape(b, skv, h * hd)
    raw = q_all @ k_all.transpose(1, 2)                                   # [B,Sq,Skv]
    hid = gelu_erf(sn_linear(P, pre + "linear_mask.0", raw, training))
    mask = sn_linear(P, pre + "linear_mask.2", hid, training)             # [B,Sq,Skv]
    logits = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)) + mask.unsqueeze(1)
    p = torch.softmax(logits, dim=-1)
    o = p @ v                                                              # [B,H,Sq,hd]
    return o.transpose(1, 2).reshape(b, sq, h * hd)


def vmla_block(P, pre: str, sh: VMLAShape, input_q: Tensor, input_kv: Optional[Tensor],
               state: Optional[LatentState], training: bool,
               noise: Callable[[Tensor], Tensor]) -> Tensor:
    """VMLA_Block.forward (Vi_Tools:207-315) with mask=True (the only usable path)."""
    H = sh.heads
    residual = input_q
    xq = layer_norm(input_q, P[pre + "ln_q.weight"])
    xkv = xq if input_kv is None else layer_norm(input_kv, P[pre + "ln_kv.weight"])
    qz = qr = xq
    kz = vz = kr = xkv
    if sh.reduce:
        if sh.t_reduce:                                                   # 224-229
            xq = seq_linear(P, pre + "t_encoder_q", xq, training)
            xkv = seq_linear(P, pre + "t_encoder_kv", xkv, training)
        mv_q = sn_linear(P, pre + "encoder_q", xq, training)              # 230-231
        mv_kv = sn_linear(P, pre + "encoder_kv", xkv, training)
        mean_q, raw_q = mv_q.chunk(2, dim=-1)
        mean_kv, raw_kv = mv_kv.chunk(2, dim=-1)
        std_q = torch.nn.functional.softplus(raw_q) + SOFTPLUS_FLOOR      # 234-235
        std_kv = torch.nn.functional.softplus(raw_kv) + SOFTPLUS_FLOOR
        if training:                                                      # 237-242; q-noise first
            zq = mean_q + noise(std_q) * std_q
            zkv = mean_kv + noise(std_kv) * std_kv
        else:
            zq, zkv = mean_q, mean_kv
        if state is not None:                                             # 243-244
            zq, zkv = state.merge(zq, zkv, mean_q, std_q, mean_kv, std_kv)
        qr = qz = zq
        kz = vz = zkv
        if sh.t_reduce:                                                   # 249-264
            qz = seq_linear(P, pre + "t_qz_upsample", qz, training)
            kz = seq_linear(P, pre + "t_kz_upsample", kz, training)
            vz = seq_linear(P, pre + "t_vz_upsample", vz, training)
            qr = seq_linear(P, pre + "t_qr_proj", qr, training)
            kr = seq_linear(P, pre + "t_kr_proj", kr, training)
    qz = sn_linear(P, pre + "q_proj", qz, training)                       # 265-267
    kz = sn_linear(P, pre + "k_proj", kz, training)
    vz = sn_linear(P, pre + "v_proj", vz, training)
    B, Sq, Skv = qz.shape[0], qz.shape[1], kz.shape[1]
    dq = sh.hd_half if sh.reduce else sh.hd
    q = qz.view(B, Sq, H, dq).transpose(1, 2)
    k = kz.view(B, Skv, H, dq).transpose(1, 2)
    v = vz.view(B, Skv, H, sh.hd).transpose(1, 2)
    if sh.reduce:                                                         # 275-281 decoupled RoPE
        qr = sn_linear(P, pre + "qr_proj", qr, training).view(B, Sq, H, sh.hd_half).transpose(1, 2)
        kr = sn_linear(P, pre + "kr_proj", kr, training).view(B, Skv, H, sh.hd_half).transpose(1, 2)
        q = torch.cat((q, rope(qr, P[pre + "rope_q.inv_freq"])), dim=-1)
        k = torch.cat((k, rope(kr, P[pre + "rope_k.inv_freq"])), dim=-1)
    else:                                                                 # 283-285
        q = rope(q, P[pre + "rope_q.inv_freq"])
        k = rope(k, P[pre + "rope_k.inv_freq"])
    x = attention_with_latent_mask(P, pre, q, k, v, training)
    x = sn_linear(P, pre + "out_proj", x, training) * P[pre + "ls_att"]   # 300
    if residual.shape != x.shape:                                         # 302-308
        if sh.seq_new != sh.seq:
            residual = seq_linear(P, pre + "input_t_proj", residual, training)
        if sh.dim1 != sh.dim2:
            residual = sn_linear(P, pre + "input_proj", residual, training)
    x = x + residual
    y = layer_norm(x, P[pre + "ln_2.weight"])                             # 310-315
    y = gelu_erf(sn_linear(P, pre + "mlp.0", y, training))
    y = sn_linear(P, pre + "mlp.3", y, training) * P[pre + "ls_mlp"]
    return x + y


# ----------------------------------------------------------------------------------
# tokenisation + CNN tail (Vi_Tools:378-403; CALM_ViT_V2.py:60-67,80-83)
# ----------------------------------------------------------------------------------
def rows_from_image(img: Tensor) -> Tensor:
    """[B,3,S,S] -> [B,S,3S]: rows[b,i,3j+c] = img[b,c,i,j] (Vi_Tools:389-391)."""
    b, c, s, s2 = img.shape
    return img.permute(0, 2, 3, 1).reshape(b, s, s2 * c)


def grid_transpose(tok: Tensor) -> Tensor:
    """rows<->columns: out[b,j,3i+c] = tok[b,i,3j+c] (Vi_Tools:394-395,397-398)."""
    b, s, _ = tok.shape
    return tok.reshape(b, s, s, 3).permute(0, 2, 1, 3).reshape(b, s, 3 * s)


def cnn_residual(P, pre: str, tok: Tensor, training: bool) -> Tensor:
    """1x1(3->32) GELU dw3x3(pad 1) GELU 1x1(32->3) on the token grid seen as an image."""
    b, s, _ = tok.shape
    img = tok.reshape(b, s, s, 3).permute(0, 3, 1, 2)
    F = torch.nn.functional
    w0 = spectral_weight(P, pre + "0", training)
    w2 = spectral_weight(P, pre + "2", training)
    w4 = spectral_weight(P, pre + "4", training)
    h = gelu_erf(F.conv2d(img, w0, P[pre + "0.bias"]))
    h = gelu_erf(F.conv2d(h, w2, P[pre + "2.bias"], padding=1, groups=w2.shape[0]))
    h = F.conv2d(h, w4, P[pre + "4.bias"])
    return h.permute(0, 2, 3, 1).reshape(b, s, 3 * s)


# ----------------------------------------------------------------------------------
# Block / EncoderDecoder_8 / ViT
# ----------------------------------------------------------------------------------
def block_shapes(heads, dim1, dim_step, mvh, seq, seq_step, seq_reduce, force_reduce):
    """The three VMLA shapes of one Block (Vi_Tools:337-376)."""
    self_sh = VMLAShape(heads, dim1, dim1, mvh, seq, seq_reduce, seq, force_reduce, False, False)
    cross_sh = VMLAShape(heads, dim1, dim1 + 3 * dim_step, mvh, seq, seq_reduce, seq + 3 * seq_step,
                         force_reduce, False, True)
    return self_sh, cross_sh


def block(P, pre: str, shapes, x: Tensor, first: bool, esm, dsm, csm, training, noise) -> Tensor:
    """Block.forward (Vi_Tools:387-403)."""
    self_sh, cross_sh = shapes
    xq = rows_from_image(x) if first else x
    xq = vmla_block(P, pre + "encoder.", self_sh, xq, None, esm, training, noise)
    xkv = grid_transpose(xq)
    xkv = vmla_block(P, pre + "decoder.", self_sh, xkv, None, dsm, training, noise)
    xkv = grid_transpose(xkv)
    y = vmla_block(P, pre + "cross.", cross_sh, xq, xkv, csm, training, noise)
    return y + cnn_residual(P, pre + "proj.", y, training)


def stage_plan(cfg: ViTConfig):
    """(prefix, dim1, dim_step, seq, seq_step) for the 8 Blocks (Vi_Tools:424-493)."""
    d, s = cfg.in_features, cfg.seq_length
    plan = []
    for i in range(3):
        plan.append((f"encoder_blocks.{i}.", d, -cfg.dim_step, s, -cfg.seq_len_step))
        d -= 3 * cfg.dim_step
        s -= 3 * cfg.seq_len_step
    plan.append(("block_bottle_neck_1.", d, 0, s, 0))
    plan.append(("block_bottle_neck_2.", d, 0, s, 0))
    for i in range(3):
        plan.append((f"decoder_blocks.{i}.", d, cfg.dim_step, s, cfg.seq_len_step))
        d += 3 * cfg.dim_step
        s += 3 * cfg.seq_len_step
    return plan


def encoder_decoder_8(P, pre: str, cfg: ViTConfig, img: Tensor, training: bool,
                      noise: Callable[[Tensor], Tensor] = torch.randn_like):
    """EncoderDecoder_8.forward (Vi_Tools:496-533)."""
    esm = LatentState(mode="sum") if cfg.force_reduce else None
    dsm = LatentState(mode="sum") if cfg.force_reduce else None
    csm = LatentState(mode="sum")
    plan = stage_plan(cfg)

    def run(i, x):
        bp, d, ds, s, ss = plan[i]
        sh = block_shapes(cfg.heads, d, ds, cfg.mean_var_hidden, s, ss, cfg.seq_len_reduce, cfg.force_reduce)
        return block(P, pre + bp, sh, x, i == 0, esm, dsm, csm, training, noise)

    x = run(0, img); skip_1 = x
    x = run(1, x); skip_2 = x
    x = run(2, x); skip_bn_1 = x
    x = run(3, x)
    x = x + skip_bn_1; skip_bn_2 = x                                      # 513-514
    x = run(4, x)
    x = x + (skip_bn_2 + skip_bn_1)                                        # 516
    x = run(5, x) + skip_2                                                 # 519-520
    x = run(6, x) + skip_1                                                 # 521-522
    x = run(7, x)
    x = layer_norm(x, P[pre + "ln_final.weight"])                          # 523
    kl = csm.kl_loss()
    if cfg.force_reduce:                                                   # 532
        kl = esm.kl_loss() + dsm.kl_loss() + kl
    return x, kl


def vit_forward(P, cfg: ViTConfig, img: Tensor, training: bool,
                noise: Callable[[Tensor], Tensor] = torch.randn_like):
    """ViT.forward (CALM_ViT_V2.py:70-84)."""
    x, kl = encoder_decoder_8(P, "autoencoder.", cfg, img, training, noise)
    if not cfg.generate:
        x = x.mean(dim=1)                                                  # AdaptiveAvgPool1d(1) over S
        x = gelu_erf(sn_linear(P, "head.0", x, training))
        x = sn_linear(P, "head.2", x, training)
    else:
        x = x + cnn_residual(P, "proj.", x, training)
    return x, kl


# ----------------------------------------------------------------------------------
# parameter inventory (names + shapes of the reference state dict; used to build P)
# ----------------------------------------------------------------------------------
def _sn(shapes, name, w_shape, bias=False):
    rows = w_shape[0]
    cols = 1
    for d in w_shape[1:]:
        cols *= d
    shapes[name + ".weight_orig"] = tuple(w_shape)
    shapes[name + ".weight_u"] = (rows,)
    shapes[name + ".weight_v"] = (cols,)
    if bias:
        shapes[name + ".bias"] = (rows,)


def vmla_param_shapes(pre: str, sh: VMLAShape, mlp_dim: int):
    """Names/shapes created by VMLA_Block.__init__ (Vi_Tools:117-205)."""
    s: Dict[str, tuple] = {}
    H = sh.heads
    s[pre + "ls_att"] = (sh.dim2,)
    s[pre + "ls_mlp"] = (sh.dim2,)
    s[pre + "ln_q.weight"] = (sh.dim1,)
    if sh.is_cross:
        s[pre + "ln_kv.weight"] = (sh.dim1,)
    if sh.t_reduce:
        _sn(s, pre + "t_encoder_q", (sh.seq_reduce, sh.seq))
        _sn(s, pre + "t_encoder_kv", (sh.seq_reduce, sh.seq))
    if sh.reduce:
        _sn(s, pre + "encoder_q", (2 * sh.mvh, sh.dim1))
        _sn(s, pre + "encoder_kv", (2 * sh.mvh, sh.dim1))
    if sh.t_reduce:
        for n in ("t_qz_upsample", "t_kz_upsample", "t_vz_upsample", "t_qr_proj"):
            _sn(s, pre + n, (sh.seq_new, sh.seq_reduce))
        _sn(s, pre + "t_kr_proj", (sh.seq_new, sh.seq))
    plain = sh.dim1 == sh.dim2 and not sh.force_reduce
    d_in = sh.dim2 if plain else sh.mvh
    d_qk = H * sh.hd_half if sh.reduce else H * sh.hd
    _sn(s, pre + "q_proj", (d_qk, d_in))
    _sn(s, pre + "k_proj", (d_qk, d_in))
    _sn(s, pre + "v_proj", (sh.dim2, d_in))
    if sh.reduce:
        _sn(s, pre + "qr_proj", (H * sh.hd_half, sh.mvh))
        _sn(s, pre + "kr_proj", (H * sh.hd_half, sh.dim1))
    if sh.seq_new != sh.seq:
        _sn(s, pre + "input_t_proj", (sh.seq_new, sh.seq))
    if sh.dim1 != sh.dim2:
        _sn(s, pre + "input_proj", (sh.dim2, sh.dim1))
    d_rope = sh.hd_half if sh.reduce else sh.hd
    s[pre + "rope_q.inv_freq"] = (d_rope // 2,)
    s[pre + "rope_k.inv_freq"] = (d_rope // 2,)
    _sn(s, pre + "linear_mask.0", (2 * sh.seq_new, sh.seq_new), bias=True)
    _sn(s, pre + "linear_mask.2", (sh.seq_new, 2 * sh.seq_new), bias=True)
    _sn(s, pre + "out_proj", (sh.dim2, sh.dim2))
    s[pre + "ln_2.weight"] = (sh.dim2,)
    _sn(s, pre + "mlp.0", (mlp_dim, sh.dim2))
    _sn(s, pre + "mlp.3", (sh.dim2, mlp_dim))
    return s


def cnn_param_shapes(pre: str, hidden: int = 32):
    s: Dict[str, tuple] = {}
    _sn(s, pre + "0", (hidden, 3, 1, 1), bias=True)
    _sn(s, pre + "2", (hidden, 1, 3, 3), bias=True)
    _sn(s, pre + "4", (3, hidden, 1, 1), bias=True)
    return s


def vit_param_shapes(cfg: ViTConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    for bp, d, ds, sq, ss in stage_plan(cfg):
        pre = "autoencoder." + bp
        self_sh, cross_sh = block_shapes(cfg.heads, d, ds, cfg.mean_var_hidden, sq, ss,
                                         cfg.seq_len_reduce, cfg.force_reduce)
        s.update(vmla_param_shapes(pre + "encoder.", self_sh, 2 * d))
        s.update(vmla_param_shapes(pre + "decoder.", self_sh, 2 * d))
        s.update(vmla_param_shapes(pre + "cross.", cross_sh, 2 * (d + 3 * ds)))
        s.update(cnn_param_shapes(pre + "proj."))
    s["autoencoder.ln_final.weight"] = (cfg.in_features,)
    if not cfg.generate:
        _sn(s, "head.0", (2 * cfg.in_features, cfg.in_features))
        _sn(s, "head.2", (cfg.out_features, 2 * cfg.in_features))
    else:
        s.update(cnn_param_shapes("proj."))
    return s


BUFFER_SUFFIXES = (".weight_u", ".weight_v")


def is_buffer(name: str) -> bool:
    return name.endswith(BUFFER_SUFFIXES)
