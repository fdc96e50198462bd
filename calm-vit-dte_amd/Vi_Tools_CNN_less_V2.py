"""MI355X-native drop-in for the reference's `Vi_Tools_CNN_less_V2` module.

Same class names, constructor signatures, forward signatures, attribute names and state-dict keys
as /root/reference/CALM-ViT/Vi_Tools_CNN_less_V2.py (cited file:line below), but every tensor
operation of the forward/backward path runs in hand-written HIP kernels (ops.py -> backend.py ->
libcalmvit_hip.so).  CUDA(HIP) tensors only; there is no CPU fallback.
"""
from functools import partial
from typing import Callable

import torch

from . import ops
from .backend import ACT_GELU, ACT_NONE  # noqa: F401
from .spectral_norm import SNConv2d, SNLinear, sn_scope


class ResidualStateManager():
    """Latent state carried from one reducing block to the next (Vi_Tools:7-50): every block folds its sampled
    (zq, zkv) into the running state and reads the folded state back, and the KL terms of all blocks are summed and
    averaged over the blocks seen.

    The path only ever builds it with mode="sum" (Vi_Tools:497-499): state += z, on the device through ops.add.  The
    other modes of the reference's constructor are kept with the same arithmetic, expressed as one rule — the new
    state is `w_new * z + w_old * state` with per-mode weights — and run as plain tensor expressions (they are not
    on the accelerated path):
        "sum", "sma"      w_new = w_old = 1                  ("sma" hands out state / count)
        "ema"             w_new = smooth_factor / (count + 1)
        "lp"              w_new = count / (count + 1)
        anything else     w_new = momentum (fixed)           and w_old = 1 - w_new for these three
    """

    def __init__(self, smooth_factor: float = 2.0, momentum: float = 0.9, mode: str = "ema"):
        super().__init__()
        self.zq_sum = None
        self.zkv_sum = None
        self.tot_kl_loss = 0.0
        self.count = 0
        self.smooth_factor = smooth_factor
        self.mode = mode
        self.momentum = momentum

    @staticmethod
    def _kl(mean, std):
        # KL(N(mean, std) || N(0, 1)) averaged over elements (Vi_Tools:24-25; "var" there is the softplus output used
        # as a standard deviation) — for callers that did not get the term from the fused latent kernel
        return -0.5 * torch.mean(1 + 2 * torch.log(std) - mean.pow(2) - std.pow(2))

    def _blend_weight(self):
        """Weight of the incoming sample for the averaging modes, after `count` was advanced."""
        if self.mode == "ema":
            self.momentum = self.smooth_factor / (self.count + 1)
        elif self.mode == "lp":
            self.momentum = self.count / (self.count + 1)
        return self.momentum

    def get_sums(self, zq, zkv, mean_q, var_q, mean_kv, var_kv, kl_q=None, kl_kv=None):
        if kl_q is None:
            kl_q = self._kl(mean_q, var_q)
        if kl_kv is None:
            kl_kv = self._kl(mean_kv, var_kv)
        self.tot_kl_loss = kl_q + kl_kv + self.tot_kl_loss
        first = self.zq_sum is None
        self.count = 1 if first else self.count + 1
        if first:
            self.zq_sum, self.zkv_sum = zq, zkv
        elif self.mode in ("sum", "sma"):
            self.zq_sum, self.zkv_sum = ops.add(self.zq_sum, zq), ops.add(self.zkv_sum, zkv)
        else:
            w = self._blend_weight()
            self.zq_sum = w * zq + (1 - w) * self.zq_sum
            self.zkv_sum = w * zkv + (1 - w) * self.zkv_sum
        if self.mode == "sma" and not first:
            return self.zq_sum / self.count, self.zkv_sum / self.count
        return self.zq_sum, self.zkv_sum

    def get_kl_loss(self):
        return self.tot_kl_loss / self.count if self.count > 0 else 0.0


class LayerNorm(torch.nn.Module):
    """nn.LayerNorm(dim, eps, bias=False) with the reference's `weight` key (Vi_Tools:115,131)."""

    def __init__(self, dim, eps=1e-6, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("the CALM-ViT path only uses LayerNorm(bias=False)")
        self.weight = torch.nn.Parameter(torch.ones(dim))
        self.eps = eps
        self.normalized_shape = (dim,)

    def forward(self, x, gemm_only=False):
        return ops.LayerNormFn.apply(x, self.weight, self.eps, gemm_only)

    def with_skip(self, x):
        """(LayerNorm(x), x) — take the second value for the residual path that bypasses the norm: the two input
        gradients are then summed inside the LayerNorm backward kernel (ops.LayerNormSkipFn)."""
        if not (ops.FUSE_LN_SKIP and torch.is_grad_enabled() and x.requires_grad):
            return self.forward(x, gemm_only=True), x
        return ops.LayerNormSkipFn.apply(x, self.weight, self.eps)


def _default_norm(dim, bias=False):
    return LayerNorm(dim, eps=1e-6, bias=bias)


class RoPE(torch.nn.Module):
    """Learned-frequency 1-D RoPE (Vi_Tools:55-95).  The blocks apply it through ops.RopeFn in the
    token layout; forward() keeps the reference's [B,H,S,d] calling convention."""

    def __init__(self, seq: int, dim: int, theta: float = 10000.0, learned: bool = False, training: bool = True):
        super().__init__()
        self.seq = seq
        self.dim = dim
        self.theta = theta
        self.learned = learned
        inv_freq = 1.0 / (self.theta ** (torch.arange(0, dim, 2).float() / self.dim))
        t = torch.arange(self.seq, dtype=torch.float)
        if learned:
            self.inv_freq = torch.nn.Parameter(inv_freq, requires_grad=True)
        else:
            self.register_buffer("inv_freq", inv_freq)
        self.register_buffer("t", t, persistent=False)

    def forward(self, x):
        B, H, S, d = x.shape
        xt = x.transpose(1, 2).reshape(B, S, H * d)
        y = ops.RopeFn.apply(None, xt, self.inv_freq, H, False)
        return y.view(B, S, H, d).transpose(1, 2)


class GELU(torch.nn.Module):
    """nn.GELU(approximate='none') of the Sequential containers (keeps `mlp.0/mlp.3`, `linear_mask.0/.2`,
    `proj.0/.2/.4`, `head.0/.2` key numbering).  On the path the activation is a GEMM epilogue (or lives inside the fused
    CNN tail) and this module is never called; called on its own it runs the stand-alone erf-GELU kernel."""

    def forward(self, x):
        return ops.GeluFn.apply(x)


class VMLA_Block(torch.nn.Module):
    """Multi-head latent-distribution attention block (Vi_Tools:98-315)."""

    def __init__(
        self,
        heads: int,
        dim1: int,
        dim2: int,
        mean_var_hidden: int,
        seq_length: int,
        seq_len_reduce: int,
        seq_len_new: int,
        mlp_dim: int,
        force_reduce: bool = True,
        t_force_reduce: bool = False,
        dropout: float = 0.0,
        use_mlp: bool = True,
        is_cross: bool = False,
        training: bool = True,
        norm_layer: Callable[..., torch.nn.Module] = _default_norm,
    ):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is not on the reference's training path (always 0.0)")
        self.ls_att = torch.nn.Parameter(torch.ones(dim2), requires_grad=True)
        self.ls_mlp = torch.nn.Parameter(torch.ones(dim2), requires_grad=True) if use_mlp else None
        self.training = training
        self.heads = heads
        self.head_dim_content = dim2 // heads // 2
        self.head_dim_rope = dim2 // heads // 2
        self.head_dim = self.head_dim_content + self.head_dim_rope
        self.t_reduce = seq_len_new != seq_length or t_force_reduce
        self.reduce = dim1 != dim2 or force_reduce
        self.ln_q = LayerNorm(dim1)
        self.ln_kv = LayerNorm(dim1) if is_cross else None
        self.t_encoder_q = None
        self.t_encoder_kv = None
        if self.t_reduce:
            self.t_encoder_q = SNLinear(seq_length, seq_len_reduce)
            self.t_encoder_kv = SNLinear(seq_length, seq_len_reduce)
        self.encoder_q = None
        self.encoder_kv = None
        if self.reduce:
            self.encoder_q = SNLinear(dim1, mean_var_hidden * 2)
            self.encoder_kv = SNLinear(dim1, mean_var_hidden * 2)
        self.t_qz_upsample = None
        self.t_kz_upsample = None
        self.t_vz_upsample = None
        self.t_qr_proj = None
        self.t_kr_proj = None
        if self.t_reduce:
            self.t_qz_upsample = SNLinear(seq_len_reduce, seq_len_new)
            self.t_kz_upsample = SNLinear(seq_len_reduce, seq_len_new)
            self.t_vz_upsample = SNLinear(seq_len_reduce, seq_len_new)
            self.t_qr_proj = SNLinear(seq_len_reduce, seq_len_new)
            self.t_kr_proj = SNLinear(seq_length, seq_len_new)
        self.qz_upsample = None
        self.kz_upsample = None
        self.vz_upsample = None
        d_in = dim2 if dim1 == dim2 and not force_reduce else mean_var_hidden
        d_qk = (self.heads * self.head_dim_content) if self.reduce else (self.heads * self.head_dim)
        self.q_proj = SNLinear(d_in, d_qk)
        self.k_proj = SNLinear(d_in, d_qk)
        self.v_proj = SNLinear(d_in, dim2)
        self.qr_proj = None
        self.kr_proj = None
        if self.reduce:
            self.qr_proj = SNLinear(mean_var_hidden, self.head_dim_rope * self.heads)
            self.kr_proj = SNLinear(dim1, self.head_dim_rope * self.heads)
        self.input_t_proj = None
        self.input_proj = None
        if seq_len_new != seq_length:
            self.input_t_proj = SNLinear(seq_length, seq_len_new)
        if dim1 != dim2:
            self.input_proj = SNLinear(dim1, dim2)
        d_rope = self.head_dim_rope if self.reduce else self.head_dim
        self.rope_q = RoPE(seq_len_new, d_rope, learned=True)
        self.rope_k = RoPE(seq_len_new, d_rope, learned=True)
        self.linear_mask = torch.nn.Sequential(
            SNLinear(seq_len_new, seq_len_new * 2, bias=True),
            GELU(),
            SNLinear(seq_len_new * 2, seq_len_new, bias=True),
        )
        self.out_proj = SNLinear(dim2, dim2)
        self.out_proj.layer_scaled = True            # combined with ls_att in one epilogue (see FusedClipAdamW)
        self.dropout = torch.nn.Dropout(dropout)
        self.ln_2 = LayerNorm(dim2)
        self.mlp = None
        if use_mlp:
            self.mlp = torch.nn.Sequential(
                SNLinear(dim2, mlp_dim),
                GELU(),
                torch.nn.Dropout(dropout, inplace=False),
                SNLinear(mlp_dim, dim2),
            )
            self.mlp[3].layer_scaled = True          # combined with ls_mlp

    @staticmethod
    def _seq(lin, x):
        return ops.SeqLinearFn.apply(x, lin.weight_orig, lin.weight_u, lin.weight_v, lin.sigma())

    def forward(self, input_q, input_kv=None, state_manager=None, mask=False):
        if not mask:
            # the reference dereferences mask_mat unconditionally (Vi_Tools:291) -> AttributeError
            raise AttributeError("'NoneType' object has no attribute 'unsqueeze' (mask=False is not a usable "
                                 "path of the reference; every caller passes mask=True)")
        with sn_scope(self):
            return self._forward(input_q, input_kv, state_manager)

    def _project_qkv(self, qz, kz, vz, out16=False):
        """q_proj / k_proj / v_proj (Vi_Tools:265-267).  Projections that read the SAME tensor — all three in a plain
        self-attention block, k and v whenever the key and value inputs coincide — run as one grouped launch."""
        pq, pk, pv = self.q_proj, self.k_proj, self.v_proj
        pack = lambda *ps: [t for p_ in ps for t in (p_.weight_orig, p_.weight_u, p_.weight_v, p_.sigma())]
        same_shape = pq.weight_orig.shape == pk.weight_orig.shape == pv.weight_orig.shape
        if any(p_.bias is not None for p_ in (pq, pk, pv)) or not same_shape or not ops.GROUP_PROJECTIONS:
            return pq(qz, out16=out16), pk(kz, out16=out16), pv(vz, out16=out16)
        if qz is kz and kz is vz:
            return ops.SNLinearGroupFn.apply(qz, out16, *pack(pq, pk, pv))
        if kz is vz:
            k, v = ops.SNLinearGroupFn.apply(kz, out16, *pack(pk, pv))
            return pq(qz, out16=out16), k, v
        return pq(qz, out16=out16), pk(kz, out16=out16), pv(vz, out16=out16)

    def _forward(self, input_q, input_kv, state_manager):
        H = self.heads
        xq, residual = self.ln_q.with_skip(input_q)                      # 209-215 (residual = input_q)
        xkv = xq if input_kv is None else self.ln_kv(input_kv, gemm_only=True)
        qz = qr = xq
        kz = vz = kr = xkv
        if self.reduce:
            if self.t_reduce:                                            # 224-229
                xq = self._seq(self.t_encoder_q, xq)
                xkv = self._seq(self.t_encoder_kv, xkv)
            mv_q = self.encoder_q(xq)                                    # 230-231
            mv_kv = self.encoder_kv(xkv)
            mvh = mv_q.shape[-1] // 2
            nq = ops.draw_noise(mv_q[..., :mvh]) if self.training else None      # 237-239 (q first)
            nkv = ops.draw_noise(mv_kv[..., :mvh]) if self.training else None
            zq, std_q, kl_q = ops.LatentFn.apply(mv_q, nq)               # 232-242 + KL of 24-25
            zkv, std_kv, kl_kv = ops.LatentFn.apply(mv_kv, nkv)
            if state_manager is not None:                                # 243-244
                zq, zkv = state_manager.get_sums(zq, zkv, mv_q[..., :mvh], std_q, mv_kv[..., :mvh], std_kv,
                                                 kl_q=kl_q, kl_kv=kl_kv)
            qr = qz = zq
            kz = vz = zkv
            if self.t_reduce:                                            # 249-264
                qz = self._seq(self.t_qz_upsample, qz)
                kz = self._seq(self.t_kz_upsample, kz)
                vz = self._seq(self.t_vz_upsample, vz)
                qr = self._seq(self.t_qr_proj, qr)
                kr = self._seq(self.t_kr_proj, kr)
        # bf16 pipeline: q, k, v are bf16 from the projections through RoPE into the bf16 attention kernels
        a16 = qz.shape[1] == kz.shape[1] and ops.use_attention16(qz.shape[1], H, self.head_dim)
        qz, kz, v = self._project_qkv(qz, kz, vz, out16=a16)             # 265-267
        if self.reduce:                                                  # 275-281 decoupled RoPE
            qr = self.qr_proj(qr, out16=a16)
            kr = self.kr_proj(kr, out16=a16)
            q = ops.RopeFn.apply(qz, qr, self.rope_q.inv_freq, H, a16)
            k = ops.RopeFn.apply(kz, kr, self.rope_k.inv_freq, H, a16)
        else:                                                            # 283-285
            q = ops.RopeFn.apply(None, qz, self.rope_q.inv_freq, H, a16)
            k = ops.RopeFn.apply(None, kz, self.rope_k.inv_freq, H, a16)
        m0, m2 = self.linear_mask[0], self.linear_mask[2]
        attention = ops.LatentMaskAttention16Fn if a16 else ops.LatentMaskAttentionFn
        x = attention.apply(                                             # 288-299
            q, k, v, m0.weight_orig, m0.bias, m2.weight_orig, m2.bias,
            m0.weight_u, m0.weight_v, m0.sigma(), m2.weight_u, m2.weight_v, m2.sigma(), H)
        if residual.shape != (x.shape[0], x.shape[1], self.out_proj.out_features):   # 302-308
            if self.input_t_proj is not None:
                residual = self._seq(self.input_t_proj, residual)
            if self.input_proj is not None:
                residual = self.input_proj(residual)
        x = self.out_proj(x, ls=self.ls_att, residual=residual)          # 300, 309
        if self.mlp is None:                                             # 310-315
            return self.ln_2(x)                                          # block output: stays fp32
        y, x = self.ln_2.with_skip(x)
        l0, l3 = self.mlp[0], self.mlp[3]
        return ops.MlpFn.apply(y, l0.weight_orig, None, l3.weight_orig, None, self.ls_mlp, x,
                               l0.weight_u, l0.weight_v, l0.sigma(), l3.weight_u, l3.weight_v, l3.sigma())


class CnnResidual(torch.nn.Sequential):
    """`proj` of Block / ViT: sn(Conv1x1 3->32) GELU sn(dwConv3x3) GELU sn(Conv1x1 32->3)
    (Vi_Tools:378-385; CALM_ViT_V2.py:60-67) — keys proj.0 / proj.2 / proj.4."""

    def __init__(self, hidden_channels=32):
        super().__init__(
            SNConv2d(3, hidden_channels, kernel_size=1, groups=1, bias=True),
            GELU(),
            SNConv2d(hidden_channels, hidden_channels, kernel_size=3, padding=1, bias=True, groups=hidden_channels),
            GELU(),
            SNConv2d(hidden_channels, 3, kernel_size=1, bias=True),
        )

    def residual_forward(self, tokens, residual=True):
        """tokens [B,S,3S] -> tokens + proj(tokens as [B,3,S,S] image), back in token layout."""
        c0, c2, c4 = self[0], self[2], self[4]
        return ops.CnnResidualFn.apply(tokens, c0.weight_orig, c0.bias, c2.weight_orig, c2.bias, c4.weight_orig,
                                       c4.bias, c0.weight_u, c0.weight_v, c0.sigma(), c2.weight_u, c2.weight_v,
                                       c2.sigma(), c4.weight_u, c4.weight_v, c4.sigma(), residual)

    def forward(self, x):
        """`proj(img)` as the reference's Sequential computes it (Vi_Tools:378-385): img [B,3,S,S] -> [B,3,S,S], no skip
        connection — the same fused kernel with its residual term switched off, between two index-only layout changes."""
        with sn_scope(self):
            y = self.residual_forward(ops.image_to_rows(x), residual=False)
            return ops.RowsToImageFn.apply(y)


class Block(torch.nn.Module):
    """Row self-attention -> column self-attention -> cross attention -> CNN residual (Vi_Tools:317-403)."""

    def __init__(
        self,
        heads: int,
        dim1: int,
        dim_step: int,
        mean_var_hidden: int,
        seq_length: int,
        seq_len_step: int,
        is_first_block: bool,
        is_last_block: bool,
        seq_len_reduce: int,
        force_reduce: bool = False,
        training: bool = True,
        use_ape: bool = True,
        norm_layer: Callable[..., torch.nn.Module] = _default_norm,
        out_features_override: int = None,
    ):
        super().__init__()
        self.is_first_block = is_first_block
        common = dict(heads=heads, dim1=dim1, mean_var_hidden=mean_var_hidden, seq_length=seq_length,
                      seq_len_reduce=seq_len_reduce, force_reduce=force_reduce, training=training, use_mlp=True)
        self.encoder = VMLA_Block(dim2=dim1, seq_len_new=seq_length, mlp_dim=dim1 * 2, **common)
        self.decoder = VMLA_Block(dim2=dim1, seq_len_new=seq_length, mlp_dim=dim1 * 2, **common)
        self.cross = VMLA_Block(
            dim2=(dim1 + (dim_step * 3)) if out_features_override is None else out_features_override,
            seq_len_new=seq_length + (seq_len_step * 3),
            mlp_dim=(dim1 + (dim_step * 3)) * 2,
            is_cross=True, **common)
        self.proj = CnnResidual(32)

    def forward(self, x, esm=None, dsm=None, csm=None, mask=True):
        with sn_scope(self):
            xq = x
            if self.is_first_block and x.dim() == 4:
                xq = ops.image_to_rows(xq)                               # 389-391
            # (a 3-d input to the first block IS the row-token tensor [B,S,3S]: what trainer.DeviceCollate(tokens=True)
            # emits straight from the uint8 batch — SURVEY 8f-3 "feeding K1 directly")
            xq = self.encoder(xq, state_manager=esm, mask=mask)
            xkv = ops.grid_transpose(xq)                                 # 394-395
            xkv = self.decoder(xkv, state_manager=dsm, mask=mask)
            xkv = ops.grid_transpose(xkv)                                # 397-398
            x = self.cross(xq, input_kv=xkv, state_manager=csm, mask=mask)
            return self.proj.residual_forward(x)                         # 400-403


class EncoderDecoder_8(torch.nn.Module):
    """3 down Blocks, 2 bottlenecks, 3 up Blocks with U-net skips (Vi_Tools:407-533)."""

    def __init__(
        self,
        heads: int = 12,
        dim1: int = 768,
        dim_step: int = 48,
        mean_var_hidden: int = 192,
        seq_length: int = 256,
        seq_len_step: int = 16,
        seq_len_reduce: int = 128,
        out_features_override: int = None,
        force_reduce: bool = False,
        training: bool = True,
        norm_layer: Callable[..., torch.nn.Module] = _default_norm,
    ):
        super().__init__()
        self.force_reduce = force_reduce
        kw = dict(heads=heads, mean_var_hidden=mean_var_hidden, seq_len_reduce=seq_len_reduce,
                  force_reduce=force_reduce, training=training)
        self.encoder_blocks = torch.nn.ModuleList()
        for i in range(3):
            self.encoder_blocks.append(Block(dim1=dim1, dim_step=-dim_step, is_first_block=(i == 0),
                                             is_last_block=False, seq_length=seq_length,
                                             seq_len_step=-seq_len_step, out_features_override=None, **kw))
            dim1 -= (dim_step * 3)
            seq_length -= (seq_len_step * 3)
        self.block_bottle_neck_1 = Block(dim1=dim1, dim_step=0, is_first_block=False, is_last_block=False,
                                         seq_length=seq_length, seq_len_step=0, out_features_override=None, **kw)
        self.block_bottle_neck_2 = Block(dim1=dim1, dim_step=0, is_first_block=False, is_last_block=False,
                                         seq_length=seq_length, seq_len_step=0, out_features_override=None, **kw)
        self.decoder_blocks = torch.nn.ModuleList()
        for i in range(3):
            self.decoder_blocks.append(Block(dim1=dim1, dim_step=dim_step, is_first_block=False,
                                             is_last_block=(i == 2), seq_length=seq_length,
                                             seq_len_step=seq_len_step,
                                             out_features_override=out_features_override if i == 2 else None, **kw))
            dim1 += (dim_step * 3)
            seq_length += (seq_len_step * 3)
        self.ln_final = LayerNorm(dim1)

    def forward(self, x):
        with sn_scope(self):
            esm = ResidualStateManager(mode="sum") if self.force_reduce else None
            dsm = ResidualStateManager(mode="sum") if self.force_reduce else None
            csm = ResidualStateManager(mode="sum")
            skip_1 = skip_2 = skip_bn_1 = skip_bn_2 = None
            for i, block in enumerate(self.encoder_blocks):
                x = block(x, esm=esm, dsm=dsm, csm=csm, mask=True)
                if i == 0:
                    skip_1 = x
                elif i == 1:
                    skip_2 = x
                else:
                    skip_bn_1 = x
            x = self.block_bottle_neck_1(x, esm=esm, dsm=dsm, csm=csm, mask=True)
            x = ops.add(x, skip_bn_1)                                    # 513
            skip_bn_2 = x
            x = self.block_bottle_neck_2(x, esm=esm, dsm=dsm, csm=csm, mask=True)
            x = ops.add(x, ops.add(skip_bn_2, skip_bn_1))                # 516
            for i, block in enumerate(self.decoder_blocks):
                x = block(x, esm=esm, dsm=dsm, csm=csm, mask=True)
                if i == 0:
                    x = ops.add(x, skip_2)                               # 520
                elif i == 1:
                    x = ops.add(x, skip_1)                               # 522
            x = self.ln_final(x)                                         # 523
            kl_loss = csm.get_kl_loss()
            kl_loss = esm.get_kl_loss() + dsm.get_kl_loss() + kl_loss if self.force_reduce else kl_loss
            return x, kl_loss
