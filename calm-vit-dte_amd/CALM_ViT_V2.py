"""MI355X-native drop-in for the reference's `CALM_ViT_V2.ViT` (CALM_ViT_V2.py:21-84).

`ViT(device, type=8, heads=..., ...)` keeps the reference constructor/forward signatures,
attribute names (`autoencoder`, `pool`, `head` / `proj`) and state-dict keys, so
`distributed_trainer_cls.py:123-126,148-152` can build it unchanged.  Dataset / sample-saving /
`__main__` code of the reference file is out of scope.
"""
import torch

from . import Vi_Tools_CNN_less_V2 as vt
from . import ops
from .spectral_norm import SNLinear, sn_scope


class SequencePool(torch.nn.Module):
    """Stands where the reference keeps `AdaptiveAvgPool1d(1)` (CALM_ViT_V2.py:48); the pooling over
    the sequence axis is done directly on [B,S,D] (no permute)."""

    def forward(self, x):
        return ops.mean_seq(x)


class ViT(torch.nn.Module):
    def __init__(self, device, type=8, heads=12, seq_length=256, in_features=768,
                 dim_step=48, mean_var_hidden=192,
                 seq_len_step=16, seq_len_reduce=128, out_features=1000,
                 force_reduce=False, generate=True):
        super().__init__()
        self.device = device
        self.generate = generate
        self.num_classes = out_features
        self.seq_length = seq_length
        if type == 8:                                                     # CALM_ViT_V2.py:35-46
            self.autoencoder = vt.EncoderDecoder_8(
                heads=heads,
                dim1=in_features,
                dim_step=dim_step,
                mean_var_hidden=mean_var_hidden,
                seq_length=seq_length,
                seq_len_step=seq_len_step,
                seq_len_reduce=seq_len_reduce,
                out_features_override=None,
                force_reduce=force_reduce,
            ).to(device)
        if not generate:                                                  # 47-53
            self.pool = SequencePool().to(device)
            self.head = torch.nn.Sequential(
                SNLinear(in_features, in_features * 2, bias=False),
                vt.GELU(),
                SNLinear(in_features * 2, out_features, bias=False),
            ).to(device)
        else:                                                             # 60-67
            self.proj = vt.CnnResidual(32)

    def forward(self, q):
        with sn_scope(self):
            x, kl_loss = self.autoencoder(q)
            if not self.generate:                                         # 71-76
                x = self.pool(x)
                h0, h2 = self.head[0], self.head[2]
                x = ops.MlpFn.apply(x, h0.weight_orig, None, h2.weight_orig, None, None, None,
                                    h0.weight_u, h0.weight_v, h0.sigma(), h2.weight_u, h2.weight_v, h2.sigma())
            else:                                                         # 78-83
                x = self.proj.residual_forward(x)
            return x, kl_loss
