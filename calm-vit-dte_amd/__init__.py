"""calm-vit-dte_amd: MI355X-native CALM-ViT cross-axial latent-masking attention path.

Drop-in use (mirrors `import CALM_ViT_V2 as rvh` of distributed_trainer_cls.py:8):
    import calm_vit_dte_amd as calm            # repo-root shim for the hyphenated directory
    model = calm.CALM_ViT_V2.ViT(device, type=8, heads=12, seq_length=224, ...)
"""
from . import backend, ops, spectral_norm, synthetic_weights  # noqa: F401
from . import Vi_Tools_CNN_less_V2, CALM_ViT_V2  # noqa: F401
from .CALM_ViT_V2 import ViT  # noqa: F401
from .build import build_library  # noqa: F401

__all__ = ["ViT", "CALM_ViT_V2", "Vi_Tools_CNN_less_V2", "ops", "backend", "build_library"]
