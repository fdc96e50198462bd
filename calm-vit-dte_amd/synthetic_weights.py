"""Deterministic, torch-RNG-independent synthetic weights for fixtures, tests and bench.

Every tensor is drawn from its own numpy Generator seeded by (seed, crc32(name)), so the
result does not depend on iteration order and can be regenerated anywhere (the GPU box
included) from the name->shape inventory alone.  Used by bench.py (random-init weights of the
benchmarked architecture, synthetic batches), and — through tests/golden/weights.py — by make_golden.py
(to fill the imported reference) and the parity tests.
"""
import zlib

import numpy as np


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def make_tensor(name: str, shape, seed: int) -> np.ndarray:
    g = _rng(seed, name)
    shape = tuple(shape)
    if name.endswith(".weight_orig"):
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        x = g.uniform(-b, b, size=shape)
    elif name.endswith((".weight_u", ".weight_v")):
        x = g.standard_normal(shape)
        x = x / max(np.linalg.norm(x), 1e-12)
    elif name.endswith(".bias"):
        x = 0.05 * g.standard_normal(shape)
    elif name.endswith("inv_freq"):
        half = shape[0]
        d = 2 * half
        base = 1.0 / (10000.0 ** (np.arange(0, d, 2, dtype=np.float64) / d))
        x = base * (1.0 + 0.05 * g.standard_normal(shape))
    elif name.endswith(("ls_att", "ls_mlp")) or (name.endswith(".weight") and "ln_" in name):
        x = 1.0 + 0.1 * g.standard_normal(shape)
    else:
        raise KeyError(f"no init rule for {name}")
    return np.ascontiguousarray(x, dtype=np.float32)


def make_params(shapes: dict, seed: int) -> dict:
    return {k: make_tensor(k, v, seed) for k, v in shapes.items()}


def make_input(shape, seed: int, name: str = "input") -> np.ndarray:
    return _rng(seed, name).standard_normal(tuple(shape)).astype(np.float32)


class NoiseStream:
    """Stand-in for torch.randn_like with numpy-seeded draws (call order = reference order:
    per reducing block q-noise then kv-noise, Vi_Tools_CNN_less_V2.py:238-239)."""

    def __init__(self, seed: int):
        self.seed = seed
        self.n = 0

    def __call__(self, like):
        import torch
        x = _rng(self.seed, f"noise{self.n}").standard_normal(tuple(like.shape)).astype(np.float32)
        self.n += 1
        return torch.from_numpy(x).to(device=like.device, dtype=like.dtype)
