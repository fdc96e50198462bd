"""Shared helpers for the test-suite (fixtures loading, parameter dicts, error metrics)."""
import json
import os

import numpy as np
import torch

import weights as W
from oracle import calm_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WEIGHT_SEED = 1234

CONFIGS = {
    "nano48_cls": O.ViTConfig(heads=3, seq_length=48, in_features=144, dim_step=12, mean_var_hidden=24,
                              seq_len_step=4, seq_len_reduce=16, out_features=10, force_reduce=False, generate=False),
    "nano48_gen": O.ViTConfig(heads=3, seq_length=48, in_features=144, dim_step=12, mean_var_hidden=24,
                              seq_len_step=4, seq_len_reduce=16, out_features=10, force_reduce=False, generate=True),
    "tiny32_cls": O.ViTConfig(heads=4, seq_length=32, in_features=96, dim_step=0, mean_var_hidden=24,
                              seq_len_step=0, seq_len_reduce=16, out_features=10, force_reduce=False, generate=False),
    "tiny32_fr": O.ViTConfig(heads=4, seq_length=32, in_features=96, dim_step=0, mean_var_hidden=24,
                             seq_len_step=0, seq_len_reduce=16, out_features=10, force_reduce=True, generate=False),
    "small224_cls": O.ViTConfig(heads=6, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=120,
                                seq_len_step=16, seq_len_reduce=40, out_features=1000, force_reduce=False, generate=False),
    "base224_cls": O.ViTConfig(heads=12, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=240,
                               seq_len_step=16, seq_len_reduce=80, out_features=1000, force_reduce=False, generate=False),
}


def load_inventory(name):
    with open(os.path.join(GOLDEN, f"state_dict_{name}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))


def oracle_params(name, golden=None, requires_grad=True):
    """Flat param dict for the oracle: numpy-seeded weights, warm u/v from the fixture."""
    shapes = O.vit_param_shapes(CONFIGS[name])
    P = {k: torch.from_numpy(v) for k, v in W.make_params(shapes, WEIGHT_SEED).items()}
    if golden is not None:
        for k in P:
            if O.is_buffer(k):
                P[k] = torch.from_numpy(golden["warm/" + k].copy())
    if requires_grad:
        for k in P:
            if not O.is_buffer(k):
                P[k].requires_grad_(True)
    return P


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)
