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
    "base384_cls": O.ViTConfig(heads=12, seq_length=384, in_features=1152, dim_step=48, mean_var_hidden=240,
                               seq_len_step=16, seq_len_reduce=80, out_features=1000, force_reduce=False, generate=False),
    "large224_cls": O.ViTConfig(heads=6, seq_length=224, in_features=672, dim_step=24, mean_var_hidden=480,
                                seq_len_step=8, seq_len_reduce=160, out_features=1000, force_reduce=False, generate=False),
}
# BASELINE.json configs #3-#5: fixtures minted from the reference at batch 1 (golden_<name>_b1.npz)
REAL_SIZE_CFGS = ["small224_cls", "base224_cls", "base384_cls", "large224_cls"]
# single reference VMLA_Blocks at the real head dims (golden_block_<name>.npz; kwargs as in make_golden.BLOCKS)
BLOCK_FIXTURES = {
    "A_hd56": dict(heads=12, dim1=672, dim2=672, mean_var_hidden=240, seq_length=224, seq_len_reduce=80,
                   seq_len_new=224, is_cross=False),
    "B_hd44": dict(heads=12, dim1=672, dim2=528, mean_var_hidden=240, seq_length=224, seq_len_reduce=80,
                   seq_len_new=176, is_cross=True),
    "A_hd32": dict(heads=12, dim1=384, dim2=384, mean_var_hidden=240, seq_length=128, seq_len_reduce=80,
                   seq_len_new=128, is_cross=False),
    "B_hd20": dict(heads=12, dim1=384, dim2=240, mean_var_hidden=240, seq_length=128, seq_len_reduce=80,
                   seq_len_new=80, is_cross=True),
}
BLOCK_WEIGHT_SEED = 77


def load_inventory(name):
    with open(os.path.join(GOLDEN, f"state_dict_{name}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))


def block_shape(kw):
    return O.VMLAShape(kw["heads"], kw["dim1"], kw["dim2"], kw["mean_var_hidden"], kw["seq_length"],
                       kw["seq_len_reduce"], kw["seq_len_new"], False, False, kw["is_cross"])


def block_fixture_params(name, g):
    """Flat parameter dict of a block fixture: numpy-seeded weights, warm u/v from the fixture."""
    shapes = {str(k): tuple(int(d) for d in g["shape/" + str(k)]) for k in g["shape_names"]}
    P = {k: torch.from_numpy(v) for k, v in W.make_params(shapes, BLOCK_WEIGHT_SEED).items()}
    for k in P:
        if O.is_buffer(k):
            P[k] = torch.from_numpy(g["warm/" + k].copy())
    return shapes, P


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


def rel_err_elem(a, b, floor=1e-2):
    """Element-wise relative error with a magnitude floor: max |a - b| / max(|b|, floor * max|b|).  rel_err below is a
    normalised inf-norm (an element 100x smaller than the tensor's largest may be off by 100 % at rel_err = 1e-2 ... );
    this one bounds every element that is at least `floor` of the largest INDIVIDUALLY."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    scale = b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / torch.maximum(b.abs(), floor * scale)).max())


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)
