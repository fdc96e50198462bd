"""The CPU oracle (oracle/calm_oracle.py) against fixtures minted from the imported reference
(tests/golden/make_golden.py).  Tolerance 1e-5 relative (max-abs / max-abs), fp32."""
import numpy as np
import pytest
import torch

import weights as W
from helpers import (BLOCK_FIXTURES, CONFIGS, REAL_SIZE_CFGS, block_fixture_params, block_shape, load_golden,
                     load_inventory, oracle_params, rel_err, rel_err_elem)
from oracle import calm_oracle as O

TOL = 1e-5
GOLDEN_CFGS = ["nano48_cls", "nano48_gen", "tiny32_cls", "tiny32_fr"]


@pytest.mark.parametrize("name", list(CONFIGS))
def test_param_inventory_matches_reference_state_dict(name):
    assert O.vit_param_shapes(CONFIGS[name]) == load_inventory(name)


@pytest.mark.parametrize("name", GOLDEN_CFGS)
def test_eval_forward(name):
    g = load_golden(name)
    cfg = CONFIGS[name]
    P = oracle_params(name, g, requires_grad=False)
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2))
    with torch.no_grad():
        y, kl = O.vit_forward(P, cfg, x, training=False)
    assert rel_err(y, g["eval/y"]) < TOL
    assert rel_err_elem(y, g["eval/y"]) < 100 * TOL           # element-wise, every element >= 1 % of the largest
    assert abs(float(kl) - float(g["eval/kl"])) <= TOL * max(1.0, abs(float(g["eval/kl"])))
    if name == "tiny32_cls":
        assert isinstance(kl, float) and kl == 0.0          # Vi_Tools:49-50 python 0.0


@pytest.mark.parametrize("name", GOLDEN_CFGS)
def test_train_forward_backward(name):
    g = load_golden(name)
    cfg = CONFIGS[name]
    P = oracle_params(name, g)
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2)).requires_grad_(True)
    y, kl = O.vit_forward(P, cfg, x, training=True, noise=W.NoiseStream(7))
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy"))
    loss = (y * gy).sum() + 0.5 * kl
    loss.backward()
    assert rel_err(y.detach(), g["train/y"]) < TOL
    assert rel_err_elem(y.detach(), g["train/y"]) < 100 * TOL
    assert abs(float(kl) - float(g["train/kl"])) < TOL * max(1.0, abs(float(g["train/kl"])))
    assert rel_err(x.grad, g["train/dx"]) < 5 * TOL
    assert rel_err_elem(x.grad, g["train/dx"]) < 500 * TOL
    names = [str(n) for n in g["train/grad_names"]]
    norms = g["train/grad_norms"]
    for n, ref in zip(names, norms):
        got = float(P[n].grad.norm())
        assert abs(got - ref) <= 2e-4 * max(abs(ref), 1e-6) + 1e-9, (n, got, ref)
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(P[key[5:]].grad, g[key]) < 5 * TOL, key
        if key.startswith("post/"):                         # in-place power-iteration result
            assert rel_err(P[key[5:]], g[key]) < TOL, key


def _check_grads(P, g, names_key, norms_key, tol_norm=2e-4):
    for n, ref in zip([str(s) for s in g[names_key]], g[norms_key]):
        got = float(P[n].grad.norm())
        assert abs(got - ref) <= tol_norm * max(abs(ref), 1e-6) + 1e-9, (n, got, ref)
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(P[key[5:]].grad, g[key]) < 5 * TOL, key
        if key.startswith("post/"):
            assert rel_err(P[key[5:]], g[key]) < TOL, key


@pytest.mark.parametrize("name", list(BLOCK_FIXTURES))
def test_single_block_at_real_head_dims(name):
    """Mode A / mode B VMLA_Block fixtures minted from the reference at hd 56/44/32/20."""
    g = load_golden("block_" + name)
    kw = BLOCK_FIXTURES[name]
    sh = block_shape(kw)
    shapes, P = block_fixture_params(name, g)
    assert shapes == O.vmla_param_shapes("", sh, 2 * kw["dim2"])
    for k in P:
        if not O.is_buffer(k):
            P[k].requires_grad_(True)
    S, D1 = kw["seq_length"], kw["dim1"]
    xq = torch.from_numpy(W.make_input((1, S, D1), 5, "xq")).requires_grad_(True)
    xkv = torch.from_numpy(W.make_input((1, S, D1), 6, "xkv")).requires_grad_(True) if kw["is_cross"] else None
    st = O.LatentState(mode="sum")
    y = O.vmla_block(P, "", sh, xq, xkv, st, True, W.NoiseStream(9))
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 8, "gy"))
    kl = st.kl_loss()
    ((y * gy).sum() + 0.5 * kl).backward()
    assert rel_err(y.detach(), g["y"]) < TOL
    assert abs(float(kl) - float(g["kl"])) < TOL * max(1.0, abs(float(g["kl"])))
    assert rel_err(xq.grad, g["dxq"]) < 5 * TOL
    if kw["is_cross"]:
        assert rel_err(xkv.grad, g["dxkv"]) < 5 * TOL
    _check_grads(P, g, "grad_names", "grad_norms")


@pytest.mark.parametrize("name", REAL_SIZE_CFGS)
def test_real_size_configs_batch1(name):
    """BASELINE configs #3-#5 (Base-224, Base-384, Large-224) at batch 1 against the reference's outputs."""
    g = load_golden(name + "_b1")
    cfg = CONFIGS[name]
    S = cfg.seq_length
    P = oracle_params(name, g, requires_grad=False)
    x = torch.from_numpy(W.make_input((1, 3, S, S), 2))
    with torch.no_grad():
        y, kl = O.vit_forward(P, cfg, x, training=False)
    assert rel_err(y, g["eval/y"]) < 2 * TOL
    assert abs(float(kl) - float(g["eval/kl"])) <= TOL * max(1.0, abs(float(g["eval/kl"])))
    for k in P:
        if not O.is_buffer(k):
            P[k].requires_grad_(True)
    x = x.clone().requires_grad_(True)
    y, kl = O.vit_forward(P, cfg, x, training=True, noise=W.NoiseStream(7))
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy"))
    ((y * gy).sum() + 0.5 * kl).backward()
    assert rel_err(y.detach(), g["train/y"]) < 2 * TOL
    assert abs(float(kl) - float(g["train/kl"])) < TOL * max(1.0, abs(float(g["train/kl"])))
    assert rel_err(x.grad, g["train/dx"]) < 5 * TOL
    _check_grads(P, g, "train/grad_names", "train/grad_norms", tol_norm=5e-4)
