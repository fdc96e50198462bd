"""BASELINE.json configs #3-#5 at their real sizes on the HIP path (through the C-ABI, cuda:0) against fixtures minted
from the imported reference at batch 1 (tests/golden/make_golden.py: golden_<cfg>_b1.npz, golden_block_<name>.npz):

  * fp32 matmuls: north_star's 1e-3 relative (max-abs error / max-abs reference) on logits, KL, dL/dx, every
    parameter-gradient norm, the small gradients in full and the power-iteration vectors;
  * bf16 matmuls (what `autocast(bfloat16)` selects — the precision these three configs are quoted in):
      - against the fp32 reference fixture within BF16_VS_FP32 (stated below; bf16 has 8 significant bits and the
        logits pass through 24 blocks), and
      - against an emulation of exactly the library's rounding points (the package's host logic over
        tests/emulated_backend.py on CPU, same inputs and noise): only the fp32 accumulation order differs, but a
        1e-6 difference in a GEMM input flips its bf16 rounding now and then (one flip moves an output by
        ~2^-8/sqrt(K)), and the network amplifies such perturbations block after block.  Measured on MI355X:
        one block 2e-4 (output) / 7e-4 (input gradient) — BF16_VS_EMULATION_BLOCK is the tight bound — growing to
        3e-3 ... 1.1e-2 behind 24 blocks, where it is bounded by BF16_VS_EMULATION_MODEL.
"""
import pytest
import torch

import calm_vit_dte_amd as calm
import weights as W
from emulated_backend import EmulatedBackend
from helpers import (BLOCK_FIXTURES, BLOCK_WEIGHT_SEED, CONFIGS, REAL_SIZE_CFGS, block_fixture_params, load_golden,
                     rel_err, rel_err_elem)
from oracle import calm_oracle as O
from test_host_logic_cpu import build_model

pytestmark = pytest.mark.gpu
TOL = 1e-3                    # north_star: 1e-3 rel fp32
BF16_VS_FP32 = 4e-2           # bf16-operand matmuls against the fp32 reference (logits / dL/dx, max-abs relative)
BF16_VS_EMULATION_BLOCK = 3e-3  # one block against the emulation of the same rounding points: output (x6 for input gradients; measured 1.3e-3)
BF16_VS_EMULATION = 2.5e-2    # = BF16_VS_EMULATION_MODEL: the full model (24 blocks) against that emulation
BF16_GRAD_EACH = 6e-2         # ... each single parameter's gradient norm against the emulation


@pytest.fixture(autouse=True)
def _restore_precision():
    yield
    calm.backend.set_matmul_precision("fp32")
    calm.ops.set_noise_override(None)


def _train_pass(m, x, noise_seed=7):
    calm.ops.set_noise_override(W.NoiseStream(noise_seed))
    try:
        y, kl = m(x)
        gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy")).to(y.device)
        ((y * gy).sum() + 0.5 * kl).backward()
    finally:
        calm.ops.set_noise_override(None)
    return y.detach(), kl.detach()


@pytest.mark.parametrize("name", REAL_SIZE_CFGS)
def test_full_model_fp32_matches_reference_fixture(name):
    g = load_golden(name + "_b1")
    cfg = CONFIGS[name]
    S = cfg.seq_length
    m = build_model(name, g, "cuda").eval()
    x = torch.from_numpy(W.make_input((1, 3, S, S), 2)).cuda()
    with torch.no_grad():
        y, kl = m(x)
    assert rel_err(y, g["eval/y"]) < TOL
    assert rel_err_elem(y, g["eval/y"]) < TOL                 # every logit >= 1 % of the largest, individually
    assert abs(float(kl) - float(g["eval/kl"])) < TOL * max(1.0, abs(float(g["eval/kl"])))
    m.train()
    x = x.clone().requires_grad_(True)
    y, kl = _train_pass(m, x)
    assert rel_err(y, g["train/y"]) < TOL
    assert rel_err_elem(y, g["train/y"]) < TOL
    assert abs(float(kl) - float(g["train/kl"])) < TOL * max(1.0, abs(float(g["train/kl"])))
    assert rel_err(x.grad, g["train/dx"]) < TOL
    assert rel_err_elem(x.grad, g["train/dx"]) < TOL
    params = dict(m.named_parameters())
    for n, ref in zip([str(s) for s in g["train/grad_names"]], g["train/grad_norms"]):
        got = float(params[n].grad.norm())
        assert abs(got - ref) <= TOL * max(abs(ref), 1e-6) + 1e-8, (n, got, ref)
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(params[key[5:]].grad, g[key]) < TOL, key
        if key.startswith("post/"):
            assert rel_err(sd[key[5:]], g[key]) < TOL, key


@pytest.mark.parametrize("name", REAL_SIZE_CFGS)
def test_full_model_bf16_vs_fp32_fixture_and_vs_rounding_emulation(name):
    g = load_golden(name + "_b1")
    cfg = CONFIGS[name]
    S = cfg.seq_length
    calm.backend.set_matmul_precision("bf16")
    x0 = torch.from_numpy(W.make_input((1, 3, S, S), 2))
    # HIP, bf16 matmuls
    m = build_model(name, g, "cuda").train()
    x = x0.cuda().requires_grad_(True)
    y, kl = _train_pass(m, x)
    assert torch.isfinite(y).all()
    assert rel_err(y, g["train/y"]) < BF16_VS_FP32
    assert abs(float(kl) - float(g["train/kl"])) < BF16_VS_FP32 * max(1.0, abs(float(g["train/kl"])))
    assert rel_err(x.grad, g["train/dx"]) < BF16_VS_FP32
    # the same host logic over the CPU emulation of the library's arithmetic (same rounding points)
    me = build_model(name, g, "cpu").train()
    xe = x0.clone().requires_grad_(True)
    with calm.backend.use_backend(EmulatedBackend()):
        ye, kle = _train_pass(me, xe)
    assert rel_err(y, ye) < BF16_VS_EMULATION
    assert abs(float(kl) - float(kle)) < BF16_VS_EMULATION * max(1.0, abs(float(kle)))
    assert rel_err(x.grad, xe.grad) < BF16_VS_EMULATION
    # gradients: the global norm tightly; each parameter's own norm within BF16_GRAD_EACH (a bias gradient is a sum
    # with heavy cancellation: the order of the fp32 accumulation — atomics on the GPU — moves it more than a matrix)
    pe = dict(me.named_parameters())
    tot_h = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()) ** 0.5
    tot_e = sum(float(p.grad.double().pow(2).sum()) for p in me.parameters()) ** 0.5
    worst = max((abs(float(p.grad.norm()) - float(pe[n].grad.norm())) / max(float(pe[n].grad.norm()), 1e-6), n)
                for n, p in m.named_parameters())
    print(f"\n[{name}] bf16 vs fp32 fixture: y {rel_err(y, g['train/y']):.2e} dx {rel_err(x.grad, g['train/dx']):.2e}; "
          f"vs emulation: y {rel_err(y, ye):.2e} dx {rel_err(x.grad, xe.grad):.2e} |grad| {abs(tot_h - tot_e) / tot_e:.2e} "
          f"worst param {worst[0]:.2e} {worst[1]}")
    assert abs(tot_h - tot_e) < 1e-3 * tot_e
    assert worst[0] < BF16_GRAD_EACH, worst


def _block_on_gpu(name, precision, device="cuda"):
    vt = calm.Vi_Tools_CNN_less_V2
    g = load_golden("block_" + name)
    kw = BLOCK_FIXTURES[name]
    shapes, P = block_fixture_params(name, g)
    blk = vt.VMLA_Block(mlp_dim=2 * kw["dim2"], force_reduce=False, **kw)
    assert {k: tuple(v.shape) for k, v in blk.state_dict().items()} == shapes
    blk.load_state_dict({k: v.clone() for k, v in P.items()})
    blk = blk.to(device).train()
    S, D1 = kw["seq_length"], kw["dim1"]
    xq = torch.from_numpy(W.make_input((1, S, D1), 5, "xq")).to(device).requires_grad_(True)
    xkv = torch.from_numpy(W.make_input((1, S, D1), 6, "xkv")).to(device).requires_grad_(True) if kw["is_cross"] else None
    sm = vt.ResidualStateManager(mode="sum")
    calm.backend.set_matmul_precision(precision)
    calm.ops.set_noise_override(W.NoiseStream(9))
    try:
        y = blk(xq, input_kv=xkv, state_manager=sm, mask=True)
        gy = torch.from_numpy(W.make_input(tuple(y.shape), 8, "gy")).to(device)
        kl = sm.get_kl_loss()
        ((y * gy).sum() + 0.5 * kl).backward()
    finally:
        calm.ops.set_noise_override(None)
    return g, kw, blk, y.detach(), kl, xq, xkv


@pytest.mark.parametrize("name", list(BLOCK_FIXTURES))
def test_single_block_fp32_matches_reference_fixture(name):
    """mode A / mode B VMLA_Block at hd 56/44/32/20 against the reference's own block."""
    g, kw, blk, y, kl, xq, xkv = _block_on_gpu(name, "fp32")
    assert rel_err(y, g["y"]) < TOL
    assert rel_err_elem(y, g["y"]) < TOL
    assert abs(float(kl) - float(g["kl"])) < TOL * max(1.0, abs(float(g["kl"])))
    assert rel_err(xq.grad, g["dxq"]) < TOL
    assert rel_err_elem(xq.grad, g["dxq"]) < TOL
    if kw["is_cross"]:
        assert rel_err(xkv.grad, g["dxkv"]) < TOL
    params = dict(blk.named_parameters())
    for n, ref in zip([str(s) for s in g["grad_names"]], g["grad_norms"]):
        got = float(params[n].grad.norm())
        assert abs(got - ref) <= TOL * max(abs(ref), 1e-6) + 1e-8, (n, got, ref)
    sd = blk.state_dict()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(params[key[5:]].grad, g[key]) < TOL, key
        if key.startswith("post/"):
            assert rel_err(sd[key[5:]], g[key]) < TOL, key


@pytest.mark.parametrize("name", list(BLOCK_FIXTURES))
def test_single_block_bf16_within_stated_tolerance_of_reference_fixture(name):
    g, kw, blk, y, kl, xq, xkv = _block_on_gpu(name, "bf16")
    assert rel_err(y, g["y"]) < BF16_VS_FP32
    assert rel_err(xq.grad, g["dxq"]) < BF16_VS_FP32
    if kw["is_cross"]:
        assert rel_err(xkv.grad, g["dxkv"]) < BF16_VS_FP32


@pytest.mark.parametrize("name", list(BLOCK_FIXTURES))
def test_single_block_bf16_tight_against_emulation_of_the_rounding_points(name):
    g, kw, blk, y, kl, xq, xkv = _block_on_gpu(name, "bf16")
    with calm.backend.use_backend(EmulatedBackend()):
        _, _, blk_e, y_e, kl_e, xq_e, xkv_e = _block_on_gpu(name, "bf16", device="cpu")
    assert rel_err(y, y_e) < BF16_VS_EMULATION_BLOCK
    assert rel_err(xq.grad, xq_e.grad) < 3 * BF16_VS_EMULATION_BLOCK
    if kw["is_cross"]:
        assert rel_err(xkv.grad, xkv_e.grad) < 3 * BF16_VS_EMULATION_BLOCK
    pe = dict(blk_e.named_parameters())
    for n, p in blk.named_parameters():
        if p.numel() >= 4096:                          # matrices; the small vectors are sums with heavy cancellation
            assert rel_err(p.grad, pe[n].grad) < 5 * BF16_VS_EMULATION_BLOCK, n
