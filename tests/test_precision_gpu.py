"""bf16-operand GEMM family of calm_gemm (tensors fp32 in HBM, operands staged as bf16, fp32 accumulate):
  * 'bf16'   — against a torch emulation that rounds both operands to bf16 (tight: only the fp32
               accumulation order differs), i.e. exactly the arithmetic of autocast(bfloat16) matmuls;
  * 'bf16x3' — hi/lo split, 3 MFMA passes — against the exact fp32 product (it must be fp32-accurate),
and model-level parity against the reference's golden fixtures:
  bf16x3 within north_star's 1e-3 rel fp32; bf16 within 3e-2 (bf16 operand rounding through 24 blocks)."""
import pytest
import torch

import calm_vit_dte_amd as calm
import weights as W
from emulated_backend import EmulatedBackend
from helpers import CONFIGS, load_golden, rel_err
from test_host_logic_cpu import build_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_precision():
    yield
    calm.backend.set_matmul_precision("fp32")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _operand(rows, K, batch, kcontig, seed):
    b0, b1 = batch
    if kcontig:
        return rnd(b0, b1, rows, K, seed=seed), (K, 1, b1 * rows * K, rows * K)
    return rnd(b0, b1, K, rows, seed=seed), (1, rows, b1 * rows * K, rows * K)


CASES = [
    # M, N, K, batch, a_kcontig, b_kcontig  (all multiples of 4: the 16-byte staging path)
    (256, 384, 128, (1, 1), True, True),      # forward linear
    (200, 136, 72, (2, 3), True, False),      # dgrad (weight read "transposed" via ds_read_b64_tr_b16)
    (132, 260, 40, (3, 1), False, True),
    (128, 96, 256, (1, 2), False, False),     # wgrad layout: both operands row-contiguous
    (672, 672, 1024, (1, 1), False, False),
    (224, 224, 112, (2, 6), True, True),      # per-head logits
    (36, 20, 44, (1, 1), True, True),         # ragged everything
]


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", CASES)
@pytest.mark.parametrize("prec,tol", [("bf16", 2e-4), ("bf16x3", 5e-5)])
@pytest.mark.parametrize("epi", ["plain", "full"])
def test_bf16_operand_gemm(M, N, K, batch, akc, bkc, prec, tol, epi):
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    b0, b1 = batch
    A, a = _operand(M, K, batch, akc, 1)
    B, b = _operand(N, K, batch, bkc, 2)
    c = (N, b1 * M * N, M * N)
    kw = {}
    if epi == "full":
        kw = dict(alpha=0.5, inv_scale=torch.tensor([1.3]), bias=rnd(N, seed=3), col_scale=rnd(N, seed=4),
                  residual=rnd(b0, b1, M, N, seed=5), r=c, act=1)
    kw_hip = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    C_ref, C_hip = torch.zeros(b0, b1, M, N), torch.zeros(b0, b1, M, N).cuda()
    calm.backend.set_matmul_precision(prec)
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, batch=batch, **kw)
    hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c, batch=batch, **kw_hip)
    assert rel_err(C_hip, C_ref) < tol
    if prec == "bf16x3":                                   # and it really is fp32-accurate
        calm.backend.set_matmul_precision("fp32")
        C_exact = torch.zeros(b0, b1, M, N)
        emu.gemm(A, B, C_exact, M, N, K, a, b, c, batch=batch, **kw)
        assert rel_err(C_hip, C_exact) < 5e-5


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
def test_split_k_weight_gradient_in_bf16_modes(prec):
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    M, N, K = 672, 528, 8192
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    G_ref, G_hip = torch.zeros(M, N), torch.full((M, N), 3.0).cuda()
    args = (M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0))
    calm.backend.set_matmul_precision(prec)
    emu.gemm(dy, x, G_ref, *args)
    hip.gemm(dy.cuda(), x.cuda(), G_hip, *args)
    assert rel_err(G_hip, G_ref) < 2e-4


@pytest.mark.parametrize("name", ["nano48_cls", "tiny32_fr"])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-3), ("bf16", 3e-2)])
def test_model_train_step_matches_reference_golden(name, prec, tol):
    g = load_golden(name)
    cfg = CONFIGS[name]
    calm.backend.set_matmul_precision(prec)
    m = build_model(name, g, "cuda").train()
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2)).cuda().requires_grad_(True)
    calm.ops.set_noise_override(W.NoiseStream(7))
    try:
        y, kl = m(x)
        gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy")).cuda()
        ((y * gy).sum() + 0.5 * kl).backward()
    finally:
        calm.ops.set_noise_override(None)
    assert rel_err(y.detach(), g["train/y"]) < tol
    assert rel_err(x.grad, g["train/dx"]) < tol
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(params[key[5:]].grad, g[key]) < 3 * tol, key


def test_small224_block_bf16x3_vs_oracle():
    """Stage-0 plain block at the bench model's size in bf16x3: every gradient within 1e-3 of the CPU oracle."""
    from test_model_gpu import _block_vs_oracle
    calm.backend.set_matmul_precision("bf16x3")
    _block_vs_oracle(6, 672, 672, 120, 224, 40, 224, False)
