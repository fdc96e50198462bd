"""fp32 instantiation of the pipelined persistent GEMM family (csrc/gemm_bf16p.h with element type float:
gemm_f32p_kernel, LDS-DMA staging of 128-byte k-rows = 32 values, v_mfma_f32_16x16x4_f32) against a float64 product.
The products are exact fp32 multiplies accumulated in fp32 (only the order of the reduction differs from the 128 x 128
kernels), so the tolerance is fp32 rounding: 2e-6 x sqrt(K / 1024) of the largest output.  Cases: every operand-layout
pair (k/k forward, k/row data gradient, row/row weight gradient — the [k][row] image is read with ds_read_b32), every tile
width, ragged M / N / K (K tails of 4 ... 28), batches, grouped launches, k-split through atomics and through the
workspace, the fused epilogue, and equivalence with the 128 x 128 family through calm_gemm_set_option."""
import pytest
import torch

import calm_vit_dte_amd as calm
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_pipeline_on():
    """The fp32 instantiation is off by default (it ties with the 128 x 128 kernels: the fp32 matrix pipe is clock-limited);
    calm_gemm_set_option switches it on for these cases."""
    hip = calm.backend.get_backend()
    prev = hip.gemm_set_option(hip.GEMM_OPT_PIPE32, 1)
    yield
    hip.gemm_set_option(hip.GEMM_OPT_PIPE32, prev)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _operand(rows, K, batch, kcontig, seed):
    b0, b1 = batch
    if kcontig:
        return rnd(b0, b1, rows, K, seed=seed), (K, 1, b1 * rows * K, rows * K)
    return rnd(b0, b1, K, rows, seed=seed), (1, rows, b1 * rows * K, rows * K)


def _product(A, B, akc, bkc):
    a = A.double() if akc else A.double().transpose(-1, -2)
    b = B.double() if bkc else B.double().transpose(-1, -2)
    return a @ b.transpose(-1, -2)


CASES = [
    # M, N, K, batch, a_kcontig, b_kcontig
    (4096, 672, 672, (1, 1), True, True),       # NT 7, K = 21 k-tiles
    (4096, 528, 528, (1, 1), True, True),       # NT 6 (576 padded), K tail of 16
    (4096, 384, 388, (1, 1), True, True),       # K tail of 4
    (4096, 480, 240, (1, 1), True, True),       # NT 5, K tail of 16
    (4096, 768, 96, (1, 1), True, True),        # NT 8, three k-tiles
    (2056, 240, 476, (1, 1), True, True),       # ragged M (rows clamped), N 240 in a 256 tile, K tail of 28
    (1000, 136, 72, (2, 3), True, True),        # batches, ragged everything
    (4096, 672, 1344, (1, 1), True, False),     # data gradient: weight in the [k][row] image
    (3000, 528, 1056, (1, 1), True, False),
    (2048, 240, 264, (1, 2), True, False),
    (1024, 352, 176, (3, 1), True, False),
    (224, 112, 224, (4, 6), False, False),      # per-image, per-head products of the composed attention backward
    (672, 672, 8192, (1, 1), False, False),     # weight gradient: k-split, both operands row-contiguous
    (1344, 672, 4096, (1, 1), False, False),
    (528, 1056, 4108, (1, 1), False, False),    # K tail of 12
    (240, 480, 20480, (1, 1), False, False),    # many slices: workspace reduction
    (384, 384, 2048, (2, 1), False, False),     # row/row batches without split
    (136, 264, 640, (1, 1), False, False),
]


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", CASES)
@pytest.mark.parametrize("epi", ["plain", "full"])
def test_fp32_pipelined_gemm_against_float64(M, N, K, batch, akc, bkc, epi):
    hip = calm.backend.get_backend()
    b0, b1 = batch
    A, a = _operand(M, K, batch, akc, 1)
    B, b = _operand(N, K, batch, bkc, 2)
    c = (N, b1 * M * N, M * N)
    ref = _product(A, B, akc, bkc)
    C = torch.full((b0, b1, M, N), 7.0).cuda()
    tol = 2e-6 * max(1.0, (K / 1024) ** 0.5)
    if epi == "plain":
        hip.gemm(A.cuda(), B.cuda(), C, M, N, K, a, b, c, batch=batch)
    else:
        bias, cs, res, aux = rnd(N, seed=3), rnd(N, seed=4), rnd(b0, b1, M, N, seed=5), rnd(b0, b1, M, N, seed=6)
        hip.gemm(A.cuda(), B.cuda(), C, M, N, K, a, b, c, batch=batch, alpha=0.5, inv_scale=torch.tensor([1.3]).cuda(),
                 bias=bias.cuda(), col_scale=cs.cuda(), residual=res.cuda(), r=c, act=2, aux=aux.cuda(), split_k=1)
        x = aux.double()
        gelu_grad = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5
        ref = ((ref * (0.5 / 1.3) + bias.double()) * gelu_grad) * cs.double() + res.double()
        tol *= 4
    assert torch.isfinite(C).all()
    assert rel_err(C.cpu().double(), ref) < tol


def test_fp32_pipelined_mlp_epilogue_and_accumulate():
    """bias + GELU with the pre-activation saved, then accumulation into an existing C."""
    hip = calm.backend.get_backend()
    M, N, K = 3000, 1344, 672
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5
    bias, sigma = rnd(N, seed=3) * 0.1, 1.3
    lin = (K, 1, 0, 0)
    y, pre = torch.zeros(M, N).cuda(), torch.zeros(M, N).cuda()
    hip.gemm(x.cuda(), w.cuda(), y, M, N, K, lin, lin, (N, 0, 0), inv_scale=torch.tensor([sigma]).cuda(), bias=bias.cuda(),
             act=1, C_pre=pre, split_k=1)
    p_ref = x.double() @ w.double().T / sigma + bias.double()
    assert rel_err(pre.cpu().double(), p_ref) < 2e-6
    assert rel_err(y.cpu().double(), torch.nn.functional.gelu(p_ref)) < 2e-6
    c0 = rnd(M, N, seed=7)
    y2 = c0.clone().cuda()
    hip.gemm(x.cuda(), w.cuda(), y2, M, N, K, lin, lin, (N, 0, 0), accumulate=True, split_k=1)
    assert rel_err(y2.cpu().double(), c0.double() + x.double() @ w.double().T) < 2e-6


def test_fp32_pipelined_grouped_projections_and_their_weight_gradients():
    hip = calm.backend.get_backend()
    M, D = 8192, 672
    x = rnd(M, D, seed=1).cuda()
    ws = [(rnd(D, D, seed=10 + i) * D ** -0.5).cuda() for i in range(3)]
    sg = [torch.tensor([1.0 + 0.3 * i], device="cuda") for i in range(3)]
    outs = [torch.empty(M, D, device="cuda") for _ in range(3)]
    lin = (D, 1, 0, 0)
    hip.gemm(x, ws, outs, M, D, D, lin, lin, (D, 0, 0), batch=(3, 1), inv_scale=sg, split_k=1)
    for i in range(3):
        assert rel_err(outs[i].double(), (x.double() @ ws[i].double().T) / (1.0 + 0.3 * i)) < 2e-6
    dys = [rnd(M, D, seed=20 + i).cuda() for i in range(3)]
    Gs = [torch.full((D, D), 3.0, device="cuda") for _ in range(3)]
    hip.gemm(dys, x, Gs, D, D, M, (1, D, 0, 0), (1, D, 0, 0), (D, 0, 0), batch=(3, 1))
    for i in range(3):
        assert rel_err(Gs[i].double(), dys[i].double().T @ x.double()) < 6e-6


def test_fp32_pipelined_gemm_propagates_nan_and_ignores_padding():
    hip = calm.backend.get_backend()
    M, N, K = 1000, 200, 76
    A, B = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2).cuda()
    A[17, 74] = float("nan")
    C = torch.zeros(M, N, device="cuda")
    hip.gemm(A, B, C, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1)
    bad = torch.isnan(C)
    assert bad[17].all() and int(bad.sum()) == N


def test_fp32_family_switch_gives_the_same_products():
    """calm_gemm_set_option routes the same call to either fp32 family: they agree to fp32 rounding (the reduction orders
    differ), and an unknown option / value is refused."""
    hip = calm.backend.get_backend()
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(3000, 672, generator=g).cuda(), torch.randn(528, 672, generator=g).cuda()
    outs = []
    for mode in (1, 0, 2):
        assert hip.gemm_set_option(hip.GEMM_OPT_PIPE32, mode) in (0, 1, 2)
        C = torch.zeros(3000, 528, device="cuda")
        hip.gemm(A, B, C, 3000, 528, 672, (672, 1, 0, 0), (672, 1, 0, 0), (528, 0, 0))
        outs.append(C.cpu())
    assert rel_err(outs[0], outs[1]) < 2e-6 and torch.equal(outs[0], outs[2])
    assert not torch.equal(outs[0], outs[1])              # really two different kernels
    with pytest.raises(ValueError):
        hip.gemm_set_option(7, 1)
    with pytest.raises(ValueError):
        hip.gemm_set_option(hip.GEMM_OPT_PIPE32, 3)
