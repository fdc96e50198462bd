"""Pipelined persistent bf16-tensor GEMM family of calm_gemm (csrc/gemm_bf16p.h: LDS-DMA staging, (64 MT) x (32 NT) x 64
tiles, swapped-operand accumulators) against the emulation of exactly its arithmetic: bf16 operands are exact in fp32,
accumulation is fp32, the output is rounded once when stored.  The cases walk every operand-layout pair the family
instantiates (k/k: forward, k/row: data gradient, row/row: weight gradient), every tile width (N picks NT, M x N picks
MT), ragged M / N / K tails (rows clamped, columns masked, the k tail fed from the zero block), batches, independent
groups, k-split launches through atomics and through the workspace, and the full fused epilogue."""
import pytest
import torch

import calm_vit_dte_amd as calm
from emulated_backend import EmulatedBackend
from helpers import rel_err
from locate import check_gemm, item_of_linear

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_mode():
    calm.backend.set_matmul_precision("bf16")
    yield
    calm.backend.set_matmul_precision("fp32")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _operand(rows, K, batch, kcontig, seed):
    b0, b1 = batch
    if kcontig:
        return rnd(b0, b1, rows, K, seed=seed).bfloat16(), (K, 1, b1 * rows * K, rows * K)
    return rnd(b0, b1, K, rows, seed=seed).bfloat16(), (1, rows, b1 * rows * K, rows * K)


CASES = [
    # M, N, K, batch, a_kcontig, b_kcontig
    (4096, 672, 672, (1, 1), True, True),       # NT 7, K tail of 32
    (4096, 528, 528, (1, 1), True, True),       # NT 6 (576 padded), K tail of 16
    (4096, 384, 384, (1, 1), True, True),       # NT 6 exact, K multiple of 64
    (4096, 480, 240, (1, 1), True, True),       # NT 5
    (4096, 768, 96, (1, 1), True, True),        # NT 8, two k-tiles
    (2056, 240, 480, (1, 1), True, True),       # ragged M (rows clamped), N 240 in a 256 tile
    (1000, 136, 72, (2, 3), True, True),        # batches, ragged everything
    (4096, 672, 1344, (1, 1), True, False),     # data gradient: weight read through the transposing LDS reads
    (3000, 528, 1056, (1, 1), True, False),
    (2048, 240, 264, (1, 2), True, False),
    (1024, 352, 176, (3, 1), True, False),
    (672, 672, 8192, (1, 1), False, False),     # weight gradient: k-split, both operands row-contiguous
    (1344, 672, 4096, (1, 1), False, False),
    (528, 1056, 4104, (1, 1), False, False),    # K tail of 8
    (240, 480, 20480, (1, 1), False, False),    # many slices: workspace reduction
    (384, 384, 2048, (2, 1), False, False),     # row/row batches without split
    (136, 264, 640, (1, 1), False, False),
]


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", CASES)
@pytest.mark.parametrize("epi", ["plain", "full_bf16_out"])
def test_pipelined_gemm_against_emulation(M, N, K, batch, akc, bkc, epi):
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    b0, b1 = batch
    A, a = _operand(M, K, batch, akc, 1)
    B, b = _operand(N, K, batch, bkc, 2)
    c = (N, b1 * M * N, M * N)
    if epi == "plain":
        kw, cdt, tol = {}, torch.float32, 2e-4
    else:
        kw = dict(alpha=0.5, inv_scale=torch.tensor([1.3]), bias=rnd(N, seed=3), col_scale=rnd(N, seed=4),
                  residual=rnd(b0, b1, M, N, seed=5), r=c, act=2, aux=rnd(b0, b1, M, N, seed=6).bfloat16(), split_k=1)
        cdt, tol = torch.bfloat16, 6e-3
    kw_hip = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    C_ref = torch.zeros(b0, b1, M, N, dtype=cdt)
    C_hip = torch.full((b0, b1, M, N), 7.0, dtype=cdt).cuda()
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, batch=batch, **kw)
    args, kwa = (A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c), dict(batch=batch, **kw_hip)
    hip.gemm(*args, **kwa)
    assert torch.isfinite(C_hip.float()).all()
    bound = tol * max(1.0, (K / 1024) ** 0.5)
    if batch == (1, 1):          # self-locating on a violation (tests/locate.py): tile / workgroup slot / XCD / wave / strip
        check_gemm(f"pipe_{M}x{N}x{K}_{int(akc)}{int(bkc)}_{epi}", hip, C_hip[0, 0], C_ref[0, 0].cuda(), bound, args, kwa,
                   cross_family=False)
    assert rel_err(C_hip.float(), C_ref.float()) < bound
    if cdt == torch.bfloat16:
        assert (C_hip.cpu() == C_ref).float().mean() > 0.97


@pytest.mark.parametrize("act,pre", [(1, True), (0, False)])
def test_pipelined_gemm_gelu_with_preactivation_and_accumulate(act, pre):
    """MLP pattern (bias + GELU, pre-activation saved) and accumulation into an existing fp32 C."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    M, N, K = 3000, 1344, 672
    x, w = rnd(M, K, seed=1).bfloat16(), (rnd(N, K, seed=2) * K ** -0.5).bfloat16()
    bias, sigma = rnd(N, seed=3) * 0.1, torch.tensor([1.3])
    lin = (K, 1, 0, 0)
    if pre:
        y_r, p_r = torch.zeros(M, N).bfloat16(), torch.zeros(M, N).bfloat16()
        y_h, p_h = torch.zeros(M, N).bfloat16().cuda(), torch.zeros(M, N).bfloat16().cuda()
        emu.gemm(x, w, y_r, M, N, K, lin, lin, (N, 0, 0), inv_scale=sigma, bias=bias, act=act, C_pre=p_r, split_k=1)
        hip.gemm(x.cuda(), w.cuda(), y_h, M, N, K, lin, lin, (N, 0, 0), inv_scale=sigma.cuda(), bias=bias.cuda(), act=act,
                 C_pre=p_h, split_k=1)
        assert rel_err(p_h.float(), p_r.float()) < 6e-3 and rel_err(y_h.float(), y_r.float()) < 6e-3
    else:
        c0 = rnd(M, N, seed=7)
        y_r, y_h = c0.clone(), c0.clone().cuda()
        emu.gemm(x, w, y_r, M, N, K, lin, lin, (N, 0, 0), inv_scale=sigma, accumulate=True, split_k=1)
        hip.gemm(x.cuda(), w.cuda(), y_h, M, N, K, lin, lin, (N, 0, 0), inv_scale=sigma.cuda(), accumulate=True, split_k=1)
        assert rel_err(y_h, y_r) < 2e-4


def test_pipelined_grouped_projections_and_their_weight_gradients():
    """q/k/v as one grouped launch (per-group sigma, separate allocations) and the grouped k-split weight gradients."""
    hip = calm.backend.get_backend()
    M, D = 8192, 672
    x = rnd(M, D, seed=1).bfloat16().cuda()
    ws = [(rnd(D, D, seed=10 + i) * D ** -0.5).bfloat16().cuda() for i in range(3)]
    sg = [torch.tensor([1.0 + 0.3 * i], device="cuda") for i in range(3)]
    outs = [torch.empty(M, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    lin = (D, 1, 0, 0)
    hip.gemm(x, ws, outs, M, D, D, lin, lin, (D, 0, 0), batch=(3, 1), inv_scale=sg, split_k=1)
    for i in range(3):
        ref = (x.float() @ ws[i].float().T) / (1.0 + 0.3 * i)
        assert rel_err(outs[i].float(), ref) < 6e-3
    dys = [rnd(M, D, seed=20 + i).bfloat16().cuda() for i in range(3)]
    Gs = [torch.full((D, D), 3.0, device="cuda") for _ in range(3)]
    hip.gemm(dys, x, Gs, D, D, M, (1, D, 0, 0), (1, D, 0, 0), (D, 0, 0), batch=(3, 1))
    for i in range(3):
        assert rel_err(Gs[i], dys[i].float().T @ x.float()) < 3e-4


def test_pipelined_gemm_propagates_nan_and_ignores_padding():
    """A NaN in the operands reaches exactly the outputs that depend on it (GradScaler's inf check relies on it); the
    zero-fed k tail and the clamped edge rows add nothing."""
    hip = calm.backend.get_backend()
    M, N, K = 1000, 200, 72
    A = rnd(M, K, seed=1).bfloat16().cuda()
    B = rnd(N, K, seed=2).bfloat16().cuda()
    A[17, 70] = float("nan")
    C = torch.zeros(M, N, device="cuda")
    hip.gemm(A, B, C, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1)
    bad = torch.isnan(C)
    assert bad[17].all() and int(bad.sum()) == N


def test_describe_reports_the_launch_decomposition_and_the_item_order_inverts():
    """calm_gemm_describe (ABI v7): the plan of the bench-size GELU' data gradient is the pipelined family on 256 x 224
    tiles, 1344 items on 256 persistent workgroups; tests/locate.py's inverse of the XCD-aware item order is a bijection
    whose items with equal index mod 8 are consecutive tiles (gemm_bf16p.h::decode)."""
    hip = calm.backend.get_backend()
    M, N = 57344, 1344
    dy = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    w = torch.empty(N, N, dtype=torch.bfloat16, device="cuda")
    plan = hip.gemm_describe(dy, w, torch.empty_like(dy), M, N, N, (N, 1, 0, 0), (1, N, 0, 0), (N, 0, 0), act=2, aux=dy,
                             split_k=1)
    assert plan["family"] == 3 and (plan["tile_m"], plan["tile_n"]) == (256, 224)
    assert plan["items"] == 224 * 6 and plan["grid"] == 256 and plan["epi_unit"] == 8 and plan["k_slices"] == 1
    for n_items in (5, 8, 13, 1344, 1345, 1351):
        items = [item_of_linear(lin, n_items) for lin in range(n_items)]
        assert sorted(items) == list(range(n_items))
        if n_items >= 8:
            by_x = {}
            for lin, it in enumerate(items):
                by_x.setdefault(it % 8, []).append(lin)
            assert all(v == list(range(v[0], v[0] + len(v))) for v in by_x.values())
    # a weight gradient: k-split, and with the deterministic option every split launch asks for a workspace
    x = torch.empty(M, 672, dtype=torch.bfloat16, device="cuda")
    G = torch.empty(N, 672, device="cuda")
    wg = (dy, x, G, N, 672, M, (1, N, 0, 0), (1, 672, 0, 0), (672, 0, 0))
    p0 = hip.gemm_describe(*wg)
    assert p0["k_slices"] > 1
    prev = hip.gemm_set_option(hip.GEMM_OPT_DETERMINISTIC, 1)
    try:
        g, _ = hip._gemm_args(*wg)
        import ctypes
        assert hip.lib.calm_gemm_workspace_bytes(ctypes.byref(g)) == 4 * p0["k_slices"] * N * 672
    finally:
        hip.gemm_set_option(hip.GEMM_OPT_DETERMINISTIC, prev)
