"""Per-entry-point parity of libcalmvit_hip.so (through the C-ABI, on a real MI355X) against the
torch emulation of the same ABI evaluated on CPU fp32.  Tolerances are fp32: 1e-4 relative
(max-abs / max-abs) for contractions and reductions, bit-exact for the index permutations."""
import math

import pytest
import torch

import calm_vit_dte_amd as calm
from emulated_backend import EmulatedBackend
from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    return calm.backend.get_backend()


@pytest.fixture(scope="module")
def emu():
    return EmulatedBackend()


def gpu(t):
    return None if t is None else t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_library_loads_and_reports_gfx950(hip):
    assert hip.lib.calm_abi_version() == calm._lib.ABI_VERSION == 7
    assert b"gfx950" in hip.lib.calm_build_info()


GEMM_CASES = [
    # M, N, K, batch, a_kcontig, b_kcontig   (vector path: everything multiple of 4)
    (256, 384, 128, (1, 1), True, True),
    (200, 136, 72, (2, 3), True, False),
    (132, 260, 40, (3, 1), False, True),
    (128, 128, 256, (1, 2), False, False),
    # scalar path (odd sizes / K=3 as in the 1x1 convs / hd/2=18 as in Nano)
    (67, 33, 3, (1, 1), True, True),
    (50, 18, 19, (2, 2), True, False),
    (45, 77, 18, (1, 3), False, True),
    (31, 29, 37, (2, 1), False, False),
    (1, 5, 7, (1, 1), True, True),
]


def _operand(rows, K, batch, kcontig, seed):
    b0, b1 = batch
    if kcontig:
        t = rnd(b0, b1, rows, K, seed=seed)
        strides = (K, 1, b1 * rows * K, rows * K)
    else:
        t = rnd(b0, b1, K, rows, seed=seed)
        strides = (1, rows, b1 * rows * K, rows * K)
    return t, strides


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", GEMM_CASES)
@pytest.mark.parametrize("epi", ["plain", "full", "gelu_bwd", "accumulate"])
def test_gemm_layouts_and_epilogues(hip, emu, M, N, K, batch, akc, bkc, epi):
    b0, b1 = batch
    A, a = _operand(M, K, batch, akc, 1)
    B, b = _operand(N, K, batch, bkc, 2)
    c = (N, b1 * M * N, M * N)
    kw = {}
    if epi == "full":
        kw = dict(alpha=0.37, inv_scale=torch.tensor([1.7]), bias=rnd(N, seed=3), col_scale=rnd(N, seed=4),
                  residual=rnd(b0, b1, M, N, seed=5), r=c, C_pre=torch.zeros(b0, b1, M, N), act=1)
    elif epi == "gelu_bwd":
        kw = dict(inv_scale=torch.tensor([0.9]), aux=rnd(b0, b1, M, N, seed=6), act=2)
    elif epi == "accumulate":
        kw = dict(accumulate=True, alpha=2.0)
    C_ref = rnd(b0, b1, M, N, seed=7)
    C_hip = C_ref.clone().cuda()
    kw_hip = {k: (gpu(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, batch=batch, **kw)
    hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c, batch=batch, **kw_hip)
    assert rel_err(C_hip, C_ref) < TOL
    if epi == "full":
        assert rel_err(kw_hip["C_pre"], kw["C_pre"]) < TOL


def test_gemm_epilogue_tensors_off_16_byte_alignment(hip, emu):
    """The vector epilogue needs every epilogue tensor addressable in aligned groups of 4 columns; a C / bias /
    residual that starts 4 bytes into an allocation (a slice) must fall back to the one-element form, same results."""
    M, N, K = 200, 136, 72
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias_buf, res_buf, c_buf = rnd(N + 1, seed=3), rnd(M * N + 1, seed=4), rnd(M * N + 1, seed=5)
    for off_c, off_b, off_r in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)):
        bias = bias_buf[off_b:off_b + N]
        res = res_buf[off_r:off_r + M * N].view(M, N)
        C_ref = c_buf[off_c:off_c + M * N].view(M, N).clone()
        cg = c_buf.clone().cuda()
        C_hip = cg[off_c:off_c + M * N].view(M, N)
        kw = dict(alpha=0.5, bias=bias, residual=res, r=(N, 0, 0), act=1)
        emu.gemm(A, B, C_ref, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), **kw)
        bg, rg = bias_buf.cuda(), res_buf.cuda()
        hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), alpha=0.5,
                 bias=bg[off_b:off_b + N], residual=rg[off_r:off_r + M * N].view(M, N), r=(N, 0, 0), act=1)
        assert rel_err(C_hip, C_ref) < TOL, (off_c, off_b, off_r)


def test_gemm_head_strided_batches(hip, emu):
    """The attention GEMMs address heads through strides inside [B,S,H*hd] tensors."""
    B_, H, S, hd = 2, 3, 48, 20
    D = H * hd
    q, k = rnd(B_, S, D, seed=1), rnd(B_, S, D, seed=2)
    mask = rnd(B_, S, S, seed=3)
    P_ref = torch.zeros(B_, H, S, S)
    P_hip = P_ref.clone().cuda()
    args = (S, S, hd, (D, 1, S * D, hd), (D, 1, S * D, hd), (S, H * S * S, S * S))
    kw = dict(batch=(B_, H), alpha=1 / math.sqrt(hd), r=(S, S * S, 0))
    emu.gemm(q, k, P_ref, *args, residual=mask, **kw)
    hip.gemm(q.cuda(), k.cuda(), P_hip, *args, residual=mask.cuda(), **kw)
    assert rel_err(P_hip, P_ref) < TOL
    ref = torch.einsum("bihd,bjhd->bhij", q.view(B_, S, H, hd), k.view(B_, S, H, hd)) / math.sqrt(hd) + mask[:, None]
    assert rel_err(P_hip, ref) < TOL


@pytest.mark.parametrize("M,N,K", [(96, 80, 5000), (672, 672, 4096), (3, 32, 9000), (240, 480, 20480), (528, 1056, 6000)])
def test_gemm_split_k_weight_gradient(hip, emu, M, N, K):
    """G = dY^T X with a long reduction: split-K partials combined with fp32 atomics."""
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    G_ref = torch.zeros(M, N)
    G_hip = torch.full((M, N), 7.0).cuda()                      # must be overwritten, not accumulated
    args = (M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0))
    emu.gemm(dy, x, G_ref, *args)
    hip.gemm(dy.cuda(), x.cuda(), G_hip, *args)
    assert rel_err(G_hip, G_ref) < TOL


def test_gemm_small_split_k_outputs_ask_for_no_workspace(hip):
    """Below ~100k output elements the second launch costs more than the atomics: the query says 0."""
    import ctypes as C
    from importlib import import_module
    binding = import_module("calm_vit_dte_amd._lib")
    for (M, N, K), expect in (((80, 160, 20480), False), ((264, 120, 45056), False), ((384, 384, 32768), True),
                              ((672, 672, 57344), False)):             # last: large output, few slices
        dy, x, G = torch.zeros(K, M).cuda(), torch.zeros(K, N).cuda(), torch.zeros(M, N).cuda()
        g = binding.GemmArgs()
        g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), G.data_ptr()
        g.M, g.N, g.K, g.batch0, g.batch1 = M, N, K, 1, 1
        g.a_rs, g.a_cs, g.b_rs, g.b_cs, g.c_rs = 1, M, 1, N, N
        g.alpha = 1.0
        assert (hip.lib.calm_gemm_workspace_bytes(C.byref(g)) > 0) == expect, (M, N, K)


@pytest.mark.parametrize("M,N,K,acc", [(384, 384, 32768, False), (240, 480, 20480, True), (384, 768, 32768, False)])
def test_gemm_split_k_through_the_workspace(hip, emu, monkeypatch, M, N, K, acc):
    """Mid-sized weight gradients (>= 48 k-slices per output): the slices' partial tiles go to a caller-owned workspace
    and one reduction pass writes C (strided rows, optional accumulate) — deterministic, unlike the atomics it replaces;
    without a workspace the same call falls back to atomics."""
    import ctypes
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    ld = N + 8                                                   # output rows strided: the 8 pad columns must survive
    base = rnd(M, ld, seed=3)
    args = (M, N, K, (1, M, 0, 0), (1, N, 0, 0), (ld, 0, 0))
    G_ref = base.clone()
    if not acc:
        G_ref[:, :N] = 0
    emu.gemm(dy, x, G_ref, *args, accumulate=True)
    dyc, xc = dy.cuda(), x.cuda()

    asked = []
    real_query = hip.lib.calm_gemm_workspace_bytes
    def query(a):
        asked.append(real_query(a))
        return asked[-1]
    monkeypatch.setattr(hip.lib, "calm_gemm_workspace_bytes", query)
    outs = []
    for _ in range(2):
        G = base.cuda()
        hip.gemm(dyc, xc, G, *args, accumulate=acc)
        outs.append(G)
    assert asked and all(b > 0 and b % (4 * M * N) == 0 for b in asked)      # whole partial tiles, >= 48 of them
    assert asked[0] // (4 * M * N) >= 48
    assert rel_err(outs[0][:, :N], G_ref[:, :N]) < TOL
    assert torch.equal(outs[0], outs[1])                                      # no atomics: bit-reproducible
    assert torch.equal(outs[0][:, N:].cpu(), base[:, N:])

    monkeypatch.setattr(hip.lib, "calm_gemm_workspace_bytes", lambda a: 0)    # no workspace offered: atomics
    G = base.cuda()
    hip.gemm(dyc, xc, G, *args, accumulate=acc)
    assert rel_err(G[:, :N], G_ref[:, :N]) < TOL
    assert torch.equal(G[:, N:].cpu(), base[:, N:])


@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(160, 100, 512), (528, 96, 1028), (90, 52, 300)])
def test_gemm_split_k_all_layouts(hip, emu, M, N, K, akc, bkc):
    """Explicit split-K (fp32 atomics) over every operand layout, 16-byte and scalar staging (K=300 / M=90 are not
    multiples of 4 on the strided side)."""
    A, a = _operand(M, K, (1, 1), akc, 1)
    B, b = _operand(N, K, (1, 1), bkc, 2)
    c = (N, 0, 0)
    C_ref, C_hip = torch.zeros(M, N), torch.full((M, N), 3.0).cuda()
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, split_k=4)
    hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c, split_k=4)
    assert rel_err(C_hip, C_ref) < TOL


def test_gemm_reduce_batch(hip, emu):
    B_, S2, S, D = 5, 44, 36, 72
    dy, x = rnd(B_, S2, D, seed=1), rnd(B_, S, D, seed=2)
    G_ref, G_hip = torch.zeros(S2, S), torch.ones(S2, S).cuda()
    args = (S2, S, D, (D, 1, S2 * D, 0), (D, 1, S * D, 0), (S, 0, 0))
    emu.gemm(dy, x, G_ref, *args, batch=(B_, 1), reduce_batch=True)
    hip.gemm(dy.cuda(), x.cuda(), G_hip, *args, batch=(B_, 1), reduce_batch=True)
    assert rel_err(G_hip, G_ref) < TOL


def test_gemm_rejects_bad_arguments(hip):
    A = torch.zeros(8, 8).cuda()
    with pytest.raises(RuntimeError):
        hip.gemm(A, A, A, 8, 8, 8, (8, 2, 0, 0), (8, 1, 0, 0), (8, 0, 0))       # neither stride is 1
    with pytest.raises(RuntimeError):
        hip.gemm(A, A, A, 0, 8, 8, (8, 1, 0, 0), (8, 1, 0, 0), (8, 0, 0))       # empty problem
    with pytest.raises(RuntimeError):
        hip.gemm(A.cpu(), A, A, 8, 8, 8, (8, 1, 0, 0), (8, 1, 0, 0), (8, 0, 0))  # CPU tensor
    with pytest.raises(RuntimeError):
        hip.gemm(A, A, A, 8, 8, 8, (1 << 21, 1, 0, 0), (8, 1, 0, 0), (8, 0, 0))  # row stride beyond the 32-bit tile offsets


@pytest.mark.parametrize("rows,D", [(37, 144), (1000, 672), (5, 36), (64, 1152), (2500, 528), (300, 240),
                                    (33, 30), (9, 1302)])      # last two: scalar (D%4 != 0 / D > 1280) kernels
def test_layernorm(hip, emu, rows, D):
    x, w, dy = rnd(rows, D, seed=1) * 2 + 0.5, 1 + 0.1 * rnd(D, seed=2), rnd(rows, D, seed=3)
    skip = rnd(rows, D, seed=4)
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        X, Wt, DY = x.to(dev), w.to(dev), dy.to(dev)
        y, mean, rstd = torch.empty_like(X), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        be.layernorm_fwd(X, Wt, y, mean, rstd, rows, D, 1e-6)
        dx, dw = torch.empty_like(X), torch.zeros(D, device=dev)
        be.layernorm_bwd(DY, X, Wt, mean, rstd, dx, dw, rows, D)
        dx2, dw2 = torch.empty_like(X), torch.zeros(D, device=dev)        # with the skip-connection gradient folded in
        be.layernorm_bwd(DY, X, Wt, mean, rstd, dx2, dw2, rows, D, dx_add=skip.to(dev))
        outs.append((y, mean, rstd, dx, dw, dx2, dw2))
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL
    assert rel_err(outs[1][5], outs[1][3] + skip.cuda()) < 1e-6


@pytest.mark.parametrize("B,S,H,dc,dr", [(2, 48, 3, 0, 48), (2, 44, 3, 18, 18), (1, 224, 6, 56, 56), (3, 7, 2, 0, 6),
                                          (2, 40, 4, 28, 28), (3, 176, 12, 0, 44), (2, 80, 12, 10, 10), (5, 33, 12, 0, 20)])
def test_rope(hip, emu, B, S, H, dc, dr):
    content = rnd(B, S, H * dc, seed=1) if dc else None
    xr, inv = rnd(B, S, H * dr, seed=2), torch.rand(dr // 2, generator=torch.Generator().manual_seed(3)) + 0.01
    g = rnd(B, S, H * (dc + dr), seed=4)
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        mv = lambda t: None if t is None else t.to(dev)
        table = torch.empty(2 * S * (dr // 2), device=dev)
        out = torch.empty(B, S, H * (dc + dr), device=dev)
        be.rope_fwd(mv(content), mv(xr), mv(inv), table, out, B, S, H, dc, dr)
        d_c = torch.empty(B, S, H * dc, device=dev) if dc else None
        d_x, d_f = torch.empty(B, S, H * dr, device=dev), torch.zeros(dr // 2, device=dev)
        be.rope_bwd(mv(g), mv(xr), table, d_c, d_x, d_f, B, S, H, dc, dr)
        outs.append([out, d_x, d_f] + ([d_c] if dc else []))
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL


@pytest.mark.parametrize("B,S,H,dc,dr", [(2, 224, 12, 0, 56), (2, 176, 12, 22, 22), (2, 128, 12, 16, 16), (3, 80, 12, 0, 20)])
@pytest.mark.parametrize("in16,out16", [(True, True), (False, True), (True, False)])
def test_rope_bf16_tensors(hip, emu, B, S, H, dc, dr, in16, out16):
    """bf16 pipeline: projection outputs (content, xr) and / or q, k (out) stored as bf16; gradients take the type of
    their tensors.  Same arithmetic in fp32 on both sides, one rounding at each bf16 store: 2^-8 relative."""
    tin, tout = (torch.bfloat16 if in16 else torch.float32), (torch.bfloat16 if out16 else torch.float32)
    content = rnd(B, S, H * dc, seed=1).to(tin) if dc else None
    xr, inv = rnd(B, S, H * dr, seed=2).to(tin), torch.rand(dr // 2, generator=torch.Generator().manual_seed(3)) + 0.01
    g = rnd(B, S, H * (dc + dr), seed=4).to(tout)
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        mv = lambda t: None if t is None else t.to(dev)
        table = torch.empty(2 * S * (dr // 2), device=dev)
        out = torch.empty(B, S, H * (dc + dr), device=dev, dtype=tout)
        be.rope_fwd(mv(content), mv(xr), mv(inv), table, out, B, S, H, dc, dr)
        d_c = torch.empty(B, S, H * dc, device=dev, dtype=tin) if dc else None
        d_x, d_f = torch.empty(B, S, H * dr, device=dev, dtype=tin), torch.zeros(dr // 2, device=dev)
        be.rope_bwd(mv(g), mv(xr), table, d_c, d_x, d_f, B, S, H, dc, dr)
        outs.append([out, d_x, d_f] + ([d_c] if dc else []))
    for a, b in zip(outs[1], outs[0]):
        assert a.dtype == b.dtype
        assert rel_err(a.float(), b.float()) < (2.0 ** -7 if a.dtype == torch.bfloat16 else TOL)


@pytest.mark.parametrize("rows,cols", [(100, 48), (999, 224), (17, 80), (64, 384), (3, 1)])
def test_softmax_and_sum_heads(hip, emu, rows, cols):
    x, g = rnd(rows, cols, seed=1) * 3, rnd(rows, cols, seed=2)
    p_ref, p_hip = x.clone(), x.clone().cuda()
    emu.softmax_fwd(p_ref, rows, cols)
    hip.softmax_fwd(p_hip, rows, cols)
    assert rel_err(p_hip, p_ref) < TOL
    g_ref, g_hip = g.clone(), g.clone().cuda()
    emu.softmax_bwd(p_ref, g_ref, rows, cols)
    hip.softmax_bwd(p_hip, g_hip, rows, cols)
    assert rel_err(g_hip, g_ref) < TOL


def test_sum_heads(hip, emu):
    B_, H, per = 3, 5, 48 * 48
    dl = rnd(B_, H, per, seed=1)
    a, b = torch.empty(B_, per), torch.empty(B_, per).cuda()
    emu.sum_heads(dl, a, B_, H, per)
    hip.sum_heads(dl.cuda(), b, B_, H, per)
    assert rel_err(b, a) < TOL


@pytest.mark.parametrize("B_,H,Sq,cols", [(2, 6, 48, 48), (3, 4, 20, 200), (1, 12, 37, 384), (2, 3, 5, 7)])
def test_softmax_backward_with_head_sum(hip, emu, B_, H, Sq, cols):
    """dL = softmax backward of [B,H,Sq,cols] and dM = sum over the heads in one pass, against the two-step form."""
    P = torch.softmax(rnd(B_, H, Sq, cols, seed=1) * 2, dim=-1)
    dP = rnd(B_, H, Sq, cols, seed=2)
    g_ref, g_hip = dP.clone(), dP.clone().cuda()
    m_ref, m_hip = torch.empty(B_, Sq, cols), torch.full((B_, Sq, cols), 3.0).cuda()
    emu.softmax_bwd_heads(P, g_ref, m_ref, B_, H, Sq, cols)
    hip.softmax_bwd_heads(P.cuda(), g_hip, m_hip, B_, H, Sq, cols)
    assert rel_err(g_hip, g_ref) < TOL and rel_err(m_hip, m_ref) < TOL
    assert rel_err(m_hip, g_hip.sum(dim=1)) < 1e-5


@pytest.mark.parametrize("with_noise", [True, False])
def test_latent(hip, emu, with_noise):
    rows, mvh = 2 * 16, 24
    mv = rnd(rows, 2 * mvh, seed=1) * 2
    mv[0, mvh] = 30.0                                   # softplus threshold branch (x > 20)
    noise = rnd(rows, mvh, seed=2) if with_noise else None
    dz, dk = rnd(rows, mvh, seed=3), torch.tensor([0.3])
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        mvd = lambda t: None if t is None else t.to(dev)
        z, std, kl = torch.empty(rows, mvh, device=dev), torch.empty(rows, mvh, device=dev), torch.zeros((), device=dev)
        be.latent_fwd(mvd(mv), mvd(noise), z, std, kl, rows, mvh)
        dmv = torch.empty(rows, 2 * mvh, device=dev)
        be.latent_bwd(mvd(dz), mvd(dk), mvd(mv), mvd(noise), std, dmv, rows, mvh)
        outs.append((z, std, kl.reshape(1), dmv))
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL


@pytest.mark.parametrize("training", [True, False])
def test_spectral_norm_batched_power_iteration(hip, emu, training):
    shapes = [(144, 144), (288, 144), (48, 96), (32, 3), (32, 9), (3, 32), (1344, 672), (16, 48), (200, 1)]
    def make(dev):
        ls = []
        for i, (r, c) in enumerate(shapes):
            w = rnd(r, c, seed=10 + i) / math.sqrt(c)
            u = torch.nn.functional.normalize(rnd(r, seed=50 + i), dim=0)
            v = torch.nn.functional.normalize(rnd(c, seed=90 + i), dim=0)
            ls.append((w.to(dev), u.to(dev), v.to(dev), torch.zeros(1, device=dev)))
        return ls
    ref, got = make("cpu"), make("cuda")
    for _ in range(2):                                  # two iterations: state carried in u,v
        emu.sn_power_iter(emu.sn_plan(ref), training)
        plan = hip.sn_plan(got)
        hip.sn_power_iter(plan, training)
    for (w, u, v, s), (w2, u2, v2, s2) in zip(ref, got):
        assert rel_err(s2, s) < TOL and rel_err(u2, u) < TOL and rel_err(v2, v) < TOL


@pytest.mark.parametrize("rows,cols,with_ls", [(144, 288, True), (96, 48, False), (32, 9, False), (3, 32, False)])
def test_sn_weight_backward(hip, emu, rows, cols, with_ls):
    G, w = rnd(rows, cols, seed=1), rnd(rows, cols, seed=2)
    u, v = torch.nn.functional.normalize(rnd(rows, seed=3), dim=0), torch.nn.functional.normalize(rnd(cols, seed=4), dim=0)
    sigma, ls = torch.tensor([1.3]), (1 + 0.1 * rnd(rows, seed=5)) if with_ls else None
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        mv = lambda t: None if t is None else t.to(dev)
        dW, d_ls = torch.empty(rows, cols, device=dev), (torch.empty(rows, device=dev) if with_ls else None)
        be.sn_weight_bwd(mv(G), mv(w), mv(u), mv(v), mv(sigma), mv(ls), dW, d_ls, rows, cols)
        outs.append([dW] + ([d_ls] if with_ls else []))
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL


@pytest.mark.parametrize("B,S", [(2, 48), (1, 224), (3, 33), (2, 80)])
def test_tokenisation_is_bit_exact(hip, emu, B, S):
    img = rnd(B, 3, S, S, seed=1)
    rows_ref, rows_hip = torch.empty(B, S, 3 * S), torch.empty(B, S, 3 * S).cuda()
    emu.image_to_rows(img, rows_ref, B, S)
    hip.image_to_rows(img.cuda(), rows_hip, B, S)
    assert torch.equal(rows_hip.cpu(), rows_ref)
    back = torch.empty(B, 3, S, S).cuda()
    hip.rows_to_image(rows_hip, back, B, S)
    assert torch.equal(back.cpu(), img)                 # round trip
    t_ref, t_hip = torch.empty_like(rows_ref), torch.empty_like(rows_hip)
    emu.grid_transpose(rows_ref, t_ref, B, S)
    hip.grid_transpose(rows_hip, t_hip, B, S)
    assert torch.equal(t_hip.cpu(), t_ref)
    tt = torch.empty_like(rows_hip)
    hip.grid_transpose(t_hip, tt, B, S)
    assert torch.equal(tt, rows_hip)                    # involution


@pytest.mark.parametrize("B,S", [(2, 44), (1, 80)])
def test_dwconv3x3(hip, emu, B, S):
    Cch = 32
    x, w, bias = rnd(B, S, S, Cch, seed=1), rnd(Cch, 9, seed=2) * 0.3, rnd(Cch, seed=3) * 0.1
    sigma, dz = torch.tensor([0.8]), rnd(B, S, S, Cch, seed=4)
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        X, Wt, Bi, Sg, DZ = x.to(dev), w.to(dev), bias.to(dev), sigma.to(dev), dz.to(dev)
        y, yp = torch.empty_like(X), torch.empty_like(X)
        be.dwconv_fwd(X, Wt, Sg, Bi, y, yp, 1, B, S, Cch)
        dx, dw, db = torch.empty_like(X), torch.zeros(Cch, 9, device=dev), torch.zeros(Cch, device=dev)
        be.dwconv_bwd(DZ, X, Wt, Sg, dx, dw, db, B, S, Cch)
        outs.append((y, yp, dx, dw, db))
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL


def test_streaming_helpers(hip, emu):
    n = 3 * 1000 + 3
    a, b = rnd(n, seed=1), rnd(n, seed=2)
    out = torch.empty(n).cuda()
    hip.add(a.cuda(), b.cuda(), out, n)
    assert torch.equal(out.cpu(), a + b)
    dz_ref, dz = torch.empty(n), torch.empty(n).cuda()
    emu.gelu_bwd(a, b, dz_ref, n)
    hip.gelu_bwd(a.cuda(), b.cuda(), dz, n)
    assert rel_err(dz, dz_ref) < TOL
    for rows, cols in ((500, 3), (300, 32), (77, 448), (40, 1000), (4096, 448), (1000, 120), (3000, 1344), (513, 2000), (100, 4)):
        x = rnd(rows, cols, seed=3)
        s_ref, s_hip = torch.zeros(cols), torch.zeros(cols).cuda()
        emu.colsum(x, s_ref, rows, cols)
        hip.colsum(x.cuda(), s_hip, rows, cols)
        assert rel_err(s_hip, s_ref) < TOL
    x, s = rnd(40, 24, seed=4), rnd(40, seed=5)
    r_hip = torch.empty(40, 24).cuda()
    hip.row_scale(x.cuda(), s.cuda(), r_hip, 40, 24)
    assert rel_err(r_hip, x * s[:, None]) < 1e-6
    x = rnd(3, 48, 144, seed=6)
    y_hip, dx_hip = torch.empty(3, 144).cuda(), torch.empty(3, 48, 144).cuda()
    hip.mean_seq_fwd(x.cuda(), y_hip, 3, 48, 144)
    assert rel_err(y_hip, x.mean(dim=1)) < TOL
    hip.mean_seq_bwd(y_hip, dx_hip, 3, 48, 144)
    assert rel_err(dx_hip, (y_hip.cpu() / 48)[:, None, :].expand(3, 48, 144)) < TOL


@pytest.mark.parametrize("B,S", [(2, 48), (1, 80), (3, 36), (2, 12), (1, 176)])
def test_fused_cnn_residual(hip, emu, B, S):
    """Fused conv1x1-GELU-dw3x3-GELU-conv1x1 + residual (LDS-tiled, backward recomputes) incl. ragged tiles."""
    Ch = 32
    x, dy = rnd(B, S, 3 * S, seed=1), rnd(B, S, 3 * S, seed=2)
    w0, b0 = rnd(Ch, 3, seed=3) * 0.6, rnd(Ch, seed=4) * 0.1
    w2, b2 = rnd(Ch, 9, seed=5) * 0.4, rnd(Ch, seed=6) * 0.1
    w4, b4 = rnd(3, Ch, seed=7) * 0.3, rnd(3, seed=8) * 0.1
    s0, s2, s4 = torch.tensor([0.9]), torch.tensor([1.2]), torch.tensor([0.7])
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        t = [v.to(dev) for v in (x, w0, s0, b0, w2, s2, b2, w4, s4, b4)]
        out = torch.empty(B, S, 3 * S, device=dev)
        be.cnn_fwd(*t, out, B, S, Ch)
        dx = torch.empty(B, S, 3 * S, device=dev)
        gs = [torch.zeros(n, device=dev) for n in (Ch * 3, Ch, Ch * 9, Ch, 3 * Ch, 3)]
        be.cnn_bwd(dy.to(dev), *t, dx, *gs, B, S, Ch)
        outs.append([out, dx] + gs)
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b) < TOL


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("M,D", [(300, 96), (1000, 240), (130, 52)])
def test_gemm_grouped_projections_and_their_input_gradient(hip, emu, n, M, D):
    """q/k/v projections of one activation as ONE launch (separately allocated weights, outputs and sigmas), and their
    input gradient dX = sum_g dY_g W_g / sigma_g as one pass over the concatenated reduction (deterministic)."""
    x = rnd(M, D, seed=1)
    ws = [rnd(D, D, seed=10 + g) / 8 for g in range(n)]
    sig = [torch.tensor([0.7 + 0.4 * g]) for g in range(n)]
    lin = (D, 1, 0, 0)
    # forward
    y_ref, y_hip = [torch.zeros(M, D) for _ in range(n)], [torch.full((M, D), 5.0).cuda() for _ in range(n)]
    emu.gemm(x, ws, y_ref, M, D, D, lin, lin, (D, 0, 0), batch=(n, 1), inv_scale=sig, split_k=1)
    hip.gemm(x.cuda(), [w.cuda() for w in ws], y_hip, M, D, D, lin, lin, (D, 0, 0), batch=(n, 1),
             inv_scale=[s.cuda() for s in sig], split_k=1)
    for g in range(n):
        assert rel_err(y_hip[g], y_ref[g]) < TOL
        assert rel_err(y_hip[g], x @ ws[g].T / sig[g]) < TOL
    # input gradient
    dys = [rnd(M, D, seed=20 + g) for g in range(n)]
    dx_ref, dx_hip = torch.zeros(M, D), torch.full((M, D), 5.0).cuda()
    args = (M, D, D, lin, (1, D, 0, 0), (D, 0, 0))
    emu.gemm(dys, ws, dx_ref, *args, batch=(n, 1), inv_scale=sig, reduce_batch=True, split_k=1)
    run = lambda out: hip.gemm([d.cuda() for d in dys], [w.cuda() for w in ws], out, *args, batch=(n, 1),
                               inv_scale=[s.cuda() for s in sig], reduce_batch=True, split_k=1)
    run(dx_hip)
    assert rel_err(dx_hip, dx_ref) < TOL
    assert rel_err(dx_hip, sum(dys[g] @ ws[g] / sig[g] for g in range(n))) < TOL
    again = torch.empty_like(dx_hip)
    run(again)
    assert torch.equal(dx_hip, again)                     # unsplit: no atomics


@pytest.mark.parametrize("n,T,D", [(3, 8192, 96), (2, 5000, 240), (3, 3000, 52)])
def test_gemm_grouped_weight_gradients_split_per_group(hip, emu, n, T, D):
    """dW_g = dY_g^T X for the projections that share X: one grouped launch in which every group is split over its
    own k-slices (batched split-K, fp32 atomics per group output)."""
    x = rnd(T, D, seed=1)
    dys = [rnd(T, D, seed=20 + g) for g in range(n)]
    G_ref = [torch.zeros(D, D) for _ in range(n)]
    G_hip = [torch.full((D, D), 9.0).cuda() for _ in range(n)]          # must be overwritten
    args = (D, D, T, (1, D, 0, 0), (1, D, 0, 0), (D, 0, 0))
    emu.gemm(dys, x, G_ref, *args, batch=(n, 1))
    hip.gemm([d.cuda() for d in dys], x.cuda(), G_hip, *args, batch=(n, 1))
    for g in range(n):
        assert rel_err(G_hip[g], G_ref[g]) < TOL
        assert rel_err(G_hip[g], dys[g].T @ x) < TOL


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_device_collate_mixup_cutmix_normalize_flip(hip, emu, mode):
    """calm_collate_mix against the torch restatement of ToDtype(scale) + Normalize + flip + MixUp / CutMix
    (distributed_trainer_cls.py:58-61,128-139): uint8 arithmetic in fp32 is exact up to rounding of the normalise."""
    B, H, W = 5, 40, 56
    g = torch.Generator().manual_seed(mode)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8)
    flip = (torch.rand(B, generator=g) < 0.5).to(torch.uint8)
    box = (7, 29, 10, 41)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    ref, out = torch.empty(B, 3, H, W), torch.empty(B, 3, H, W).cuda()
    emu.collate_mix(img, flip, ref, mode, 0.3, box, mean, std)
    hip.collate_mix(img.cuda(), flip.cuda(), out, mode, 0.3, box, mean, std)
    assert rel_err(out, ref) < 1e-6
    if mode == 2:          # outside the box nothing is mixed, inside it is the rolled partner
        plain = torch.empty(B, 3, H, W).cuda()
        hip.collate_mix(img.cuda(), flip.cuda(), plain, 0, 1.0, None, mean, std)
        assert torch.equal(out[..., :7, :], plain[..., :7, :])
        assert torch.equal(out[..., 7:29, 10:41], plain.roll(1, 0)[..., 7:29, 10:41])
