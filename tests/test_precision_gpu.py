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


WIDE_CASES = [
    # shapes with >= 512 tiles of 256x128: the wide bf16 kernel (ragged M, N and K tails in every layout)
    (16388, 1000, 200, (1, 1), True, True),
    (16388, 1000, 72, (1, 1), True, False),
    (8200, 2060, 40, (1, 1), False, True),
    (8200, 1924, 96, (1, 1), False, False),
    (4100, 1000, 64, (2, 2), True, False),    # batched
    (300, 200, 48, (32, 8), True, True),      # many small entries: 2 x 2 tiles each
]


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", WIDE_CASES)
@pytest.mark.parametrize("epi", ["plain", "full"])
def test_bf16_wide_tile_gemm(M, N, K, batch, akc, bkc, epi):
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
    C_ref, C_hip = torch.zeros(b0, b1, M, N), torch.full((b0, b1, M, N), 7.0).cuda()
    calm.backend.set_matmul_precision("bf16")
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, batch=batch, split_k=1, **kw)
    hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c, batch=batch, split_k=1, **kw_hip)
    assert rel_err(C_hip, C_ref) < 2e-4
    assert (C_hip.cpu() - C_ref).abs().max() < 2e-3 * C_ref.abs().max()     # no stray tile


@pytest.mark.parametrize("M,N,K", [(672, 528, 8192), (1344, 100, 6000), (768, 672, 4104), (1056, 96, 12288)])
def test_bf16_split_k_weight_gradient_on_the_wide_tile(M, N, K):
    """Output rows that pad by <= 1/4 in 256-row tiles: the split-K weight gradient runs on the 256x128 kernel."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    G_ref, G_hip = torch.zeros(M, N), torch.full((M, N), 3.0).cuda()
    args = (M, N, K, (1, M, 0, 0), (1, N, 0, 0), (N, 0, 0))
    calm.backend.set_matmul_precision("bf16")
    emu.gemm(dy, x, G_ref, *args)
    hip.gemm(dy.cuda(), x.cuda(), G_hip, *args)
    assert rel_err(G_hip, G_ref) < 2e-4


@pytest.mark.parametrize("n,T,Dout,Din", [(3, 8192, 672, 96), (2, 4100, 768, 240)])
def test_bf16_grouped_weight_gradients_on_the_wide_tile(n, T, Dout, Din):
    """dW_g = dY_g^T X (grouped, batched split-K) with 256-row output tiles."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    x = rnd(T, Din, seed=1)
    dys = [rnd(T, Dout, seed=20 + g) for g in range(n)]
    G_ref = [torch.zeros(Dout, Din) for _ in range(n)]
    G_hip = [torch.full((Dout, Din), 9.0).cuda() for _ in range(n)]
    args = (Dout, Din, T, (1, Dout, 0, 0), (1, Din, 0, 0), (Din, 0, 0))
    calm.backend.set_matmul_precision("bf16")
    emu.gemm(dys, x, G_ref, *args, batch=(n, 1))
    hip.gemm([d.cuda() for d in dys], x.cuda(), G_hip, *args, batch=(n, 1))
    for g in range(n):
        assert rel_err(G_hip[g], G_ref[g]) < 2e-4


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


def test_autocast_bfloat16_is_the_bf16_mode_forward_and_backward():
    """`with autocast(device_type="cuda", dtype=bfloat16): y_hat, _ = model(x)` (distributed_trainer_cls.py:84-85) must
    run the same arithmetic as set_matmul_precision('bf16') — in forward AND in backward, which autograd runs after
    the `with` block has been left — and hand back fp32 tensors.  eval(): bit-identical (no power iteration); train():
    the batched power iteration sums with fp32 atomics, so sigma moves in its last bit from run to run and the bf16
    operand rounding amplifies that to ~1e-3 — compared against the distance to the fp32 arithmetic instead."""
    name = "nano48_cls"
    g = load_golden(name)
    cfg = CONFIGS[name]
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2)).cuda()
    gy = torch.from_numpy(W.make_input((2, cfg.out_features), 3, "gy")).cuda()
    sd = {k: v.clone() for k, v in build_model(name, g, "cuda").state_dict().items()}

    def fresh():
        m = build_model(name, g, "cuda").train()
        m.load_state_dict(sd)
        return m

    # eval: deterministic
    m = fresh().eval()
    with torch.no_grad():
        calm.backend.set_matmul_precision("bf16")
        y_mode = m(x)[0]
        calm.backend.set_matmul_precision("fp32")
        y_fp32 = m(x)[0]
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            y_auto = m(x)[0]
            assert calm.backend.effective_precision() == "bf16"
        assert calm.backend.effective_precision() == "fp32"
    assert y_auto.dtype == torch.float32
    assert torch.equal(y_auto, y_mode) and not torch.equal(y_auto, y_fp32)

    # backward of one spectral-normed Linear: autograd runs it outside the `with` block, it must still use the pipe of
    # its forward (dgrad is launched unsplit: bit-identical; the weight gradient sums k-slices with fp32 atomics)
    def lin(mode):
        xx = rnd(512, 96, seed=1).cuda().requires_grad_(True)
        w = (rnd(144, 96, seed=2) / 10).cuda().requires_grad_(True)
        u, v, sigma = rnd(144, seed=3).cuda(), rnd(96, seed=4).cuda(), torch.tensor([1.3], device="cuda")
        args = (xx, w, None, None, None, u, v, sigma, calm.backend.ACT_NONE)
        if mode == "autocast":
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                out = calm.ops.SNLinearFn.apply(*args)
        else:
            calm.backend.set_matmul_precision(mode)
            out = calm.ops.SNLinearFn.apply(*args)
        out.backward(rnd(512, 144, seed=5).cuda())         # outside the context, as in the reference (cls:87)
        calm.backend.set_matmul_precision("fp32")
        return out.detach(), xx.grad, w.grad

    o_a, dx_a, dw_a = lin("autocast")
    o_m, dx_m, dw_m = lin("bf16")
    o_f, dx_f, dw_f = lin("fp32")
    assert torch.equal(o_a, o_m) and torch.equal(dx_a, dx_m) and rel_err(dw_a, dw_m) < 1e-5
    assert rel_err(dx_a, dx_f) > 1e-4 and rel_err(dw_a, dw_f) > 1e-4


def test_reference_amp_step_with_grad_scaler():
    """cls:84-96: autocast forward, GradScaler.scale(loss).backward(), unscale_, clip_grad_norm_, scaler.step/update."""
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")
    name = "nano48_cls"
    g = load_golden(name)
    cfg = CONFIGS[name]
    m = build_model(name, g, "cuda").train()
    opt = trainer.make_optimizer(m, lr=1e-3)
    scaler = torch.amp.GradScaler("cuda")
    x = torch.from_numpy(W.make_input((4, 3, cfg.seq_length, cfg.seq_length), 2)).cuda()
    y = torch.zeros(4, cfg.out_features, device="cuda")
    y[torch.arange(4), torch.arange(4) % cfg.out_features] = 1.0
    losses = []
    torch.manual_seed(0)
    for _ in range(6):
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            y_hat, _ = m(x)
            loss = trainer.soft_target_cross_entropy(y_hat.squeeze(), y)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0, error_if_nonfinite=False)
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("prec,tol,M,D", [("bf16", 2e-4, 520, 96), ("bf16x3", 5e-5, 520, 96),
                                          ("bf16", 2e-4, 65540, 256)])     # last: enough 256x128 tiles for the wide kernel
def test_grouped_projection_gemm_on_the_bf16_pipe(prec, tol, M, D):
    """Grouped q/k/v launch and its one-pass input gradient (per-group sigma) in the bf16-operand families."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    n = 3
    x = rnd(M, D, seed=1)
    ws = [rnd(D, D, seed=10 + g) / 8 for g in range(n)]
    sig = [torch.tensor([0.7 + 0.4 * g]) for g in range(n)]
    dys = [rnd(M, D, seed=20 + g) for g in range(n)]
    lin = (D, 1, 0, 0)
    calm.backend.set_matmul_precision(prec)
    y_ref, y_hip = [torch.zeros(M, D) for _ in range(n)], [torch.zeros(M, D).cuda() for _ in range(n)]
    emu.gemm(x, ws, y_ref, M, D, D, lin, lin, (D, 0, 0), batch=(n, 1), inv_scale=sig, split_k=1)
    hip.gemm(x.cuda(), [w.cuda() for w in ws], y_hip, M, D, D, lin, lin, (D, 0, 0), batch=(n, 1),
             inv_scale=[s.cuda() for s in sig], split_k=1)
    dx_ref, dx_hip = torch.zeros(M, D), torch.zeros(M, D).cuda()
    args = (M, D, D, lin, (1, D, 0, 0), (D, 0, 0))
    emu.gemm(dys, ws, dx_ref, *args, batch=(n, 1), inv_scale=sig, reduce_batch=True, split_k=1)
    hip.gemm([d.cuda() for d in dys], [w.cuda() for w in ws], dx_hip, *args, batch=(n, 1),
             inv_scale=[s.cuda() for s in sig], reduce_batch=True, split_k=1)
    for g in range(n):
        assert rel_err(y_hip[g], y_ref[g]) < tol
    assert rel_err(dx_hip, dx_ref) < tol


# ---- bf16 tensors in HBM (ABI v4 storage types): activations / weight copies of the bf16 pipeline ---------------------
TYPED_CASES = [
    # M, N, K, batch, a_kcontig, b_kcontig   (bf16 operands: contiguous extent and strides are multiples of 8)
    (256, 384, 128, (1, 1), True, True),
    (200, 136, 72, (2, 3), True, False),
    (136, 264, 40, (3, 1), False, True),
    (128, 96, 256, (1, 2), False, False),
    (672, 672, 1024, (1, 1), False, False),      # weight-gradient layout (split over k: fp32 output)
    (224, 224, 112, (2, 6), True, True),
    (40, 24, 48, (1, 1), True, True),
    (16392, 1000, 200, (1, 1), True, True),      # wide 256x128 tile, ragged M and N
    (8200, 1928, 96, (1, 1), False, False),
    (8200, 2064, 40, (1, 1), False, True),
]


@pytest.mark.parametrize("M,N,K,batch,akc,bkc", TYPED_CASES)
@pytest.mark.parametrize("ta,tb", [("bf16", "bf16"), ("bf16", "f32"), ("f32", "bf16")])
@pytest.mark.parametrize("epi", ["plain", "full_bf16_out"])
def test_gemm_with_bf16_tensors_in_hbm(M, N, K, batch, akc, bkc, ta, tb, epi):
    """calm_gemm on operands that are already bf16 in HBM (no conversion while staging, 8 elements per 16-byte vector)
    and with bf16 outputs / epilogue operands, against the emulation: bf16 inputs are exact in fp32, fp32 inputs are
    rounded while staged, the output is rounded once when stored."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    b0, b1 = batch
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    A, a = _operand(M, K, batch, akc, 1)
    B, b = _operand(N, K, batch, bkc, 2)
    A, B = A.to(dt[ta]), B.to(dt[tb])
    c = (N, b1 * M * N, M * N)
    calm.backend.set_matmul_precision("bf16")
    if epi == "plain":
        kw, cdt, tol = {}, torch.float32, 2e-4
    else:
        kw = dict(alpha=0.5, inv_scale=torch.tensor([1.3]), bias=rnd(N, seed=3), col_scale=rnd(N, seed=4),
                  residual=rnd(b0, b1, M, N, seed=5), r=c, act=2, aux=rnd(b0, b1, M, N, seed=6).bfloat16(), split_k=1)
        cdt, tol = torch.bfloat16, 6e-3                     # one bf16 ulp of the largest output
    kw_hip = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    C_ref = torch.zeros(b0, b1, M, N, dtype=cdt)
    C_hip = torch.full((b0, b1, M, N), 7.0, dtype=cdt).cuda()
    emu.gemm(A, B, C_ref, M, N, K, a, b, c, batch=batch, **kw)
    hip.gemm(A.cuda(), B.cuda(), C_hip, M, N, K, a, b, c, batch=batch, **kw_hip)
    assert rel_err(C_hip.float(), C_ref.float()) < tol
    if cdt == torch.bfloat16:                               # rounding-flip free elements agree exactly: most of them
        same = (C_hip.cpu() == C_ref).float().mean()
        assert same > 0.97, float(same)


def test_gemm_mlp_pattern_with_bf16_hidden_and_preactivation():
    """The MLP's first GEMM in the bf16 pipeline: bf16 LayerNorm output x bf16 weight copy -> bf16 GELU output and
    bf16 pre-activation (C and C_pre), bias in fp32."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    M, K, N = 3000, 672, 1344
    calm.backend.set_matmul_precision("bf16")
    x, w, bias, sg = rnd(M, K, seed=1).bfloat16(), (rnd(N, K, seed=2) * 0.05).bfloat16(), rnd(N, seed=3), torch.tensor([0.7])
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        P = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        be.gemm(x.to(dev), w.to(dev), C, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sg.to(dev),
                bias=bias.to(dev), act=1, C_pre=P, split_k=1)
        outs.append((C.float().cpu(), P.float().cpu()))
    assert rel_err(outs[1][0], outs[0][0]) < 6e-3 and rel_err(outs[1][1], outs[0][1]) < 6e-3
    assert (outs[1][0] == outs[0][0]).float().mean() > 0.97


def test_gemm_bf16_tensors_that_cannot_be_staged_take_the_generic_path():
    """K = 36 rules out 16-byte bf16 vectors: the library answers CALM_E_LAYOUT and the binding reruns the (tiny, rare)
    launch on fp32 copies through the generic exact kernels; bf16 tensors outside the bf16 matrix pipe are refused."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    calm.backend.set_matmul_precision("bf16")
    A, B = rnd(64, 36).bfloat16(), rnd(64, 36, seed=1)
    C, C_ref = torch.zeros(64, 64).cuda(), torch.zeros(64, 64)
    hip.gemm(A.cuda(), B.cuda(), C, 64, 64, 36, (36, 1, 0, 0), (36, 1, 0, 0), (64, 0, 0))
    emu.gemm(A, B, C_ref, 64, 64, 36, (36, 1, 0, 0), (36, 1, 0, 0), (64, 0, 0))
    assert rel_err(C, C_ref) < 1e-2                          # exact fp32 product of the same operands vs bf16-rounded B
    calm.backend.set_matmul_precision("fp32")
    A = rnd(64, 64).bfloat16().cuda()
    with pytest.raises(RuntimeError):                       # bf16 tensors need the bf16 matrix pipe
        hip.gemm(A, rnd(64, 64).cuda(), C, 64, 64, 64, (64, 1, 0, 0), (64, 1, 0, 0), (64, 0, 0))


def test_backward_after_a_later_forward_refuses_stale_bf16_weight_copies():
    """The per-step bf16 weight copies are rewritten in place by every forward and reach backward outside
    save_for_backward: a backward that runs after another forward must raise (as torch's in-place version check does)
    instead of silently differentiating with the new weights."""
    name = "nano48_cls"
    g = load_golden(name)
    m = build_model(name, g, "cuda").train()
    cfg = CONFIGS[name]
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2)).cuda()
    calm.backend.set_matmul_precision("bf16")
    calm.ops.set_noise_override(W.NoiseStream(7))
    try:
        y1, _ = m(x)
        y2, _ = m(x)                       # refreshes the copies: generation moves on
        with pytest.raises(RuntimeError, match="bf16 weight copies"):
            y1.sum().backward()
        y2.sum().backward()                # the newest forward's backward is fine
    finally:
        calm.ops.set_noise_override(None)
