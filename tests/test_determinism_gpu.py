"""ABI v7: no atomics on the path.  Every cross-workgroup sum (LayerNorm dw, RoPE d_inv_freq, bias column sums, the latent
KL sum, the CNN tail's weight gradients) goes through per-workgroup partial rows and a fixed-order second pass, and with
CALM_GEMM_OPT_DETERMINISTIC the k-split weight gradients go through the workspace reduction: a forward + backward of
the model repeats BIT FOR BIT (VERDICT r3 #7 — rounds 1-3 had fp32 atomics in these sums, which capped what any
regression test of the trainer could detect).  Also here: the first Block consuming row tokens and the
stand-alone GELU / proj(img) calls of the reference's Sequential containers."""
from importlib import import_module

import numpy as np
import pytest
import torch

import calm_vit_dte_amd as calm
import weights as W
from emulated_backend import EmulatedBackend
from helpers import BLOCK_FIXTURES, CONFIGS, load_golden, rel_err
from test_host_logic_cpu import build_model

pytestmark = pytest.mark.gpu
trainer = import_module("calm_vit_dte_amd.trainer")
vt = import_module("calm_vit_dte_amd.Vi_Tools_CNN_less_V2")


def rnd(*shape, seed=0, dev="cuda"):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(dev)


@pytest.fixture
def deterministic_gemm():
    be = calm.backend.get_backend()
    prev = be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, 1)
    yield be
    be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, prev)


def _busy(side, a, b):
    with torch.cuda.stream(side):
        for _ in range(3):
            b.copy_(a)
            a.add_(1.0)


def _reduction_cases(be):
    """One call of every entry point with a cross-workgroup reduction (ABI v7), at bench sizes and at ragged toy sizes
    (vector and scalar kernels); each returns the tensors the call produced."""

    def ln(rows, D, g16):
        x, w, dy = rnd(rows, D, seed=1) * 2 + 0.5, 1 + 0.1 * rnd(D, seed=2), rnd(rows, D, seed=3)
        if g16:
            dy = dy.bfloat16()
        y, mean, rstd = torch.empty(rows, D, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        be.layernorm_fwd(x, w, y, mean, rstd, rows, D, 1e-6)
        dx, dw = torch.empty_like(x), torch.zeros(D, device="cuda")
        be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, rows, D)
        return [dx, dw]

    def rope(B, S, H, dc, dr, t16):
        dt = torch.bfloat16 if t16 else torch.float32
        xr, inv = rnd(B, S, H * dr, seed=2).to(dt), torch.rand(dr // 2, generator=torch.Generator().manual_seed(3)).cuda() + 0.01
        g = rnd(B, S, H * (dc + dr), seed=4).to(dt)
        content = rnd(B, S, H * dc, seed=1).to(dt) if dc else None
        table, out = torch.empty(2 * S * (dr // 2), device="cuda"), torch.empty(B, S, H * (dc + dr), device="cuda", dtype=dt)
        be.rope_fwd(content, xr, inv, table, out, B, S, H, dc, dr)
        d_c = torch.empty(B, S, H * dc, device="cuda", dtype=dt) if dc else None
        d_x, d_f = torch.empty(B, S, H * dr, device="cuda", dtype=dt), torch.zeros(dr // 2, device="cuda")
        be.rope_bwd(g, xr, table, d_c, d_x, d_f, B, S, H, dc, dr)
        return [d_x, d_f]

    def colsum(rows, cols, t16):
        x = rnd(rows, cols, seed=5)
        if t16:
            x = x.bfloat16()
        out = torch.zeros(cols, device="cuda")
        be.colsum(x, out, rows, cols)
        return [out]

    def latent(rows, mvh):
        mv, noise = rnd(rows, 2 * mvh, seed=6), rnd(rows, mvh, seed=7)
        z, std, kl = torch.empty(rows, mvh, device="cuda"), torch.empty(rows, mvh, device="cuda"), torch.zeros((), device="cuda")
        be.latent_fwd(mv, noise, z, std, kl, rows, mvh)
        return [z, kl]

    def cnn(B, S):
        Ch = 32
        t = [rnd(B, S, 3 * S, seed=1), rnd(Ch, 3, seed=3) * 0.6, torch.tensor([0.9], device="cuda"), rnd(Ch, seed=4) * 0.1,
             rnd(Ch, 9, seed=5) * 0.4, torch.tensor([1.2], device="cuda"), rnd(Ch, seed=6) * 0.1,
             rnd(3, Ch, seed=7) * 0.3, torch.tensor([0.7], device="cuda"), rnd(3, seed=8) * 0.1]
        dx = torch.empty(B, S, 3 * S, device="cuda")
        gs = [torch.zeros(n, device="cuda") for n in (Ch * 3, Ch, Ch * 9, Ch, 3 * Ch, 3)]
        be.cnn_bwd(rnd(B, S, 3 * S, seed=2), *t, dx, *gs, B, S, Ch)
        return [dx] + gs

    cases = [lambda: ln(57344, 672, True), lambda: ln(20480, 240, False), lambda: ln(33, 30, False),
             lambda: rope(64, 224, 12, 0, 56, True), lambda: rope(16, 176, 12, 22, 22, False), lambda: rope(3, 7, 2, 0, 6, False),
             lambda: colsum(57344, 1344, True), lambda: colsum(4096, 448, False), lambda: colsum(500, 3, False),
             lambda: colsum(40, 1000, False), lambda: latent(20480, 240), lambda: latent(64, 24),
             lambda: cnn(32, 224), lambda: cnn(3, 36)]
    return cases


def test_cross_workgroup_reductions_repeat_bit_for_bit():
    """Each reduction entry point twice (the second time beside a memory-bound stream that perturbs workgroup timing):
    identical bits; the sums themselves are checked against the emulation in test_kernels_gpu.py."""
    be = calm.backend.get_backend()
    side = torch.cuda.Stream()
    a, b = torch.randn(16 << 20, device="cuda"), torch.empty(16 << 20, device="cuda")
    for i, case in enumerate(_reduction_cases(be)):
        ref = case()
        for rep in range(3):
            if rep:
                _busy(side, a, b)
            for got, want in zip(case(), ref):
                assert torch.equal(got, want), (i, rep)
    torch.cuda.synchronize()


def test_reduction_scratch_contract(monkeypatch):
    """calm_reduce_scratch_floats(op, rows, cols) is the contract between caller and library for the `partials` argument
    (include/calm_vit.h, ABI v7): with a buffer of EXACTLY that many floats every entry point produces the bits it
    produces with the backend's large shared buffer, and it writes nothing behind the end (guard words intact)."""
    be = calm.backend.get_backend()
    guard, sentinel = 4096, 12345.0
    made = []

    def exact(op, rows, cols, device):
        need = int(be.lib.calm_reduce_scratch_floats(op, rows, cols))
        assert need > 0
        buf = torch.full((need + guard,), sentinel, dtype=torch.float32, device=device)
        made.append((buf, need))
        return buf.data_ptr()

    cases = _reduction_cases(be)
    refs = [case() for case in cases]
    monkeypatch.setattr(be, "_partials", exact)
    for i, (case, ref) in enumerate(zip(cases, refs)):
        made.clear()
        got = case()
        torch.cuda.synchronize()
        assert made, i
        for buf, need in made:
            assert bool((buf[need:] == sentinel).all()), (i, need)
        for g, w in zip(got, ref):
            assert torch.equal(g, w), i


def test_split_k_weight_gradient_is_reproducible_in_deterministic_mode(deterministic_gemm):
    be = deterministic_gemm
    calm.backend.set_matmul_precision("bf16")
    try:
        M, N, K = 57344, 1344, 672
        dy, x = rnd(M, N, seed=1).bfloat16(), rnd(M, K, seed=2).bfloat16()
        args = (dy, x, None, N, K, M, (1, N, 0, 0), (1, K, 0, 0), (K, 0, 0))
        plan = be.gemm_describe(dy, x, torch.empty(N, K, device="cuda"), *args[3:])
        assert plan["k_slices"] > 1
        outs = []
        for _ in range(3):
            G = torch.zeros(N, K, device="cuda")
            be.gemm(dy, x, G, *args[3:], accumulate=True)
            outs.append(G)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert rel_err(outs[0], dy.float().T @ x.float()) < 1e-4
        # batch-reduced sequence-axis weight gradient (fp32 tensors under the bf16 pipe) and a small output
        B, S2, S, D = 64, 80, 224, 672
        dyb, xb = rnd(B, S2, D, seed=3), rnd(B, S, D, seed=4)
        outs = []
        for _ in range(2):
            G = torch.zeros(S2, S, device="cuda")
            be.gemm(dyb, xb, G, S2, S, D, (D, 1, S2 * D, 0), (D, 1, S * D, 0), (S, 0, 0), batch=(B, 1), reduce_batch=True,
                    accumulate=True)
            outs.append(G)
        assert torch.equal(outs[0], outs[1])
        assert rel_err(outs[0], torch.einsum("bid,bjd->ij", dyb.bfloat16().float(), xb.bfloat16().float())) < 1e-3
    finally:
        calm.backend.set_matmul_precision("fp32")


@pytest.mark.parametrize("name,autocast", [("nano48_cls", False), ("nano48_cls", True), ("tiny32_fr", False)])
def test_model_forward_backward_repeats_bit_for_bit(deterministic_gemm, name, autocast):
    """Two runs of the same training forward + backward (same weights, batch and injected latent noise): logits, KL,
    dL/dx and EVERY parameter gradient identical in every bit — fp32 pipeline and the reference trainer's autocast."""
    g = load_golden(name)
    cfg = CONFIGS[name]
    x0 = torch.from_numpy(W.make_input((4, 3, cfg.seq_length, cfg.seq_length), 2)).cuda()
    runs = []
    for _ in range(2):
        m = build_model(name, g, "cuda").train()
        x = x0.clone().requires_grad_(True)
        calm.ops.set_noise_override(W.NoiseStream(7))
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                y, kl = m(x)
                gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy")).cuda()
                loss = (y.float() * gy).sum() + 0.5 * kl
            loss.backward()
        finally:
            calm.ops.set_noise_override(None)
        torch.cuda.synchronize()
        runs.append((y.detach(), torch.as_tensor(kl).detach(), x.grad, {n: p.grad for n, p in m.named_parameters()}))
    (y0, k0, dx0, g0), (y1, k1, dx1, g1) = runs
    assert torch.equal(y0, y1) and torch.equal(k0, k1) and torch.equal(dx0, dx1)
    differing = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not differing, differing[:8]


@pytest.mark.parametrize("fixture", ["A_hd56", "B_hd44"])
def test_real_size_block_backward_repeats_bit_for_bit_in_the_bf16_pipeline(deterministic_gemm, fixture):
    """One VMLA_Block at Base-224's stage sizes, batch 16, autocast(bfloat16): the pipelined GEMM family (k-split weight
    gradients through the workspace), the pipelined attention kernels and every reduction kernel at real shapes."""
    kw = BLOCK_FIXTURES[fixture]
    torch.manual_seed(0)
    blk = vt.VMLA_Block(kw["heads"], kw["dim1"], kw["dim2"], kw["mean_var_hidden"], kw["seq_length"], kw["seq_len_reduce"],
                        kw["seq_len_new"], kw["dim2"] * 2, force_reduce=False, is_cross=kw["is_cross"]).cuda().train()
    B = 16
    xq0, xkv0 = rnd(B, kw["seq_length"], kw["dim1"], seed=1), rnd(B, kw["seq_length"], kw["dim1"], seed=2)
    state = {k: v.clone() for k, v in blk.state_dict().items()}
    runs = []
    for _ in range(2):
        blk.load_state_dict(state)                      # the training forward advances u, v: both runs start from the same
        for p in blk.parameters():
            p.grad = None
        xq, xkv = xq0.clone().requires_grad_(True), xkv0.clone().requires_grad_(True)
        calm.ops.set_noise_override(W.NoiseStream(3))
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(xq, input_kv=xkv if kw["is_cross"] else None, state_manager=vt.ResidualStateManager(mode="sum"),
                        mask=True)
            (y.float() * rnd(*y.shape, seed=5)).sum().backward()
        finally:
            calm.ops.set_noise_override(None)
        torch.cuda.synchronize()
        runs.append((y.detach().clone(), xq.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}))
    (y0, d0, g0), (y1, d1, g1) = runs
    assert torch.equal(y0, y1) and torch.equal(d0, d1)
    differing = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not differing, differing[:8]


def test_first_block_consumes_row_tokens_bit_for_bit():
    """SURVEY 8f-3 / VERDICT r3 missing #1: the model fed the row tokens [B,S,3S] (what DeviceCollate(tokens=True) writes
    straight from the uint8 batch) computes exactly what it computes from the image [B,3,S,S] — logits, KL and every
    parameter gradient in every bit (deterministic GEMM mode for the comparison of the gradients)."""
    be = calm.backend.get_backend()
    prev = be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, 1)
    try:
        name = "nano48_cls"
        g = load_golden(name)
        cfg = CONFIGS[name]
        S = cfg.seq_length
        col = trainer.DeviceCollate(num_classes=cfg.out_features, seed=4)
        u8 = torch.randint(0, 256, (4, 3, S + 6, S + 6), dtype=torch.uint8, device="cuda")
        labels = torch.tensor([1, 2, 3, 4], device="cuda")
        dec = col.draw(4, S, S)
        img, y_img = col(u8, labels, decisions=dec, crop=(S, S), tokens=False)
        corners = col.last_corners
        # the same decisions and crop corners, written as tokens
        out = torch.empty(4, S, 3 * S, device="cuda")
        be.collate_crop_mix(u8, corners, dec[3].cuda(), out, dec[0], dec[1], dec[2], col.MEAN, col.STD, tokens=True)
        rows = torch.empty_like(out)
        be.image_to_rows(img, rows, 4, S)
        assert torch.equal(out, rows)
        runs = []
        for inp in (img, out):
            m = build_model(name, g, "cuda").train()
            calm.ops.set_noise_override(W.NoiseStream(7))
            try:
                y, kl = m(inp)
                loss = trainer.soft_target_cross_entropy(y.squeeze(), y_img) + 0.1 * kl
                loss.backward()
            finally:
                calm.ops.set_noise_override(None)
            torch.cuda.synchronize()
            runs.append((y.detach(), torch.as_tensor(kl).detach(), {n: p.grad for n, p in m.named_parameters()}))
        (ya, ka, ga), (yb, kb, gb) = runs
        assert torch.equal(ya, yb) and torch.equal(ka, kb)
        assert not [n for n in ga if not torch.equal(ga[n], gb[n])]
    finally:
        be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, prev)


def test_train_launcher_with_device_collate_consumes_uint8_batches():
    """trainer.train(device_collate=True): uint8 images from the DataLoader -> H2D -> DeviceCollate(crop, tokens=True) ->
    the model's first Block (distributed_trainer_cls.py:58-62,128-139 with Vi_Tools:389-391)."""
    name = "tiny32_cls"
    g = load_golden(name)
    cfg = CONFIGS[name]
    S = cfg.seq_length
    gen = torch.Generator().manual_seed(0)
    data = torch.utils.data.TensorDataset(torch.randint(0, 256, (16, 3, S + 4, S + 4), generator=gen, dtype=torch.uint8),
                                          torch.randint(0, cfg.out_features, (16,), generator=gen))
    m = build_model(name, g, "cpu")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    out = trainer.train(m, "fused", scheduler=False, use_gpu=True, dataset=data, epochs=1, batch_size=8,
                        num_classes=cfg.out_features, device_collate=True, crop=(S, S), log_every=1000)
    after = out.state_dict()
    assert all(torch.isfinite(v.float()).all() for v in after.values())
    moved = [k for k in before if k.endswith("weight_orig") and not torch.equal(before[k], after[k])]
    assert len(moved) > 50


def test_train_launcher_graph_mode_replays_the_eager_trajectory():
    """trainer.train(graph=True): the step (autocast + GradScaler + fused optimizer) is captured on the first batch and
    replayed; the capture's warm-up steps are undone (GraphedTrainStep.restore_after_warmup: parameters, u / v buffers,
    AdamW moments, device step counter, scale, RNG state), so four replayed steps leave exactly the model that four
    eager steps leave (deterministic GEMM mode: no atomics anywhere in the step)."""
    name = "tiny32_cls"
    g = load_golden(name)
    cfg = CONFIGS[name]
    S = cfg.seq_length
    gen = torch.Generator().manual_seed(3)
    data = torch.utils.data.TensorDataset(torch.randn(16, 3, S, S, generator=gen),
                                          torch.randint(0, cfg.out_features, (16,), generator=gen))
    be = calm.backend.get_backend()
    prev = be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, 1)
    try:
        outs = []
        for graph in (False, True):
            torch.manual_seed(11)
            torch.cuda.manual_seed(11)
            m = build_model(name, g, "cpu")
            before = {k: v.clone() for k, v in m.state_dict().items()}
            out = trainer.train(m, "fused", scheduler=None, use_gpu=True, dataset=data, epochs=2, batch_size=8,
                                num_classes=cfg.out_features, log_every=1000, graph=graph)
            outs.append({k: v.clone() for k, v in out.state_dict().items()})
    finally:
        be.gemm_set_option(be.GEMM_OPT_DETERMINISTIC, prev)
    eager, replay = outs
    moved = [k for k in eager if k.endswith("weight_orig") and not torch.equal(eager[k], before[k])]
    assert len(moved) > 50                                   # both runs trained
    worst = max(rel_err(replay[k].float(), eager[k].float()) for k in eager)
    assert worst == 0.0, worst


def test_gelu_module_and_bare_proj_are_callable_like_the_reference_sequential():
    """`GELU()(x)` and `block.proj(img)` (Vi_Tools:378-385: conv1x1 -> GELU -> dw3x3 -> GELU -> conv1x1, NO residual) called
    on their own, forward and backward against the emulation."""
    emu = EmulatedBackend()
    x = rnd(5, 37, seed=1).requires_grad_(True)
    y = vt.GELU()(x)
    gy = rnd(5, 37, seed=2)
    y.backward(gy)
    xr = x.detach().cpu().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(gy.cpu())
    assert rel_err(y.detach(), yr.detach()) < 1e-6 and rel_err(x.grad, xr.grad) < 1e-5
    torch.manual_seed(1)
    proj = vt.CnnResidual(32).cuda().train()
    B, S = 2, 36
    img = rnd(B, 3, S, S, seed=3).requires_grad_(True)
    out = proj(img)
    assert out.shape == (B, 3, S, S)
    gout = rnd(B, 3, S, S, seed=4)
    out.backward(gout)
    c0, c2, c4 = proj[0], proj[2], proj[4]
    cpu = lambda t: t.detach().cpu()
    tok = cpu(img).permute(0, 2, 3, 1).reshape(B, S, 3 * S)
    args = (cpu(c0.weight_orig).view(32, 3), cpu(c0._sigma), cpu(c0.bias), cpu(c2.weight_orig).view(32, 9), cpu(c2._sigma),
            cpu(c2.bias), cpu(c4.weight_orig).view(3, 32), cpu(c4._sigma), cpu(c4.bias))
    ref = torch.empty(B, S, 3 * S)
    emu.cnn_fwd(tok, *args, ref, B, S, 32, residual=False)
    assert rel_err(out.detach().permute(0, 2, 3, 1).reshape(B, S, 3 * S), ref) < 1e-5
    dx = torch.empty(B, S, 3 * S)
    gs = [torch.zeros(n) for n in (96, 32, 288, 32, 96, 3)]
    emu.cnn_bwd(cpu(gout).permute(0, 2, 3, 1).reshape(B, S, 3 * S), tok, *args, dx, *gs, B, S, 32, residual=False)
    assert rel_err(img.grad.permute(0, 2, 3, 1).reshape(B, S, 3 * S), dx) < 1e-4
    # with the residual: proj.residual_forward(tokens) = tokens + proj(img) in token layout (eval: u, v, sigma stay put)
    proj.eval()
    toks = img.detach().permute(0, 2, 3, 1).reshape(B, S, 3 * S).contiguous()
    with torch.no_grad():
        bare = proj(img.detach()).permute(0, 2, 3, 1).reshape(B, S, 3 * S)
        both = proj.residual_forward(toks)
    assert rel_err(both, toks + bare) < 1e-6
