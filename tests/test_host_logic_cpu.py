"""Host-side logic of the package (module tree, manual autograd formulas, stride bookkeeping,
state-dict surface) checked on CPU by running it over tests/emulated_backend.py — a torch emulation
of the C-ABI — against the golden fixtures of the reference.  The HIP kernels themselves are checked
on the GPU (test_kernels_gpu.py / test_model_gpu.py)."""
import numpy as np
import pytest
import torch

import calm_vit_dte_amd as calm
import weights as W
from emulated_backend import EmulatedBackend
from helpers import CONFIGS, WEIGHT_SEED, load_golden, load_inventory, rel_err
from oracle import calm_oracle as O

GOLDEN_CFGS = ["nano48_cls", "nano48_gen", "tiny32_cls", "tiny32_fr"]


def build_model(name, golden, device="cpu"):
    cfg = CONFIGS[name]
    m = calm.ViT(torch.device(device), type=8, heads=cfg.heads, seq_length=cfg.seq_length,
                 in_features=cfg.in_features, dim_step=cfg.dim_step, mean_var_hidden=cfg.mean_var_hidden,
                 seq_len_step=cfg.seq_len_step, seq_len_reduce=cfg.seq_len_reduce,
                 out_features=cfg.out_features, force_reduce=cfg.force_reduce, generate=cfg.generate)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in W.make_params(shapes, WEIGHT_SEED).items()}
    if golden is not None:
        for k in sd:
            if k.endswith(("weight_u", "weight_v")):
                sd[k] = torch.from_numpy(golden["warm/" + k].copy())
    m.load_state_dict(sd, strict=True)
    return m.to(device)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_state_dict_surface_matches_reference(name):
    cfg = CONFIGS[name]
    m = calm.ViT(torch.device("cpu"), type=8, heads=cfg.heads, seq_length=cfg.seq_length,
                 in_features=cfg.in_features, dim_step=cfg.dim_step, mean_var_hidden=cfg.mean_var_hidden,
                 seq_len_step=cfg.seq_len_step, seq_len_reduce=cfg.seq_len_reduce,
                 out_features=cfg.out_features, force_reduce=cfg.force_reduce, generate=cfg.generate)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == load_inventory(name)
    assert all(p.requires_grad for p in m.parameters())


def test_product_path_fails_loudly_on_cpu_tensors():
    m = build_model("tiny32_cls", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize("name", GOLDEN_CFGS)
def test_eval_forward_matches_golden(name):
    g = load_golden(name)
    cfg = CONFIGS[name]
    m = build_model(name, g).eval()
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2))
    with calm.backend.use_backend(EmulatedBackend()), torch.no_grad():
        y, kl = m(x)
    assert rel_err(y, g["eval/y"]) < 2e-5
    assert abs(float(kl) - float(g["eval/kl"])) < 2e-5 * max(1.0, abs(float(g["eval/kl"])))
    if name == "tiny32_cls":
        assert isinstance(kl, float) and kl == 0.0


@pytest.mark.parametrize("name", GOLDEN_CFGS)
def test_train_forward_backward_matches_golden(name):
    g = load_golden(name)
    cfg = CONFIGS[name]
    m = build_model(name, g).train()
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2)).requires_grad_(True)
    calm.ops.set_noise_override(W.NoiseStream(7))
    try:
        with calm.backend.use_backend(EmulatedBackend()):
            y, kl = m(x)
            gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy"))
            loss = (y * gy).sum() + 0.5 * kl
            loss.backward()
    finally:
        calm.ops.set_noise_override(None)
    assert rel_err(y.detach(), g["train/y"]) < 2e-5
    assert abs(float(kl) - float(g["train/kl"])) < 2e-5 * max(1.0, abs(float(g["train/kl"])))
    assert rel_err(x.grad, g["train/dx"]) < 1e-4
    params = dict(m.named_parameters())
    for n, ref in zip([str(s) for s in g["train/grad_names"]], g["train/grad_norms"]):
        got = float(params[n].grad.norm())
        assert abs(got - ref) <= 5e-4 * max(abs(ref), 1e-6) + 1e-9, (n, got, ref)
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(params[key[5:]].grad, g[key]) < 1e-4, key
        if key.startswith("post/"):
            assert rel_err(sd[key[5:]], g[key]) < 2e-5, key


def test_fused_optimizer_step_defers_spectral_norm_gradient_and_matches_torch():
    """trainer.FusedClipAdamW (host side of calm_optim_step) with the emulated backend: backward leaves the gradient
    w.r.t. the normalised weights, the step corrects it, and the parameters end where
    backward -> clip_grad_norm_(1.0) -> torch.optim.AdamW.step() (distributed_trainer_cls.py:87-96) puts them."""
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")
    name = "tiny32_cls"
    g = load_golden(name)
    cfg = CONFIGS[name]
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((4, 3, cfg.seq_length, cfg.seq_length)).astype(np.float32))
    y = torch.nn.functional.one_hot(torch.from_numpy(rng.integers(0, cfg.out_features, 4)), cfg.out_features).float()
    results = []
    with calm.backend.use_backend(EmulatedBackend()):
        for fused in (False, True):
            m = build_model(name, g).train()
            opt = trainer.FusedClipAdamW(m) if fused else trainer.make_optimizer(m)
            step = trainer.TrainStep(m, opt, None)
            try:
                for _ in range(2):
                    step(x, y)
                if fused:
                    assert float(opt.stats[0]) > 0 and float(opt.stats[1]) == 0
                    assert len(opt._deferred) > 100
                    assert all(getattr(p, calm.ops.DEFER_ATTR, False) for p in opt._deferred)
                    assert opt.step_count == 2
            finally:
                if fused:
                    opt.close()
            results.append({k: v.clone() for k, v in m.state_dict().items()})
            assert not any(hasattr(p, calm.ops.DEFER_ATTR) for p in m.parameters())     # close() restored them
    worst = max((rel_err(results[1][k], results[0][k]), k) for k in results[0])
    assert worst[0] < 1e-4, worst                          # Adam amplifies fp32 rounding of near-zero gradients


def test_generative_trainer_step_and_eval_loop():
    """trainer.RegTrainStep (distributed_trainer_reg.py:71-95: Huber(img, x) + 0.1*kl on the generate=True model)
    against the same iteration driven through the CPU oracle, and trainer.evaluate (CALM_ViT_V2.py:228-239)."""
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")
    name = "nano48_gen"
    g = load_golden(name)
    cfg = CONFIGS[name]
    x = torch.from_numpy(W.make_input((2, 3, cfg.seq_length, cfg.seq_length), 2))
    # oracle iteration
    P = {k: torch.from_numpy(v) for k, v in W.make_params(O.vit_param_shapes(cfg), WEIGHT_SEED).items()}
    for k in P:
        if O.is_buffer(k):
            P[k] = torch.from_numpy(g["warm/" + k].copy())
    leaves = [P[k].requires_grad_(True) for k in P if not O.is_buffer(k)]
    y_o, kl_o = O.vit_forward(P, cfg, x, True, W.NoiseStream(7))
    img = y_o.reshape(-1, cfg.seq_length, cfg.seq_length, 3).permute(0, 3, 1, 2)
    loss_o = torch.nn.functional.huber_loss(img, x) + 0.1 * kl_o
    loss_o.backward()
    norm_o = torch.nn.utils.clip_grad_norm_(leaves, 1.0)
    with calm.backend.use_backend(EmulatedBackend()):
        m = build_model(name, g).train()
        opt = trainer.FusedClipAdamW(m)
        step = trainer.RegTrainStep(m, opt, None)
        calm.ops.set_noise_override(W.NoiseStream(7))
        try:
            loss_h, img_h = step(x)
        finally:
            calm.ops.set_noise_override(None)
            opt.close()
        assert img_h.shape == x.shape
        assert abs(float(loss_h) - float(loss_o)) < 1e-5 * max(1.0, abs(float(loss_o)))
        assert abs(float(opt.stats[0]) - float(norm_o)) < 1e-4 * float(norm_o)
        # eval loop on the classification fixture
        mc = build_model("nano48_cls", load_golden("nano48_cls"))
        xs = torch.from_numpy(W.make_input((4, 3, 48, 48), 5))
        with torch.no_grad():
            labels = mc.eval()(xs)[0].reshape(4, -1).argmax(dim=1)
        assert trainer.evaluate(mc, [(xs[:2], labels[:2]), (xs[2:], labels[2:])]) == 1.0
        assert trainer.evaluate(mc, [(xs, (labels + 1) % 10)]) == 0.0


def test_device_collate_soft_labels_and_decisions():
    """trainer.DeviceCollate: CutMix's lam is re-derived from the clipped box, labels are lam*onehot + (1-lam)*rolled
    (rows sum to 1), MixUp / CutMix are both drawn, and the image pass goes through the backend."""
    from importlib import import_module
    trainer = import_module("calm_vit_dte_amd.trainer")
    dc = trainer.DeviceCollate(num_classes=10, seed=3)
    img = torch.randint(0, 256, (6, 3, 32, 32), dtype=torch.uint8)
    labels = torch.tensor([0, 1, 2, 3, 4, 5])
    modes = set()
    with calm.backend.use_backend(EmulatedBackend()):
        for _ in range(12):
            d = dc.draw(6, 32, 32)
            mode, lam, box, flips = d
            modes.add(mode)
            assert 0.0 <= lam <= 1.0 and flips.shape == (6,)
            if mode == 2:
                y1, y2, x1, x2 = box
                assert 0 <= y1 <= y2 <= 32 and 0 <= x1 <= x2 <= 32
                assert abs(lam - (1 - (y2 - y1) * (x2 - x1) / 1024.0)) < 1e-6
            x, y = dc(img, labels, decisions=d)
            assert x.shape == (6, 3, 32, 32) and x.dtype == torch.float32
            assert torch.allclose(y.sum(dim=1), torch.ones(6), atol=1e-6)
            assert abs(float(y[1, 1]) - lam) < 1e-6 and abs(float(y[1, 0]) - (1 - lam)) < 1e-6
    assert modes == {1, 2}


@pytest.mark.parametrize("mode", ["sum", "sma", "ema", "lp", "static"])
def test_residual_state_manager_modes_match_the_oracle_restatement(mode):
    """ResidualStateManager (Vi_Tools:7-50): every mode of the constructor against oracle.LatentState over four merges."""
    vt = calm.Vi_Tools_CNN_less_V2
    sm, st = vt.ResidualStateManager(mode=mode), O.LatentState(mode=mode)
    gen = torch.Generator().manual_seed(0)
    with calm.backend.use_backend(EmulatedBackend()):
        for _ in range(4):
            zq, zkv, mq, mk = (torch.randn(2, 5, 6, generator=gen) for _ in range(4))
            sq, sk = torch.rand(2, 5, 6, generator=gen) + 0.1, torch.rand(2, 5, 6, generator=gen) + 0.1
            a = sm.get_sums(zq, zkv, mq, sq, mk, sk)
            b = st.merge(zq, zkv, mq, sq, mk, sk)
            assert torch.allclose(a[0], b[0], atol=1e-6) and torch.allclose(a[1], b[1], atol=1e-6)
    assert abs(float(sm.get_kl_loss()) - float(st.kl_loss())) < 1e-6


def test_bench_finds_its_kernels_in_the_committed_pmc_summaries():
    """bench.py reads HBM traffic / MFMA utilisation of the dominant kernels from the committed rocprofv3 PMC summaries
    by kernel-name prefix: every prefix it uses must still name rows there (a renamed kernel would silently turn the
    `traffic` field of the bench line into null), and each summary must carry its source stamp."""
    import glob
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    for prefix in list(bench.GEMM_PMC_PREFIX.values()) + list(bench.ATTN_PMC_PREFIX.values()):
        rows, source, stale = bench.pmc_rows(prefix)
        assert rows and source, prefix
        newest = max(int(os.path.basename(f)[5]) for f in glob.glob(os.path.join(root, "profiles", "round?_*pmc_summary.csv")))
        assert source.startswith(f"profiles/round{newest}_"), (prefix, source)      # the newest committed round
        assert stale in (True, False)
    for f in glob.glob(os.path.join(root, "profiles", "round[2-9]_*pmc_summary.csv")):
        assert os.path.exists(f + ".stamp.json"), f


def test_locate_inverts_the_kernels_xcd_aware_item_order():
    """tests/locate.py::item_of_linear (used by the self-locating GEMM mismatch reports) must be the inverse of the item ->
    linear-tile map of the persistent kernels (gemm_bf16p.h::decode: items with equal index mod 8 — one XCD — walk
    consecutive tiles), for every item count: restated here from the kernel source and checked for bijectivity."""
    from locate import item_of_linear

    def lin_of_item(item, n_items):                     # gemm_bf16p.h:483-488
        if n_items < 8:
            return item
        q, rem, x, idx = n_items >> 3, n_items & 7, item & 7, item >> 3
        return (x * (q + 1) if x < rem else rem * (q + 1) + (x - rem) * q) + idx

    for n in list(range(1, 70)) + [255, 256, 257, 528, 705, 1344, 4099]:
        lins = [lin_of_item(i, n) for i in range(n)]
        assert sorted(lins) == list(range(n)), n                      # the kernel's order is a permutation
        assert all(item_of_linear(lins[i], n) == i for i in range(n)), n
