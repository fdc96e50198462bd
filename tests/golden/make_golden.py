#!/usr/bin/env python3
"""Mint golden vectors from the REFERENCE implementation (build container only).

Imports /root/reference/CALM-ViT/{Vi_Tools_CNN_less_V2,CALM_ViT_V2}.py (torchvision, which the
image lacks and only the reference's dataset/__main__ code touches, is stubbed in sys.modules),
fills it with numpy-seeded weights (tests/golden/weights.py), runs 5 train() warm-up forwards so
the spectral-norm u,v converge (fresh-init eval() output is NaN), then records
eval and train forward/backward results.  Only DATA is written: name->shape inventories (json)
and tensors (npz).  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import weights as W  # noqa: E402

REF = "/root/reference/CALM-ViT"

CONFIGS = {
    # SURVEY.md 8(d) variant table
    "nano48_cls": dict(heads=3, seq_length=48, in_features=144, dim_step=12, mean_var_hidden=24,
                       seq_len_step=4, seq_len_reduce=16, out_features=10, force_reduce=False, generate=False),
    "nano48_gen": dict(heads=3, seq_length=48, in_features=144, dim_step=12, mean_var_hidden=24,
                       seq_len_step=4, seq_len_reduce=16, out_features=10, force_reduce=False, generate=True),
    "tiny32_cls": dict(heads=4, seq_length=32, in_features=96, dim_step=0, mean_var_hidden=24,
                       seq_len_step=0, seq_len_reduce=16, out_features=10, force_reduce=False, generate=False),
    "tiny32_fr": dict(heads=4, seq_length=32, in_features=96, dim_step=0, mean_var_hidden=24,
                      seq_len_step=0, seq_len_reduce=16, out_features=10, force_reduce=True, generate=False),
}
INVENTORY_ONLY = {}
# BASELINE.json configs #3-#5 at their real sizes, batch 1 ("logits of Base-224 at bs=1"):
# inventory + golden_<name>_b1.npz with eval / train outputs, dL/dx, every gradient norm, the small gradients in full
REAL_SIZE = {
    # BASELINE configs[1], the configuration the headline number is quoted on (pinned by the reference itself)
    "small224_cls": dict(heads=6, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=120,
                         seq_len_step=16, seq_len_reduce=40, out_features=1000, force_reduce=False, generate=False),
    "base224_cls": dict(heads=12, seq_length=224, in_features=672, dim_step=48, mean_var_hidden=240,
                        seq_len_step=16, seq_len_reduce=80, out_features=1000, force_reduce=False, generate=False),
    "base384_cls": dict(heads=12, seq_length=384, in_features=1152, dim_step=48, mean_var_hidden=240,
                        seq_len_step=16, seq_len_reduce=80, out_features=1000, force_reduce=False, generate=False),
    "large224_cls": dict(heads=6, seq_length=224, in_features=672, dim_step=24, mean_var_hidden=480,
                         seq_len_step=8, seq_len_reduce=160, out_features=1000, force_reduce=False, generate=False),
}
# single VMLA_Block fixtures at the real head dims of Base-224: mode A (plain) and
# mode B (latent, seq + feature reduction) — kwargs of the reference's VMLA_Block
BLOCKS = {
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
FULL_GRAD_MAX_REAL = 20000        # real-size fixtures keep only the gradients up to this many elements in full
BATCH = 2
WEIGHT_SEED = 1234
FULL_GRAD_PREFIXES = ("autoencoder.encoder_blocks.0.encoder.", "autoencoder.encoder_blocks.0.cross.",
                      "autoencoder.encoder_blocks.0.proj.", "autoencoder.decoder_blocks.2.cross.",
                      "autoencoder.ln_final.", "head.", "proj.")


def import_reference():
    for m in ("torchvision", "torchvision.transforms", "torchvision.transforms.v2", "torchvision.datasets",
              "torchvision.transforms.functional", "torchvision.models"):
        sys.modules.setdefault(m, MagicMock())
    sys.path.insert(0, REF)
    import CALM_ViT_V2 as rvh
    return rvh


def run_with_noise(fn, seed):
    orig = torch.randn_like
    torch.randn_like = W.NoiseStream(seed)
    try:
        return fn()
    finally:
        torch.randn_like = orig


def kl_value(kl):
    return np.float32(kl.item() if torch.is_tensor(kl) else kl)


def mint(name, kw, rvh, batch=BATCH, suffix="", full_grad_max=None):
    model = rvh.ViT(torch.device("cpu"), type=8, **kw)
    sd = model.state_dict()
    shapes = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(HERE, f"state_dict_{name}.json"), "w") as f:
        json.dump(shapes, f, indent=0, sort_keys=True)
    params = W.make_params(shapes, WEIGHT_SEED)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    S = kw["seq_length"]
    out = {}
    # 1. warm-up: 5 train forwards
    xw = torch.from_numpy(W.make_input((batch, 3, S, S), 1))
    model.train()
    for i in range(5):
        with torch.no_grad():
            run_with_noise(lambda: model(xw), 100 + i)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith(("weight_u", "weight_v")):
            out["warm/" + k] = v.detach().numpy().copy()
    # 2. eval forward
    x = torch.from_numpy(W.make_input((batch, 3, S, S), 2)).requires_grad_(True)
    model.eval()
    with torch.no_grad():
        y, kl = model(x)
    out["eval/y"] = y.numpy().copy()
    out["eval/kl"] = kl_value(kl)
    # 3. train forward + backward
    model.train()
    y, kl = run_with_noise(lambda: model(x), 7)
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 3, "gy"))
    loss = (y * gy).sum() + 0.5 * kl
    loss.backward()
    out["train/y"] = y.detach().numpy().copy()
    out["train/kl"] = kl_value(kl)
    out["train/loss"] = np.float32(loss.item())
    out["train/dx"] = x.grad.numpy().copy()
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        if k.startswith(FULL_GRAD_PREFIXES) and (full_grad_max is None or p.numel() <= full_grad_max):
            out["grad/" + k] = p.grad.numpy().copy()
    out["train/grad_names"] = np.array(names)
    out["train/grad_norms"] = np.array(norms, dtype=np.float32)
    for k, v in model.state_dict().items():
        if k.endswith(("weight_u", "weight_v")) and k.startswith(FULL_GRAD_PREFIXES):
            out["post/" + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"golden_{name}{suffix}.npz"), **out)
    print(name + suffix, "params", sum(int(np.prod(s)) for k, s in shapes.items()), "loss", out["train/loss"],
          "kl", out["train/kl"], "|y|max", float(np.abs(out["eval/y"]).max()))


def mint_block(name, kw, vtools):
    """One reference VMLA_Block (Vi_Tools_CNN_less_V2.py:98-315) at batch 1: numpy-seeded weights, 3 train-mode
    warm-up forwards, then a train forward with injected noise, KL through a ResidualStateManager("sum") and the
    backward of sum(y*gy) + 0.5*kl.  Stored: warm u/v, y, dL/dxq (dL/dxkv), kl, every parameter-gradient norm, the
    gradients up to FULL_GRAD_MAX_REAL elements in full, u/v after the forward."""
    blk = vtools.VMLA_Block(mlp_dim=2 * kw["dim2"], force_reduce=False, **kw)
    shapes = {k: list(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_params(shapes, BLOCK_WEIGHT_SEED).items()})
    S, D1, cross = kw["seq_length"], kw["dim1"], kw["is_cross"]
    xq = torch.from_numpy(W.make_input((1, S, D1), 5, "xq"))
    xkv = torch.from_numpy(W.make_input((1, S, D1), 6, "xkv")) if cross else None
    blk.train()
    for i in range(3):
        with torch.no_grad():
            run_with_noise(lambda: blk(xq, input_kv=xkv, state_manager=vtools.ResidualStateManager(mode="sum"),
                                       mask=True), 50 + i)
    out = {"shape_names": np.array(sorted(shapes)), }
    for k in sorted(shapes):
        out["shape/" + k] = np.array(shapes[k], dtype=np.int64)
    for k, v in blk.state_dict().items():
        if k.endswith(("weight_u", "weight_v")):
            out["warm/" + k] = v.detach().numpy().copy()
    xq = xq.clone().requires_grad_(True)
    xkv = xkv.clone().requires_grad_(True) if cross else None
    sm = vtools.ResidualStateManager(mode="sum")
    y = run_with_noise(lambda: blk(xq, input_kv=xkv, state_manager=sm, mask=True), 9)
    gy = torch.from_numpy(W.make_input(tuple(y.shape), 8, "gy"))
    kl = sm.get_kl_loss()
    loss = (y * gy).sum() + 0.5 * kl
    loss.backward()
    out["y"] = y.detach().numpy().copy()
    out["kl"] = kl_value(kl)
    out["loss"] = np.float32(loss.item())
    out["dxq"] = xq.grad.numpy().copy()
    if cross:
        out["dxkv"] = xkv.grad.numpy().copy()
    names, norms = [], []
    for k, p in blk.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()))
        if p.numel() <= FULL_GRAD_MAX_REAL:
            out["grad/" + k] = p.grad.numpy().copy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float32)
    for k, v in blk.state_dict().items():
        if k.endswith(("weight_u", "weight_v")):
            out["post/" + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"golden_block_{name}.npz"), **out)
    print("block", name, "loss", out["loss"], "kl", out["kl"], "|y|max", float(np.abs(out["y"]).max()))


def main():
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "4")))
    rvh = import_reference()
    only = set(sys.argv[1:])
    import Vi_Tools_CNN_less_V2 as vtools          # the reference's module (REF is on sys.path)
    for name, kw in CONFIGS.items():
        if not only or name in only:
            mint(name, kw, rvh)
    for name, kw in BLOCKS.items():
        if not only or "block_" + name in only:
            mint_block(name, kw, vtools)
    for name, kw in REAL_SIZE.items():
        if not only or name in only:
            mint(name, kw, rvh, batch=1, suffix="_b1", full_grad_max=FULL_GRAD_MAX_REAL)
    for name, kw in INVENTORY_ONLY.items():
        model = rvh.ViT(torch.device("cpu"), type=8, **kw)
        shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
        with open(os.path.join(HERE, f"state_dict_{name}.json"), "w") as f:
            json.dump(shapes, f, indent=0, sort_keys=True)
        print(name, "entries", len(shapes))


if __name__ == "__main__":
    main()
