"""Fused cross-axial latent-mask attention kernel (calm_attention_fwd) on the MI355X against the
torch emulation on CPU, at every (S, heads, head-dim) the Small-224 / Base-224 / Nano-48 / Tiny-32
models use, plus the composite fallback for a shape without a fused instantiation.  fp32, 1e-4 rel."""
import math

import pytest
import torch

import calm_vit_dte_amd as calm
from emulated_backend import EmulatedBackend
from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4

SHAPES = [
    # B, S, H, hd
    (2, 224, 6, 112), (2, 176, 6, 88), (2, 128, 6, 64), (3, 80, 6, 40),        # Small-224 stages
    (1, 224, 12, 56), (2, 176, 12, 44), (2, 128, 12, 32), (2, 80, 12, 20),     # Base-224 stages
    (2, 48, 3, 48), (2, 32, 4, 24),                                            # Nano-48 / Tiny-32 first stage
]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _inputs(B, S, H, hd):
    D = H * hd
    q, k, v = rnd(B, S, D, seed=1) * 0.5, rnd(B, S, D, seed=2) * 0.5, rnd(B, S, D, seed=3)
    w1, b1 = rnd(2 * S, S, seed=4) / math.sqrt(S), rnd(2 * S, seed=5) * 0.1
    w2, b2 = rnd(S, 2 * S, seed=6) / math.sqrt(2 * S), rnd(S, seed=7) * 0.1
    s1, s2 = torch.tensor([0.8]), torch.tensor([1.3])
    return q, k, v, w1, b1, s1, w2, b2, s2


@pytest.mark.parametrize("B,S,H,hd", SHAPES)
def test_fused_attention_forward(B, S, H, hd):
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    assert hip.attn_fwd_supported(S, S, H, hd) and emu.attn_fwd_supported(S, S, H, hd)
    ins = _inputs(B, S, H, hd)
    D = H * hd
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        t = [x.to(dev) for x in ins]
        out, R = torch.empty(B, S, D, device=dev), torch.empty(B, S, S, device=dev)
        hp, hg = torch.empty(B, S, 2 * S, device=dev), torch.empty(B, S, 2 * S, device=dev)
        Mk, P = torch.empty(B, S, S, device=dev), torch.empty(B, H, S, S, device=dev)
        be.attn_fwd(*t, out, R, hp, hg, Mk, P, B, S, S, H, hd)
        outs.append((out, R, hp, hg, Mk, P))
    for name, a, b in zip(("out", "R", "hp", "hg", "Mk", "P"), outs[1], outs[0]):
        assert rel_err(a, b) < TOL, name
    assert torch.allclose(outs[1][5].sum(dim=-1).cpu(), torch.ones(B, H, S), atol=1e-5)   # rows of P sum to 1


@pytest.mark.parametrize("B,S,H,hd", SHAPES)
def test_fused_attention_backward_core(B, S, H, hd):
    """dP / softmax backward / head-sum / dQ, dK, dV in the two fused launches vs the emulation."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    D = H * hd
    q, k, v = rnd(B, S, D, seed=1) * 0.5, rnd(B, S, D, seed=2) * 0.5, rnd(B, S, D, seed=3)
    dout = rnd(B, S, D, seed=8)
    P = torch.softmax(rnd(B, H, S, S, seed=9) * 2, dim=-1)
    outs = []
    for be, dev in ((emu, "cpu"), (hip, "cuda")):
        t = [x.to(dev) for x in (q, k, v, dout, P)]
        e = lambda *s: torch.full(s, float("nan"), device=dev)
        dS, dq, dk, dv, dM = e(B, H, S, S), e(B, S, D), e(B, S, D), e(B, S, D), e(B, S, S)
        be.attn_bwd(*t, dS, dq, dk, dv, dM, B, S, S, H, hd)
        outs.append((dS, dq, dk, dv, dM))
    for name, a, b in zip(("dS", "dq", "dk", "dv", "dM"), outs[1], outs[0]):
        assert rel_err(a, b) < TOL, name


def test_fused_attention_without_probability_output():
    hip = calm.backend.get_backend()
    B, S, H, hd = 2, 80, 6, 40
    t = [x.cuda() for x in _inputs(B, S, H, hd)]
    D = H * hd
    mk = lambda *s: torch.empty(*s, device="cuda")
    out1, out2 = mk(B, S, D), mk(B, S, D)
    hip.attn_fwd(*t, out1, mk(B, S, S), mk(B, S, 2 * S), mk(B, S, 2 * S), mk(B, S, S), mk(B, H, S, S), B, S, S, H, hd)
    hip.attn_fwd(*t, out2, mk(B, S, S), mk(B, S, 2 * S), mk(B, S, 2 * S), mk(B, S, S), None, B, S, S, H, hd)
    assert torch.equal(out1, out2)


def test_unsupported_shapes_are_reported_not_run():
    hip = calm.backend.get_backend()
    assert not hip.attn_fwd_supported(36, 36, 3, 36)      # S not a multiple of 16 (Nano-48 inner stages)
    assert not hip.attn_fwd_supported(64, 64, 4, 24)      # no instantiation for 4 key tiles
    assert not hip.attn_fwd_supported(224, 176, 6, 112)   # Sq != Skv never occurs in the model
    t = [x.cuda() for x in _inputs(1, 36, 3, 36)]
    e = lambda *s: torch.empty(*s, device="cuda")
    with pytest.raises(RuntimeError):
        hip.attn_fwd(*t, e(1, 36, 108), e(1, 36, 36), e(1, 36, 72), e(1, 36, 72), e(1, 36, 36), None, 1, 36, 36, 3, 36)
