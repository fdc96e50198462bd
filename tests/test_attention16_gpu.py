"""The bf16 latent-mask attention kernels (csrc/attention_bf16.hip: calm_attention16_fwd / _bwd) through the C-ABI
against the torch emulation of their rounding points (tests/emulated_backend.py) at every (S, H, hd) of the five
BASELINE configs and of the fixture models — including key counts that are not multiples of 32 (pad keys masked
in-kernel) and head dims that are not multiples of 32 (zero-padded), and against exact fp32 math with a bf16-level
tolerance.  Tolerances are max-abs relative: one bf16 ulp of the largest element is 2^-8 = 3.9e-3."""
import math

import pytest
import torch

import calm_vit_dte_amd as calm
from emulated_backend import EmulatedBackend
from helpers import rel_err

pytestmark = pytest.mark.gpu

SHAPES = [
    # S, H, hd                       config / stage
    (224, 12, 56), (176, 12, 44), (128, 12, 32), (80, 12, 20),          # Base-224
    (224, 6, 112), (176, 6, 88), (128, 6, 64), (80, 6, 40),             # Small-224
    (384, 12, 96), (336, 12, 84), (288, 12, 72), (240, 12, 60),         # Base-384
    (200, 6, 100), (152, 6, 76),                                        # Large-224 (S not a multiple of 16)
    (48, 3, 48), (24, 3, 24), (32, 4, 24),                              # Nano-48 / Tiny-32 stages with S % 8 == 0
]


def _inputs(B, S, H, hd, seed=0):
    g = torch.Generator().manual_seed(seed)
    D = H * hd
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    q, k, v = rn(B, S, D, sc=0.5).bfloat16(), rn(B, S, D, sc=0.5).bfloat16(), rn(B, S, D).bfloat16()
    w1, w2 = rn(2 * S, S, sc=S ** -0.5).bfloat16(), rn(S, 2 * S, sc=(2 * S) ** -0.5).bfloat16()
    b1, b2 = rn(2 * S, sc=0.1), rn(S, sc=0.1)
    s1, s2 = torch.tensor([1.3]), torch.tensor([0.8])
    return q, k, v, w1, b1, s1, w2, b2, s2


def _outputs(B, S, H, hd, dev):
    """out, R, hp, hg, Mk, lse (NaN-filled: every element must be written by the kernel)"""
    bf = lambda *s: torch.full(s, float("nan"), dtype=torch.bfloat16, device=dev)
    return (bf(B, S, H * hd), bf(B, S, S), bf(B, S, 2 * S), bf(B, S, 2 * S), bf(B, S, S),
            torch.full((B, H, S), float("nan"), device=dev))


@pytest.mark.parametrize("S,H,hd", SHAPES)
def test_attention16_forward_matches_emulation(S, H, hd):
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    assert hip.attn16_supported(S, H, hd) and emu.attn16_supported(S, H, hd)
    B = 3
    ins = _inputs(B, S, H, hd)
    ref = list(_outputs(B, S, H, hd, "cpu"))
    got = list(_outputs(B, S, H, hd, "cuda"))
    ref.insert(5, torch.empty_like(ref[4]))                 # MkT
    got.insert(5, torch.full_like(got[4], float("nan")))
    emu.attn16_fwd(*ins, *ref, B, S, H, hd)
    hip.attn16_fwd(*[t.cuda() for t in ins], *got, B, S, H, hd)
    assert torch.equal(got[5], got[4].transpose(1, 2))      # the transposed copy of the mask
    names = ["out", "R", "hp", "hg", "Mk", "MkT", "lse"]
    for n, a, b in zip(names, got, ref):
        assert torch.isfinite(a.float()).all(), n
        # R / hp / hg differ by bf16 rounding flips only (<= 1 ulp of the largest element); a flip of a mask element
        # (one ulp of a mask value of magnitude ~20 is ~0.08) moves the logits, hence lse and the output, further
        tol = 8e-3 if n == "lse" else 3e-2 if n == "out" else 1.2e-2 if n in ("Mk", "MkT") else 6e-3
        assert rel_err(a.float(), b.float()) < tol, (n, rel_err(a.float(), b.float()))
    # most elements are bit-identical (the rest differ by a rounding flip)
    assert (got[1].cpu() == ref[1]).float().mean() > 0.98
    assert (got[0].cpu() == ref[0]).float().mean() > 0.80
    # and against exact fp32 attention math on the same bf16 inputs: bf16-level agreement
    q, k, v, w1, b1, s1, w2, b2, s2 = [t.float() for t in ins]
    raw = q @ k.transpose(1, 2)
    mask = torch.nn.functional.gelu(raw @ (w1 / s1).t() + b1) @ (w2 / s2).t() + b2
    qh, kh, vh = (t.view(B, S, H, hd).transpose(1, 2) for t in (q, k, v))
    o = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd) + mask[:, None], dim=-1) @ vh
    assert rel_err(got[0].float(), o.transpose(1, 2).reshape(B, S, H * hd)) < 6e-2


@pytest.mark.parametrize("S,H,hd", SHAPES)
def test_attention16_backward_matches_emulation(S, H, hd):
    """dq, dk, dv, dM and delta of calm_attention16_bwd (P recomputed in-kernel from q, k, the saved mask and the row
    log-sum-exp) against the emulation, both fed the SAME saved tensors (the emulation's forward outputs), and dq/dk/dv
    against autograd through exact fp32 attention math on the same bf16 inputs (bf16-level tolerance)."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    B = 2
    ins = _inputs(B, S, H, hd, seed=1)
    q, k, v, w1, b1, s1, w2, b2, s2 = ins
    out, R, hp, hg, Mk, lse = _outputs(B, S, H, hd, "cpu")
    MkT = torch.empty_like(Mk)
    emu.attn16_fwd(*ins, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd)
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(B, S, H * hd, generator=g).bfloat16()
    mk = lambda dev: [torch.full((B, S, H * hd), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3)] + \
        [torch.full((B, S, S), float("nan"), dtype=torch.bfloat16, device=dev)]
    dq_r, dk_r, dv_r, dM_r = mk("cpu")
    dq_h, dk_h, dv_h, dM_h = mk("cuda")
    delta_r, delta_h = torch.zeros(B, H, S), torch.full((B, H, S), float("nan"), device="cuda")
    emu.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta_r, dq_r, dk_r, dv_r, dM_r, B, S, H, hd)
    c = lambda t: t.cuda()
    hip.attn16_bwd(c(q), c(k), c(v), c(out), c(dout), c(Mk), c(MkT), c(lse), delta_h, dq_h, dk_h, dv_h, dM_h, B, S, H, hd)
    assert rel_err(delta_h, delta_r) < 1e-4
    for n, a, b_ in (("dq", dq_h, dq_r), ("dk", dk_h, dk_r), ("dv", dv_h, dv_r), ("dM", dM_h, dM_r)):
        assert torch.isfinite(a.float()).all(), n
        assert rel_err(a.float(), b_.float()) < 1.2e-2, (n, rel_err(a.float(), b_.float()))
        assert (a.cpu() == b_).float().mean() > 0.90, (n, float((a.cpu() == b_).float().mean()))
    # exact math: autograd through fp32 attention with the SAME (rounded) mask held fixed
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, S, H, hd).transpose(1, 2) for t in (qf, kf, vf))
    o = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd) + Mk.float()[:, None], dim=-1) @ vh
    o.transpose(1, 2).reshape(B, S, H * hd).backward(dout.float())
    for n, a, ref in (("dq", dq_h, qf.grad), ("dk", dk_h, kf.grad), ("dv", dv_h, vf.grad)):
        assert rel_err(a.float(), ref) < 4e-2, (n, rel_err(a.float(), ref))


@pytest.mark.parametrize("S,H,hd", [(176, 12, 44), (80, 12, 20), (128, 12, 32),            # pipelined kernels (Base-224 stages)
                                    (224, 6, 112), (200, 6, 100), (288, 12, 72)])          # register-staged kernels (hd > 64)
def test_attention16_eight_images_take_the_xcd_paired_order(S, H, hd):
    """B % 8 == 0 switches the attn16 kernels to the XCD-paired workgroup order (image = 8 (idx / groups) + id % 8), and
    with hd % 8 == 4 the last image's last head stages its last row through the zero block + the by-hand half chunk:
    forward and backward at B = 8 against the emulation, and image by image against the B = 3 order (a batch is a set of
    independent images: the same image must give the same bits whichever order its workgroups ran in)."""
    hip, emu = calm.backend.get_backend(), EmulatedBackend()
    B = 8
    ins = _inputs(B, S, H, hd, seed=3)
    q, k, v, w1, b1, s1, w2, b2, s2 = ins
    ref = list(_outputs(B, S, H, hd, "cpu"))
    got = list(_outputs(B, S, H, hd, "cuda"))
    ref.insert(5, torch.empty_like(ref[4]))
    got.insert(5, torch.full_like(got[4], float("nan")))
    emu.attn16_fwd(*ins, *ref, B, S, H, hd)
    hip.attn16_fwd(*[t.cuda() for t in ins], *got, B, S, H, hd)
    for n, a, b_ in zip(["out", "R", "hp", "hg", "Mk", "MkT", "lse"], got, ref):
        assert torch.isfinite(a.float()).all(), n
        tol = 8e-3 if n == "lse" else 3e-2 if n == "out" else 1.2e-2 if n in ("Mk", "MkT") else 6e-3
        assert rel_err(a.float(), b_.float()) < tol, (n, rel_err(a.float(), b_.float()))
    # the last three images alone (B = 3: plain order, and image 2 is then the tensor-end image as well)
    sub = [t[5:].contiguous().cuda() if t.dim() == 3 and t.shape[0] == B else t.cuda() for t in ins]
    got3 = list(_outputs(3, S, H, hd, "cuda"))
    got3.insert(5, torch.full_like(got3[4], float("nan")))
    hip.attn16_fwd(*sub, *got3, 3, S, H, hd)
    for a8, a3 in zip(got, got3):
        assert torch.equal(a8[5:], a3)
    # backward on the emulation's saved tensors
    out, R, hp, hg, Mk, MkT, lse = ref
    dout = torch.randn(B, S, H * hd, generator=torch.Generator().manual_seed(9)).bfloat16()
    mk = lambda dev: [torch.full((B, S, H * hd), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3)] + \
        [torch.full((B, S, S), float("nan"), dtype=torch.bfloat16, device=dev)]
    dq_r, dk_r, dv_r, dM_r = mk("cpu")
    dq_h, dk_h, dv_h, dM_h = mk("cuda")
    delta_r, delta_h = torch.zeros(B, H, S), torch.full((B, H, S), float("nan"), device="cuda")
    emu.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta_r, dq_r, dk_r, dv_r, dM_r, B, S, H, hd)
    c = lambda t: t.cuda()
    hip.attn16_bwd(c(q), c(k), c(v), c(out), c(dout), c(Mk), c(MkT), c(lse), delta_h, dq_h, dk_h, dv_h, dM_h, B, S, H, hd)
    assert rel_err(delta_h, delta_r) < 1e-4
    for n, a, b_ in (("dq", dq_h, dq_r), ("dk", dk_h, dk_r), ("dv", dv_h, dv_r), ("dM", dM_h, dM_r)):
        assert torch.isfinite(a.float()).all(), n
        assert rel_err(a.float(), b_.float()) < 1.2e-2, (n, rel_err(a.float(), b_.float()))


def test_attention16_forward_v3_experimental_kernels_match_in_a_child_process():
    """CALM_ATTN16_V3=1 (read once per process): mask kernel + persistent per-(image, head) core kernel
    (csrc/attention_bf16_fwd3.h) against the emulation of ITS rounding points, on the four stage shapes of Base-224."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, CALM_ATTN16_V3="1")
    nodes = [f"{os.path.abspath(__file__)}::test_attention16_forward_matches_emulation[{S}-{H}-{hd}]" for S, H, hd in SHAPES[:4]]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x"] + nodes,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
