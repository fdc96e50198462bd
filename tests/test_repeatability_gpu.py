"""The LDS-DMA pipelines (pipelined GEMM family, attn16 forward / backward v2) order a stage's reads behind its requests
with counted `s_waitcnt vmcnt` + a workgroup barrier; a read that is not covered returns the OLD LDS bytes without a
stall and shows up as rare wrong tiles that come and go with memory load (programming guide, "Pipelining across
barriers").  These kernels have no atomics on the launches used here, so every output must repeat bit for bit — also
while a second stream keeps HBM busy."""
import pytest
import torch

import calm_vit_dte_amd as calm

pytestmark = pytest.mark.gpu


def _load(side, a, b):
    with torch.cuda.stream(side):
        for _ in range(3):
            b.copy_(a)
            a.add_(1.0)


def test_pipelined_gemm_repeats_bit_for_bit_under_memory_load():
    be = calm.backend.get_backend()
    calm.backend.set_matmul_precision("bf16")
    try:
        gen = torch.Generator(device="cuda").manual_seed(3)
        rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=gen) * sc).bfloat16()
        M, K, N = 28672, 672, 1344
        x, w1, dy, w2 = rn(M, K), rn(N, K, sc=K ** -0.5), rn(M, N), rn(N, N, sc=N ** -0.5)
        bias, sigma = torch.randn(N, device="cuda", generator=gen) * 0.1, torch.tensor([1.3], device="cuda")

        def run():
            e = lambda: torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            hg, hp, dz = e(), e(), e()
            be.gemm(x, w1, hg, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), inv_scale=sigma, bias=bias, act=1, C_pre=hp, split_k=1)
            be.gemm(dy, w2, dz, M, N, N, (N, 1, 0, 0), (1, N, 0, 0), (N, 0, 0), inv_scale=sigma, act=2, aux=hp, split_k=1)
            return hg, hp, dz
        ref = run()
        assert all(torch.isfinite(t.float()).all() for t in ref)
        side = torch.cuda.Stream()
        a, b = torch.randn(32 << 20, device="cuda"), torch.empty(32 << 20, device="cuda")
        for it in range(12):
            if it >= 4:
                _load(side, a, b)
            for got, want in zip(run(), ref):
                assert torch.equal(got, want), it
        torch.cuda.synchronize()
    finally:
        calm.backend.set_matmul_precision("fp32")


@pytest.mark.parametrize("S,H,hd", [(224, 12, 56), (176, 12, 44), (80, 12, 20)])
def test_pipelined_attention_repeats_bit_for_bit_under_memory_load(S, H, hd):
    be = calm.backend.get_backend()
    B, D = 32, H * hd
    gen = torch.Generator(device="cuda").manual_seed(S)
    bf = lambda *s, sc=0.5: (torch.randn(*s, device="cuda", generator=gen) * sc).bfloat16()
    q, k, v, dout = bf(B, S, D), bf(B, S, D), bf(B, S, D, sc=1.0), bf(B, S, D, sc=1.0)
    w1, w2 = bf(2 * S, S, sc=S ** -0.5), bf(S, 2 * S, sc=(2 * S) ** -0.5)
    b1, b2 = torch.randn(2 * S, device="cuda", generator=gen) * 0.1, torch.randn(S, device="cuda", generator=gen) * 0.1
    s1, s2 = torch.tensor([1.3], device="cuda"), torch.tensor([0.8], device="cuda")

    def run():
        e = lambda *s: torch.full(s, float("nan"), dtype=torch.bfloat16, device="cuda")
        out, R, hp, hg, Mk, MkT = e(B, S, D), e(B, S, S), e(B, S, 2 * S), e(B, S, 2 * S), e(B, S, S), e(B, S, S)
        lse, delta = torch.full((B, H, S), float("nan"), device="cuda"), torch.full((B, H, S), float("nan"), device="cuda")
        dq, dk, dv, dM = e(B, S, D), e(B, S, D), e(B, S, D), e(B, S, S)
        be.attn16_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, MkT, lse, B, S, H, hd)
        be.attn16_bwd(q, k, v, out, dout, Mk, MkT, lse, delta, dq, dk, dv, dM, B, S, H, hd)
        return [out, R, hp, hg, Mk, MkT, lse, delta, dq, dk, dv, dM]
    ref = run()
    assert all(torch.isfinite(t.float()).all() for t in ref)
    side = torch.cuda.Stream()
    a, b = torch.randn(32 << 20, device="cuda"), torch.empty(32 << 20, device="cuda")
    for it in range(12):
        if it >= 4:
            _load(side, a, b)
        for got, want in zip(run(), ref):
            assert torch.equal(got, want), it
    torch.cuda.synchronize()
