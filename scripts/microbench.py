#!/usr/bin/env python3
"""In-process A/B timing of individual kernels of libcalmvit_hip.so (HIP events, median of N)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calm_vit_dte_amd as calm

be = calm.backend.get_backend()


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3   # us


def ln(rows, D):
    def mk(aligned):
        base = torch.randn(rows * D + 4, device="cuda")
        return (base[:rows * D] if aligned else base[1:rows * D + 1]).view(rows, D)
    for aligned in (False, True):
        x, dy, y, dx = mk(aligned), mk(aligned), mk(aligned), mk(aligned)
        w = torch.ones(D + 4, device="cuda")[(0 if aligned else 1):][:D]
        mean, rstd, dw = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda"), torch.zeros(D, device="cuda")
        tf = timeit(lambda: be.layernorm_fwd(x, w, y, mean, rstd, rows, D, 1e-6))
        tb = timeit(lambda: be.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, rows, D))
        gb = rows * D * 4 / 1e9
        print(f"LN [{rows}x{D}] {'vec' if aligned else 'scalar'}: fwd {tf:7.1f} us ({2*gb/tf*1e6/1e3:6.2f} TB/s)  "
              f"bwd {tb:7.1f} us ({3*gb/tb*1e6/1e3:6.2f} TB/s)")


if __name__ == "__main__":
    for rows, D in ((57344, 672), (45056, 528), (32768, 384), (20480, 240)):
        ln(rows, D)


def cnn(B, S):
    Ch = 32
    g = lambda *s: torch.randn(*s, device="cuda")
    x, dy = g(B, S, 3 * S), g(B, S, 3 * S)
    w0, b0, w2, b2, w4, b4 = g(Ch, 3), g(Ch), g(Ch, 9), g(Ch), g(3, Ch), g(3)
    s0 = s2 = s4 = torch.ones(1, device="cuda")
    out, dx = torch.empty_like(x), torch.empty_like(x)
    gs = [torch.zeros(n, device="cuda") for n in (Ch * 3, Ch, Ch * 9, Ch, 3 * Ch, 3)]
    tf = timeit(lambda: be.cnn_fwd(x, w0, s0, b0, w2, s2, b2, w4, s4, b4, out, B, S, Ch), n=8)
    tb = timeit(lambda: be.cnn_bwd(dy, x, w0, s0, b0, w2, s2, b2, w4, s4, b4, dx, *gs, B, S, Ch), n=8)
    print(f"CNN B={B} S={S}: fwd {tf:8.1f} us  bwd {tb:8.1f} us   ({B*S*S/1e6:.1f} Mpix)")


def attn(B, S, H, hd):
    D = H * hd
    g = lambda *s: torch.randn(*s, device="cuda") * 0.3
    q, k, v = g(B, S, D), g(B, S, D), g(B, S, D)
    w1, b1, w2, b2 = g(2 * S, S) * 0.1, g(2 * S), g(S, 2 * S) * 0.1, g(S)
    s1 = s2 = torch.ones(1, device="cuda")
    e = lambda *s: torch.empty(*s, device="cuda")
    out, R, hp, hg, Mk, P = e(B, S, D), e(B, S, S), e(B, S, 2 * S), e(B, S, 2 * S), e(B, S, S), e(B, H, S, S)
    t = timeit(lambda: be.attn_fwd(q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, P, B, S, S, H, hd), n=8)
    fl = B * (2 * S * S * D + 8 * S * S * S + 4 * S * S * D)
    print(f"ATTN fwd B={B} S={S} H={H} hd={hd}: {t:8.1f} us  {fl/t/1e6:6.1f} TFLOP/s")


def streaming():
    B, S, H, hd = 256, 224, 6, 112
    D = H * hd
    n = B * S * D
    gb = n * 4 / 1e9
    g = lambda *s: torch.randn(*s, device="cuda")
    a, b_, c = g(n), g(n), g(n)
    t = timeit(lambda: be.add(a, b_, c, n)); print(f"add            {t:7.1f} us {3*gb/t*1e3:6.2f} TB/s")
    xr, out, inv, tab = g(B, S, D), g(B, S, D), torch.rand(hd // 2, device="cuda"), torch.empty(S * hd, device="cuda")
    t = timeit(lambda: be.rope_fwd(None, xr, inv, tab, out, B, S, H, 0, hd)); print(f"rope_fwd       {t:7.1f} us {2*gb/t*1e3:6.2f} TB/s")
    dxr, dif = g(B, S, D), torch.zeros(hd // 2, device="cuda")
    t = timeit(lambda: be.rope_bwd(out, xr, tab, None, dxr, dif, B, S, H, 0, hd)); print(f"rope_bwd       {t:7.1f} us {3*gb/t*1e3:6.2f} TB/s")
    P = g(B * H * S, S); dP = g(B * H * S, S); pg = P.numel() * 4 / 1e9
    t = timeit(lambda: be.softmax_fwd(P, B * H * S, S)); print(f"softmax_fwd    {t:7.1f} us {2*pg/t*1e3:6.2f} TB/s")
    t = timeit(lambda: be.softmax_bwd(P, dP, B * H * S, S)); print(f"softmax_bwd    {t:7.1f} us {3*pg/t*1e3:6.2f} TB/s")
    dm = g(B, S * S)
    t = timeit(lambda: be.sum_heads(dP, dm, B, H, S * S)); print(f"sum_heads      {t:7.1f} us {pg/t*1e3:6.2f} TB/s")
    x2 = g(B * S, 448); o = torch.zeros(448, device="cuda")
    t = timeit(lambda: be.colsum(x2, o, B * S, 448)); print(f"colsum 448     {t:7.1f} us {x2.numel()*4/1e9/t*1e3:6.2f} TB/s")
    tok, tok2 = g(B, S, 3 * S), g(B, S, 3 * S)
    t = timeit(lambda: be.grid_transpose(tok, tok2, B, S)); print(f"grid_transpose {t:7.1f} us {2*gb/t*1e3:6.2f} TB/s")
    z = g(B * S, 1344); dz = g(B * S, 1344); o2 = g(B * S, 1344)
    t = timeit(lambda: be.gelu_bwd(dz, z, o2, z.numel())); print(f"gelu_bwd       {t:7.1f} us {3*z.numel()*4/1e9/t*1e3:6.2f} TB/s")


if __name__ == "__main__":
    for B, S in ((256, 224), (256, 176), (256, 80)):
        cnn(B, S)
    for S, hd in ((224, 112), (176, 88), (128, 64), (80, 40)):
        attn(256, S, 6, hd)
    streaming()
