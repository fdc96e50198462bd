import os, sys
sys.path.insert(0, os.getcwd())
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")
def t_med(fn, n=12, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]
g = lambda *s: torch.randn(*s, device="cuda").bfloat16()
f = lambda *s: torch.randn(*s, device="cuda")
for M, N in ((57344, 672), (57344, 1344), (45056, 1056)):
    for K in (32, 64, 128, 256, 672):
        x, w = g(M, K), g(N, K)
        y16, y32, aux = g(M, N), f(M, N), g(M, N)
        bias = f(N)
        t16 = t_med(lambda: be.gemm(x, w, y16, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1))
        t32 = t_med(lambda: be.gemm(x, w, y32, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1))
        tg = t_med(lambda: be.gemm(x, w, y16, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), bias=bias, act=1, aux=aux, split_k=1))
        tb = t_med(lambda: be.gemm(x, w, y16, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), act=2, aux=aux, split_k=1))
        print(f"M={M} N={N} K={K:4d}: bf16 out {1e3*t16:7.1f} us | fp32 out {1e3*t32:7.1f} us | bias+GELU+aux {1e3*tg:7.1f} us | GELU' {1e3*tb:7.1f} us   (out bytes bf16 {M*N*2/1e6:.0f} MB)")
