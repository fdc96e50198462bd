import os, sys
sys.path.insert(0, os.getcwd())
import torch
import calm_vit_dte_amd as calm
be = calm.backend.get_backend()
calm.backend.set_matmul_precision("bf16")
def t_med(fn, n=20, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]
g = lambda *s: torch.randn(*s, device="cuda").bfloat16()
f = lambda *s: torch.randn(*s, device="cuda")
# empty-ish kernel launch cost reference
x = f(1024)
print("torch add_ tiny:", 1e3 * t_med(lambda: x.add_(1.0)), "us")
for (Ms, Ns, nb) in ((176, 528, 256), (176, 528, 64), (176, 528, 16), (128, 384, 256), (80, 240, 256)):
    for Ks in (32, 64, 96, 176, 352):
        a_, b_, c_ = g(nb, Ms, Ks), g(nb, Ks, Ns), g(nb, Ms, Ns)
        c32 = f(nb, Ms, Ns)
        t1 = t_med(lambda: be.gemm(a_, b_, c_, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(nb, 1), split_k=1))
        t2 = t_med(lambda: be.gemm(a_, b_, c32, Ms, Ns, Ks, (Ks, 1, Ms * Ks, 0), (1, Ns, Ks * Ns, 0), (Ns, Ms * Ns, 0), batch=(nb, 1), split_k=1, accumulate=True))
        print(f"per-image {Ms}x{Ns}x{Ks} x{nb}: bf16 out {1e3*t1:6.1f} us | fp32 accumulate {1e3*t2:6.1f} us")
