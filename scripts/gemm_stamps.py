#!/usr/bin/env python3
"""Timing experiment.  Needs a library with the in-kernel phase stamps, which live in the history, not in the product
source: build calm-vit-dte_amd/csrc/gemm.hip of commit d96fccc…a953380 with -DCALM_GEMM_STAMP (and -DCALM_GEMM_ABLATE=n
for the main-loop ablations) and point CALM_VIT_LIB at it.
 where does a workgroup's time go inside the fp32
GEMM kernel?  Per-phase s_memtime sums of wave 0 of every workgroup, plus the per-CU timeline (how many co-resident
workgroups are inside their k-loop at any moment).
   CALM_VIT_LIB=$PWD/ab/lib_stamp.so python3 scripts/gemm_stamps.py [M N K]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import calm_vit_dte_amd as calm
from importlib import import_module

be = calm.backend.get_backend()
lib = import_module("calm_vit_dte_amd._lib").load()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (57344, 672, 672)
kind = sys.argv[4] if len(sys.argv) > 4 else "fwd"
g = lambda *s: torch.randn(*s, device="cuda")
x, w, y = g(M, K), g(N, K), g(M, N)
if kind == "fwd":
    run = lambda: be.gemm(x, w, y, M, N, K, (K, 1, 0, 0), (K, 1, 0, 0), (N, 0, 0), split_k=1)
else:
    run = lambda: be.gemm(y, w, x, M, K, N, (N, 1, 0, 0), (1, K, 0, 0), (K, 0, 0), split_k=1)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
Nn = K if kind != "fwd" else N
n_wg = ((M + 127) // 128) * ((Nn + 95) // 96)
n_wg = min(n_wg, 8192)
buf = np.zeros((n_wg, 12), dtype=np.int64)
rc = lib.calm_debug_stamps(buf.ctypes.data_as(C.c_void_p), n_wg)
assert rc == 0, rc
t0 = buf[:, 0].min()
begin, loop, epi, end = (buf[:, i] - t0 for i in range(4))
span = end.max()
tick_us = 0.01                                   # s_memrealtime: 100 MHz
print(f"{kind} {M}x{N}x{K}: {ms:.3f} ms by events, {n_wg} workgroups, kernel span {span*tick_us:.1f} us by s_memrealtime")
mt_us = np.sum(epi - loop) * tick_us / buf[:, 11].sum()      # us per s_memtime tick
print(f"s_memtime runs at {1/mt_us:.0f} MHz")
life = (end - begin)
print(f"workgroup lifetime  mean {life.mean()*tick_us:7.2f} us   prologue {np.mean(loop-begin)*tick_us:6.2f} us   k-loop {np.mean(epi-loop)*tick_us:7.2f} us   epilogue {np.mean(end-epi)*tick_us:6.2f} us")
tot = buf[:, 4:9].sum(axis=1).astype(float)
names = ["issue global loads", "LDS reads + MFMA", "wait vmcnt(0)", "LDS stores", "barrier"]
for i, nm in enumerate(names):
    print(f"  k-loop phase {nm:20s} {100*buf[:, 4+i].sum()/tot.sum():5.1f}%   ({buf[:, 4+i].mean()*mt_us:7.2f} us per workgroup)")
hw, xcc = buf[:, 9], buf[:, 10] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
print("distinct CUs seen:", len(np.unique(cu)))
# per-CU timeline: fraction of the kernel span with n workgroups resident / n inside the k-loop
res_hist, loop_hist = np.zeros(10), np.zeros(10)
for c in np.unique(cu):
    sel = cu == c
    ev = []
    for b, l, e, d in zip(begin[sel], loop[sel], epi[sel], end[sel]):
        ev += [(b, 0, 1), (d, 0, -1), (l, 1, 1), (e, 1, -1)]
    ev.sort()
    cur = [0, 0]; last = 0
    for t, which, dlt in ev:
        res_hist[max(0, min(cur[0], 9))] += t - last; loop_hist[max(0, min(cur[1], 9))] += t - last
        cur[which] += dlt; last = t
    res_hist[0] += span - last; loop_hist[0] += span - last
print("fraction of CU-time with n workgroups resident :", np.round(res_hist / res_hist.sum(), 3)[:7])
print("fraction of CU-time with n workgroups in k-loop:", np.round(loop_hist / loop_hist.sum(), 3)[:7])
# lock-step: spread of epilogue start times among workgroups sharing a CU in the first round
first = np.argsort(begin)[:1024]
print(f"first-round begin spread {np.ptp(begin[first])*tick_us:.2f} us; their epilogue-start spread (std) {np.std(epi[first])*tick_us:.2f} us, lifetime std {np.std(life[first])*tick_us:.2f} us")
