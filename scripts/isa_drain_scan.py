#!/usr/bin/env python3
"""Flag kernels whose global loads are followed directly by a full drain (`s_waitcnt vmcnt(0)`): each such pair is one
exposed memory round trip per execution of that code (round 4: run-time type branches in the RoPE kernels, conversions inside
masked branches in the LayerNorm backward, one register quad reused for a row of P loads in the fp32 attention backward).
Runs on the build machine (no GPU):   python scripts/isa_drain_scan.py calm-vit-dte_amd/csrc/norm_act.hip [name-regex]"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else None
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
rows = []
for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M):
    body, name = m.group(0), m.group(1)
    lines = body.split("\n")
    is_load = lambda x: "global_load" in x or "buffer_load" in x
    loads = sum(is_load(x) for x in lines)
    drains = sum("s_waitcnt vmcnt(0)" in x for x in lines)
    serial = sum(1 for i, x in enumerate(lines) if is_load(x) and any("vmcnt(0)" in y for y in lines[i + 1:i + 4])
                 and not any(is_load(y) for y in lines[i + 1:i + 4]))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(anonymous namespace\)::|calm_gemm_detail::", "", dn)
    if filt and not re.search(filt, dn):
        continue
    rows.append((serial, loads, drains, dn))
for serial, loads, drains, dn in sorted(rows, reverse=True):
    if serial >= 2:
        print(f"{serial:3d} of {loads:3d} loads drained at once ({drains:3d} full drains)  {dn[:110]}")
