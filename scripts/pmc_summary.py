#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (sums over dispatches) and derive
MFMA utilisation and HBM traffic as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs
      mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)
  * FETCH_SIZE (KB) under-reports wide coalesced reads by exactly 2x on gfx950 -> doubled; WRITE_SIZE (KB) exact.
Also writes OUT.csv.stamp.json: sha256 of every kernel source (calm-vit-dte_amd/csrc/*) the counters were taken with —
bench.py refuses counter values whose kernel source has changed since.
usage: pmc_summary.py OUT.csv DIR [DIR ...]"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
dur = defaultdict(float)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = k.split("(")[0][:80]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
                seen.add(key)
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    calls[k] += 1
                    if "Start_Timestamp" in r and r.get("End_Timestamp"):
                        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = mfma / (gui / 8 * 256 * 4) if gui else 0.0
    rd, wr = 2 * c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    t = dur.get(k, 0.0) * 1e-9
    rows.append((gui, k, calls[k], util, rd, wr, t, c))
rows.sort(reverse=True)
with open(out, "w") as f:
    f.write("kernel,dispatches,gui_active_sum,mfma_busy_cycles,mfma_util,hbm_read_bytes(2xFETCH),hbm_write_bytes,duration_s,hbm_GBps\n")
    for gui, k, n, util, rd, wr, t, c in rows:
        bw = (rd + wr) / t / 1e9 if t else 0.0
        f.write(f"\"{k}\",{n},{gui:.0f},{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{util:.4f},{rd:.0f},{wr:.0f},{t:.6f},{bw:.1f}\n")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stamp = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()
         for f in sorted(glob.glob(os.path.join(root, "calm-vit-dte_amd", "csrc", "*")))}
with open(out + ".stamp.json", "w") as f:
    json.dump({"sources_sha256": stamp, "argv": sys.argv[1:]}, f, indent=1, sort_keys=True)

try:
    for gui, k, n, util, rd, wr, t, c in rows[:16]:
        print(f"{k[:58]:58s} n={n:5d} mfma_util={100*util:5.1f}%  rd={rd/1e9:8.2f}GB wr={wr/1e9:8.2f}GB t={1e3*t:8.2f}ms")
except BrokenPipeError:
    pass
