#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd $R
CALM_VIT_LIB=$R/ab/libcalmvit_stamp3.so timeout -k 10 120 python scripts/attn3_stamps.py 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_a3
C="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_a3 -o s -- python3 $R/scripts/ab_attn16.py 4 > $R/gpurun_out/pmc_a3.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc_a3.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(list)
for f in glob.glob('$R/gpurun_out/pmc_a3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for f in glob.glob('$R/gpurun_out/pmc_a3/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':44s} {'n':>4s} {'us(med)':>8s} {'mfma_util':>9s} {'valu_busy':>9s} {'valu':>10s} {'salu':>10s} {'lds':>9s}  (instructions per dispatch)")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:18]:
    if 'attn16' not in k and 'mask_' not in k: continue
    d = max(n[k], 1); cyc = (c['GRBM_GUI_ACTIVE'] or 1) / 8 * 1024
    t = sorted(dur[k]); 
    print(f"{k:44s} {n[k]:4d} {t[len(t)//2]:8.1f} {c['SQ_VALU_MFMA_BUSY_CYCLES']/cyc:9.3f} {4*c['SQ_ACTIVE_INST_VALU']/cyc:9.3f} {c['SQ_INSTS_VALU']/d:10.0f} {c['SQ_INSTS_SALU']/d:10.0f} {c['SQ_INSTS_LDS']/d:9.0f}")
PY
