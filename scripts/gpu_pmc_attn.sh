#!/bin/bash
# LDS bank-conflict / stall counters of the fused attention kernels (forward and backward) on the micro-benchmarks
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_attn*
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_attn1 -o s -- python3 $R/scripts/microbench.py > $R/gpurun_out/pmc_attn1.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc_attn1.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_attn2 -o s -- python3 $R/scripts/ab_attn_bwd.py > $R/gpurun_out/pmc_attn2.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc_attn2.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob('$R/gpurun_out/pmc_attn*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
        if 'attn' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
print(f"{'kernel':34s} {'n':>4s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'lds_conf/idx':>12s} {'wait_lds':>8s} {'mfma/wave_cyc':>13s}")
for k, c in sorted(agg.items()):
    w = c['SQ_WAVE_CYCLES'] or 1
    print(f"{k:34s} {n[k]:4d} {c['SQ_WAIT_ANY']/w:8.2f} {c['SQ_WAIT_INST_ANY']/w:9.2f} {c['SQ_ACTIVE_INST_ANY']/w:7.2f} "
          f"{(c['SQ_LDS_BANK_CONFLICT']/(c['SQ_LDS_IDX_ACTIVE'] or 1)):12.3f} {c['SQ_WAIT_INST_LDS']/w:8.3f} {c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*w):13.3f}")
PY
