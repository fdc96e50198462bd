#!/bin/bash
# stall / latency counters of the attention forward kernels (separate --pmc passes, one micro-benchmark shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_a3_$i
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_a3_$i -o s -- python3 $R/scripts/ab_attn16.py 1 > $R/gpurun_out/pmc_a3_$i.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc_a3_$i.log; }
done
python3 - <<PY
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob('$R/gpurun_out/pmc_a3_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
        if 'attn16' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, c in agg.items():
    print(k)
    for name in sorted(c):
        print(f"    {name:34s} {c[name] / max(n[k][name], 1):16.0f}")
PY
