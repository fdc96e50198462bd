#!/bin/bash
# SQ stall-reason counters for the CNN tail kernels (one --pmc pass over scripts/ab_cnn.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_cnn
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_cnn -o s -- python3 $R/scripts/ab_cnn.py > $R/gpurun_out/pmc_cnn.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc_cnn.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob('$R/gpurun_out/pmc_cnn/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
print(f"{'kernel':40s} {'n':>4s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'valu':>6s} {'lds_conf/idx':>12s} {'wait_lds':>8s}")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:6]:
    w = c['SQ_WAVE_CYCLES'] or 1
    print(f"{k:40s} {n[k]:4d} {c['SQ_WAIT_ANY']/w:8.2f} {c['SQ_WAIT_INST_ANY']/w:9.2f} {c['SQ_ACTIVE_INST_ANY']/w:7.2f} {c['SQ_ACTIVE_INST_VALU']/w:6.2f} "
          f"{(c['SQ_LDS_BANK_CONFLICT']/(c['SQ_LDS_IDX_ACTIVE'] or 1)):12.3f} {c['SQ_WAIT_INST_LDS']/w:8.3f}")
PY
