#!/bin/bash
# iteration loop on the GPU box: tests, then bench (+ per-shape GEMM report), optional rocprof
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q --timeout 900 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -n ${PYTEST_TAIL:-6} gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ -z "$KEEP_GOING" ]; then exit $rc; fi
timeout -k 10 900 python3 bench.py --steps ${STEPS:-4} --warmup 2 --gemm-report gpurun_out/gemm_report.csv ${BENCH_ARGS} > gpurun_out/bench_full.log 2>&1 || { tail -n 30 gpurun_out/bench_full.log; exit 1; }
tail -n 1 gpurun_out/bench_full.log
head -n 45 gpurun_out/gemm_report.csv
if [ -n "$ROCPROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 ${BENCH_ARGS} > $R/gpurun_out/rocprof.log 2>&1 || { tail -n 30 $R/gpurun_out/rocprof.log; exit 1; }
  find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/prof/r1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:64]:64s} n={r['Calls']:>6} ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print('total ms (3 steps)', tot/1e6)
PY
fi
