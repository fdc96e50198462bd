#!/bin/bash
# full-size bench + rocprofv3 kernel-trace summary of the same workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 bench.py --steps ${STEPS:-4} --warmup 2 ${BENCH_ARGS} > gpurun_out/bench_full.log 2>&1 || { tail -n 30 gpurun_out/bench_full.log; exit 1; }
tail -n 1 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 ${BENCH_ARGS} > $R/gpurun_out/rocprof.log 2>&1 || { tail -n 30 $R/gpurun_out/rocprof.log; exit 1; }
tail -n 1 $R/gpurun_out/rocprof.log
ls -la $R/gpurun_out/prof | head; find $R/gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -n 40 "$f"
# the per-dispatch trace is large: keep only the stats
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
