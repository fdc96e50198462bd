#!/bin/bash
# first GPU pass: kernel parity, model parity, smoke, small bench runs (each step only if the previous passed)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 40 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -n 30 gpurun_out/smoke.log; exit 1; }
tail -n 3 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --workload nano48 --steps 3 --warmup 1 > gpurun_out/bench_nano.log 2>&1 || { tail -n 30 gpurun_out/bench_nano.log; exit 1; }
tail -n 2 gpurun_out/bench_nano.log
timeout -k 10 600 python bench.py --workload small224 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_s32.log 2>&1 || { tail -n 30 gpurun_out/bench_s32.log; exit 1; }
tail -n 2 gpurun_out/bench_s32.log
