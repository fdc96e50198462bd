#!/bin/bash
# same-box A/B of the whole Base-224 autocast step: round-2 GEMM kernels (CALM_GEMM_PIPE=0) vs the pipelined family
mkdir -p gpurun_out
CALM_GEMM_PIPE=0 timeout -k 10 400 python bench.py --workload base224 --autocast --steps 8 --warmup 3 --no-cpu-baseline --gemm-report gpurun_out/gemm_old.csv > gpurun_out/step_old.json 2> gpurun_out/step_old.err || { tail -5 gpurun_out/step_old.err; exit 1; }
timeout -k 10 400 python bench.py --workload base224 --autocast --steps 8 --warmup 3 --no-cpu-baseline --gemm-report gpurun_out/gemm_new.csv > gpurun_out/step_new.json 2> gpurun_out/step_new.err || { tail -5 gpurun_out/step_new.err; exit 1; }
python - <<'PY'
import json
for n in ("old","new"):
    d=json.loads(open(f"gpurun_out/step_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["roofline"])
PY
