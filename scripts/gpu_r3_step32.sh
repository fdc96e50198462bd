#!/bin/bash
# same-box A/B of the Small-224 fp32 step: 128 x 128 fp32 kernels (CALM_GEMM_PIPE32=0) vs the pipelined fp32 family
mkdir -p gpurun_out
for m in 0 1 2; do  # (CALM_GEMM_PIPE32 env = initial value of calm_gemm_set_option(CALM_GEMM_OPT_PIPE32))
CALM_GEMM_PIPE32=$m timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --gemm-report gpurun_out/gemm32_$m.csv > gpurun_out/step32_$m.json 2> gpurun_out/step32_$m.err || { tail -5 gpurun_out/step32_$m.err; exit 1; }
done
python - <<'PY'
import json
for n in "012":
    d=json.loads(open(f"gpurun_out/step32_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["gemm_ms_per_step"])
PY
