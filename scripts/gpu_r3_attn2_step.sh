#!/bin/bash
# round 3: the pipelined attention forward inside the tests that reach it and inside the Base-224 autocast step
# (same-box A/B: CALM_ATTN16_V2=0 is the register-staged forward)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attention16_gpu.py tests/test_fullsize_gpu.py tests/test_precision_gpu.py -x -q > gpurun_out/attn2_suite.log 2>&1
rc=$?
tail -5 gpurun_out/attn2_suite.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for m in 0 1; do
CALM_ATTN16_V2=$m timeout -k 10 400 python bench.py --workload base224 --autocast --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/attn2_step_$m.json 2> gpurun_out/attn2_step_$m.err || { tail -5 gpurun_out/attn2_step_$m.err; exit 1; }
done
python - <<'PY'
import json
for n in "01":
    d=json.loads(open(f"gpurun_out/attn2_step_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["attention"]["ms_per_step"], d["attention"]["largest_shape"])
PY
