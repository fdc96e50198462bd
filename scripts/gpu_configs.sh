#!/bin/bash
# the other BASELINE configs on one GPU (record only; the bench default stays configs[1])
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
run() { timeout -k 10 500 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --prof-steps 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print(d['config']['workload'][:70], '|', d['ms_per_step'], 'ms |', d['value'], 'img/s | gemm', d['roofline']['achieved'], 'TF | model', d['model_tflops'], 'TF')
    open('gpurun_out/configs.jsonl', 'a').write(json.dumps(d) + chr(10))
except Exception as e:
    print('FAILED', e)
"; }
rm -f gpurun_out/configs.jsonl
run --workload small224
run --workload small224 --precision bf16x3
run --workload small224 --autocast
run --workload base224
run --workload base224 --autocast
run --workload large224
run --workload large224 --autocast
run --workload base384
run --workload base384 --autocast
run --workload nano48
