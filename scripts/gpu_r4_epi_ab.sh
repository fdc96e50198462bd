#!/bin/bash
# A/B of two builds on the per-shape GEMM tables of both bench lines (same box): base = ab/libcalmvit_base.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r4epi; mkdir -p $O
for tag in base new; do
  for cfg in "small224" "base224 --autocast"; do
    set -- $cfg
    if [ $tag = base ]; then export CALM_VIT_LIB=$R/ab/libcalmvit_base.so; else unset CALM_VIT_LIB; fi
    timeout -k 10 300 python3 bench.py --workload $1 ${2:-} --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --gemm-report $O/${tag}_$1.csv > $O/${tag}_$1.json 2> $O/${tag}_$1.err || tail -3 $O/${tag}_$1.err
    python3 -c "
import json
j=json.loads(open('$O/${tag}_$1.json').read().strip().splitlines()[-1]); print('$tag $1', j['ms_per_step'], 'gemm', j['roofline']['gemm_ms_per_step'], 'loss', j['config']['loss'])"
  done
done
python3 - <<'PY'
import csv
for wl in ("small224", "base224"):
    key = ("M","N","K","batch","a_kc","b_kc","reduce")
    a = {tuple(r[k] for k in key): r for r in csv.DictReader(open(f"gpurun_out/r4epi/base_{wl}.csv"))}
    b = {tuple(r[k] for k in key): r for r in csv.DictReader(open(f"gpurun_out/r4epi/new_{wl}.csv"))}
    rows = sorted(((float(a[k]["ms_per_step"]) - float(b[k]["ms_per_step"]), k) for k in a if k in b), reverse=True)
    print(wl, "base - new:", round(sum(d for d, _ in rows), 2), "ms/step")
    for d, k in rows[:8] + rows[-4:]:
        print("   ", k, a[k]["ms_per_step"], "->", b[k]["ms_per_step"])
PY
