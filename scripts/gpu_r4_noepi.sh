#!/bin/bash
# Timing experiment: how much of the 256-thread GEMM families' time is their epilogue?  Same step, library built with
# -DCALM_GEMM_NO_EPILOGUE (gemm_common.h: the epilogue returns at once — results are garbage, shapes and launches are not).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r4noepi; mkdir -p $O
for tag in base noepi; do
  for cfg in "small224" "base224 --autocast"; do
    set -- $cfg
    if [ $tag = noepi ]; then export CALM_VIT_LIB=$R/ab/libcalmvit_noepi.so; else unset CALM_VIT_LIB; fi
    timeout -k 10 300 python3 bench.py --workload $1 ${2:-} --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --gemm-report $O/${tag}_$1.csv > $O/${tag}_$1.json 2> $O/${tag}_$1.err || tail -3 $O/${tag}_$1.err
    python3 -c "
import json,sys
j=json.loads(open('$O/${tag}_$1.json').read().strip().splitlines()[-1]); print('$tag $1', j['ms_per_step'], 'gemm', j['roofline']['gemm_ms_per_step'])"
  done
done
python3 - <<'PY'
import csv
for wl in ("small224", "base224"):
    a = {tuple(r[k] for k in ("M","N","K","batch","a_kc","b_kc","reduce")): r for r in csv.DictReader(open(f"gpurun_out/r4noepi/base_{wl}.csv"))}
    b = {tuple(r[k] for k in ("M","N","K","batch","a_kc","b_kc","reduce")): r for r in csv.DictReader(open(f"gpurun_out/r4noepi/noepi_{wl}.csv"))}
    rows = sorted(((float(a[k]["ms_per_step"]) - float(b[k]["ms_per_step"]), k) for k in a if k in b), reverse=True)
    print(wl, "epilogue share, largest first:", round(sum(d for d, _ in rows), 2), "ms/step in total")
    for d, k in rows[:14]:
        print("   ", k, a[k]["ms_per_step"], "->", b[k]["ms_per_step"])
PY
