#!/bin/bash
# Per-launch view of the fixed-order reduction launches (calm_reduce_partials_kernel) in the Base-224 autocast step:
# duration by producer kernel (the launch right before on the same stream) and grid.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4red; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --prof-steps 0 --workload base224 --autocast > $O/rocprof.log 2>&1 || { tail -n 5 $O/rocprof.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0.0])
for i, r in enumerate(rows):
    if "calm_reduce_partials" not in r["Kernel_Name"]:
        continue
    prev = rows[i - 1]["Kernel_Name"].split("(")[0][-60:] if i else "?"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0
    k = (prev, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    agg[k][0] += 1; agg[k][1] += d
    agg[k].append(gap) if len(agg[k]) < 3 else None
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
print("reduce launches", n, "total us", round(tot, 1), "avg", round(tot / max(n, 1), 2))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, v[0], round(v[1] / v[0], 2), "gap", v[2] if len(v) > 2 else None)
st = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:12]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -rf $O/kt
