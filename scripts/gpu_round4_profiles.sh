#!/bin/bash
# Round-4 evidence in one GPU call: for the bench default (BASELINE configs[1], Small-224 fp32) and for the reference's
# own configuration under its trainer's call pattern (Base-224, autocast bf16): bench JSON + per-shape GEMM table,
# rocprofv3 kernel-trace stats of the same command, PMC passes (separate --pmc runs; FETCH_SIZE doubled per the gfx950
# correction) summarised with a stamp of the kernel sources.  Results land in gpurun_out/r4/ (copy to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() {   # tag, bench args
  tag=$1; shift
  timeout -k 10 400 python3 $R/bench.py --steps 6 --warmup 2 --gemm-report $O/${tag}_gemm_shapes.csv "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { tail -n 5 $O/${tag}_bench.err; return 1; }
  tail -c 600 $O/${tag}_bench.json; echo
  CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --prof-steps 0 $*"
  rm -rf $O/tmp_$tag; mkdir -p $O/tmp_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$tag/kt -o kt -- $CMD > $O/${tag}_rocprof.log 2>&1 || { tail -n 5 $O/${tag}_rocprof.log; return 1; }
  cp $(find $O/tmp_$tag/kt -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
  echo "$tag kernel stats done"
  CMD1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --prof-steps 0 $*"
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/tmp_$tag/p1 -o p1 -- $CMD1 > $O/${tag}_pmc1.log 2>&1 || { tail -n 5 $O/${tag}_pmc1.log; return 1; }
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tmp_$tag/p2 -o p2 -- $CMD1 > $O/${tag}_pmc2.log 2>&1 || { tail -n 5 $O/${tag}_pmc2.log; return 1; }
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tmp_$tag/p3 -o p3 -- $CMD1 > $O/${tag}_pmc3.log 2>&1 || { tail -n 5 $O/${tag}_pmc3.log; return 1; }
  python3 $R/scripts/pmc_summary.py $O/${tag}_pmc_summary.csv $O/tmp_$tag/p1 $O/tmp_$tag/p2 $O/tmp_$tag/p3 | head -12
  rm -rf $O/tmp_$tag
}
one round4_small224_fp32 && one round4_base224_bf16 --workload base224 --autocast --no-secondary
ls -la $O | head -30
