#!/bin/bash
timeout -k 5 300 scripts/micro/pipe_gemm_check > gpurun_out/pipe_check.log 2>&1 || { tail -50 gpurun_out/pipe_check.log; exit 1; }
tail -31 gpurun_out/pipe_check.log
export LD_LIBRARY_PATH=$PWD/ab STAMP=1
for shp in "57344 672 672 1 1" "45056 528 528 1 1"; do timeout -k 5 60 scripts/micro/pipe_gemm_check $shp 1 1 | grep -v "item 1\|item 2"; done
