#!/bin/bash
# round 3: parity of the pipelined bf16 GEMM family, then same-box timing against the round-2 kernels (CALM_GEMM_PIPE=0)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gemm_pipe_gpu.py -x -q > gpurun_out/pipe_test.log 2>&1
rc=$?
tail -5 gpurun_out/pipe_test.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc (crash): stopping"; exit $rc; fi
CALM_GEMM_PIPE=0 timeout -k 10 300 python scripts/ab_gemm16.py > gpurun_out/ab_old.log 2>&1 && \
timeout -k 10 300 python scripts/ab_gemm16.py > gpurun_out/ab_new.log 2>&1
paste -d'\n' gpurun_out/ab_old.log gpurun_out/ab_new.log | tail -40
exit $rc
