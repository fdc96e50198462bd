#!/bin/bash
# round 3: parity of the fp32 pipelined GEMM family, then same-box timing against the 128 x 128 kernels (CALM_GEMM_PIPE32=0)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gemm_pipe32_gpu.py -x -q > gpurun_out/pipe32_test.log 2>&1
rc=$?
tail -8 gpurun_out/pipe32_test.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
CALM_GEMM_PIPE32=0 timeout -k 10 300 python scripts/ab_gemm32.py > gpurun_out/ab32_old.log 2>&1 && \
timeout -k 10 300 python scripts/ab_gemm32.py > gpurun_out/ab32_new.log 2>&1
paste -d'\n' gpurun_out/ab32_old.log gpurun_out/ab32_new.log | tail -44
