#!/bin/bash
# round 3: parity of the pipelined attention backward (attn16_bwd2_kernel), then same-box timing against the
# register-staged pair (CALM_ATTN16_BWD2=0)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attention16_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/attn_bwd2_suite.log 2>&1
rc=$?
tail -12 gpurun_out/attn_bwd2_suite.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
CALM_ATTN16_BWD2=0 timeout -k 10 200 python scripts/ab_attn16.py 4 > gpurun_out/attn_bwd2_old.log 2>&1 && \
timeout -k 10 200 python scripts/ab_attn16.py 4 > gpurun_out/attn_bwd2_new.log 2>&1
paste -d'\n' gpurun_out/attn_bwd2_old.log gpurun_out/attn_bwd2_new.log | grep -v amdgpu.ids
