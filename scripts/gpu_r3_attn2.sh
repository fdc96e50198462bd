#!/bin/bash
# round 3: parity of the pipelined attention forward (attn16_fwd2_kernel), then same-box timing against the
# register-staged forward (CALM_ATTN16_V2=0)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_attention16_gpu.py -x -q -k forward > gpurun_out/attn2_test.log 2>&1
rc=$?
tail -15 gpurun_out/attn2_test.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
CALM_ATTN16_V2=0 timeout -k 10 200 python scripts/ab_attn16.py 4 > gpurun_out/attn2_old.log 2>&1 && \
timeout -k 10 200 python scripts/ab_attn16.py 4 > gpurun_out/attn2_new.log 2>&1
paste -d'\n' gpurun_out/attn2_old.log gpurun_out/attn2_new.log | grep -v amdgpu.ids
