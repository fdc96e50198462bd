#!/bin/bash
# round 3: XCD-aware image pairing in every attn16 kernel — parity, then timing (compare with gpurun_out/attn2_new.log of the run before)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attention16_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/attn_xcd_suite.log 2>&1
rc=$?
tail -3 gpurun_out/attn_xcd_suite.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 200 python scripts/ab_attn16.py 9 2>&1 | grep -v amdgpu.ids
