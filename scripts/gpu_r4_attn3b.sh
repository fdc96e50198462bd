#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout -k 10 300 python -m pytest tests/test_attention16_gpu.py tests/test_fullsize_gpu.py -q -x > gpurun_out/r4_attn3_test.log 2>&1; echo "rc=$?" >> gpurun_out/r4_attn3_test.log
tail -3 gpurun_out/r4_attn3_test.log
grep -q "rc=0" gpurun_out/r4_attn3_test.log || { grep -n "Error\|assert" gpurun_out/r4_attn3_test.log | head -20; }
echo "== v2"; CALM_ATTN16_V3=0 timeout -k 10 120 python scripts/ab_attn16.py 4 2>&1 | grep "^B"
echo "== v3"; timeout -k 10 120 python scripts/ab_attn16.py 4 2>&1 | grep "^B"
CALM_VIT_LIB=$R/ab/libcalmvit_stamp3.so timeout -k 10 120 python scripts/attn3_stamps.py 2>&1 | tail -4
