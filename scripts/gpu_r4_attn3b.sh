#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd $R
python -m pytest tests/test_attention16_gpu.py tests/test_fullsize_gpu.py -q -x > gpurun_out/r4_attn3_test.log 2>&1; echo "rc=$?" >> gpurun_out/r4_attn3_test.log
tail -3 gpurun_out/r4_attn3_test.log
echo "== v2"; CALM_ATTN16_V3=0 python scripts/ab_attn16.py 4 2>&1 | grep "^B"
for st in 0 40; do echo "== v3 stagger $st"; CALM_ATTN16_STAGGER=$st python scripts/ab_attn16.py 4 2>&1 | grep "^B"; done
CALM_VIT_LIB=$R/ab/libcalmvit_stamp3.so python scripts/attn3_stamps.py 2>&1 | tail -7
