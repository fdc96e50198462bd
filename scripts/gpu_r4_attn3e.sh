#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CALM_ATTN16_V3=1 timeout -k 10 300 python -m pytest tests/test_attention16_gpu.py -q -x -k "forward_matches" > gpurun_out/r4_attn3_test.log 2>&1; echo "rc=$?" >> gpurun_out/r4_attn3_test.log
tail -3 gpurun_out/r4_attn3_test.log
echo "== v2"; timeout -k 10 120 python scripts/ab_attn16.py 4 2>&1 | grep "^B"
for st in 0 30 60 100; do echo "== v3 stagger $st"; CALM_ATTN16_V3=1 CALM_ATTN16_STAGGER=$st timeout -k 10 120 python scripts/ab_attn16.py 4 2>&1 | grep "^B"; done
CALM_ATTN16_V3=1 CALM_ATTN16_STAGGER=60 CALM_VIT_LIB=$R/ab/libcalmvit_stamp3.so timeout -k 10 120 python scripts/attn3_stamps.py 2>&1 | tail -3
timeout -k 10 300 python scripts/rccl_capture_check.py > gpurun_out/r4_rccl_capture.log 2>&1; echo "rccl capture rc=$?"; tail -5 gpurun_out/r4_rccl_capture.log
