#!/bin/bash
for st in 0 1 2; do
  echo "== stagger $st"
  for shp in "57344 672 672 1 1" "45056 528 528 1 1" "57344 1344 672 1 1"; do
    CALM_PIPE_STAGGER=$st timeout -k 5 60 scripts/micro/pipe_gemm_check $shp 1 1 || exit 1
  done
done
