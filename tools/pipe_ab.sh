#!/bin/bash
# On the GPU box: parity of every experimental second-order build in ab_libs/ first, then alternating benches.
set -o pipefail
for lib in ab_libs/*.so; do
  case $lib in *base.so) continue;; esac
  echo "== parity $lib"
  RDYHIP_LIB=$PWD/$lib timeout -k 10 500 python3 -m pytest tests/test_gpu_second_order.py tests/test_gpu_multirank.py -x -q -k "second or muscl" 2>&1 | tail -3 || { echo "PARITY FAILED $lib"; exit 1; }
done
for w in "" "--workload dambreak_quads" "--workload houston_refined"; do
  echo "== bench second order $w"
  tools/ab_libs_run.sh --second-order --steps 100 --warmup 10 $w
done
