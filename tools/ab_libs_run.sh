#!/bin/bash
# On the GPU box: bench every library in ab_libs/ alternately (same process tree, same device).
BARGS=${BENCH_ARGS:---steps 100 --warmup 10 --no-cpu-baseline}
for rep in 1 2 3; do
  for lib in ab_libs/*.so; do
    RDYHIP_LIB=$PWD/$lib python3 bench.py $BARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib rep$rep', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'])"
  done
done
