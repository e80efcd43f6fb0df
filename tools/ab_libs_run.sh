#!/bin/bash
# On the GPU box: bench every library in ab_libs/ alternately (same box, same call).  usage: tools/ab_libs_run.sh [bench args]
# AB_ARGS_<libname without .so>: extra bench arguments for that library only (e.g. the cell numbering an older build wants)
BARGS=${@:---steps 100 --warmup 10}
for rep in 1 2; do
  for lib in ab_libs/*.so; do
    name=$(basename $lib .so); extra_var="AB_ARGS_$name"; extra=${!extra_var}
    RDYHIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-order-study $BARGS $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib rep$rep', '$extra', d['value'], d['ms_per_step'], d['roofline']['steady_state_period_median_ms'], d['euler_step']['fused_ms_per_step'], d['roofline'].get('persistent_workgroups'), d['roofline'].get('lds_bytes_per_workgroup'))"
  done
done
