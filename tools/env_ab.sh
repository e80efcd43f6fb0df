#!/bin/bash
# On the GPU box: A/B of one environment knob of the library inside one call, alternating runs.
# usage: tools/env_ab.sh KNOB "v0 v1" [bench args]      e.g. tools/env_ab.sh RDYHIP_BLOCKS_PER_CU "2 3" --hr
KNOB=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
env $KNOB=$v python3 bench.py --no-cpu-baseline --no-order-study --steps 100 --warmup 10 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KNOB=$v rep$rep', '$*', d['value'], d['ms_per_step'], d['roofline']['steady_state_period_median_ms'], d['euler_step']['fused_ms_per_step'], d['roofline'].get('persistent_workgroups'), d['roofline'].get('lds_bytes_per_workgroup'))"
done; done
