#!/bin/bash
# one box: A/B of the non-temporal hint choices (first-order path; `euler_step` shows the RHS + axpy pair too)
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d.get('euler_step'))"; }
B="--steps 100 --warmup 10 --no-cpu-baseline"
for flags in "$@"; do
  RDYHIP_EXTRA_HIPCC_FLAGS="$flags" python3 -c "from rdycore_amd import build; build.build_native(force=True)" || exit 1
  python3 bench.py $B 2>/dev/null | tail -1 | show "[$flags] first "
done
python3 -c "from rdycore_amd import build; build.build_native(force=True)"
