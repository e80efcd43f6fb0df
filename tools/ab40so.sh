#!/bin/bash
# second-order path: hipcc flag sets at 10 M and 40 M cells on one box
for flags in "$@"; do
  RDYHIP_EXTRA_HIPCC_FLAGS="$flags" python3 -c "from rdycore_amd import build; build.build_native(force=True)" || exit 1
  for size in "--nx 2500 --ny 2000 --steps 100 --warmup 10" "--nx 5000 --ny 4000 --steps 40 --warmup 4"; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --second-order $size 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', d['config']['cells_per_gpu'], d['value'], d['ms_per_step'], d['euler_step'])"
  done
done
python3 -c "from rdycore_amd import build; build.build_native(force=True)"
