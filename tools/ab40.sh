#!/bin/bash
# A/B of hipcc flag sets at two sizes on one box: tools/ab40.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  RDYHIP_EXTRA_HIPCC_FLAGS="$flags" python3 -c "from rdycore_amd import build; build.build_native(force=True)" || exit 1
  for size in "--nx 2500 --ny 2000 --steps 100 --warmup 10" "--nx 5000 --ny 4000 --steps 50 --warmup 5"; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline $size 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', d['config']['cells_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['euler_step'])"
  done
done
python3 -c "from rdycore_amd import build; build.build_native(force=True)"
