#!/bin/bash
# A/B of compile-time kernel variants on the GPU box: tools/ab_variants.sh "<hipcc flags A>" "<hipcc flags B>" ...
# Each variant is built in-tree and benched (N=1, 10 M cells); prints value / ms / frac per variant.
BARGS=${BENCH_ARGS:---steps 100 --warmup 10 --no-cpu-baseline}
for flags in "$@"; do
  RDYHIP_EXTRA_HIPCC_FLAGS="$flags" python3 -c "from rdycore_amd import build; build.build_native(force=True)" || exit 1
  for rep in 1 2; do
    python3 bench.py $BARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[%s] rep$rep' % '''$flags''', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'])"
  done
done
python3 -c "from rdycore_amd import build; build.build_native(force=True)"
