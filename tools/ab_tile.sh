#!/bin/bash
# tile size A/B (Hilbert-ordered mesh, so that any tile size gets compact tiles)
for flags in "$@"; do
  RDYHIP_EXTRA_HIPCC_FLAGS="$flags" python3 -c "from rdycore_amd import build; build.build_native(force=True)" || exit 1
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --order hilbert --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['tile_edge_records_per_cell'])"
  done
done
python3 -c "from rdycore_amd import build; build.build_native(force=True)"
