#!/bin/bash
# On the GPU box: the unprofiled bench line of every reported variant -> gpurun_out/<round>_bench_<tag>.json (copied to profiles/).
# usage: tools/bench_all.sh <round tag> [variant ...]
ROUND=${1:-r05}; shift || true
declare -A V
V[c3]="--steps 20 --warmup 5"                 # the driver's flags
V[c3_200]=""
V[hr]="--hr --no-cpu-all-cores"
V[xq]="--source implicit_xq2018 --no-cpu-all-cores"
V[c2]="--workload c2 --no-cpu-all-cores"
V[quads]="--workload dambreak_quads --no-cpu-all-cores"
V[c5]="--workload c5 --emulate-world 8 --emulate-rank 3 --no-cpu-all-cores"
V[so]="--second-order --no-cpu-all-cores"
V[so_quads]="--second-order --workload dambreak_quads --no-cpu-all-cores"
V[houston]="--workload houston_refined"
V[houston_hr]="--workload houston_refined --hr --no-cpu-all-cores"
V[houston_natural]="--workload houston_refined --order natural --no-cpu-baseline"
V[houston_so]="--workload houston_refined --second-order --no-cpu-all-cores"
V[houston_l7]="--workload houston_refined --levels 7 --no-cpu-all-cores"
V[houston_l7_so]="--workload houston_refined --levels 7 --second-order --no-cpu-all-cores"
V[houston_l7_hr]="--workload houston_refined --levels 7 --hr --no-cpu-all-cores"
V[delaunay]="--workload delaunay --no-cpu-all-cores"
V[self_exchange]="--emulate-world 3 --emulate-rank 1 --self-exchange --no-cpu-baseline"
V[self_exchange_so]="--emulate-world 3 --emulate-rank 1 --self-exchange --second-order --no-cpu-baseline"
ORDER="${@:-c3 c3_200 hr xq c2 quads c5 so so_quads houston houston_hr houston_natural houston_so delaunay self_exchange self_exchange_so}"
mkdir -p gpurun_out
for tag in $ORDER; do
  echo "bench $tag: ${V[$tag]}"
  timeout -k 10 420 python3 bench.py ${V[$tag]} > gpurun_out/${ROUND}_bench_$tag.json 2> gpurun_out/${ROUND}_bench_$tag.err || echo "  FAILED"
  python3 -c "import json,sys; d=json.load(open('gpurun_out/${ROUND}_bench_$tag.json')); r=d['roofline']; print('  ', d['value'], 'M cell-updates/s', d['ms_per_step'], 'ms  frac', r['frac'], 'traffic', r['traffic'], r['traffic_source']['status'] if r.get('traffic_source') else None)" 2>/dev/null || tail -3 gpurun_out/${ROUND}_bench_$tag.err
done
