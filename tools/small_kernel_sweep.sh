#!/bin/bash
# On the GPU box: the RHS kernel and the fused Euler step over a sweep of mesh sizes (0.36 M ... 10 M triangles), for every
# library in ab_libs/ and two tile sizes (RDYHIP_TILE_CELLS) -- the small-part regime of VERDICT r4 item 3: where do plain
# (cached) loads of the per-cell streams / edge records beat the non-temporal ones, where do 128-cell tiles beat 256.
# usage: tools/small_kernel_sweep.sh [extra bench args] > gpurun_out/small_kernel_sweep.txt
SIZES=${SIZES:-"600x300 700x500 1000x500 1000x700 1450x1000 2500x1000 2500x2000"}
TILES=${TILES:-"256 128"}
echo "# cells lib tile_cells  rhs_period_us  euler_fused_us  M_cell_updates_per_s  effective_GBps(176B)  persistent_wgs cells_per_tile"
for sz in $SIZES; do
  nx=${sz%x*}; ny=${sz#*x}
  for lib in ab_libs/*.so; do
    for tc in $TILES; do
      RDYHIP_TILE_CELLS=$tc RDYHIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-order-study --steps 200 --warmup 20 --condition-seconds 0.3 --nx $nx --ny $ny "$@" 2>/dev/null | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['cells_per_gpu']; p=r['steady_state_period_median_ms']; print(n, '$(basename $lib .so)', $tc, round(1e3*p,2), round(1e3*d['euler_step']['fused_ms_per_step'],2), round(n/p/1e3,1), round(n*176/p/1e6,1), r['persistent_workgroups'], r.get('cells_per_tile'))"
    done
  done
done
