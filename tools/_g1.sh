mkdir -p gpurun_out/g17
run() { lib=$1; shift; RDYHIP_LIB=$PWD/ab_libs/$lib.so python3 bench.py --no-cpu-baseline --no-order-study "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$*', d['value'], d['ms_per_step'], d['roofline']['steady_state_period_median_ms'], d['euler_step']['fused_ms_per_step'], d['roofline'].get('cells_per_tile'), d['config']['cells_per_gpu'])"; }
(for lib in a_r4 b_r5 a_r4 b_r5; do run $lib --steps 50 --warmup 10 --nx 5000 --ny 4000; done
for lib in a_r4 b_r5; do run $lib --steps 30 --warmup 10 --workload houston_refined --levels 7; done) > gpurun_out/g17/ab.txt 2>&1
cat gpurun_out/g17/ab.txt
