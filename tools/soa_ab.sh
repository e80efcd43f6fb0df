for rep in 1 2; do for soa in 0 1; do
RDYHIP_MUSCL_SOA=$soa python3 bench.py --no-cpu-baseline --no-order-study --steps 100 --warmup 10 --second-order "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('soa=$soa rep$rep', d['value'], d['ms_per_step'], d['roofline']['steady_state_period_median_ms'], d['euler_step']['fused_ms_per_step'], d['roofline'].get('persistent_workgroups'), d['roofline'].get('lds_bytes_per_workgroup'))"
done; done
