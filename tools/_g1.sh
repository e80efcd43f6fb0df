mkdir -p gpurun_out/g18
(tools/ab_libs_run.sh --steps 100 --warmup 10 --workload houston_refined; tools/ab_libs_run.sh --steps 100 --warmup 10 --workload houston_refined --order natural; tools/ab_libs_run.sh --steps 100 --warmup 10) > gpurun_out/g18/ab.txt 2>&1
cat gpurun_out/g18/ab.txt
bash tools/bench_all.sh r05 houston_l7 houston_l7_so houston_l7_hr 2>&1 | tee gpurun_out/bench_all_r05_l7.log
