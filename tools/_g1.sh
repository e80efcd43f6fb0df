mkdir -p gpurun_out/g16
timeout -k 10 900 python -m pytest tests/test_gpu_numbering.py tests/test_gpu_second_order.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/g16/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a gpurun_out/g16/rc.txt
tail -4 gpurun_out/g16/tests1.log
(AB_ARGS_a_r4="--quad-block 16x16" tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads; AB_ARGS_a_r4="--quad-block 16x16" tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads --second-order; tools/ab_libs_run.sh --steps 100 --warmup 10; tools/ab_libs_run.sh --steps 100 --warmup 10 --second-order; tools/ab_libs_run.sh --steps 100 --warmup 10 --hr) > gpurun_out/g16/ab.txt 2>&1
cat gpurun_out/g16/ab.txt
