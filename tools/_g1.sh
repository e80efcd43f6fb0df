mkdir -p gpurun_out/g4
timeout -k 10 900 python -m pytest tests/test_gpu_numbering.py tests/test_gpu_parity.py tests/test_gpu_second_order.py tests/test_gpu_known_answers.py tests/test_gpu_fuzz.py -q -m gpu > gpurun_out/g4/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a gpurun_out/g4/rc.txt
tail -15 gpurun_out/g4/tests1.log
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -q -m gpu -x -k "not bench_self" > gpurun_out/g4/tests2.log 2>&1; echo "tests2 rc=$?" | tee -a gpurun_out/g4/rc.txt
tail -15 gpurun_out/g4/tests2.log
(tools/ab_libs_run.sh --steps 100 --warmup 10; tools/ab_libs_run.sh --steps 100 --warmup 10 --hr; tools/ab_libs_run.sh --steps 100 --warmup 10 --second-order; AB_ARGS_a_r4="--quad-block 16x16" tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads; AB_ARGS_a_r4="--quad-block 16x16" tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads --second-order) > gpurun_out/g4/ab.txt 2>&1
cat gpurun_out/g4/ab.txt
