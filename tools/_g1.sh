mkdir -p gpurun_out/g7
timeout -k 10 600 python -m pytest tests/test_gpu_numbering.py -q -m gpu -k "per_rank" > gpurun_out/g7/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a gpurun_out/g7/rc.txt
tail -5 gpurun_out/g7/tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -q -m gpu -k "flips_the_form or second_order" > gpurun_out/g7/tests2.log 2>&1; echo "tests2 rc=$?" | tee -a gpurun_out/g7/rc.txt
tail -12 gpurun_out/g7/tests2.log
timeout -k 10 600 python -m pytest tests/test_gpu_second_order.py -q -m gpu > gpurun_out/g7/tests3.log 2>&1; echo "tests3 rc=$?" | tee -a gpurun_out/g7/rc.txt
tail -5 gpurun_out/g7/tests3.log
mkdir -p ab_hold; mv ab_libs/a_r4.so ab_hold/
(tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads --second-order --moving-state; tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads --second-order;  tools/ab_libs_run.sh --steps 100 --warmup 10 --workload dambreak_quads --second-order --nx 2560 --ny 1280 --moving-state) > gpurun_out/g7/ab.txt 2>&1
mv ab_hold/a_r4.so ab_libs/
cat gpurun_out/g7/ab.txt
