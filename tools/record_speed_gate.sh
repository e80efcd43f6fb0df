#!/bin/bash
# On the GPU box, after a change that is MEANT to move a kernel: measure the ratios tests/test_gpu_speed_gate.py holds the
# kernels to and write them to gpurun_out/speed_gate.json (copy to tests/golden/speed_gate.json to make them the reference).
RDYHIP_RECORD_SPEED_GATE=$PWD/gpurun_out/speed_gate.json python -m pytest tests/test_gpu_speed_gate.py -q -m gpu && cat gpurun_out/speed_gate.json
