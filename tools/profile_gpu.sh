#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -e
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-order-study $@"
PARGS="$BARGS --condition-seconds 0.2"   # counter passes: fewer launches
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $BARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o pmc -- python3 $REPO/tools/calib_traffic.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o pmc -- python3 $REPO/tools/calib_traffic.py > $OUT/calib_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_sq2.log 2>&1 || echo "sq2 pass failed"
cd $REPO
# the machine code that was measured: hash of every kernel in the library this run loaded (profiles/traffic.json ties its figures to it)
python3 -c "import json; from rdycore_amd import build, codeobj; json.dump(codeobj.kernel_hashes(build.lib_path()), open('$OUT/kernel_code_hashes.json', 'w'), indent=0)"
find $OUT -name "*.csv" | head -30
python3 tools/parse_rocprof.py $OUT $TAG
