#!/bin/bash
# On the GPU box: rocprofv3 kernel stats + PMC traffic + SQ counters for EVERY reported variant.
# One calibration of FETCH_SIZE / WRITE_SIZE per session (tools/calib_traffic.py), then per variant: a traced run
# (--kernel-trace --stats), a FETCH_SIZE pass, a WRITE_SIZE pass and two SQ passes -- each pass on its own, never a --pmc
# pass combined with a trace domain beyond --kernel-trace.  usage: tools/profile_all.sh <round tag> [variant ...]
ROUND=${1:-r05}; shift || true
REPO=$(pwd); export TMPDIR=/tmp
CAL=$REPO/gpurun_out/prof_${ROUND}_cal; mkdir -p $CAL
declare -A V
V[c3]=""
V[hr]="--hr"
V[xq]="--source implicit_xq2018"
V[c2]="--workload c2"
V[quads]="--workload dambreak_quads"
V[c5]="--workload c5 --emulate-world 8 --emulate-rank 3"
V[so]="--second-order"
V[so_quads]="--second-order --workload dambreak_quads"
V[houston]="--workload houston_refined"
V[houston_hr]="--workload houston_refined --hr"
V[houston_so]="--workload houston_refined --second-order"
V[delaunay]="--workload delaunay"
V[self_exchange]="--emulate-world 3 --emulate-rank 1 --self-exchange"
V[self_exchange_so]="--emulate-world 3 --emulate-rank 1 --self-exchange --second-order"
V[houston_natural]="--workload houston_refined --order natural"
V[houston_l7]="--workload houston_refined --levels 7"
V[houston_l7_so]="--workload houston_refined --levels 7 --second-order"
V[houston_l7_hr]="--workload houston_refined --levels 7 --hr"
ORDER="${@:-c3 hr xq c2 quads c5 so so_quads houston houston_hr houston_so delaunay self_exchange}"
cd /tmp
echo "calibration"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $CAL/cal_fetch -o pmc -- python3 $REPO/tools/calib_traffic.py > $CAL/calib_fetch.log 2>&1 || echo "calibration (fetch) failed"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $CAL/cal_write -o pmc -- python3 $REPO/tools/calib_traffic.py > $CAL/calib_write.log 2>&1 || echo "calibration (write) failed"
for tag in $ORDER; do
  OUT=$REPO/gpurun_out/prof_${ROUND}_$tag; mkdir -p $OUT
  BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-order-study ${V[$tag]}"
  PARGS="$BARGS --condition-seconds 0.2"
  echo "variant $tag: $BARGS"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $BARGS > $OUT/bench_trace.log 2>&1 || echo "  trace failed"
  echo "  fetch"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_fetch.log 2>&1 || echo "  fetch failed"
  echo "  write"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_write.log 2>&1 || echo "  write failed"
  echo "  sq"; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_sq.log 2>&1 || echo "  sq failed"
  echo "  sq2"; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $REPO/bench.py $PARGS > $OUT/bench_pmc_sq2.log 2>&1 || echo "  sq2 failed"
  # the calibration passes are shared: parse_rocprof.py looks for cal_fetch / cal_write under the variant's directory
  ln -sfn $CAL/cal_fetch $OUT/cal_fetch; ln -sfn $CAL/cal_write $OUT/cal_write
  # the machine code that was measured: hash of every kernel of the library these runs loaded (tools/make_traffic.py stamps the entry with it)
  (cd $REPO && python3 -c "import json; from rdycore_amd import build, codeobj; json.dump(codeobj.kernel_hashes(build.lib_path()), open('$OUT/kernel_code_hashes.json', 'w'), indent=0)")
  (cd $REPO && python3 tools/parse_rocprof.py $OUT ${ROUND}_$tag > $OUT/summary.txt 2>&1; tail -4 $OUT/summary.txt)
  # gpurun copies back at most 64 MiB per call: once parsed, the raw per-dispatch counter rows (~8 MB per variant) stay on the box
  if [ -s $OUT/summary.json ] && [ -z "$KEEP_RAW" ]; then rm -f $OUT/pmc_*/pmc_counter_collection.csv; fi
done
