#!/bin/bash
# Memory-pipeline counters of the RHS kernel (texture addresser TA, vector L1 = TCP, L2 = TCC, fabric) from rocprofv3 PMC
# passes on the GPU box: where a kernel that saturates neither HBM bytes nor VALU waits.  usage: tools/mem_pipe.sh <tag> [bench args]
TAG=${1:-mem}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/mem_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-order-study --condition-seconds 0.1 $@"
cd /tmp
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_HIT_sum" \
           "TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "pass $i: $set"       # progress: a silent run is taken for a hung one
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py $BARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "swe_rhs" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    if n < 20:
        continue
    print(k, f"({n} launches)")
    for c, v in sorted(d.items()):
        print(f"   {c:42s} {sum(v) / len(v):16.0f}")
PY
