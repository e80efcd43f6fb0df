#!/bin/bash
# Dynamic instruction mix of the RHS kernel from rocprofv3 PMC passes (GPU box).  usage: tools/inst_mix.sh <tag> [bench args]
TAG=${1:-mix}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/mix_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-order-study --condition-seconds 0.1 $@"
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py $BARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "swe_rhs" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"][-8:]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    tot = None
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        if c == "SQ_INSTS_VALU": tot = m
        print(f"   {c:28s} {m:14.0f}" + (f"  {m / tot:6.1%} of VALU" if tot and c.startswith("SQ_INSTS_VALU_") else ""))
PY
