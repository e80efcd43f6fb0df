#!/bin/bash
# On the GPU box: what a rank of a strong-scaled run of the reference's own meshes costs per step.
# One rank's part of an 8-way RCB partition on ONE GPU, the real multi-rank step (rdyhip_rhs_overlapped) with its exchange
# looped back through a one-rank RCCL communicator (--self-exchange), beside the plain single-launch RHS of the same part.
# usage: tools/small_partitions.sh > gpurun_out/small_partitions.txt
B="--steps 200 --warmup 20 --no-cpu-baseline --no-order-study"
pick='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1], "cells", c["cells_per_gpu"], "ms/step", d["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_avg_ms"], "halo_B", c.get("halo_bytes_per_rank"), "tiles/grid", r.get("persistent_workgroups"), "euler_fused_ms", d["euler_step"]["fused_ms_per_step"])'
run() { tag=$1; shift; python3 bench.py $B "$@" 2>/dev/null | python3 -c "$pick" "$tag"; }
# the reference's 2.88 M-quad dam break (docs/user/example-cases/dam-break/index.md:24-26) and a 2.8 M-cell refined Houston mesh, 8 ranks
for rank in 0 3; do
  run "dambreak_2560x1280 rank$rank/8 single-launch" --workload dambreak_quads --nx 2560 --ny 1280 --emulate-world 8 --emulate-rank $rank
  run "dambreak_2560x1280 rank$rank/8 overlapped+self-exchange" --workload dambreak_quads --nx 2560 --ny 1280 --emulate-world 8 --emulate-rank $rank --self-exchange
  run "houston_L5 rank$rank/8 single-launch" --workload houston_refined --levels 5 --emulate-world 8 --emulate-rank $rank
  run "houston_L5 rank$rank/8 overlapped+self-exchange" --workload houston_refined --levels 5 --emulate-world 8 --emulate-rank $rank --self-exchange
done
# whole meshes on one GPU: 1 M (C2), 2.9 M
run "c2 1M" --workload c2
RDYHIP_BALANCE_ROUNDS=1 run "c2 1M balanced-rounds" --workload c2
for g in 768 704 656 576 512; do RDYHIP_PGRID=$g run "c2 1M pgrid=$g" --workload c2; done
run "houston_L5 2.8M" --workload houston_refined --levels 5
RDYHIP_BALANCE_ROUNDS=1 run "houston_L5 2.8M balanced-rounds" --workload houston_refined --levels 5
run "dambreak_2560x1280 2.88M" --workload dambreak_quads --nx 2560 --ny 1280
RDYHIP_BALANCE_ROUNDS=1 run "dambreak_2560x1280 2.88M balanced-rounds" --workload dambreak_quads --nx 2560 --ny 1280
