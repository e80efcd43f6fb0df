#!/bin/bash
# On the GPU box: the multi-rank step of a strip rank of growing size in its two forms (RDYHIP_OVERLAP=1: exchange on the
# library's stream behind the interior tiles; 0: everything in order on one stream), exchange looped back through a
# one-rank RCCL communicator (tools/step_breakdown.py).  Sets the size from which the overlapped form pays.
for n in "300 600" "900 600" "1300 770" "1500 1000" "2000 1000" "2500 2000"; do
  for ov in 0 1; do
    RDYHIP_OVERLAP=$ov python3 tools/step_breakdown.py $n 2>>gpurun_out/r03/overlap_threshold.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('cells', d['cells'], 'tiles', d['info']['num_tiles'], 'overlap', d['halo_overlaps'], 'step_us', round(1e3*d['overlapped_step'][0],1), 'host_us', round(1e3*d['overlapped_step'][1],1), 'single_launch_us', round(1e3*d['single_launch_rhs'][0],1), 'exchange_us', round(1e3*d['exchange_only_pack_rccl_unpack'][0],1))"
  done
done
