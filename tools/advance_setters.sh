#!/bin/bash
# On the GPU box: the driver-shaped loop (tools/advance_pattern.py) with the rain refreshed through the SETTER path each interval,
# fixed dt (nothing in an interval synchronises): stream-ordered setter vs the synchronising legacy setter vs the on-device fill,
# domain-wide (one value per owned cell from a host array) and for a region of a tenth of the cells.  VERDICT r3 item 9.
# usage: tools/advance_setters.sh [workload args] > gpurun_out/r04/advance_setters.txt
for frac in 1.0 0.1; do
  for refresh in device setter setter_sync; do
    python3 tools/advance_pattern.py --fixed-dt --refresh $refresh --region-fraction $frac --steps-per-interval 20,100 --gaps-ms 0 --intervals 24 "$@" 2>/dev/null
  done
done
