#!/bin/bash
# On the GPU box: A/B of the order of a tile's edge records (RDYHIP_EDGE_SORT: 0 = the edge numbering, 1 = by the left cell's LDS slot,
# 2 = by the smaller of the two slots) for the variants whose edge phase gathers most from LDS.  usage: tools/edge_sort_ab.sh
for args in "--second-order" "--second-order --workload dambreak_quads" "--hr" "" "--workload dambreak_quads"; do
  tools/env_ab.sh RDYHIP_EDGE_SORT "0 1 2" $args
done
