"""Prints the device layout numbers of the bench workload (first and second order)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rdycore_amd import cases as CS
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2500, 2000)
for so in (False, True):
    case = bench.build_case(nx, ny, 0, 1, "tiled", "semi_implicit", "c3", False, so, "minmod")
    op = CS.create_operator(case)
    print("second_order" if so else "first_order", json.dumps(op.layout_info()), flush=True)
    op.destroy()
