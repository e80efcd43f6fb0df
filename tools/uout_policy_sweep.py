"""The Euler-step kernels' u_out store policy over a sweep of mesh sizes, IN A TIME LOOP (the two state arrays alternate: what a
step stores is what the next one reads).  Without the non-temporal hint the new state is still in the Infinity Cache (256 MB)
when the next step gathers it -- if it fits beside what else lives there; with the hint it comes from HBM.  rdyhip_create picks
by the size of the state array (RDYHIP_UOUT_CACHED_MAX_MB); this sweep is where that number comes from.
usage (GPU box): python tools/uout_policy_sweep.py [--hr | --second-order] > gpurun_out/uout_policy_sweep.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from rdycore_amd import cases as CS

torch.cuda.set_device(0)
extra = [a for a in sys.argv[1:] if a.startswith("--")]
sizes = os.environ.get("SIZES", "600x300 1000x500 1000x700 1450x1000 1600x1250 2000x1250 2500x1250 2500x1600 2500x2000").split()
print("# cells state_MB  ping-pong Euler step us: plain stores of u_out | non-temporal stores | plain / nt")
for sz in sizes:
    nx, ny = sz.split("x")
    args = bench.parse(["--no-cpu-baseline", "--nx", nx, "--ny", ny] + extra)
    case = bench.build_case(args, 0, 1)
    res = {}
    for policy in ("1", "0"):
        os.environ["RDYHIP_UOUT_CACHED"] = policy
        op = CS.create_operator(case)
        os.environ.pop("RDYHIP_UOUT_CACHED")
        a = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
        b = a.clone()
        best = None
        for _ in range(3):
            for _ in range(100):
                op.euler_step(0.0, a, b)
                a, b = b, a
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                op.euler_step(0.0, a, b)
                a, b = b, a
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 200 * 1e3
            best = t if best is None else min(best, t)
        res[policy] = best
        op.destroy()
        del a, b
    n = case.mesh.num_owned_cells
    print(n, round(n * 24 / 1e6, 1), round(res["1"], 2), round(res["0"], 2), round(res["1"] / res["0"], 3), flush=True)
