"""What the first launches after the host-only setup cost (VERDICT r1 "what's weak": the driver's --warmup 5 --steps 20
run sits inside a ramp).  Times launches 1..N of the RHS kernel one by one with HIP events right after setup -- no
conditioning, no warm-up -- then again after a 1 s busy phase, and after an idle gap.
usage (GPU box): python tools/launch_series.py [n] > gpurun_out/launch_series.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
args = bench.parse(["--no-cpu-baseline"])
torch.cuda.set_device(0)
from rdycore_amd import cases as CS

t0 = time.time()
case = bench.build_case(args, 0, 1)
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
setup = time.time() - t0


def series(k):
    s = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    for i in range(k):
        s[i].record()
        op.rhs_function(case.dt, u, f)
        e[i].record()
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 4) for a, b in zip(s, e)], [round(s[0].elapsed_time(x), 3) for x in e]


out = {"setup_seconds": round(setup, 1), "cells": case.mesh.num_owned_cells}
d, t = series(n)
out["cold_ms"] = d
out["cold_end_time_ms"] = t
t1 = time.perf_counter()
while time.perf_counter() - t1 < 1.0:
    for _ in range(10):
        op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
d, _ = series(100)
out["after_1s_busy_ms"] = d
for gap in (0.05, 0.5, 3.0):
    time.sleep(gap)
    d, _ = series(60)
    out[f"after_{gap}s_idle_ms"] = d
summ = {k: {"first5": v[:5], "median_first20": float(np.median(v[:20])), "median_last20": float(np.median(v[-20:]))}
        for k, v in out.items() if k.endswith("_ms") and isinstance(v, list) and "time" not in k}
out["summary"] = summ
print(json.dumps(out))
