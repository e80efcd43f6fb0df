"""The product's step pattern, not back-to-back launches: a loop shaped like the driver's time loop
(driver/main.c) around RDyAdvance (src/rdyadvance.c:261-383):

    per coupling interval:  RDyApplyForcing  -> refill of the rain source on the device (one small launch)
                            RDyAdvance       -> N explicit steps (fused Euler steps, or RHS + axpy as TSEULER does)
                            diagnostics      -> the 16-byte Courant struct read back (a synchronisation)
                            host work        -> output / logging / the coupler: the GPU idles for `gap` ms

After ANY idle moment the device runs its next launches 5..40 at 20-35 % lower speed (profiles/r02_launch_series.json), so
a loop with host gaps never reaches the back-to-back rate bench.py reports.  This tool measures the effective
cell-updates/s of that pattern for several (N, gap).  (Round 3 also tried a keep-warm option inside the library -- one
sleeping wave, or 16 / 64 workgroups streaming memory, on a side stream during the host's gap: no effect on the dip
(profiles/r03_advance_pattern_keep_warm_modes.json), so it was removed again; the `keep_warm` rows only appear when a
library that still exports rdyhip_keep_warm is loaded.)

usage (GPU box): python tools/advance_pattern.py [--workload c3] > gpurun_out/advance_pattern.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--intervals", type=int, default=12)
ap.add_argument("--steps-per-interval", default="20,100,400")
ap.add_argument("--gaps-ms", default="0,2,10,50")
ap.add_argument("--keep-warm-modes", default="1", help="rdyhip_keep_warm arguments to try beside 0 (1: one sleeping wave; n > 1: n streaming workgroups)")
ap.add_argument("--pair", action="store_true", help="RHS + axpy per step (what TSEULER does) instead of the fused Euler step")
ap.add_argument("--refresh", default="device", choices=["device", "setter", "setter_sync"],
                help="how the rain source is refreshed at the start of every interval: device = rdyhip_forcing_fill_source (one small launch); "
                     "setter = RDySetDomainWaterSource's path, a host array of one value per owned cell through the stream-ordered "
                     "rdyhip_set_external_source_on; setter_sync = the same through the synchronising legacy setter")
ap.add_argument("--fixed-dt", action="store_true",
                help="time.adaptive.enable off (the reference's default, src/rdyadvance.c:303-305): no Courant read-back, i.e. nothing in an "
                     "interval synchronises; the host only waits at the very end")
ap.add_argument("--region-fraction", type=float, default=1.0, help="setter: the refreshed region as a fraction of the owned cells")
a = ap.parse_args()

args = bench.parse(["--no-cpu-baseline", "--workload", a.workload, "--levels", str(a.levels)])
torch.cuda.set_device(0)
from rdycore_amd import cases as CS
from rdycore_amd import _lib

case = bench.build_case(args, 0, 1)
op = CS.create_operator(case)
lib = _lib.load()
n_owned = case.mesh.num_owned_cells
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
u2 = torch.empty_like(u)
f = torch.empty((n_owned, 3), dtype=torch.float64, device="cuda")
st = int(torch.cuda.current_stream().cuda_stream)


n_region = max(1, int(round(a.region_fraction * n_owned)))
rain = np.full(n_region, 1e-5)
region_ids = None if n_region == n_owned else np.arange(n_region, dtype=np.int32)


def advance(nsteps, refresh=True):
    """one RDyAdvance: forcing refill, nsteps explicit steps with dt = 0 (the state stays put: every interval does the
    same work), diagnostics read-back"""
    if not refresh:
        pass
    elif a.refresh == "device":
        _lib.check(lib.rdyhip_forcing_fill_source(op._h, 0, n_owned, None, 1e-5, st))
    elif region_ids is None:
        op.set_domain_external_source(0, rain, ordered=(a.refresh == "setter"))
    else:
        op.set_regional_external_source(region_ids, 0, rain, ordered=(a.refresh == "setter"))
    cur, nxt = u, u2
    for _ in range(nsteps):
        if a.pair:
            op.rhs_function(0.0, u, f)
            op.axpy_owned(0.0, f, u)
        else:
            op.euler_step(0.0, cur, nxt)
            cur, nxt = nxt, cur
    if a.fixed_dt:
        return 0.0
    op.update_diagnostics()          # hipStreamSynchronize + 16 bytes D2H
    return op.get_diagnostics().max_courant_num


def busy_wait(ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        pass


def pattern(nsteps, gap_ms, keep_warm):
    advance(nsteps)                                   # first interval untimed (allocation, first-launch costs)
    busy_wait(gap_ms)
    t_dev = 0.0
    t0 = time.perf_counter()
    for _ in range(a.intervals):
        if keep_warm:
            _lib.check(lib.rdyhip_keep_warm(op._h, 0))    # the interval's own launches take over
        t1 = time.perf_counter()
        advance(nsteps)
        t_dev += time.perf_counter() - t1
        if keep_warm:
            _lib.check(lib.rdyhip_keep_warm(op._h, int(keep_warm)))
        busy_wait(gap_ms)                             # the host's own work between two RDyAdvance calls
    torch.cuda.synchronize()                          # fixed dt: the only wait of the whole pattern
    t_dev = t_dev if not a.fixed_dt else time.perf_counter() - t0
    wall = time.perf_counter() - t0
    if keep_warm:
        _lib.check(lib.rdyhip_keep_warm(op._h, 0))
    steps = a.intervals * nsteps
    return {"steps_per_interval": nsteps, "gap_ms": gap_ms, "keep_warm": int(keep_warm),
            "ms_per_step_in_advance": round(t_dev / steps * 1e3, 5),
            "M_cell_updates_per_s_in_advance": round(n_owned * steps / t_dev / 1e6, 1),
            "M_cell_updates_per_s_wall": round(n_owned * steps / wall / 1e6, 1)}


# the back-to-back reference: one long advance after conditioning
for _ in range(300):
    op.euler_step(0.0, u, u2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
advance(400, refresh=False)          # the steps alone: what every pattern below is compared with
e1.record()
torch.cuda.synchronize()
back_to_back = e0.elapsed_time(e1) / 400
out = {"workload": a.workload, "cells": n_owned, "step": "RHS + axpy (TSEULER)" if a.pair else "fused Euler step (rdyhip_euler_step)",
       "refresh": a.refresh, "refreshed_cells": n_region, "fixed_dt": bool(a.fixed_dt),
       "back_to_back_ms_per_step": round(back_to_back, 5), "back_to_back_M_cell_updates_per_s": round(n_owned / back_to_back / 1e3, 1),
       "intervals": a.intervals, "rows": []}
has_keep_warm = hasattr(lib, "rdyhip_keep_warm")
for nsteps in map(int, a.steps_per_interval.split(",")):
    for gap in map(float, a.gaps_ms.split(",")):
        for kw in ((0,) + tuple(int(x) for x in a.keep_warm_modes.split(",")) if (has_keep_warm and gap > 0) else (0,)):
            r = pattern(nsteps, gap, kw)
            r["vs_back_to_back"] = round(r["ms_per_step_in_advance"] / back_to_back, 4)
            out["rows"].append(r)
            print(json.dumps(r), file=sys.stderr)
print(json.dumps(out))
