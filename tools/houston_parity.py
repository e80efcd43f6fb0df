"""Whole-RHS parity of the refined Houston workload at any level against the oracle (what
tests/test_gpu_golden_and_scale.py:test_houston_refined_full_size does at level 6), for sizes kept out of the test suite.
usage (GPU box): python tools/houston_parity.py 7 [--hr | --second-order]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from rdycore_amd import cases as CS
from helpers import oracle_from_case, rel_linf

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 7
hr = "--hr" in sys.argv
so = "--second-order" in sys.argv
t0 = time.time()
case = CS.houston_refined_case(os.path.join(ROOT, "tests", "golden", "houston"), levels, "hilbert", hr=hr)
if so:
    case.config.second_order = True   # minmod, the default limiter
mesh = case.mesh
print(f"mesh: {mesh.num_cells} cells, {mesh.num_edges} edges, built in {time.time() - t0:.0f} s", flush=True)
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
op.rhs_function(case.dt, u, f)
torch.cuda.synchronize()
fh = f.cpu().numpy()
print("device RHS done", flush=True)
orc = oracle_from_case(case)
t1 = time.time()
fo = orc.apply(case.dt, case.u_local)
print(f"oracle RHS: {time.time() - t1:.1f} s", flush=True)
op.update_diagnostics()
d = op.get_diagnostics()
cmax, ce, cc = orc.diagnostics()
info = op.layout_info()
out = {"levels": levels, "hr": hr, "second_order": so, "cells": mesh.num_cells, "dry_fraction": float((case.u_local[:, 0] == 0).mean()),
       "rhs_rel_linf": rel_linf(fh, fo), "pv_rel_linf": rel_linf(op.primitive_variables.cpu().numpy(), orc.primitive_variables),
       "courant_device": d.max_courant_num, "courant_oracle": cmax, "courant_ids_equal": (d.global_edge_id, d.global_cell_id) == (ce, cc),
       "edge_records_per_cell": info["num_edge_records"] / mesh.num_cells, "halo_cells_per_tile": info["num_halo_entries"] / info["num_tiles"],
       "device_bytes": info["device_bytes"], "lds_fixed_layout": info["lds_fixed_layout"]}
import json
print(json.dumps(out))
assert out["rhs_rel_linf"] <= 1e-10 and out["pv_rel_linf"] <= 1e-10 and abs(d.max_courant_num - cmax) <= (1e-10 if so else 1e-12) * max(1.0, cmax)
