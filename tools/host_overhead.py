"""Host-side cost (Python + ctypes + HIP launches) of the calls one multi-rank step makes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdycore_amd import mesh as M, cases as CS
from rdycore_amd.operator import pack_cells, unpack_cells
m = M.strip_partition_tri_mesh(64, 64, 1, 3)
case = CS.dam_break_case(m, 192.0)
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((m.num_owned_cells, 3), dtype=torch.float64, device="cuda")
ids = torch.arange(128, dtype=torch.int32, device="cuda"); buf = torch.empty((128, 3), dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()
def t(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return dt
print("rhs_function        %.1f us" % t(lambda: op.rhs_function(1e-3, u, f)))
print("apply_phase INTERIOR %.1f us" % t(lambda: op.apply_phase(1, True, 1e-3, u, f, reset_diagnostics=True)))
print("apply_phase HALO     %.1f us" % t(lambda: op.apply_phase(2, True, 1e-3, u, f)))
print("pack_cells           %.1f us" % t(lambda: pack_cells(u, ids, buf)))
print("unpack_cells         %.1f us" % t(lambda: unpack_cells(u, ids, buf)))
def streams():
    main = torch.cuda.current_stream(); side.wait_stream(main)
    with torch.cuda.stream(side): pass
    main.wait_stream(side)
print("stream fork/join     %.1f us" % t(streams))
