"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE
on this kernel family's access width (8 B per lane, AoS triples): launches
rdyhip_axpy_owned (reads f 24 B/cell + u 24 B/cell, writes u 24 B/cell)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdycore_amd import mesh as M, cases as CS

nx, ny = 2500, 2000
mesh = M.structured_tri_mesh(nx, ny, 1.0, order="tiled")
case = CS.dam_break_case(mesh, nx)
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.zeros((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
for _ in range(10):
    op.axpy_owned(1e-3, f, u)
torch.cuda.synchronize()
print("axpy_owned: cells", mesh.num_owned_cells, "read bytes", 48 * mesh.num_owned_cells, "write bytes", 24 * mesh.num_owned_cells)
