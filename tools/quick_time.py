"""Exploratory timing of the RHS kernel on one GPU (not the bench contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdycore_amd import mesh as M, cases as CS

def run(nx, ny, order, src, iters=50, tile=16):
    K = 2*np.pi/200
    t0 = time.time()
    mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=K), order=order, tile=tile)
    case = CS.friction_slope_case(mesh, nx, ny, dt=1e-3, source_method=src, K=K)
    t1 = time.time()
    op = CS.create_operator(case)
    t2 = time.time()
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    for _ in range(5): op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): op.rhs_function(case.dt, u, f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/iters
    nc = mesh.num_owned_cells
    print(f"{nx}x{ny} {order:9s} src={src} cells={nc} mesh {t1-t0:.1f}s create {t2-t1:.1f}s  {ms*1e3:.1f} us/rhs  "
          f"{nc/ms/1e3:.0f} Mcell/s  alg {nc*176/ms/1e6:.0f} GB/s ({nc*176/ms/1e6/8000*100:.1f}% of 8TB/s) layout {op.layout_info()['bytes_per_apply']/ms/1e6:.0f} GB/s", flush=True)
    op.destroy()

if __name__ == "__main__":
    sizes = [(1000,500)] if len(sys.argv) < 2 else [tuple(map(int, a.split('x'))) for a in sys.argv[1:]]
    for nx, ny in sizes:
        for order in ("rowmajor", "tiled"):
            for src in (0, 1):
                run(nx, ny, order, src)
