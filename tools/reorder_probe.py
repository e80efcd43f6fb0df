"""RHS time for caller numberings of decreasing locality, and for the same meshes renumbered by their owner
along a Hilbert curve (rdycore_amd.mesh.hilbert_cell_order) before the operator is created."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdycore_amd import mesh as M, cases as CS

def run(nx, ny, kind, hilbert):
    K = 2*np.pi/200
    xyz, conn, _, _ = M.structured_tri_connectivity(nx, ny, order="rowmajor" if kind != "tiled" else "tiled")
    xyz[:, 2] = CS.mms_bathymetry(K=K)(xyz[:, 0], xyz[:, 1])
    rng = np.random.default_rng(1)
    if kind == "random":
        conn = conn[rng.permutation(conn.shape[0])]
    elif kind == "blocks":           # row-major blocks of 4096 cells in random order
        nb = conn.shape[0] // 4096
        order = np.concatenate([np.arange(b*4096, (b+1)*4096) for b in rng.permutation(nb)] + [np.arange(nb*4096, conn.shape[0])])
        conn = conn[order]
    if hilbert:
        cent = xyz[conn].mean(axis=1)
        conn = conn[M.hilbert_cell_order(cent)]
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.box_side_boundaries(0, nx, 0, ny))
    case = CS.friction_slope_case(mesh, nx, ny, dt=1e-3, K=K)
    op = CS.create_operator(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    for _ in range(5): op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    info = op.layout_info()
    print(f"{kind:8s} hilbert={int(hilbert)} rec/cell={info['num_edge_records']/mesh.num_owned_cells:.3f} "
          f"halo/tile={info['num_halo_entries']/info['num_tiles']:.0f} lds={info['lds_bytes']}  {ms:.4f} ms  {mesh.num_owned_cells/ms/1e3:.0f} Mcell/s", flush=True)
    op.destroy()

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1500, 1000)
for kind in ("tiled", "rowmajor", "blocks", "random"):
    for hilbert in (False, True):
        run(nx, ny, kind, hilbert)
