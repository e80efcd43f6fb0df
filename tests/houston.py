"""The reference's Houston-Harvey test in miniature (driver/tests/swe_roe/Houston1km.DirichletBC.yaml and the
add_test lines of driver/tests/swe_roe/CMakeLists.txt:110-158) driven the way driver/main.c does it: every coupling
interval of 60 s, RDyApplyForcing at the current time, then RDyAdvance (two Euler steps of 30 s), to 4200 s.

  homogeneous: -homogeneous_rain_file Houston1km.rain.*.bin  -homogeneous_bc_file Houston1km.bc.*.bin -temporally_interpolate_bc
  raster:      -raster_rain_start_date 2017,8,26,0,0 -raster_rain_dir ./   (hourly 77 x 38 rasters in mm/h, nearest neighbour)

All data files are the reference's own fixtures (tests/golden/houston/, copied from share/meshes and share/conditions).
`oracle_run` is the CPU loop on the oracle, `device_run` the same loop on the HIP operator (EulerStepper + Forcing).
"""
import os

import numpy as np

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "houston")
DT, INTERVAL, T_STOP = 30.0, 60.0, 4200.0          # Houston1km.DirichletBC.yaml:13-18


def datasets():
    rain = M.read_petsc_vec(os.path.join(DATA, "Houston1km.rain.int32.bin")).reshape(-1, 2)
    bc = M.read_petsc_vec(os.path.join(DATA, "Houston1km.bc.int32.bin")).reshape(-1, 2)
    rasters = [M.read_petsc_vec(os.path.join(DATA, f"2017-08-26:0{h}-00.int32.bin")) for h in (0, 1)]
    return rain, bc, rasters


def _case(levels: int, hr: bool):
    """the reference's test mesh itself (levels = 0) or its refinement (the unstructured benchmark workload in small:
    cases.houston_refined_case -- dt halves with every level, the coupling interval stays 60 s)"""
    if levels == 0:
        return CS.houston_case(DATA), DT
    case = CS.houston_refined_case(DATA, levels, "hilbert", hr=hr, time=0.0)
    return case, DT / 2 ** levels


def oracle_run(mode: str, t_stop: float = T_STOP, levels: int = 0, hr: bool = False):
    from oracle import oracle as O
    from helpers import oracle_from_case
    case, dt0 = _case(levels, hr)
    mesh = case.mesh
    orc = oracle_from_case(case)
    rain, bc, rasters = datasets()
    dirichlet = mesh.boundary_by_name("bottom_wall")
    oc = mesh.owned_centroids()
    if mode == "raster":
        ncols, nrows, xlc, ylc, cs = int(rasters[0][0]), int(rasters[0][1]), rasters[0][2], rasters[0][3], rasters[0][4]
        xs = xlc + np.arange(ncols) * cs + cs / 2.0
        ys = ylc + (nrows - 1 - np.arange(nrows)) * cs + cs / 2.0
        rmap = O.forcing_raster_map(oc[:, 0], oc[:, 1], ncols, nrows, cs, np.tile(xs, nrows), np.repeat(ys, ncols))
        nfile = 1
    u = case.u_local.copy()
    t = 0.0
    wet_history = []
    while t < t_stop * (1.0 - 1e-14):
        # RDyApplyForcing(rdy, forcing, time)
        if mode == "raster":
            if t / 3600.0 >= nfile * 1.0:
                nfile += 1
            orc.external_sources[:, 0] = O.forcing_set_raster(rasters[nfile - 1], 5, rmap)
        else:
            orc.external_sources[:, 0] = O.forcing_current_data(rain, t, False)[1]
            orc.boundary_values[dirichlet][:] = [O.forcing_current_data(bc, t, True)[1], 0.0, 0.0]
        # RDyAdvance: one coupling interval
        t_end = t + INTERVAL
        while t < t_end * (1.0 - 1e-14):
            h = min(dt0, t_end - t)
            u = u + h * orc.apply(h, u)
            t += h
        wet_history.append(int((u[:, 0] > 1e-7).sum()))
    return case, u, orc, wet_history


def device_run(mode: str, t_stop: float = T_STOP, fused: bool = True, second_order: bool = False, levels: int = 0, hr: bool = False):
    import torch
    from rdycore_amd import forcing as F
    from rdycore_amd.timestep import EulerStepper
    case, dt0 = _case(levels, hr)
    case.config.second_order = second_order
    mesh = case.mesh
    op = CS.create_operator(case)
    rain, bc, rasters = datasets()
    frc = F.Forcing(op)
    ras = None
    if mode == "raster":
        oc = mesh.owned_centroids()
        ras = F.RasterDataset(rasters[0], oc[:, 0], oc[:, 1], "cuda")
        frc.add_raster_source(None, ras)
    else:
        frc.add_homogeneous_source(None, F.HomogeneousDataset(rain, temporally_interpolate=False))
        frc.add_homogeneous_boundary(mesh.boundary_by_name("bottom_wall"), F.HomogeneousDataset(bc, temporally_interpolate=True))
    st = EulerStepper(op, forcing=frc, fused=fused)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    nfile = 1
    while st.time < t_stop * (1.0 - 1e-14):
        if ras is not None and ras.needs_next_file(st.time):
            ras.load_next(rasters[nfile])
            nfile += 1
        st.advance(u, dt0, INTERVAL)
    torch.cuda.synchronize()
    return case, u.cpu().numpy(), op, st
