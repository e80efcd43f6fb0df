"""GPU: the reference's Houston-Harvey test in miniature (tests/houston.py) -- real DEM, file initial state, rain and
stage series / hourly rain rasters ingested on the device, 140 Euler steps with the forcing re-applied every coupling
interval -- the device loop (EulerStepper + Forcing, fused Euler steps) against the same loop on the oracle."""
import numpy as np
import pytest

import houston
from helpers import rel_linf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["homogeneous", "raster"])
@pytest.mark.parametrize("fused", [True, False])
def test_houston_time_loop_matches_the_oracle(mode, fused, rdyhip_kernel):
    if rdyhip_kernel == "cell" and not fused:
        pytest.skip("one unfused run is enough")
    case, u_ref, orc, wet = houston.oracle_run(mode)
    _, u, op, st = houston.device_run(mode, fused=fused)
    assert st.step == 140 and abs(st.time - houston.T_STOP) < 1e-9
    assert np.isfinite(u).all()
    assert rel_linf(u, u_ref) <= 1e-10, rel_linf(u, u_ref)
    # the dt-weighted boundary fluxes of the whole run (time_series: boundary_fluxes in the reference's yaml)
    b = case.mesh.boundary_by_name("bottom_wall")
    assert rel_linf(op.boundary_fluxes(b, accumulated=True), orc.boundary_fluxes_accum[b]) <= 1e-10
    assert rel_linf(op.external_sources.cpu().numpy(), orc.external_sources) == 0.0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hr", [False, True])
def test_refined_houston_time_loop_matches_the_oracle(hr, rdyhip_kernel):
    """the unstructured benchmark workload as a RUN, in small: the Houston mesh refined three times (175 744 triangles, Hilbert
    order, 38 % of the cells dry at the start), the reference's rain and stage series re-applied every 60 s coupling interval
    on the device, 48 fused Euler steps with wet / dry fronts moving over the real DEM -- against the same loop on the oracle,
    with and without hydrostatic reconstruction"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough (HR lives in the tiled kernel)")
    t_stop = 180.0
    case, u_ref, orc, wet = houston.oracle_run("homogeneous", t_stop=t_stop, levels=3, hr=hr)
    _, u, op, st = houston.device_run("homogeneous", t_stop=t_stop, levels=3, hr=hr)
    assert st.step == 48 and abs(st.time - t_stop) < 1e-9
    assert np.isfinite(u).all() and np.isfinite(u_ref).all()
    h0 = case.u_local[:, 0]
    assert ((h0 == 0) != (u_ref[:, 0] <= 1e-7)).sum() > 100          # the wet / dry front has moved over hundreds of cells
    assert rel_linf(u, u_ref) <= 1e-10, rel_linf(u, u_ref)
    b = case.mesh.boundary_by_name("bottom_wall")
    assert rel_linf(op.boundary_fluxes(b, accumulated=True), orc.boundary_fluxes_accum[b]) <= 1e-10
    assert rel_linf(op.external_sources.cpu().numpy(), orc.external_sources) == 0.0


def test_houston_second_order(rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("tiled kernels only")
    import houston as H
    from helpers import oracle_from_case
    from oracle import oracle as O
    from rdycore_amd import cases as CS
    # the same loop with numerics.second_order on both sides, for the four intervals (8 steps) the second-order scheme
    # survives this 30 s step on the real DEM (not a configuration the reference runs; both sides blow up alike later)
    case = CS.houston_case(H.DATA)
    case.config.second_order = True
    orc = oracle_from_case(case)
    rain, bc, _ = H.datasets()
    d = case.mesh.boundary_by_name("bottom_wall")
    u_ref, t = case.u_local.copy(), 0.0
    while t < 240.0 * (1 - 1e-14):
        orc.external_sources[:, 0] = O.forcing_current_data(rain, t, False)[1]
        orc.boundary_values[d][:] = [O.forcing_current_data(bc, t, True)[1], 0.0, 0.0]
        for _ in range(2):
            u_ref = u_ref + H.DT * orc.apply(H.DT, u_ref)
            t += H.DT
    _, u, op, st = H.device_run("homogeneous", t_stop=240.0, second_order=True)
    assert st.step == 8 and np.isfinite(u_ref).all() and np.abs(u_ref).max() < 1e3
    assert rel_linf(u, u_ref) <= 1e-10


def test_levee_lake_at_rest_with_hydrostatic_reconstruction(rdyhip_kernel):
    """the reference's HR test (driver/tests/swe_roe/levee.hr.yaml) on its own fixtures: 600 fused Euler steps of the HR
    kernel leave the lake behind the dry levee at rest, as the oracle does"""
    if rdyhip_kernel == "cell":
        pytest.skip("tiled kernels only")
    import os
    import torch
    from rdycore_amd import cases as CS
    from rdycore_amd.timestep import EulerStepper
    from helpers import oracle_from_case
    case = CS.levee_hr_case(os.path.join(os.path.dirname(houston.DATA), "levee"))
    op = CS.create_operator(case)
    st = EulerStepper(op)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    st.advance(u, case.dt, 60.0)
    torch.cuda.synchronize()
    assert st.step == 600
    orc = oracle_from_case(case)
    u_ref = case.u_local.copy()
    for _ in range(600):
        u_ref = u_ref + case.dt * orc.apply(case.dt, u_ref)
    got = u.cpu().numpy()
    assert np.abs(got - case.u_local).max() < 1e-12          # at rest
    assert rel_linf(got, u_ref) <= 1e-10


@pytest.mark.parametrize("fused", [True, False])
def test_mixed_elements_dam_break_trajectory(fused, rdyhip_kernel):
    """driver/tests/swe_roe/mixed_elements_ic_file.yaml on its own fixtures (tests/golden/mixed/): 20 quads + 96 triangles,
    initial state and Manning n from the reference's binary files, 1000 Euler steps of 0.018 s, device against oracle"""
    import os
    import torch
    from rdycore_amd import cases as CS
    from rdycore_amd.timestep import EulerStepper
    from helpers import oracle_from_case
    case = CS.mixed_elements_case(os.path.join(os.path.dirname(houston.DATA), "mixed"))
    op = CS.create_operator(case)
    assert op.layout_info()["slots_per_cell"] == 4
    st = EulerStepper(op, fused=fused)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    st.advance(u, case.dt, 1000 * case.dt)
    torch.cuda.synchronize()
    assert st.step == 1000
    orc = oracle_from_case(case)
    u_ref = case.u_local.copy()
    for _ in range(1000):
        u_ref = u_ref + case.dt * orc.apply(case.dt, u_ref)
    assert rel_linf(u.cpu().numpy(), u_ref) <= 1e-10
    # reflecting walls all around: the water volume is the initial one
    a = case.mesh.cell_areas
    assert abs(u.cpu().numpy()[:, 0] @ a - case.u_local[:, 0] @ a) <= 1e-10 * (case.u_local[:, 0] @ a)


def test_quad_tri_mesh_with_multi_homogeneous_forcing(rdyhip_kernel):
    """driver/tests/swe_roe/quad_tri_mesh.yaml with the flags of its add_test line (CMakeLists.txt:233): the rain series on
    regions 1 and 3, the stage series on boundaries 1, 2 and 4 (FORCING_DATASET_MULTI_HOMOGENEOUS, src/forcing/rdyforcing.c:
    728-735, 757-769), a runoff source on region 2, a critical-outflow boundary -- all three boundary kinds, regional
    sources and mixed elements in twelve cells; 10 steps, device (Forcing + fused Euler) against oracle"""
    import os
    import torch
    from oracle import oracle as O
    from rdycore_amd import cases as CS
    from rdycore_amd import forcing as F
    from rdycore_amd.timestep import EulerStepper
    from helpers import oracle_from_case
    case, region = CS.quad_tri_case(os.path.join(os.path.dirname(houston.DATA), "quad_tri"))
    mesh = case.mesh
    rain, bc, _ = houston.datasets()
    rain_regions = [np.nonzero(region == r)[0].astype(np.int32) for r in (1, 3)]
    bc_boundaries = [mesh.boundary_by_name(n) for n in ("right", "left", "bottom")]
    op = CS.create_operator(case)
    frc = F.Forcing(op)
    for ids in rain_regions:
        frc.add_homogeneous_source(ids, F.HomogeneousDataset(rain, temporally_interpolate=False))
    for b in bc_boundaries:
        frc.add_homogeneous_boundary(b, F.HomogeneousDataset(bc, temporally_interpolate=False))
    st = EulerStepper(op, forcing=frc)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    st.advance(u, case.dt, 0.005)
    torch.cuda.synchronize()
    assert st.step == 10
    orc = oracle_from_case(case)
    for ids in rain_regions:
        orc.external_sources[ids, 0] = O.forcing_current_data(rain, 0.0, False)[1]
    for b in bc_boundaries:
        orc.boundary_values[b][:] = [O.forcing_current_data(bc, 0.0, False)[1], 0.0, 0.0]
    assert np.array_equal(op.external_sources.cpu().numpy(), orc.external_sources)
    assert orc.external_sources[region == 2, 0].tolist() == [0.0002] * 4            # the yaml's runoff stays on tri_1
    u_ref = case.u_local.copy()
    for _ in range(10):
        u_ref = u_ref + case.dt * orc.apply(case.dt, u_ref)
    assert rel_linf(u.cpu().numpy(), u_ref) <= 1e-10
    for b in range(len(mesh.boundaries)):
        assert rel_linf(np.nan_to_num(op.boundary_fluxes(b, accumulated=True)), np.nan_to_num(orc.boundary_fluxes_accum[b])) <= 1e-10


def test_parabolic_bowl_against_the_analytic_solution(rdyhip_kernel):
    """Thacker's oscillating planar surface (tests/bowl.py) on the device with hydrostatic reconstruction: one full period on
    80 x 80 and 160 x 160 squares, fused Euler steps; the L1 depth error against the ANALYTIC solution falls with the mesh
    size, the volume is conserved, and the 80 x 80 run equals the oracle's"""
    if rdyhip_kernel == "cell":
        pytest.skip("tiled kernels only")
    import torch
    import bowl
    from rdycore_amd import cases as CS
    from helpers import oracle_from_case
    errs = {}
    for n in (80, 160):
        case, nsteps = bowl.case_and_steps(n)
        op = CS.create_operator(case)
        a, b = torch.tensor(case.u_local, dtype=torch.float64, device="cuda"), None
        b = torch.empty_like(a)
        for _ in range(nsteps):
            op.euler_step(case.dt, a, b)
            a, b = b, a
        torch.cuda.synchronize()
        u = a.cpu().numpy()
        errs[n], mass = bowl.error_after_one_period(case, u)
        assert abs(mass - 1.0) < 1e-11
        if n == 80:
            orc = oracle_from_case(case)
            u_ref = case.u_local.copy()
            for _ in range(nsteps):
                u_ref = u_ref + case.dt * orc.apply(case.dt, u_ref)
            assert rel_linf(u, u_ref) <= 1e-9
    assert errs[80] < 0.08 and errs[160] < 0.75 * errs[80], errs
