"""GPU: committed golden vectors, end-to-end trajectories (ex2b, MMS), and
size-independent properties at BASELINE.json's full sizes (1 M and 10 M cells),
where the oracle would take too long to be the checker."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden
import mms

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from helpers import oracle_from_case, rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def gpu_rhs(case, op=None):
    torch = _torch()
    own = op is None
    if own:
        op = CS.create_operator(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    return f.cpu().numpy(), op


@pytest.mark.parametrize("name", sorted(make_golden.golden_cases()))
def test_golden_vectors(name, rdyhip_kernel):
    case = make_golden.golden_cases()[name]
    if rdyhip_kernel == "cell" and (case.config.second_order or case.config.well_balancing):
        pytest.skip("second order and hydrostatic reconstruction are implemented by the tiled kernels")
    g = np.load(os.path.join(ROOT, "tests", "golden", f"rhs_{name}.npz"))
    f, op = gpu_rhs(case)
    assert rel_linf(f, g["f"]) <= TOL
    assert rel_linf(op.primitive_variables.cpu().numpy(), g["pv"]) <= TOL
    op.update_diagnostics()
    d = op.get_diagnostics()
    assert abs(d.max_courant_num - g["courant"][0]) <= 1e-12
    assert [d.global_edge_id, d.global_cell_id] == g["courant_ids"].tolist()
    for b in range(len(case.mesh.boundaries)):
        ref = g[f"bflux{b}"]
        got = op.boundary_fluxes(b)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert rel_linf(np.nan_to_num(got), np.nan_to_num(ref)) <= TOL


def test_ex2b_trajectory_matches_oracle():
    """C1: 300 forward-Euler steps of ex2b (dt = 0.018 s) on the device
    (rhs_function + axpy_owned) against the same loop on the oracle."""
    torch = _torch()
    case = CS.ex2b_case(os.path.join(ROOT, "tests", "golden", "planar_dam_10x5.msh"))
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    uc = case.u_local.copy()
    for _ in range(300):
        op.rhs_function(case.dt, u, f)
        op.axpy_owned(case.dt, f, u)
        uc = uc + case.dt * orc.apply(case.dt, uc)
    torch.cuda.synchronize()
    assert np.isfinite(uc).all() and uc[:, 0].min() > 0
    assert np.abs(uc[:, 1]).max() > 1.0                  # the dam has broken
    assert rel_linf(u.cpu().numpy(), uc) <= 1e-10
    # critical-outflow boundary has passed water: accumulated boundary flux matches
    b = case.mesh.boundary_by_name("bottom_wall")
    assert rel_linf(np.nan_to_num(op.boundary_fluxes(b, accumulated=True)), np.nan_to_num(orc.boundary_fluxes_accum[b])) <= 1e-10


def test_mms_convergence_on_the_gpu():
    """The reference's accuracy gate (mms_conv_study.yaml:48-64) with the HIP
    operator in the loop (two refinement levels fewer steps than the CPU pin:
    dt = 0.01, t = 5, levels 1..3)."""
    torch = _torch()
    from rdycore_amd.operator import Operator, RDyFlowConfig

    def make_apply(mesh, bc_types):
        op = Operator.create(RDyFlowConfig(), mesh, bc_types)
        f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")

        def apply(dt, u, src, bvals):
            for c in range(3):
                op.set_domain_external_source(c, src[:, c])
            op.set_boundary_values(0, bvals)
            ud = torch.tensor(u, dtype=torch.float64, device="cuda")
            op.rhs_function(dt, ud, f)
            return f.cpu().numpy()

        return apply, op.set_domain_mannings_n

    rates = mms.convergence_rates(make_apply, base_refinement=1, num_refinements=2)
    ref = mms.convergence_rates(mms.oracle_make_apply, base_refinement=1, num_refinements=2)
    for comp in rates:
        assert np.allclose(rates[comp], ref[comp], atol=1e-9), (rates[comp], ref[comp])
        assert all(r > 0.75 for r in rates[comp])


@pytest.mark.parametrize("nx,ny", [(1000, 500), (2500, 2000)])
def test_full_size_properties(nx, ny):
    """C2 / C3 sizes (1 M and 10 M cells): properties that need no oracle --
    (a) the RHS is independent of the cell numbering (row-major vs tiled mesh:
        same per-cell values after un-permuting, bitwise, since a cell's edges
        are summed in a numbering-independent... order given by edge position),
    (b) water mass: sum F_h * area = -(boundary outflow) + sources,
    (c) the whole RHS against the oracle (1 M and 10 M cells),
    (d) the phased apply equals the full apply."""
    torch = _torch()
    K = 2 * np.pi / 200.0
    z = CS.mms_bathymetry(K=K)
    m1 = M.structured_tri_mesh(nx, ny, 1.0, zfunc=z, order="rowmajor")
    c1 = CS.friction_slope_case(m1, nx, ny, K=K)
    f1, op1 = gpu_rhs(c1)
    assert np.isfinite(f1).all()
    # (b) mass balance from the boundary fluxes
    flux_out = 0.0
    for b, bnd in enumerate(m1.boundaries):
        bf = op1.boundary_fluxes(b)
        wet = ~np.isnan(bf[:, 0])
        flux_out += (bf[wet, 0] * m1.edge_lengths[bnd.edge_ids][wet]).sum()
    lhs = (f1[:, 0] * m1.cell_areas).sum()
    rhs = -flux_out + (c1.ext_src[:, 0] * m1.cell_areas).sum()
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(rhs), np.abs(f1[:, 0] * m1.cell_areas).sum())
    # (a) numbering independence
    m2 = M.structured_tri_mesh(nx, ny, 1.0, zfunc=z, order="tiled")
    c2 = CS.friction_slope_case(m2, nx, ny, K=K)
    f2, op2 = gpu_rhs(c2)
    k1 = np.lexsort((m1.cell_centroids[:, 1].round(6), m1.cell_centroids[:, 0].round(6)))
    k2 = np.lexsort((m2.cell_centroids[:, 1].round(6), m2.cell_centroids[:, 0].round(6)))
    assert rel_linf(f1[k1], f2[k2]) <= 1e-13
    op1.update_diagnostics(); op2.update_diagnostics()
    assert abs(op1.get_diagnostics().max_courant_num - op2.get_diagnostics().max_courant_num) <= 1e-14
    # (c) the WHOLE right-hand side against the oracle (the C oracle does ~9 M cells/s, so even the
    # 10 M-cell case is one second of CPU), including primitive variables, Courant number and ids
    orc = oracle_from_case(c1)
    fo = orc.apply(c1.dt, c1.u_local)
    assert rel_linf(f1, fo) <= TOL
    assert rel_linf(op1.primitive_variables.cpu().numpy(), orc.primitive_variables) <= TOL
    d = op1.get_diagnostics()
    cmax, ce, cc = orc.diagnostics()
    assert abs(d.max_courant_num - cmax) <= 1e-12 and (d.global_edge_id, d.global_cell_id) == (ce, cc)
    del orc
    # (d) phases
    u = torch.tensor(c1.u_local, dtype=torch.float64, device="cuda")
    f = torch.full((m1.num_owned_cells, 3), -1.0, dtype=torch.float64, device="cuda")
    op1.apply_phase(1, True, c1.dt, u, f)
    op1.apply_phase(2, True, c1.dt, u, f)
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy(), f1)


@pytest.mark.parametrize("fused", [True, False])
def test_adaptive_euler_advance_matches_oracle_loop(fused):
    """RDyAdvance with adaptive dt (src/rdyadvance.c:303-343) on ex2b: the
    device-resident stepper against the same rules driven by the oracle."""
    torch = _torch()
    from rdycore_amd.timestep import AdaptiveTime, EulerStepper
    case = CS.ex2b_case(os.path.join(ROOT, "tests", "golden", "planar_dam_10x5.msh"))
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    ad = AdaptiveTime(target_courant_number=0.4, max_increase_factor=1.5)
    st = EulerStepper(op, adaptive=ad, fused=fused)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    uc = case.u_local.copy()
    dt_g = dt_c = 0.002
    interval, t_c, cmax = 0.2, 0.0, None
    for _ in range(12):
        dt_g = st.advance(u, dt_g, interval)
        # the same on the CPU
        if cmax is not None and cmax > 0:
            if cmax < ad.target_courant_number:
                dt_c = min(dt_c * min(ad.target_courant_number / cmax, ad.max_increase_factor), interval)
            else:
                dt_c *= ad.target_courant_number / cmax
        t_end = t_c + interval
        while t_c < t_end * (1.0 - 1e-14):
            h = min(dt_c, t_end - t_c)
            orc.reset_diagnostics()
            uc = uc + h * orc.apply(h, uc)
            t_c += h
        cmax = orc.diagnostics()[0]
        assert abs(dt_g - dt_c) <= 1e-12 * dt_c
        assert abs(st.max_courant - cmax) <= 1e-10
    torch.cuda.synchronize()
    assert st.step > 100 and abs(st.time - 12 * interval) < 1e-9
    assert rel_linf(u.cpu().numpy(), uc) <= 1e-9


@pytest.mark.parametrize("variant", ["first", "hr", "second"])
def test_fused_euler_step_equals_rhs_plus_axpy(variant, rdyhip_kernel):
    """rdyhip_euler_step (update fused into the RHS kernel's stores, or the fallback pair) against
    rdyhip_rhs_function + rdyhip_axpy_owned, on one rank of a partition (ghost cells, phased) and with F
    requested or not."""
    if rdyhip_kernel == "cell" and variant != "first":
        pytest.skip("tiled kernels only")
    torch = _torch()
    nxg, ny = 24, 10
    K = 2 * np.pi / 15
    xyz, conn, cqi, _ = M.structured_tri_connectivity(nxg, ny)
    xyz[:, 2] = CS.mms_bathymetry(K=K)(xyz[:, 0], xyz[:, 1])
    owned = (cqi >= 8) & (cqi < 16) if variant != "second" else np.ones(conn.shape[0], dtype=bool)
    mesh = M.extract_local_mesh(xyz, conn, owned, boundary_classifier=M.box_side_boundaries(0, nxg, 0, ny), ghosts="interleaved",
                                project_2d=(variant == "hr"))
    case = CS.friction_slope_case(mesh, nxg, ny, dt=1e-2, K=K)
    case.config.well_balancing = 2 if variant == "hr" else 0
    case.config.second_order = variant == "second"
    op = CS.create_operator(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    op.rhs_function(case.dt, u, f)
    ref = u.clone()
    op.axpy_owned(case.dt, f, ref)
    op.update_diagnostics()
    c_ref = op.get_diagnostics().max_courant_num
    pv_ref = op.primitive_variables.clone()
    ghost = torch.as_tensor(mesh.cell_is_owned == 0, device="cuda")
    for with_f in (False, True):
        out = torch.full_like(u, -7.0)
        f2 = torch.full_like(f, 3.0) if with_f else None
        op.primitive_variables.zero_()
        if variant == "second":
            op.euler_step(case.dt, u, out, f2)
        else:
            op.reset_boundary_fluxes_accum()
            op.euler_step(case.dt, u, out, f2, phase=1)
            op.euler_step(case.dt, u, out, f2, phase=2, reset_diagnostics=False)
        torch.cuda.synchronize()
        assert torch.all(out[ghost] == -7.0)                               # ghost rows are the next exchange's
        assert torch.allclose(out[~ghost], ref[~ghost], rtol=0, atol=1e-14)
        assert torch.equal(op.primitive_variables, pv_ref)
        if with_f:
            assert torch.equal(f2, f)
        op.update_diagnostics()
        assert op.get_diagnostics().max_courant_num == c_ref
    from rdycore_amd.operator import RDyHipError
    with pytest.raises(RDyHipError):
        op.euler_step(case.dt, u, u)                                       # not in place


@pytest.mark.timeout(1500)
def test_reference_dam_break_benchmark_full_size(rdyhip_kernel):
    """The reference's own published benchmark problem at full size (docs/user/example-cases/dam-break: 5120 x 2560 quads
    minus the dam = 11,534,336 cells, h = 10 / 5 m, n = 0.015, dt = 1.5625e-5 s, reflecting walls): the whole RHS against
    the oracle at the benchmark's initial state (at rest: only the breach edges carry a flux jump) and at a moving
    state, water mass balance, and the state after the benchmark's first Euler steps (device loop = oracle loop)."""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough at this size")
    torch = _torch()
    mesh = CS.dam_break_quads_mesh()
    assert mesh.num_cells == 11_534_336 and (mesh.cell_nverts == 4).all()
    case = CS.dam_break_quads_case(mesh)
    op = CS.create_operator(case)
    assert op.layout_info()["slots_per_cell"] == 4
    orc = oracle_from_case(case)
    dev = "cuda"
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device=dev)
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    moving = case.u_local.copy()
    moving[:, 1] = 0.3 * moving[:, 0] * np.sin(1.7 * xc + 0.9 * yc)
    moving[:, 2] = 0.2 * moving[:, 0] * np.cos(1.1 * xc - 2.3 * yc)
    for state in (case.u_local, moving):
        u = torch.tensor(state, dtype=torch.float64, device=dev)
        op.rhs_function(case.dt, u, f)
        fo = orc.apply(case.dt, state)
        fh = f.cpu().numpy()
        assert rel_linf(fh, fo) <= TOL
        assert rel_linf(op.primitive_variables.cpu().numpy(), orc.primitive_variables) <= TOL
        op.update_diagnostics()
        d = op.get_diagnostics()
        cmax = orc.diagnostics()[0]
        assert abs(d.max_courant_num - cmax) <= 1e-12 * max(1.0, cmax)
        # closed basin: the water mass does not change
        assert abs((fh[:, 0] * mesh.cell_areas).sum()) <= 1e-9 * np.abs(fh[:, 0] * mesh.cell_areas).sum() + 1e-12
        orc.reset_diagnostics()
    # the benchmark's first steps (of its 100): fused device Euler steps against the oracle's F + axpy loop
    from rdycore_amd.timestep import EulerStepper
    u = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
    st = EulerStepper(op)
    nsteps = 5
    st.advance(u, case.dt, nsteps * case.dt)
    assert st.step == nsteps
    uo = case.u_local.copy()
    for _ in range(nsteps):
        uo += case.dt * orc.apply(case.dt, uo)
    assert rel_linf(u.cpu().numpy(), uo) <= TOL
    op.destroy()


def test_flat_dam_break_state_one_million_cells(rdyhip_kernel):
    """BASELINE.json configs[1] with its own state: the 1 M-cell flat-bed dam break (h = 10 / 5, perturbed momenta,
    Manning 0.015, reflecting walls) against the oracle at full size"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough at this size")
    m = M.structured_tri_mesh(1000, 500, 1.0, order="tiled")
    c = CS.dam_break_case(m, 1000.0, dt=1e-3)
    f, op = gpu_rhs(c)
    orc = oracle_from_case(c)
    fo = orc.apply(c.dt, c.u_local)
    assert rel_linf(f, fo) <= TOL
    op.update_diagnostics()
    assert abs(op.get_diagnostics().max_courant_num - orc.diagnostics()[0]) <= 1e-12


def test_rk4_advance_matches_oracle_loop():
    """numerics.temporal: rk4 (TSRK4, src/rdysetup.c:1187-1189) on ex2b: the device-resident stages of
    rdycore_amd/timestep.py against the oracle-driven Runge-Kutta loop, 40 steps"""
    torch = _torch()
    from rdycore_amd.timestep import EulerStepper
    from helpers import oracle_rk4
    case = CS.ex2b_case(os.path.join(ROOT, "tests", "golden", "planar_dam_10x5.msh"))
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    st = EulerStepper(op, temporal="rk4")
    dt, n = 0.01, 40
    st.advance(u, dt, n * dt)
    torch.cuda.synchronize()
    assert st.step == n
    ref = oracle_rk4(orc, case.u_local, dt, n)
    assert rel_linf(u.cpu().numpy(), ref) <= 1e-10
    # and it is not the Euler trajectory
    ue = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    EulerStepper(op).advance(ue, dt, n * dt)
    assert rel_linf(ue.cpu().numpy(), ref) > 1e-6
    op.destroy()


def _check_whole_rhs_against_oracle(case, second_state=None, ids=True):
    """the WHOLE right-hand side of `case` on the device against the oracle: F, primitive variables, Courant number (and
    ids), every boundary's fluxes with their NaN pattern, the water-mass balance; returns the operator"""
    torch = _torch()
    mesh = case.mesh
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    for state in [case.u_local] + ([second_state] if second_state is not None else []):
        u = torch.tensor(state, dtype=torch.float64, device="cuda")
        op.rhs_function(case.dt, u, f)
        torch.cuda.synchronize()
        fh = f.cpu().numpy()
        fo = orc.apply(case.dt, state)
        assert np.isfinite(fh).all()
        assert rel_linf(fh, fo) <= TOL
        assert rel_linf(op.primitive_variables.cpu().numpy(), orc.primitive_variables) <= TOL
        op.update_diagnostics()
        d = op.get_diagnostics()
        cmax, ce, cc = orc.diagnostics()
        # second order: gradients formed on the chip differ from the oracle's by cond(M) x rounding
        assert abs(d.max_courant_num - cmax) <= (1e-10 if case.config.second_order else 1e-12) * max(1.0, cmax)
        if ids:
            assert (d.global_edge_id, d.global_cell_id) == (ce, cc)
        flux_out = 0.0
        for b, bnd in enumerate(mesh.boundaries):
            got, ref = op.boundary_fluxes(b), orc.boundary_fluxes[b]
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            assert rel_linf(np.nan_to_num(got), np.nan_to_num(ref)) <= TOL
            wet = ~np.isnan(got[:, 0])
            flux_out += (got[wet, 0] * mesh.edge_lengths[bnd.edge_ids][wet]).sum()
        own = mesh.cell_owned_to_local
        lhs = (fh[:, 0] * mesh.cell_areas[own]).sum()
        rhs = (case.ext_src[:, 0] * mesh.cell_areas[own]).sum() - flux_out
        if mesh.num_cells == mesh.num_owned_cells:      # one rank of a partition exchanges mass with its neighbours
            assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(rhs), np.abs(fh[:, 0] * mesh.cell_areas[own]).sum())
        orc.reset_diagnostics()
    return op


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("hr", [False, True])
def test_houston_refined_full_size(hr, rdyhip_kernel):
    """the unstructured real-DEM workload of bench.py (--workload houston_refined): the reference's Houston1km mesh refined six
    times as -dm_refine does (src/rdydm.c:82-188) = 11,247,616 triangles, Hilbert-ordered, wet/dry fronts, the reference's rain
    and stage series -- the whole RHS against the oracle, with and without hydrostatic reconstruction"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough at this size")
    case = CS.houston_refined_case(os.path.join(ROOT, "tests", "golden", "houston"), 6, "hilbert", hr=hr)
    assert case.mesh.num_cells == 2746 * 4 ** 6
    dry = (case.u_local[:, 0] == 0).mean()
    assert 0.2 < dry < 0.6
    op = _check_whole_rhs_against_oracle(case)
    info = op.layout_info()
    assert info["num_edge_records"] / case.mesh.num_cells < 1.70 and info["max_tile_edges"] <= 512
    op.destroy()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("limiter", ["minmod", "van_leer"])
def test_houston_refined_second_order(limiter, rdyhip_kernel):
    """second order (fused MUSCL kernel, plane layout in LDS) on the unstructured real-DEM mesh: Houston1km refined five times
    = 2.81 M triangles (the size of the reference's Harvey mesh) with wet / dry fronts, against the oracle's
    ApplyInteriorFlux2R restatement"""
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    case = CS.houston_refined_case(os.path.join(ROOT, "tests", "golden", "houston"), 5, "hilbert")
    case.config.second_order = True
    case.config.limiter = {"minmod": 0, "van_leer": 2}[limiter]
    op = _check_whole_rhs_against_oracle(case, ids=False)
    info = op.layout_info()
    assert info["second_order_fused"] == 1 and info["lds_fixed_layout"] == 1
    op.destroy()


@pytest.mark.timeout(900)
def test_delaunay_unstructured_full_size(rdyhip_kernel):
    """bench.py --workload delaunay: a Delaunay triangulation of 1211^2 jittered, graded points (2.93 M triangles, the size of the
    reference's Turning_30m Harvey mesh; vertex valences 3..11) with the C5 physics (HR, ~40 % dry, rain, critical outflow)"""
    if rdyhip_kernel == "cell":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    mesh = CS.delaunay_mesh(1210)
    val = np.bincount(mesh.cell_conn[:, :3].ravel())
    assert val.min() <= 3 and val.max() >= 10 and 2_900_000 < mesh.num_cells < 2_960_000
    case = CS.c5_case(mesh, 1210.0, 1210.0)
    op = _check_whole_rhs_against_oracle(case)
    op.destroy()


@pytest.mark.timeout(900)
def test_hydrostatic_reconstruction_ten_million_cells(rdyhip_kernel):
    """bench.py --hr at its own size: the 2500 x 2000 x 2 C3 mesh with x-y projected geometry and ApplyInteriorFluxHR
    (src/swe/swe_petsc.c:1000-1161) against the oracle, all cells"""
    if rdyhip_kernel == "cell":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    from rdycore_amd.operator import WELL_BALANCING_HR
    K = 2 * np.pi / 200.0
    mesh = M.structured_tri_mesh(2500, 2000, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", project_2d=True)
    case = CS.friction_slope_case(mesh, 2500.0, 2000.0, dt=1e-3, K=K)
    case.config.well_balancing = WELL_BALANCING_HR
    op = _check_whole_rhs_against_oracle(case)
    op.destroy()


@pytest.mark.timeout(900)
def test_c5_rank_3_of_8_full_size(rdyhip_kernel):
    """bench.py --workload c5 --emulate-world 8 --emulate-rank 3: one rank's 3.1 M-square part of the 5000 x 5000 C5 mesh (6.25 M
    owned triangles + its ghost layer, HR, wet/dry) against the oracle on the same local mesh"""
    if rdyhip_kernel == "cell":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    mesh = CS.c5_mesh(5000, 5000, 3, 8)
    assert mesh.num_owned_cells == 6_250_000 and mesh.num_cells > mesh.num_owned_cells
    case = CS.c5_case(mesh, 5000.0, 5000.0)
    op = _check_whole_rhs_against_oracle(case)
    op.destroy()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("variant", ["hr", "second_order"])
def test_houston_refined_45M(variant, rdyhip_kernel):
    """BASELINE.json configs[4] at its own scale: the reference's Houston1km real-DEM mesh refined SEVEN times = 44,990,464
    triangles, 38 % of them dry (wet / dry fronts along every slope), rain + Dirichlet stage from the reference's series --
    with hydrostatic reconstruction (ApplyInteriorFluxHR, src/swe/swe_petsc.c:1000-1161: configs[4]'s kernel) and, second
    variant, the MUSCL path (ApplyInteriorFlux2R, 98-213) -- the whole RHS against the oracle, all 45 M cells"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough at this size")
    case = CS.houston_refined_case(os.path.join(ROOT, "tests", "golden", "houston"), 7, "hilbert", hr=(variant == "hr"))
    assert case.mesh.num_cells == 2746 * 4 ** 7
    if variant == "second_order":
        case.config.second_order = True
    else:
        assert case.config.well_balancing == 2
    dry = (case.u_local[:, 0] == 0).mean()
    assert 0.2 < dry < 0.6
    op = _check_whole_rhs_against_oracle(case, ids=(variant == "hr"))
    info = op.layout_info()
    assert info["lds_fixed_layout"] == 1 and info["num_edge_records"] / case.mesh.num_cells < 1.70
    op.destroy()


@pytest.mark.timeout(900)
def test_c3_strip_rank_with_ghosts_full_size(rdyhip_kernel):
    """BASELINE.json configs[3] as one rank sees it: rank 1 of 3 of the weak-scaled strips -- 10 M owned cells, a ghost column on
    both sides (2 x 2000 cells numbered peer by peer after the owned ones) -- the whole RHS against the oracle on the same local
    mesh, the INTERIOR + HALO phases the overlapped step is made of = the single launch, and the step of the C ABI itself
    (rdyhip_rhs_overlapped, ghost rows refreshed in place by a looped-back exchange) = the plain RHS"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough at this size")
    torch = _torch()
    import ctypes as C
    from rdycore_amd import _lib
    K = 2 * np.pi / 200.0
    mesh = M.strip_partition_tri_mesh(2500, 2000, 1, 3, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled")
    assert mesh.num_owned_cells == 10_000_000 and mesh.num_cells == 10_004_000
    assert np.array_equal(mesh.cell_owner_rank[mesh.num_owned_cells:], np.repeat([0, 2], 2000))      # ghosts: peer by peer
    case = CS.friction_slope_case(mesh, 7500.0, 2000.0, dt=1e-3, K=K)
    op = _check_whole_rhs_against_oracle(case)
    info = op.layout_info()
    assert 0 < info["num_halo_tiles"] < 0.03 * info["num_tiles"]
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    f2 = torch.full_like(f, -1.0)
    op.rhs_function(case.dt, u, f)
    op.apply_phase(1, True, case.dt, u, f2, reset_diagnostics=True)
    op.apply_phase(2, True, case.dt, u, f2)
    torch.cuda.synchronize()
    assert torch.equal(f, f2)
    # the multi-rank step itself, its exchange looped back through a one-rank RCCL communicator (as bench.py --self-exchange):
    # the ghost-adjacent owned cells travel to the ghost rows, in place, behind the interior tiles; the result must be the
    # plain RHS of the state that exchange leaves behind
    lib = _lib.load()
    uid = C.create_string_buffer(128)
    _lib.check(lib.rdyhip_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
    ghost = np.ascontiguousarray(np.arange(mesh.num_owned_cells, mesh.num_cells, dtype=np.int32))
    gset = np.zeros(mesh.num_cells, dtype=bool)
    gset[ghost] = True
    cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
    cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
    sendc = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut])).astype(np.int32)
    n = min(sendc.size, ghost.size)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    h = C.c_void_p()
    _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, p(i32([0])), p(i32([n])), p(i32(sendc[:n])), p(i32([n])), p(i32(ghost[:n])), C.byref(h)))
    assert lib.rdyhip_halo_direct_receive(h) == 1
    st = int(torch.cuda.current_stream().cuda_stream)
    expect = u.clone()
    expect[mesh.num_owned_cells:mesh.num_owned_cells + n] = u[torch.as_tensor(sendc[:n].astype(np.int64), device="cuda")]
    op.rhs_function(case.dt, expect, f2)
    for form in (1, 0):                  # two streams (exchange beside the interior tiles), then everything in order
        _lib.check(lib.rdyhip_halo_set_form(h, 0, form))
        assert lib.rdyhip_halo_overlaps(h) == form
        w = u.clone()
        f.fill_(-3.0)
        _lib.check(lib.rdyhip_rhs_overlapped(op._h, h, case.dt, int(w.data_ptr()), int(f.data_ptr()), st))
        torch.cuda.synchronize()
        assert torch.equal(w, expect) and torch.equal(f, f2), form
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
    _lib.check(lib.rdyhip_comm_destroy(comm))
    op.destroy()
