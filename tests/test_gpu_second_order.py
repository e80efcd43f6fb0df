"""GPU parity of the second-order (MUSCL) path -- rdyhip_compute_gradients and
swe_rhs_muscl_kernel through the C ABI -- against the oracle's restatement of
ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213).  Tolerance: RHS L-inf <= 1e-10
relative to max(1, |F|_inf), as for the first-order path."""
import numpy as np
import pytest

import mms
from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import (LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER, SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT,
                                  Operator, RDyFlowConfig, RDyHipError)

from helpers import oracle_from_case, rel_linf
from test_gpu_parity import check_all, run_both, tri_mms_case

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(autouse=True)
def muscl_mode(rdyhip_kernel):
    """second order lives in the tiled kernel alone (the gradients formed in LDS by the flux kernel; the split form of rounds
    1-4 -- a gradient launch, a flux kernel reading the gradients -- is tools/probes/split_muscl_form.patch)"""
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    yield "fused"


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def second_order(case, limiter=LIMITER_MINMOD):
    case.config.second_order, case.config.limiter = True, limiter
    return case


@pytest.mark.parametrize("limiter", [LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER])
@pytest.mark.parametrize("source_method", [SOURCE_SEMI_IMPLICIT, SOURCE_IMPLICIT_XQ2018])
def test_tri_all_bcs_sources_limiters(limiter, source_method, muscl_mode):
    case = second_order(tri_mms_case(40, 28, source_method, order="tiled"), limiter)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)
    # the gradients themselves (the fused kernel keeps them on chip: ask for them)
    torch = _torch()
    op.compute_gradients(torch.tensor(case.u_local, dtype=torch.float64, device="cuda"))
    g = op.gradients.cpu().numpy()
    assert rel_linf(g, orc.gradients6()) <= TOL
    assert op.layout_info()["second_order_fused"] == 1
    # and the scheme differs from first order on this state
    case.config.second_order = False
    assert np.abs(oracle_from_case(case).apply(case.dt, case.u_local) - fr).max() > 1e-8


def test_dam_break_with_dry_cells_and_clamped_depths():
    mesh = M.structured_tri_mesh(48, 20)
    case = second_order(CS.dam_break_case(mesh, 24.0), LIMITER_NONE)   # unlimited extrapolation over the front: negative depths get clamped
    case.u_local[mesh.cell_centroids[:, 0] > 30.0, 0] = 0.0            # dry bed downstream
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)


@pytest.mark.parametrize("limiter", [LIMITER_MINMOD, LIMITER_VANLEER])
def test_quad_mesh(limiter):
    K = 2 * np.pi / 17
    mesh = M.structured_quad_mesh(30, 22, 1.0, 1.0, zfunc=CS.mms_bathymetry(K=K))
    case = second_order(CS.friction_slope_case(mesh, 30, 22, dt=1e-2, K=K), limiter)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)
    assert op.layout_info()["slots_per_cell"] == 4


def test_accumulate_semantics():
    case = second_order(tri_mms_case(24, 16, SOURCE_SEMI_IMPLICIT))
    rng = np.random.default_rng(3)
    f0 = rng.normal(size=(case.mesh.num_owned_cells, 3)) * 0.1
    f, fr, op, orc = run_both(case, accumulate_from=f0)
    assert rel_linf(f, fr) <= TOL


def test_random_cell_numbering():
    rng = np.random.default_rng(5)
    nx, ny = 30, 20
    K = 2 * np.pi / 13
    xyz, conn, _, _ = M.structured_tri_connectivity(nx, ny)
    xyz[:, 2] = CS.mms_bathymetry(K=K)(xyz[:, 0], xyz[:, 1])
    perm = rng.permutation(conn.shape[0])
    mesh = M.build_mesh(xyz, conn[perm], boundary_classifier=M.box_side_boundaries(0, nx, 0, ny))
    case = second_order(CS.friction_slope_case(mesh, nx, ny, dt=1e-2, K=K))
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)


@pytest.mark.parametrize("ghosts", ["tail", "interleaved"])
def test_one_rank_of_a_partition(ghosts):
    """A rank's local mesh with ghost cells: owned gradients from the kernel, ghost gradients as their owners
    would send them (here: taken from the global oracle), then the flux kernel over ALL edges of the owned
    cells -- equals the reference's owner-computes + reverse-add result, i.e. the global RHS."""
    torch = _torch()
    nxg, ny = 24, 10
    K = 2 * np.pi / 15
    z = CS.mms_bathymetry(K=K)
    g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
    gc = second_order(CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K))
    og = oracle_from_case(gc)
    fg = og.apply(gc.dt, gc.u_local)
    gg = og.gradients6()
    xyz, conn, cqi, _ = M.structured_tri_connectivity(nxg, ny)
    xyz[:, 2] = z(xyz[:, 0], xyz[:, 1])
    owned = (cqi >= 8) & (cqi < 16)
    mesh = M.extract_local_mesh(xyz, conn, owned, boundary_classifier=M.box_side_boundaries(0, nxg, 0, ny), ghosts=ghosts)
    case = second_order(CS.friction_slope_case(mesh, nxg, ny, dt=1e-2, K=K))
    op = CS.create_operator(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.full((mesh.num_owned_cells, 3), 7.0, dtype=torch.float64, device="cuda")
    # without the exchange the call is refused (the ghost gradients would be stale)
    with pytest.raises(RDyHipError):
        op.rhs_function(case.dt, u, f)
    op.compute_gradients(u)
    grads = op.gradients
    ghost = np.nonzero(mesh.cell_is_owned == 0)[0]
    own = mesh.cell_owned_to_local
    torch.cuda.synchronize()
    assert rel_linf(grads.cpu().numpy()[own], gg[mesh.cell_global_ids[own]]) <= TOL
    grads[torch.as_tensor(ghost, device="cuda")] = torch.as_tensor(gg[mesh.cell_global_ids[ghost]], device="cuda")
    op.apply_phase(0, True, case.dt, u, f, reset_diagnostics=True, gradients_ready=True)
    torch.cuda.synchronize()
    assert rel_linf(f.cpu().numpy(), fg[mesh.cell_global_ids[own]]) <= TOL
    # phased: gradients interior + halo, fluxes interior + halo, bitwise the same
    g_all = grads.clone()
    grads[torch.as_tensor(own, device="cuda")] = 0.0
    op.compute_gradients(u, phase=1)
    op.compute_gradients(u, phase=2)
    torch.cuda.synchronize()
    assert torch.equal(grads, g_all)
    f2 = torch.zeros_like(f)
    op.reset_boundary_fluxes_accum()
    op.apply_phase(1, True, case.dt, u, f2, reset_diagnostics=True, gradients_ready=True)
    op.apply_phase(2, True, case.dt, u, f2, gradients_ready=True)
    torch.cuda.synchronize()
    assert torch.equal(f2, f)
    with pytest.raises(RDyHipError):
        op.apply_phase(1, True, case.dt, u, f2)          # a phased apply needs the gradients to be ready


def test_second_order_mms_convergence_on_the_gpu():
    """mms_conv_study_second_order.yaml with the HIP operator in the loop: the reference's thresholds are
    exceeded and the rates equal the oracle's."""
    torch = _torch()

    def make_apply(mesh, bc_types):
        op = Operator.create(RDyFlowConfig(second_order=True), mesh, bc_types)
        f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")

        def apply(dt, u, src, bvals):
            for c in range(3):
                op.set_domain_external_source(c, src[:, c])
            op.set_boundary_values(0, bvals)
            ud = torch.tensor(u, dtype=torch.float64, device="cuda")
            op.rhs_function(dt, ud, f)
            return f.cpu().numpy()

        return apply, op.set_domain_mannings_n

    rates = mms.second_order_rates(make_apply)
    ref = mms.second_order_rates(mms.oracle_make_apply_second_order)
    for comp, expected in mms.EXPECTED_SECOND_ORDER.items():
        assert np.allclose(rates[comp], ref[comp], atol=1e-8), (rates[comp], ref[comp])
        assert all(r > t for r, t in zip(rates[comp], expected))


def test_unsupported_combinations_are_rejected():
    _torch()
    mesh = M.structured_tri_mesh(6, 4, project_2d=True)
    with pytest.raises(RDyHipError) as e:
        Operator.create(RDyFlowConfig(second_order=True, well_balancing=2), mesh, None)      # src/operator.c:388-389
    assert e.value.code == 83
    with pytest.raises(RDyHipError):
        Operator.create(RDyFlowConfig(second_order=True, limiter=7), mesh, None)
    op = Operator.create(RDyFlowConfig(), mesh, None)
    with pytest.raises(RDyHipError):
        op.gradients                                                                          # first-order operator: no gradient field


@pytest.mark.parametrize("nx,ny", [(1000, 500)])
def test_full_size_parity_and_mass_balance(nx, ny):
    """C2 size (1 M cells): whole RHS against the oracle, and water-mass balance."""
    torch = _torch()
    K = 2 * np.pi / 200.0
    mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled")
    case = second_order(CS.friction_slope_case(mesh, nx, ny, K=K))
    f, fr, op, orc = run_both(case)
    assert rel_linf(f, fr) <= TOL
    flux_out = 0.0
    for b, bnd in enumerate(mesh.boundaries):
        bf = op.boundary_fluxes(b)
        wet = ~np.isnan(bf[:, 0])
        flux_out += (bf[wet, 0] * mesh.edge_lengths[bnd.edge_ids][wet]).sum()
    lhs = (f[:, 0] * mesh.cell_areas).sum()
    rhs = -flux_out + (case.ext_src[:, 0] * mesh.cell_areas).sum()
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(rhs))


def test_resident_workgroups_of_the_fused_kernels(muscl_mode):
    """A register-count regression guard for the two tile loops of the fused kernel (csrc/muscl_kernels.h): triangles in the
    plane layout must keep FOUR workgroups per CU (<= 128 VGPRs: the next tile's cells group in flight costs twenty), and since
    round 5 quads as well (their tiles have two flux rounds; the cross-tile pipeline of rounds 3-4 ran three workgroups)."""
    torch = _torch()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tri = second_order(CS.dam_break_case(M.structured_tri_mesh(64, 32, order="tiled"), 24.0))
    quad = second_order(CS.dam_break_case(M.structured_quad_mesh(32, 64), 32.0))   # row-major, 32 wide: tiles of 32 x 8 quads fit the planes
    for case, per_cu in ((tri, 4), (quad, 4)):
        op = CS.create_operator(case)
        info = op.layout_info()
        assert info["second_order_fused"] == 1 and info["lds_fixed_layout"] == 1, info
        assert info["persistent_grid"] == per_cu * cus, (info["persistent_grid"], per_cu, cus)


# ---------------------------------------------------------------------------
# edge cases: empty and ragged inputs
# ---------------------------------------------------------------------------
def test_rank_with_no_owned_cells():
    torch = _torch()
    xyz, conn, _, _ = M.structured_tri_connectivity(3, 2)
    mesh = M.build_mesh(xyz, conn, is_owned=np.zeros(conn.shape[0], dtype=np.int32), boundary_classifier=M.box_side_boundaries(0, 3, 0, 2))
    op = Operator.create(RDyFlowConfig(second_order=True), mesh)
    u = torch.ones((mesh.num_cells, 3), dtype=torch.float64, device="cuda")
    f = torch.zeros((0, 3), dtype=torch.float64, device="cuda")
    op.compute_gradients(u)
    op.apply_phase(0, True, 0.1, u, f, reset_diagnostics=True, gradients_ready=True)
    op.rhs_function(0.1, u, f)          # nothing to do on this rank: not refused
    op.update_diagnostics()
    assert op.get_diagnostics().max_courant_num == 0.0
    assert op.gradients.shape == (mesh.num_cells, 6)
    op.destroy()


def test_single_cell_and_two_cells_have_degenerate_stencils():
    # one triangle (no internal edge) and two triangles (one internal edge): the 2x2 least-squares matrix is
    # singular, the reference zeroes the gradient (operator_fluxes_ceed.c:926-933) and the scheme is first order
    for conn in (np.array([[0, 1, 2]], dtype=np.int32), np.array([[0, 1, 2], [1, 3, 2]], dtype=np.int32)):
        xyz = np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.1], [0.0, 1.0, 0.3], [2.0, 1.5, 0.2]])[: conn.max() + 1]
        mesh = M.build_mesh(xyz, conn, boundary_classifier=M.single_boundary())
        nc = mesh.num_cells
        case = CS.Case("tiny", mesh, RDyFlowConfig(second_order=True), [M.CONDITION_REFLECTING],
                       np.array([[1.3, 0.4, -0.2], [0.7, -0.1, 0.3]])[:nc], np.full(nc, 0.03), np.zeros((nc, 3)), {}, 0.01)
        f, fr, op, orc = run_both(case)
        check_all(case, f, fr, op, orc)
        case.config.second_order = False
        assert np.array_equal(oracle_from_case(case).apply(case.dt, case.u_local), fr)


def test_mixed_tri_quad_mesh_second_order():
    # quads and triangles in one mesh: 4 slots per cell with empty slots on the triangles
    rng = np.random.default_rng(2)
    nx, ny = 12, 9
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    xyz = np.stack([ii.ravel() * 1.0, jj.ravel() * 1.0, 0.05 * np.sin(ii.ravel() * 0.7) * np.cos(jj.ravel() * 0.5)], axis=1)
    cells = []
    for j in range(ny):
        for i in range(nx):
            v00, v10, v11, v01 = j * (nx + 1) + i, j * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i
            if (i + j) % 3 == 0:
                cells += [[v00, v10, v11, -1], [v00, v11, v01, -1]]
            else:
                cells.append([v00, v10, v11, v01])
    conn = np.array(cells, dtype=np.int32)
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.box_side_boundaries(0, nx, 0, ny))
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    h = 1.0 + 0.2 * np.sin(0.5 * xc) * np.cos(0.4 * yc)
    u = np.stack([h, h * 0.3 * np.cos(0.3 * yc), -h * 0.2 * np.sin(0.6 * xc)], axis=1) + rng.normal(size=(mesh.num_cells, 3)) * 1e-3
    case = CS.Case("mixed", mesh, RDyFlowConfig(second_order=True, limiter=LIMITER_VANLEER),
                   [M.CONDITION_REFLECTING] * len(mesh.boundaries), u, np.full(mesh.num_cells, 0.02), np.zeros((mesh.num_cells, 3)), {}, 5e-3)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)
    assert op.layout_info()["slots_per_cell"] == 4


@pytest.mark.parametrize("kind", ["tri", "quad"])
@pytest.mark.parametrize("limiter", [LIMITER_MINMOD, LIMITER_VANLEER])
def test_tile_size_changes_nothing_but_the_tiling(limiter, kind, monkeypatch):
    """the tiles a mesh is cut into (RDYHIP_TILE_CELLS: 256 / 128 / 64 cells at most) decide which edges are evaluated twice and
    which gradients three times -- all copies from the same code path, so F, the Courant number and its ids have the same bits"""
    torch = _torch()
    if kind == "tri":
        case = second_order(tri_mms_case(64, 48, SOURCE_SEMI_IMPLICIT, order="tiled"), limiter)
    else:
        K = 2 * np.pi / 37
        mesh = M.structured_quad_mesh(80, 64, 1.0, 1.0, zfunc=CS.mms_bathymetry(K=K))
        case = second_order(CS.friction_slope_case(mesh, 80, 64, dt=1e-2, K=K), limiter)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    out, ntiles = [], []
    for cells in ("256", "128", "64"):
        monkeypatch.setenv("RDYHIP_TILE_CELLS", cells)
        op = CS.create_operator(case)
        f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
        op.rhs_function(case.dt, u, f)
        torch.cuda.synchronize()
        op.update_diagnostics()
        out.append((f.cpu().numpy(), op.get_diagnostics()))
        ntiles.append(op.layout_info()["num_tiles"])
        op.destroy()
    assert ntiles[0] < ntiles[1] < ntiles[2]
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and out[0][1] == o[1]
