"""Domain properties of the RHS, checked on the oracle (CPU): conservation,
lake at rest, independence of the partition, dt-dependence of friction."""
import numpy as np

from rdycore_amd import mesh as M
from rdycore_amd import cases as CS

from helpers import oracle_from_case, rel_linf


def test_mass_is_conserved_with_reflecting_walls():
    mesh = M.structured_tri_mesh(30, 20)
    case = CS.dam_break_case(mesh, 30.0)
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    assert abs((f[:, 0] * mesh.cell_areas).sum()) < 1e-10 * np.abs(f[:, 0] * mesh.cell_areas).sum()


def test_lake_at_rest_on_flat_bed_has_zero_rhs():
    mesh = M.structured_tri_mesh(12, 9)
    case = CS.dam_break_case(mesh, 1e9, perturb=0.0)          # h = 10 everywhere
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.abs(f).max() < 1e-11


def test_rhs_is_independent_of_the_partition():
    # shared edges are computed on both ranks, each writes its own side
    nxg, ny, P = 18, 7, 3
    K = 2 * np.pi / 11
    z = CS.mms_bathymetry(K=K)
    g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
    gc = CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K)
    fg = oracle_from_case(gc).apply(gc.dt, gc.u_local)
    cmax_g = None
    seen = np.zeros(g.num_cells, dtype=bool)
    cmax = 0.0
    for r in range(P):
        m = M.strip_partition_tri_mesh(nxg // P, ny, r, P, 1.0, zfunc=z)
        c = CS.friction_slope_case(m, nxg, ny, dt=1e-2, K=K)
        orc = oracle_from_case(c)
        f = orc.apply(c.dt, c.u_local)
        gid = m.cell_global_ids[m.cell_owned_to_local]
        assert rel_linf(f, fg[gid]) < 1e-13
        seen[gid] = True
        cmax = max(cmax, orc.diagnostics()[0])
    assert seen.all()
    og = oracle_from_case(gc)
    og.apply(gc.dt, gc.u_local)
    assert abs(cmax - og.diagnostics()[0]) < 1e-14


def test_dt_enters_the_rhs_through_friction_and_courant():
    mesh = M.structured_tri_mesh(10, 8, zfunc=CS.mms_bathymetry(K=0.5))
    case = CS.friction_slope_case(mesh, 10, 8, K=0.5, dry_disc=False)
    o1 = oracle_from_case(case)
    o2 = oracle_from_case(case)
    f1 = o1.apply(1e-3, case.u_local)
    f2 = o2.apply(1e-1, case.u_local)
    assert np.abs(f1[:, 1] - f2[:, 1]).max() > 1e-6
    assert np.array_equal(f1[:, 0], f2[:, 0])
    assert abs(o2.diagnostics()[0] / o1.diagnostics()[0] - 100.0) < 1e-9


def test_hydrostatic_reconstruction_keeps_a_lake_at_rest():
    # the defining property of HR well-balancing: flat free surface over a bumpy
    # bed is a steady state (docs/theory/second_order_hydrostatic_reconstruction.md);
    # without HR the first-order scheme is not.
    from rdycore_amd.operator import WELL_BALANCING_HR
    K = 2 * np.pi / 9
    mesh = M.structured_tri_mesh(14, 10, 1.0, zfunc=lambda x, y: 0.3 * np.sin(K * x) * np.cos(K * y), project_2d=True)
    case = CS.dam_break_case(mesh, 1e9, perturb=0.0)
    case.u_local[:, 0] = 2.0 - mesh.cell_zc            # eta = h + zc = 2 everywhere
    case.config.well_balancing = WELL_BALANCING_HR
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.abs(f).max() < 1e-12
    case.config.well_balancing = 0
    f0 = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.abs(f0).max() > 1e-3
