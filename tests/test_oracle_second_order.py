"""CPU: the oracle's second-order (MUSCL) path -- ApplyInteriorFlux2R and its helpers.

Pin: the reference's accuracy gate for this path, the expected rates of
driver/tests/swe_roe/mms_conv_study_second_order.yaml:54-69 (a rate must exceed
its threshold, src/rdymms.c:1005).  The remaining tests are properties of the
algorithm (exactness on linear fields, reduction to first order, conservation,
independence of the partition under the reference's owner-computes + reverse-add
scheme)."""
import numpy as np
import pytest

import mms
from oracle import oracle as O
from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER

from helpers import oracle_from_case, rel_linf, second_order_oracle_ranks


def test_second_order_mms_rates_exceed_reference_thresholds():
    rates = mms.second_order_rates(mms.oracle_make_apply_second_order)
    for comp, expected in mms.EXPECTED_SECOND_ORDER.items():
        for got, thr, norm in zip(rates[comp], expected, ("L1", "L2", "Linf")):
            assert np.isfinite(got) and got > thr, f"{norm} rate for {comp}: {got} (expected > {thr})"
    # and it is a second-order scheme where the first-order one is not: every L1 rate beats the first-order study's by > 0.4
    first = mms.convergence_rates(mms.oracle_make_apply, base_refinement=1, num_refinements=2)
    for comp in rates:
        assert rates[comp][0] > first[comp][0] + 0.4


def test_least_squares_gradient_is_exact_for_linear_fields():
    mesh = M.structured_quad_mesh(7, 5, 1.3, 0.7)
    orc = O.OracleOperator(mesh, [2] * len(mesh.boundaries), second_order=True, limiter=LIMITER_NONE)
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    u = np.stack([1 + 0.1 * xc + 0.2 * yc, 0.3 * xc - 0.05 * yc, -0.2 * yc], 1)
    orc.compute_gradients(u)
    nn = np.zeros(mesh.num_cells, int)
    for e in mesh.edge_internal_ids:
        nn[mesh.edge_cell_ids[2 * e]] += 1
        nn[mesh.edge_cell_ids[2 * e + 1]] += 1
    g = orc.gradients6()
    assert np.abs(g[nn >= 2] - np.array([0.1, 0.2, 0.3, -0.05, 0.0, -0.2])).max() < 1e-13
    # rows of the per-edge coefficient table: cx_LR, cy_LR, cx_RL, cy_RL (operator_fluxes_ceed.c:968-975)
    assert orc.ls_grad_coeffs.shape == (mesh.num_internal_edges, 4)


@pytest.mark.parametrize("limiter", [LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER])
def test_uniform_state_reduces_to_first_order(limiter):
    mesh = M.structured_tri_mesh(9, 6)
    case = CS.dam_break_case(mesh, 1e9, perturb=0.0)
    f1 = oracle_from_case(case).apply(case.dt, case.u_local)
    case.config.second_order, case.config.limiter = True, limiter
    f2 = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.array_equal(f1, f2)


@pytest.mark.parametrize("limiter", [LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER])
def test_mass_is_conserved_and_limiters_differ(limiter):
    mesh = M.structured_tri_mesh(20, 12)
    case = CS.dam_break_case(mesh, 10.0)
    case.config.second_order, case.config.limiter = True, limiter
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    mass = (f[:, 0] * mesh.cell_areas).sum()
    assert abs(mass) < 1e-10 * np.abs(f[:, 0] * mesh.cell_areas).sum()
    case.config.second_order = False
    f1 = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.abs(f - f1).max() > 1e-6


def test_second_order_rhs_is_independent_of_the_partition():
    nxg, ny, P = 18, 7, 3
    K = 2 * np.pi / 11
    z = CS.mms_bathymetry(K=K)
    g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
    gc = CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K)
    gc.config.second_order = True
    og = oracle_from_case(gc)
    fg = og.apply(gc.dt, gc.u_local)
    cases = []
    for r in range(P):
        m = M.strip_partition_tri_mesh(nxg // P, ny, r, P, 1.0, zfunc=z)
        c = CS.friction_slope_case(m, nxg, ny, dt=1e-2, K=K)
        c.config.second_order = True
        cases.append(c)
    # every internal edge has exactly one owning rank
    owned_edges = sum(int(c.mesh.edge_is_owned()[c.mesh.edge_internal_ids].sum()) for c in cases)
    assert owned_edges == g.num_internal_edges
    fs, orcs = second_order_oracle_ranks(cases)
    for c, f in zip(cases, fs):
        gid = c.mesh.cell_global_ids[c.mesh.cell_owned_to_local]
        assert rel_linf(f, fg[gid]) < 1e-13
    assert abs(max(o.diagnostics()[0] for o in orcs) - og.diagnostics()[0]) < 1e-14


def test_second_order_with_hr_is_rejected():
    mesh = M.structured_tri_mesh(4, 3, project_2d=True)
    case = CS.dam_break_case(mesh, 2.0)
    case.config.second_order, case.config.well_balancing = True, 2
    with pytest.raises(RuntimeError):
        oracle_from_case(case).apply(case.dt, case.u_local)   # src/operator.c:388-389
