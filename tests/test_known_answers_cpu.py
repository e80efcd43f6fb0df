"""The oracle against hand-derived known answers (tests/known_answers.py) for the branches the reference's tests hold no
values for: reflecting / Dirichlet / critical-outflow boundary fluxes, both friction schemes, bed slope, sources,
hydrostatic reconstruction on a bed step."""
import numpy as np
import pytest

import known_answers as KA
from helpers import oracle_from_case


def check(name, ent, f, bflux, bitwise=True, courant=None):
    tol = ent["tol"]
    if ent.get("courant") is not None:
        assert courant is not None and abs(courant - ent["courant"]) <= 1e-14 * ent["courant"], (name, courant, ent["courant"])
    scale = lambda a: max(1.0, float(np.nanmax(np.abs(a)))) if np.size(a) else 1.0
    if ent.get("rhs") is not None:
        assert np.max(np.abs(f - ent["rhs"])) <= tol * scale(ent["rhs"]) , (name, f, ent["rhs"])
    if ent.get("flux") is not None:
        exp = ent["flux"]
        rows = ent.get("flux_rows", range(exp.shape[0]))
        for e in rows:
            assert np.max(np.abs(bflux[e] - exp[e].astype(float))) <= tol * scale(exp[e].astype(float)), (name, e, bflux[e], exp[e])
    for e in ent.get("nan_rows", []):
        assert np.isnan(bflux[e]).all(), (name, e, bflux[e])          # dry / dry: 0/0 in the Roe average, never added to F
    for kind, col in ent.get("exact", []):
        # exact in IEEE arithmetic with the reference's operation order (the oracle); the device contracts to FMAs and uses
        # refined reciprocals, which leaves the mirror symmetry intact only to rounding
        got, exp = (f, ent["rhs"]) if kind == "rhs" else (bflux, ent["flux"])
        if bitwise:
            assert np.array_equal(got[:, col], exp[:, col].astype(float)), (name, kind, col, got[:, col])
        else:
            assert np.max(np.abs(got[:, col] - exp[:, col].astype(float))) <= 1e-15 * scale(bflux if kind == "flux" else f), (name, kind, col, got[:, col])


@pytest.mark.parametrize("name", sorted(KA.entries()))
def test_oracle_reproduces_the_known_answer(name):
    ent = KA.entries()[name]
    case = ent["case"]
    orc = oracle_from_case(case)
    f = orc.apply(case.dt, case.u_local)
    check(name, ent, f, orc.boundary_fluxes[0].copy(), courant=orc.diagnostics()[0])


def test_critical_outflow_inflow_side_contributes_nothing():
    """flow pointing into the domain at a critical-outflow edge: both states are set dry, the edge is skipped; with the
    cell's other three edges removed from the picture (reflecting walls at rest would add pressure) use the -x edge alone:
    F of the critical_outflow entry = -(flux through +x) - (dam-break fluxes through +-y), and nothing from -x"""
    ent = KA.entries()["critical_outflow"]
    case = ent["case"]
    orc = oracle_from_case(case)
    f = orc.apply(case.dt, case.u_local)
    m = case.mesh
    b = m.boundaries[0]
    bf = orc.boundary_fluxes[0]
    total = np.zeros(3)
    for e in range(b.num_edges):
        if e in ent["nan_rows"]:
            continue
        total -= bf[e] * m.edge_lengths[b.edge_ids[e]] / m.cell_areas[0]
    assert np.allclose(f[0], total, rtol=0, atol=1e-13 * max(1.0, np.abs(total).max()))


def test_hydrostatic_reconstruction_is_well_balanced_on_a_bed_step():
    case, rhs = KA.hr_two_cell_step()
    orc = oracle_from_case(case)
    f = orc.apply(case.dt, case.u_local)
    assert np.max(np.abs(f - rhs)) <= 1e-14
    # and the scheme without reconstruction is not (the step's pressure imbalance g (1^2 - 0.6^2) / 2 acts on both cells)
    case.config.well_balancing = 0
    f0 = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.max(np.abs(f0[:, 1])) > 1.0


def test_second_order_is_exact_for_a_linear_state():
    """oracle-independent answer for the second-order path (gradient, reconstruction, flux): tests/known_answers.py"""
    case, rhs, interior = KA.second_order_linear_field()
    assert interior.sum() > 40
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    err = np.abs(f[interior] - rhs[interior]).max() / np.abs(rhs[interior]).max()
    assert err <= 1e-12, err
