"""CPU: the oracle on the reference's Houston1km test (real DEM, file initial state, time-varying rain and stage)."""
import numpy as np
import pytest

import houston
from rdycore_amd import mesh as M


def test_fixture_readers():
    xyz, conn, side_sets = M.read_exodus(houston.DATA + "/Houston1km_with_z.exo")
    assert xyz.shape == (1458, 3) and conn.shape == (2746, 4) and (conn[:, 3] == -1).all() and list(side_sets) == [1]
    rain, bc, rasters = houston.datasets()
    assert rain.shape == (139, 2) and bc.shape == (572, 2) and rain[1, 0] == 3600.0 and bc[1, 0] == 900.0
    assert [int(r[0]) * int(r[1]) + 5 for r in rasters] == [r.size for r in rasters] == [2931, 2931]


@pytest.mark.parametrize("mode", ["homogeneous", "raster"])
def test_oracle_runs_the_houston_case(mode):
    case, u, orc, wet = houston.oracle_run(mode, t_stop=1200.0)
    mesh = case.mesh
    assert [b.name for b in mesh.boundaries] == ["bottom_wall", "unassigned"] and mesh.boundaries[0].num_edges == 2
    assert np.isfinite(u).all() and (u[:, 0] >= 0.0).all()
    # water mass balance: what rained in minus what left through the two Dirichlet edges (reflecting walls pass none;
    # the stage boundary drains far more than the rain adds in these 20 minutes)
    added = (u[:, 0] - case.u_local[:, 0]) @ mesh.cell_areas
    assert np.isfinite(orc.boundary_fluxes_accum[0]).all()
    out = (orc.boundary_fluxes_accum[0][:, 0] * mesh.edge_lengths[mesh.boundaries[0].edge_ids]).sum()
    src = 0.0
    rain, _, rasters = houston.datasets()
    # the source integrated the way the loop applied it: piecewise constant per 60 s interval
    from oracle import oracle as O
    if mode == "homogeneous":
        for k in range(20):
            src += O.forcing_current_data(rain, 60.0 * k, False)[1] * 60.0 * mesh.cell_areas.sum()
        assert abs(added - (src - out)) <= 1e-9 * abs(src)


def test_oracle_keeps_the_levee_lake_at_rest():
    """driver/tests/swe_roe/levee.hr.yaml, the reference's hydrostatic-reconstruction test, on its own fixtures
    (tests/golden/levee/): a lake at eta = 15 m behind a dry levee.  With HR 600 steps leave it untouched; without
    HR the first-order scheme sets it in motion."""
    import os
    from rdycore_amd import cases as CS
    from helpers import oracle_from_case
    case = CS.levee_hr_case(os.path.join(os.path.dirname(houston.DATA), "levee"))
    mesh = case.mesh
    assert mesh.num_cells == 506 and [b.name for b in mesh.boundaries] == ["unassigned"]
    wet = case.u_local[:, 0] > 0
    assert 200 < wet.sum() < 300 and np.allclose(case.u_local[wet, 0] + mesh.cell_zc[wet], 15.0, atol=1e-12)
    orc = oracle_from_case(case)
    u = case.u_local.copy()
    for _ in range(600):
        u = u + case.dt * orc.apply(case.dt, u)
    assert np.abs(u - case.u_local).max() < 1e-12
    case.config.well_balancing = 0
    f = oracle_from_case(case).apply(case.dt, case.u_local)
    assert np.abs(f).max() > 1e-2


def test_mixed_elements_fixture():
    import os
    from rdycore_amd import cases as CS
    case = CS.mixed_elements_case(os.path.join(os.path.dirname(houston.DATA), "mixed"))
    m = case.mesh
    assert m.num_cells == 116 and int((m.cell_nverts == 4).sum()) == 20 and m.boundaries[0].num_edges == m.num_boundary_edges == 36
    assert sorted(np.unique(case.u_local[:, 0]).tolist()) == [5.0, 10.0] and case.mannings.shape == (116,)
