"""oracle/libswe_oracle_omp.so (the same source with -fopenmp; bench.py's all-cores CPU line) gives the serial oracle's
results bit for bit: the Riemann batch is embarrassingly parallel and each cell's flux sum is formed by one thread in
the serial loop's order."""
import os

import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from oracle import oracle as O


def run(case, openmp):
    cfg = case.config
    orc = O.OracleOperator(case.mesh, case.condition_types, cfg.tiny_h, cfg.h_anuga_regular, cfg.xq2018_threshold, cfg.source_method,
                           cfg.well_balancing, openmp=openmp)
    orc.mannings[:] = case.mannings
    orc.external_sources[:] = case.ext_src
    for b, vals in case.boundary_values.items():
        orc.boundary_values[b][:] = vals
    f = orc.apply(case.dt, case.u_local)
    return f, orc.primitive_variables.copy(), orc.diagnostics(), [bf.copy() for bf in orc.boundary_fluxes]


@pytest.mark.parametrize("source", [0, 1])
@pytest.mark.parametrize("kind", ["tri", "quad_hole", "ghosts"])
def test_openmp_oracle_is_bitwise_the_serial_oracle(kind, source):
    assert O.lib(openmp=True).oracle_set_num_threads(4) == 4 and O.lib().oracle_set_num_threads(4) == 1
    K = 2 * np.pi / 37
    if kind == "tri":
        mesh = M.structured_tri_mesh(60, 44, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=8)
        case = CS.friction_slope_case(mesh, 60.0, 44.0, dt=1e-2, source_method=source, K=K)
    elif kind == "quad_hole":
        case = CS.dam_break_quads_case(CS.dam_break_quads_mesh(160, 80))
        case.config.source_method = source
        xc, yc = case.mesh.cell_centroids[:, 0], case.mesh.cell_centroids[:, 1]
        case.u_local[:, 1] = 0.3 * case.u_local[:, 0] * np.sin(1.7 * xc + 0.9 * yc)
    else:
        mesh = M.strip_partition_tri_mesh(20, 30, 1, 3, 1.0, zfunc=CS.mms_bathymetry(K=K))
        case = CS.friction_slope_case(mesh, 60.0, 30.0, dt=1e-2, source_method=source, K=K)
    f0, pv0, d0, bf0 = run(case, False)
    f1, pv1, d1, bf1 = run(case, True)
    assert np.array_equal(f0, f1) and np.array_equal(pv0, pv1) and d0 == d1
    for a, b in zip(bf0, bf1):
        assert np.array_equal(a, b, equal_nan=True)
