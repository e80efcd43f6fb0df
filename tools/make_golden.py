"""Generates tests/golden/rhs_*.npz: inputs and expected outputs of one RHS
evaluation for a few small cases, computed by the CPU oracle
(oracle/swe_oracle.c) -- NOT by the reference, which cannot be built here.
The oracle itself is pinned to the reference in tests/test_oracle_pins.py.

    python tools/make_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from helpers import oracle_from_case


def golden_cases():
    K = 2 * np.pi / 17
    out = {}
    mesh = M.structured_tri_mesh(14, 9, 1.0, zfunc=CS.mms_bathymetry(K=K))
    out["tri_semi_implicit"] = CS.friction_slope_case(mesh, 14, 9, dt=1e-2, source_method=0, K=K)
    out["tri_xq2018"] = CS.friction_slope_case(mesh, 14, 9, dt=1e-2, source_method=1, K=K)
    mesh = M.structured_quad_mesh(9, 6, 1.0, 1.5, zfunc=CS.mms_bathymetry(K=K))
    out["quad_semi_implicit"] = CS.friction_slope_case(mesh, 9, 9, dt=5e-3, source_method=0, K=K)
    out["ex2b"] = CS.ex2b_case(os.path.join(ROOT, "tests", "golden", "planar_dam_10x5.msh"))
    # second order (MUSCL): minmod on triangles, van Leer on quads; hydrostatic reconstruction
    mesh = M.structured_tri_mesh(14, 9, 1.0, zfunc=CS.mms_bathymetry(K=K))
    c = CS.friction_slope_case(mesh, 14, 9, dt=1e-2, source_method=0, K=K)
    c.config.second_order = True
    out["tri_second_order_minmod"] = c
    mesh = M.structured_quad_mesh(9, 6, 1.0, 1.5, zfunc=CS.mms_bathymetry(K=K))
    c = CS.friction_slope_case(mesh, 9, 9, dt=5e-3, source_method=1, K=K)
    c.config.second_order, c.config.limiter = True, 2
    out["quad_second_order_vanleer"] = c
    mesh = M.structured_tri_mesh(14, 9, 1.0, zfunc=CS.mms_bathymetry(K=K), project_2d=True)
    c = CS.friction_slope_case(mesh, 14, 9, dt=1e-2, source_method=0, K=K)
    c.config.well_balancing = 2
    out["tri_hydrostatic_reconstruction"] = c
    return out


def main():
    for name, case in golden_cases().items():
        orc = oracle_from_case(case)
        f = orc.apply(case.dt, case.u_local)
        cmax, ce, cc = orc.diagnostics()
        d = dict(u_local=case.u_local, dt=case.dt, f=f, pv=orc.primitive_variables.copy(),
                 courant=np.array([cmax]), courant_ids=np.array([ce, cc]),
                 mannings=case.mannings, ext_src=case.ext_src)
        for b in range(len(case.mesh.boundaries)):
            d[f"bflux{b}"] = orc.boundary_fluxes[b].copy()
        path = os.path.join(ROOT, "tests", "golden", f"rhs_{name}.npz")
        np.savez_compressed(path, **d)
        print(path, f.shape, float(np.abs(f).max()))


if __name__ == "__main__":
    main()
