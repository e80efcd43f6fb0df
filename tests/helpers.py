"""Shared test helpers: drive the CPU oracle and the HIP operator from one Case."""
import numpy as np

from oracle import oracle as O


def oracle_from_case(case):
    cfg = case.config
    orc = O.OracleOperator(case.mesh, case.condition_types, cfg.tiny_h, cfg.h_anuga_regular, cfg.xq2018_threshold, cfg.source_method, cfg.well_balancing)
    orc.mannings[:] = case.mannings
    orc.external_sources[:] = case.ext_src
    for b, vals in case.boundary_values.items():
        orc.boundary_values[b][:] = vals
    return orc


def rel_linf(a, b):
    """max |a-b| relative to max(1, max|b|): the RHS L-inf measure of BASELINE.json."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))
