"""The oracle still reproduces the committed golden vectors (guards the test
infrastructure against silent drift; the vectors are oracle-generated, see
tools/make_golden.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden

from helpers import oracle_from_case


@pytest.mark.parametrize("name", sorted(make_golden.golden_cases()))
def test_oracle_matches_golden(name):
    case = make_golden.golden_cases()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"rhs_{name}.npz"))
    assert np.array_equal(g["u_local"], case.u_local), "case definition drifted"
    orc = oracle_from_case(case)
    f = orc.apply(case.dt, case.u_local)
    # same compiler flags -> identical up to libm version differences in pow()
    assert np.max(np.abs(f - g["f"])) <= 1e-13 * max(1.0, np.abs(g["f"]).max())
    assert abs(orc.diagnostics()[0] - g["courant"][0]) <= 1e-14
