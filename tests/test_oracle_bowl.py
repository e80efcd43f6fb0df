"""CPU: the oracle (hydrostatic-reconstruction path) against Thacker's analytic parabolic-bowl solution (tests/bowl.py)."""
import numpy as np

import bowl
from helpers import oracle_from_case


def test_oracle_tracks_the_oscillating_planar_surface():
    errs = []
    for n in (40, 80):
        case, nsteps = bowl.case_and_steps(n)
        orc = oracle_from_case(case)
        u = case.u_local.copy()
        for _ in range(nsteps):
            u = u + case.dt * orc.apply(case.dt, u)
        assert np.isfinite(u).all() and (u[:, 0] >= 0).all()
        err, mass = bowl.error_after_one_period(case, u)
        assert abs(mass - 1.0) < 1e-12           # closed basin: the volume is conserved to rounding
        errs.append(err)
    # first-order scheme with a moving shoreline: ~10 % L1 depth error after a full period on 40 x 40 squares, ~6 % on 80 x 80
    assert errs[0] < 0.15 and errs[1] < 0.08 and errs[1] < 0.75 * errs[0], errs
