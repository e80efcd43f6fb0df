"""CPU: the time-stepping logic of rdycore_amd/timestep.py around a stand-in operator (no GPU): the Runge-Kutta tableau
of temporal="rk4" (TSRK4, src/rdysetup.c:1187-1189) integrates du/dt = lambda u with fourth-order accuracy, forward Euler
with first order, and the interval logic of RDyAdvance lands on the coupling time (TS_EXACTFINALTIME_MATCHSTEP)."""
import math
import types

import numpy as np
import pytest
import torch

from rdycore_amd.timestep import EulerStepper


class LinearOp:
    """du/dt = lam * u on `n` cells, with the operator methods the stepper calls"""

    def __init__(self, n, lam):
        self.mesh = types.SimpleNamespace(num_owned_cells=n)
        self.lam = lam
        self.rhs_calls = 0

    def rhs_function(self, dt, u_local, f_global):
        self.rhs_calls += 1
        f_global.copy_(self.lam * u_local[: self.mesh.num_owned_cells])

    def axpy_owned(self, a, f_global, u_local):
        u_local[: self.mesh.num_owned_cells] += a * f_global

    def reset_diagnostics(self):
        pass


def integrate(temporal, nsteps, lam=-1.3, t_end=1.0, n=5):
    op = LinearOp(n, lam)
    st = EulerStepper(op, fused=False, temporal=temporal)
    u = torch.ones((n + 2, 3), dtype=torch.float64)      # two "ghost" rows the operator never reads
    st.advance(u, t_end / nsteps, t_end)
    assert st.step == nsteps and abs(st.time - t_end) < 1e-14
    return float(abs(u[0, 0] - math.exp(lam * t_end))), op.rhs_calls


def test_rk4_is_fourth_order_and_euler_first():
    e1, calls = integrate("rk4", 8)
    e2, _ = integrate("rk4", 16)
    assert calls == 4 * 8
    assert 14.0 < e1 / e2 < 18.0, (e1, e2)            # halving dt divides the error by 2^4
    f1, calls = integrate("euler", 64)
    f2, _ = integrate("euler", 128)
    assert calls == 64 and 1.9 < f1 / f2 < 2.1
    # one RK4 step of du/dt = lam u is the degree-4 Taylor polynomial of exp(lam dt)
    op = LinearOp(1, 0.7)
    u = torch.full((1, 3), 2.0, dtype=torch.float64)
    EulerStepper(op, fused=False, temporal="rk4").advance(u, 0.5, 0.5)
    z = 0.7 * 0.5
    assert abs(float(u[0, 1]) - 2.0 * (1 + z + z * z / 2 + z ** 3 / 6 + z ** 4 / 24)) < 1e-15


def test_last_step_is_shortened_to_land_on_the_coupling_time():
    op = LinearOp(3, -0.2)
    st = EulerStepper(op, fused=False, temporal="rk4")
    u = torch.ones((3, 3), dtype=torch.float64)
    dt = st.advance(u, 0.3, 1.0)                       # 0.3 + 0.3 + 0.3 + 0.1
    assert st.step == 4 and abs(st.time - 1.0) < 1e-14 and dt == 0.3
    with pytest.raises(ValueError):
        EulerStepper(op, temporal="beuler")
