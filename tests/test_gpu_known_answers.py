"""GPU: the HIP operator against the same hand-derived known answers the oracle is held to
(tests/known_answers.py, tests/test_known_answers_cpu.py)."""
import numpy as np
import pytest
import torch

import known_answers as KA
from rdycore_amd import cases as CS
from test_known_answers_cpu import check

pytestmark = pytest.mark.gpu


def gpu_apply(case):
    op = CS.create_operator(case)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    return f.cpu().numpy(), op


@pytest.mark.parametrize("name", sorted(KA.entries()))
def test_hip_operator_reproduces_the_known_answer(name, rdyhip_kernel):
    ent = KA.entries()[name]
    f, op = gpu_apply(ent["case"])
    op.update_diagnostics()
    check(name, ent, f, op.boundary_fluxes(0), bitwise=False, courant=op.get_diagnostics().max_courant_num)
    op.destroy()


def test_hydrostatic_reconstruction_is_well_balanced_on_a_bed_step(rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    case, rhs = KA.hr_two_cell_step()
    f, op = gpu_apply(case)
    assert np.max(np.abs(f - rhs)) <= 1e-14
    op.destroy()


def test_second_order_is_exact_for_a_linear_state(rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    case, rhs, interior = KA.second_order_linear_field()
    f, op = gpu_apply(case)
    err = np.abs(f[interior] - rhs[interior]).max() / np.abs(rhs[interior]).max()
    assert err <= 1e-12, err
    op.destroy()
