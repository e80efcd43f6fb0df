import torch
from rdycore_amd import mesh as M, cases as CS
from rdycore_amd.operator import Operator, RDyFlowConfig
from rdycore_amd.timestep import EulerStepper

mesh = M.structured_tri_mesh(250, 200, order="tiled")
op = Operator.create(RDyFlowConfig(), mesh, [M.CONDITION_REFLECTING] * len(mesh.boundaries))
op.set_domain_mannings_n(0.015 * torch.ones(mesh.num_owned_cells).numpy())
u = torch.zeros((mesh.num_cells, 3), dtype=torch.float64, device="cuda"); u[:, 0] = 1.0
f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
op.rhs_function(1e-3, u, f)
EulerStepper(op).advance(u, dt=1e-3, interval=0.1)
print("ok", float(f.abs().max()), float(u[:, 0].mean()))
