"""Device-resident explicit time stepping around the RHS operator (SURVEY.md
section 8.f row 1): what PETSc's TSEULER and RDyAdvance do between RHS
evaluations, with the state kept on the GPU.

  * forward Euler: F = RHS(t, U); U += dt F   (TSStep_Euler's VecAXPY; the RHS is
    OperatorRHSFunction, src/rdysetup.c:1120-1172).  `fused=True` (default) does both in
    one pass over the mesh (rdyhip_euler_step: the update rides on the RHS kernel's
    stores, F is never written, two state arrays ping-pong); `fused=False` keeps the
    RHS + axpy pair.
  * RDyAdvance (src/rdyadvance.c:261-383): advance to the next coupling time
    with the last step shortened to land on it (TS_EXACTFINALTIME_MATCHSTEP),
    and, when adaptive time stepping is on, rescale dt from the previous
    interval's maximum Courant number (303-343).
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import torch
import torch.distributed as dist


def reduce_courant(diag, device=None, group=None):
    """UpdateOperatorDiagnostics' cross-rank step (src/operator.c:879): the MPI_Allreduce of the
    {max_courant_num, global_edge_id, global_cell_id} struct with FindCourantNumberDiagnostics (705-715), which keeps
    the struct holding the larger value -- so the edge and cell ids of the global maximum travel with it.  One
    all-gather of the three 8-byte words per rank (the value's bit pattern, so nothing is rounded); ties go to the
    lowest rank, a rank without a wet edge contributes (0, -1, -1) as ResetOperatorDiagnostics leaves it (772-784)."""
    from .operator import CourantNumberDiagnostics
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return diag
    import struct
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    bits = struct.unpack("<q", struct.pack("<d", float(diag.max_courant_num)))[0]
    mine = torch.tensor([bits, int(diag.global_edge_id), int(diag.global_cell_id)], dtype=torch.int64,
                        device=device if on_gpu else "cpu")
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    rows = torch.stack(parts).cpu().tolist()
    best = CourantNumberDiagnostics(0.0, -1, -1)
    for b, e, c in rows:                      # rank order: a later rank replaces only on a strictly larger value
        v = struct.unpack("<d", struct.pack("<q", b))[0]
        if v > best.max_courant_num:
            best = CourantNumberDiagnostics(v, e, c)
    return best


@dataclasses.dataclass
class AdaptiveTime:
    """RDyTimeAdaptiveSection (include/private/rdyconfigimpl.h): enable,
    target_courant_number, max_increase_factor."""
    target_courant_number: float = 0.5
    max_increase_factor: float = 2.0


class EulerStepper:
    def __init__(self, op, halo=None, adaptive: Optional[AdaptiveTime] = None, fused: bool = True, forcing=None):
        self.op = op
        self.forcing = forcing   # rdycore_amd.forcing.Forcing: applied at the start of every interval, as the
                                 # driver calls RDyApplyForcing before each RDyAdvance (driver/main.c time loop)
        self.fused = fused
        self._u2 = None
        self.halo = halo
        self.adaptive = adaptive
        self.time = 0.0
        self.step = 0
        self.max_courant = None      # diagnostics of the last interval, all ranks (None = not updated yet)
        self.courant = None          # the whole struct (value + global edge / cell id of the maximum)
        self._f = None

    def rhs(self, dt, u_local, f_global):
        if self.halo is not None and self.halo.world > 1:
            self.halo.rhs_overlapped(self.op, dt, u_local, f_global)
        else:
            self.op.rhs_function(dt, u_local, f_global)

    def advance(self, u_local: torch.Tensor, dt: float, interval: float) -> float:
        """RDyAdvance: integrate from self.time to self.time + interval; returns
        the dt to use next (changed only by adaptive stepping)."""
        if self._f is None or self._f.shape[0] != self.op.mesh.num_owned_cells:
            self._f = torch.empty((self.op.mesh.num_owned_cells, 3), dtype=torch.float64, device=u_local.device)
        a = self.adaptive
        if a is not None and self.max_courant is not None:
            # src/rdyadvance.c:308-330: rescaled whenever the diagnostics are valid; a zero Courant number (nothing
            # wet) makes target / max infinite there, i.e. the factor is max_increase_factor
            if self.max_courant < a.target_courant_number:
                ratio = a.target_courant_number / self.max_courant if self.max_courant > 0.0 else float("inf")
                dt *= min(ratio, a.max_increase_factor)
                dt = min(dt, interval)
            else:
                dt *= a.target_courant_number / self.max_courant
        t_end = self.time + interval
        if self.forcing is not None:
            self.forcing.apply(self.time)
        self.op.reset_diagnostics()
        cur = u_local
        if self.fused and (self._u2 is None or self._u2.shape != u_local.shape):
            self._u2 = torch.empty_like(u_local)
        while self.time < t_end * (1.0 - 1e-14):
            h = min(dt, t_end - self.time)       # TS_EXACTFINALTIME_MATCHSTEP
            if self.fused:
                nxt = self._u2 if cur is u_local else u_local
                if self.halo is not None and self.halo.world > 1:
                    self.halo.step_overlapped(self.op, h, cur, nxt)
                else:
                    self.op.euler_step(h, cur, nxt)   # resets the Courant diagnostic like every RHS (src/rdysetup.c:1136)
                cur = nxt
            else:
                self.rhs(h, u_local, self._f)
                self.op.axpy_owned(h, self._f, u_local)
            self.time += h
            self.step += 1
        if cur is not u_local:
            u_local.copy_(cur)                   # an odd number of steps ended in the second buffer
        if a is not None:
            # UpdateOperatorDiagnostics: local 16-byte copy + the MPI_Allreduce(max) of src/operator.c:879
            self.op.update_diagnostics()
            self.courant = reduce_courant(self.op.get_diagnostics(), u_local.device)
            self.max_courant = self.courant.max_courant_num
        return dt
