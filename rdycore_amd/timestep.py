"""Device-resident explicit time stepping around the RHS operator (SURVEY.md
section 8.f row 1): what PETSc's TSEULER and RDyAdvance do between RHS
evaluations, with the state kept on the GPU.

  * forward Euler: F = RHS(t, U); U += dt F   (TSStep_Euler's VecAXPY; the RHS is
    OperatorRHSFunction, src/rdysetup.c:1120-1172).  `fused=True` (default) does both in
    one pass over the mesh (rdyhip_euler_step: the update rides on the RHS kernel's
    stores, F is never written, two state arrays ping-pong); `fused=False` keeps the
    RHS + axpy pair.
  * classical Runge-Kutta (`temporal="rk4"`: the reference's `numerics.temporal: rk4` = TSRK with TSRK4,
    src/rdysetup.c:1187-1189): four RHS evaluations per step on the stage states U, U + dt/2 k1, U + dt/2 k2, U + dt k3
    -- each one OperatorRHSFunction with its ghost update and the FULL step's dt in the friction term (TSGetTimeStep,
    src/rdysetup.c:1129) -- then U += dt/6 (k1 + 2 k2 + 2 k3 + k4).  Stage vectors stay on the device.
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
    """the explicit TS of RDyAdvance: forward Euler (default) or, with temporal="rk4", classical Runge-Kutta"""

    def __init__(self, op, halo=None, adaptive: Optional[AdaptiveTime] = None, fused: bool = True, forcing=None, temporal: str = "euler"):
        if temporal not in ("euler", "rk4"):
            raise ValueError(f"temporal = {temporal!r}: the explicit path has euler and rk4 (include/private/rdyconfigimpl.h:110-115)")
        self.temporal = temporal
        self._stage = None           # rk4: the stage state (local size) and k1..k4 (owned rows)
        self.op = op
        self.forcing = forcing   # rdycore_amd.forcing.Forcing: applied at the start of every interval, as the
                                 # driver calls RDyApplyForcing before each RDyAdvance (driver/main.c time loop)
        self.fused = fused
        self._u2 = None
        self.halo = halo
        # the stepper owns the two state arrays of the fused Euler step between the steps of one advance(): the pack of each
        # exchange can ride on the previous step's kernel (rdyhip_halo_fuse_pack); advance() invalidates at its start
        if halo is not None and fused and temporal == "euler" and getattr(halo, "_halo", None) is not None:
            halo.fuse_pack(True)
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
        if self.halo is not None and getattr(self.halo, "_halo", None) is not None:
            self.halo.invalidate()          # the caller may have written u_local since the last interval
        cur = u_local
        if self.fused and (self._u2 is None or self._u2.shape != u_local.shape):
            self._u2 = torch.empty_like(u_local)
        while self.time < t_end * (1.0 - 1e-14):
            h = min(dt, t_end - self.time)       # TS_EXACTFINALTIME_MATCHSTEP
            if self.temporal == "rk4":
                self._rk4_step(h, u_local)
            elif self.fused:
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
            u_local.copy_(cur)                   # an odd number of steps ended in the second buffer (euler, fused)
        if a is not None:
            # UpdateOperatorDiagnostics: local 16-byte copy + the MPI_Allreduce(max) of src/operator.c:879
            self.op.update_diagnostics()
            self.courant = reduce_courant(self.op.get_diagnostics(), u_local.device)
            self.max_courant = self.courant.max_courant_num
        return dt

    def _rk4_step(self, h: float, u_local: torch.Tensor):
        """TSStep_RK with the TSRK4 tableau (A = [[0], [1/2], [0, 1/2], [0, 0, 1]], b = [1/6, 1/3, 1/3, 1/6]): every stage
        is one OperatorRHSFunction on the stage state; the diagnostics left behind are the last stage's, as in the reference"""
        no = self.op.mesh.num_owned_cells
        if self._stage is None or self._stage[0].shape != u_local.shape:
            self._stage = [torch.empty_like(u_local)] + [torch.empty((no, 3), dtype=torch.float64, device=u_local.device) for _ in range(4)]
        y, k = self._stage[0], self._stage[1:]
        self.rhs(h, u_local, k[0])
        for j, a in ((1, 0.5), (2, 0.5), (3, 1.0)):
            y.copy_(u_local)                       # ghost rows are refreshed by the stage's own exchange
            self.op.axpy_owned(a * h, k[j - 1], y)
            self.rhs(h, y, k[j])
        for j, b in enumerate((1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0)):
            self.op.axpy_owned(b * h, k[j], u_local)
