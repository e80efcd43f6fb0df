"""Ghost-cell update of the local solution vector between ranks.

Stands in for DMGlobalToLocalBegin/End in OperatorRHSFunction
(src/rdysetup.c:1133-1134): before an RHS evaluation every ghost cell of
`u_local` receives the state of the rank that owns it (3 doubles per cell,
1-cell overlap, src/rdydm.c:145-157).  First-order fluxes need no reverse
exchange (src/swe/swe_petsc.c:272-274).  The second-order path also exchanges
the cell gradients (6 doubles per cell, CommunicateCellGradients,
src/operator_fluxes_ceed.c:1058-1107); its reverse ADD exchange is avoided by
evaluating cut edges on both ranks (csrc/muscl_kernels.h).

One process per GPU.  The pattern itself (who needs which cells) is planned by rdyhip_halo_plan_* behind the C ABI for BOTH
drivers below, so HaloExchange needs librdyhip.so even for the pure torch / gloo transport (the plan functions touch no
device: the CPU tests use them as they are).  Two drivers of the same exchange pattern:

  * transport="c" (what a C host gets, and the default of bench.py on RCCL): the whole overlapped step is ONE call into
    librdyhip.so -- rdyhip_rhs_overlapped / rdyhip_euler_step_overlapped (csrc/halo_exchange.h): pack, ncclSend / ncclRecv in
    one group over xGMI and unpack on the library's own stream, HIP events for the fork / join, interior
    tiles meanwhile, ghost-adjacent tiles after.  This module then only discovers the pattern (who needs which cells)
    and owns the RCCL communicator.  Under the "gloo" backend (several ranks rehearsed on ONE GPU, where RCCL cannot
    run) the bytes travel through the ABI's transport callback, host-staged -- the C orchestration is the same.
  * transport="torch": torch.distributed point-to-point driven from Python (backend "nccl" = RCCL, or "gloo" in the CPU
    tests), with the pack / unpack kernels and the phased applies called one by one; on CPU tensors (gloo tests of the
    host logic) plain indexing is used.  The pattern is neighbour point-to-point, not a collective:
per RHS each rank sends ~perimeter x 24 B to each neighbour, which is
latency-bound, so it is issued on a side stream and overlapped with the
interior cells (see `rhs_overlapped`).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .mesh import RDyMesh


class HaloExchange:
    def __init__(self, mesh: RDyMesh, device: torch.device, group=None, transport: str = "torch", op=None):
        if transport not in ("torch", "c"):
            raise ValueError(transport)
        if transport == "c" and op is None:
            raise ValueError('transport="c" needs the operator the halo belongs to')
        self.transport = transport
        self._halo = None          # RDyHipHalo
        self._comm = None          # ncclComm_t owned by this object
        self._cb = None
        self.mesh = mesh
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.send_ids: Dict[int, torch.Tensor] = {}   # peer -> local ids of owned cells to send
        self.recv_ids: Dict[int, torch.Tensor] = {}   # peer -> local ids of ghost cells to fill
        self.send_buf: Dict[int, torch.Tensor] = {}
        self.recv_buf: Dict[int, torch.Tensor] = {}
        self.timings: Dict[str, float] = {}          # wall seconds per setup stage (bench.py reports them per run)
        import time as _time
        t0 = _time.perf_counter()
        self._setup()
        self.timings["plan_s"] = _time.perf_counter() - t0
        # default priority, like the compute stream: ordering work between streams of different priorities costs 0.25 ms of
        # host time per step on this runtime (csrc/halo_exchange.h)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        if transport == "c" and self.world > 1:
            t0 = _time.perf_counter()
            self._create_c_halo(op)
            self.timings["c_halo_total_s"] = _time.perf_counter() - t0

    # -- the exchange behind the C ABI (include/rdyhip.h: rdyhip_halo_create) ------------------------------
    def _all_ok(self, ok: bool) -> bool:
        """collective: did every rank get through the last stage?"""
        on_dev = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _create_c_halo(self, op):
        """Every stage that can fail on one rank alone (drawing the id, ncclCommInitRank, rdyhip_halo_create) is followed by
        an agreement (all-reduce MIN of an ok flag) BEFORE any rank raises, so that the ranks' collectives always match: a
        failure anywhere becomes the same exception on every rank (bench.py then falls back to torch P2P on all of them),
        never a hang.  Whatever was built is torn down before raising."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        peers = self._plan_peers
        send_counts, recv_counts = i32(self._plan_send_counts), i32(self._plan_recv_counts)
        send_ids, recv_ids, peers_a = i32(self._plan_send_cells), i32(self._plan_recv_cells), i32(peers)
        comm = C.c_void_p()
        err = None
        if dist.get_backend(self.group) == "nccl":
            # an RCCL communicator of our own over the same ranks: rank 0 draws the id, everybody joins
            box = [None]
            if self.rank == 0:
                try:
                    buf = C.create_string_buffer(128)
                    _lib.check(lib.rdyhip_comm_unique_id(buf))
                    box[0] = buf.raw
                except Exception as exc:        # the broadcast still happens: a None tells everybody
                    err = exc
            dist.broadcast_object_list(box, src=0, group=self.group)
            if box[0] is None:
                raise RuntimeError(f"rank 0 could not draw an RCCL unique id: {err!r}")
            import time as _time
            t0 = _time.perf_counter()
            try:
                _lib.check(lib.rdyhip_comm_init_rank(self.world, self.rank, box[0], C.byref(comm)))
                self._comm = comm
            except Exception as exc:
                err = exc
            self.timings["rccl_comm_init_s"] = _time.perf_counter() - t0
            if not self._all_ok(err is None):
                self.destroy()
                raise RuntimeError(f"ncclCommInitRank failed on some rank (this rank: {err!r})")
        h = C.c_void_p()
        p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        try:
            _lib.check(lib.rdyhip_halo_create(op._h, comm, len(peers), p(peers_a), p(send_counts), p(send_ids), p(recv_counts), p(recv_ids),
                                              C.byref(h)))
            self._halo = h
            self._c_peers, self._c_send_counts, self._c_recv_counts = list(peers), send_counts, recv_counts
            if self._comm is None:
                # no RCCL between ranks that share a device: the bytes go through the transport callback, staged on the host
                self._cb = _lib.TRANSPORT_FN(self._host_staged_transport)
                _lib.check(lib.rdyhip_halo_set_transport(h, C.cast(self._cb, C.c_void_p), None))
        except Exception as exc:
            err = exc
        if not self._all_ok(err is None):
            self.destroy()
            raise RuntimeError(f"rdyhip_halo_create failed on some rank (this rank: {err!r})")

    # -- the two shortcuts of the exchange chain (include/rdyhip.h: direct receive, fused pack) -----------------
    @property
    def direct_receive(self) -> bool:
        """the transfer lands in the local array's ghost rows themselves (ghosts numbered peer by peer in arrival order,
        mesh.extract_local_mesh / rdyhip_local_cell_order): no unpack launch"""
        from . import _lib
        return self._halo is not None and bool(_lib.load().rdyhip_halo_direct_receive(self._halo))

    def fuse_pack(self, enable: bool = True) -> bool:
        """rdyhip_halo_fuse_pack: the Euler-step kernels store their send cells' new state into the send buffer, so the next
        `step_overlapped` on that array starts with the transfer.  The caller must not write the owned rows of the array
        between two steps itself, or call `invalidate()`.  Returns whether the fused pack is on (first-order tiled kernels)."""
        from . import _lib
        if self._halo is None:
            return False
        lib = _lib.load()
        if enable:
            rc = lib.rdyhip_halo_fuse_pack(self._halo, 1)
            if rc:                              # second order / cell-centric kernel: not available, the pack launch stays
                return False
        else:
            _lib.check(lib.rdyhip_halo_fuse_pack(self._halo, 0))
        return bool(lib.rdyhip_halo_pack_fused(self._halo))

    FORM_SOURCES = ("trial_running", "measured", "forced", "default", "locked_in_order")

    def form_info(self, kind: str = "rhs") -> dict:
        """rdyhip_halo_form_info: which form the steps of this halo take (in order on the caller's stream / two streams) and why:
        the library times both forms over the first 16 calls of a kind ("rhs": rhs_overlapped, "euler": step_overlapped) on the
        communicator it really has and keeps the faster one"""
        import ctypes as C
        from . import _lib
        if self._halo is None:
            return {"form": "none", "source": "no_halo"}
        info = _lib.RDyHipHaloFormInfo()
        _lib.check(_lib.load().rdyhip_halo_form_info(self._halo, 1 if kind == "euler" else 0, C.byref(info)))
        return {"form": "two_streams" if info.form else "in_order", "source": self.FORM_SOURCES[info.source], "trial_steps": int(info.trial_steps),
                "in_order_ms": float(info.in_order_ms), "two_stream_ms": float(info.two_stream_ms)}

    def set_form(self, kind: str, form) -> None:
        """rdyhip_halo_set_form: force a form ("in_order" / "two_streams") or run the trial again (None)"""
        from . import _lib
        if self._halo is not None:
            f = -1 if form is None else (1 if form in (1, True, "two_streams") else 0)
            _lib.check(_lib.load().rdyhip_halo_set_form(self._halo, 1 if kind == "euler" else 0, f))

    def invalidate(self):
        """the state array was written by somebody else since the last step: the next step packs again"""
        from . import _lib
        if self._halo is not None:
            _lib.check(_lib.load().rdyhip_halo_invalidate(self._halo))

    def rccl_ranks(self) -> Optional[int]:
        """ncclCommCount of the library's own communicator (None when the bytes do not travel over RCCL)"""
        if self._comm is None:
            return None
        import ctypes as C
        from . import _lib
        n = C.c_int32(0)
        _lib.check(_lib.load().rdyhip_comm_count(self._comm, C.byref(n)))
        return int(n.value)

    def _host_staged_transport(self, ctx, d_send, d_recv, ncomp, stream):
        """RDyHipTransportFn: d_send's per-peer slices -> the peers' d_recv slices through the process group (gloo)"""
        try:
            from .operator import _DeviceArray
            ns, nr = int(self._c_send_counts.sum()), int(self._c_recv_counts.sum())
            st = torch.cuda.ExternalStream(stream, device=self.device) if stream else torch.cuda.default_stream(self.device)
            st.synchronize()                                   # the pack launch has filled d_send
            send_h = torch.as_tensor(_DeviceArray(d_send, (ns * ncomp,), self), device=self.device).cpu() if ns else torch.zeros(0, dtype=torch.float64)
            recv_h = torch.empty(nr * ncomp, dtype=torch.float64)
            ops, so, ro = [], 0, 0
            for peer, cs, cr in zip(self._c_peers, self._c_send_counts, self._c_recv_counts):
                if cs:
                    ops.append(dist.P2POp(dist.isend, send_h[so * ncomp:(so + int(cs)) * ncomp], peer, group=self.group))
                    so += int(cs)
                if cr:
                    ops.append(dist.P2POp(dist.irecv, recv_h[ro * ncomp:(ro + int(cr)) * ncomp], peer, group=self.group))
                    ro += int(cr)
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if nr:
                with torch.cuda.stream(st):
                    torch.as_tensor(_DeviceArray(d_recv, (nr * ncomp,), self), device=self.device).copy_(recv_h)
            return 0
        except Exception:          # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    def destroy(self):
        """frees the C halo and the RCCL communicator (idempotent)"""
        import ctypes as C
        from . import _lib
        if self._halo is not None:
            _lib.check(_lib.load().rdyhip_halo_destroy(C.byref(self._halo)))
            self._halo = None
        if self._comm is not None:
            _lib.check(_lib.load().rdyhip_comm_destroy(self._comm))
            self._comm = None

    # -- pattern discovery (setup only): rdyhip_halo_plan_* behind the ABI (csrc/halo_plan.h) ---------------
    def _owners_by_query(self, ghost_gid: np.ndarray) -> np.ndarray:
        """Owner rank of every ghost cell when the mesh does not carry it (`RDyMesh.cell_owner_rank`): every rank publishes the
        global ids it needs, every rank answers which of them it owns.  O(world x ghosts) -- the fallback for meshes cut
        without a part array at hand; a DMPlex host reads the owners from the point SF instead (adapter/rdyhip_petsc.c)."""
        m = self.mesh
        owned_gid = np.sort(m.cell_global_ids[m.cell_owned_to_local])
        wanted: List[Optional[np.ndarray]] = [None] * self.world
        dist.all_gather_object(wanted, ghost_gid, group=self.group)
        mine = []
        for peer in range(self.world):
            req = np.asarray(wanted[peer], dtype=np.int64)
            if peer == self.rank or req.size == 0 or owned_gid.size == 0:
                mine.append(np.zeros(req.size, dtype=bool))
                continue
            idx = np.minimum(np.searchsorted(owned_gid, req), owned_gid.size - 1)
            mine.append(owned_gid[idx] == req)
        answers: List[Optional[List[np.ndarray]]] = [None] * self.world
        dist.all_gather_object(answers, mine, group=self.group)
        owner = np.full(ghost_gid.size, -1, dtype=np.int32)
        for peer in range(self.world):
            if peer != self.rank:
                hit = np.asarray(answers[peer][self.rank], dtype=bool)
                if np.any(owner[hit] >= 0):
                    raise RuntimeError(f"rank {self.rank}: a ghost cell has two owners")
                owner[hit] = peer
        if np.any(owner < 0):
            raise RuntimeError(f"rank {self.rank}: {int((owner < 0).sum())} ghost cells have no owner")
        return owner

    def _alltoall_requests(self, counts: np.ndarray, keys: np.ndarray):
        """the one exchange the plan needs (MPI_Alltoall + MPI_Alltoallv in an MPI host): request counts to everybody, the
        request keys point-to-point to the ranks actually asked"""
        on_dev = dist.get_backend(self.group) == "nccl"
        dev = self.device if on_dev else torch.device("cpu")
        mine = torch.as_tensor(counts.astype(np.int64), device=dev)
        table = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(table, mine, group=self.group)
        table = torch.stack(table).cpu().numpy()               # table[r][q] = cells rank r asks of rank q
        incoming_counts = table[:, self.rank].astype(np.int32)
        send = torch.as_tensor(keys.astype(np.int64), device=dev)
        recv = torch.empty(int(incoming_counts.sum()), dtype=torch.int64, device=dev)
        ops, so, ro = [], 0, 0
        for peer in range(self.world):
            cs, cr = int(counts[peer]), int(incoming_counts[peer])
            if cs:
                ops.append(dist.P2POp(dist.isend, send[so:so + cs], peer, group=self.group))
            if cr:
                ops.append(dist.P2POp(dist.irecv, recv[ro:ro + cr], peer, group=self.group))
            so, ro = so + cs, ro + cr
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if on_dev:
            torch.cuda.synchronize(self.device)
        return incoming_counts, recv.cpu().numpy()

    def _setup(self):
        import ctypes as C
        from . import _lib
        m = self.mesh
        ghost_local = np.nonzero(m.cell_is_owned == 0)[0].astype(np.int32)
        ghost_gid = np.ascontiguousarray(m.cell_global_ids[ghost_local], dtype=np.int64)
        self._plan_peers, self._plan_send_counts, self._plan_recv_counts = [], [], []
        self._plan_send_cells = self._plan_recv_cells = np.zeros(0, dtype=np.int32)
        if self.world == 1:
            if ghost_local.size:
                raise ValueError("mesh has ghost cells but there is only one rank")
            return
        owner_all = getattr(m, "cell_owner_rank", None)
        have = torch.tensor([1 if owner_all is not None else 0], dtype=torch.int32,
                            device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=self.group)      # the fallback is collective: all ranks or none
        if int(have.item()):
            owner = np.ascontiguousarray(np.asarray(owner_all)[ghost_local], dtype=np.int32)
        else:
            owner = self._owners_by_query(ghost_gid)
        lib = _lib.load()
        plan = C.c_void_p()
        pi = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        pl = lambda a: a.ctypes.data_as(_lib.c_int64_p)
        _lib.check(lib.rdyhip_halo_plan_create(self.world, self.rank, int(ghost_local.size), pi(ghost_local), pi(owner), pl(ghost_gid), C.byref(plan)))
        try:
            cptr, kptr = _lib.c_int32_p(), _lib.c_int64_p()
            _lib.check(lib.rdyhip_halo_plan_requests(plan, C.byref(cptr), C.byref(kptr)))
            counts = np.ctypeslib.as_array(cptr, shape=(self.world,)).copy()
            keys = np.ctypeslib.as_array(kptr, shape=(max(int(ghost_local.size), 1),))[:ghost_local.size].copy()
            incoming_counts, incoming_keys = self._alltoall_requests(counts, keys)
            incoming_counts = np.ascontiguousarray(incoming_counts, dtype=np.int32)
            incoming_keys = np.ascontiguousarray(incoming_keys, dtype=np.int64)
            is_owned = np.ascontiguousarray(m.cell_is_owned, dtype=np.int32)
            gids = np.ascontiguousarray(m.cell_global_ids, dtype=np.int64)
            _lib.check(lib.rdyhip_halo_plan_finish(plan, pi(incoming_counts), pl(incoming_keys), int(m.num_cells), pi(is_owned), pl(gids)))
            npeers = C.c_int32(0)
            ptrs = [_lib.c_int32_p() for _ in range(5)]
            _lib.check(lib.rdyhip_halo_plan_get(plan, C.byref(npeers), *[C.byref(q) for q in ptrs]))
            n = int(npeers.value)
            take = lambda q, k: np.ctypeslib.as_array(q, shape=(max(k, 1),))[:k].copy()
            peers, sc, rc = take(ptrs[0], n), take(ptrs[1], n), take(ptrs[3], n)
            send_cells, recv_cells = take(ptrs[2], int(sc.sum())), take(ptrs[4], int(rc.sum()))
        finally:
            _lib.check(lib.rdyhip_halo_plan_destroy(C.byref(plan)))
        self._plan_peers, self._plan_send_counts, self._plan_recv_counts = [int(q) for q in peers], sc, rc
        self._plan_send_cells, self._plan_recv_cells = send_cells, recv_cells
        so = ro = 0
        for q, cs, cr in zip(self._plan_peers, sc, rc):
            if cs:
                self.send_ids[q] = torch.as_tensor(send_cells[so:so + int(cs)], device=self.device)
            if cr:
                self.recv_ids[q] = torch.as_tensor(recv_cells[ro:ro + int(cr)], device=self.device)
            so, ro = so + int(cs), ro + int(cr)
        peers_s, peers_r = sorted(self.send_ids), sorted(self.recv_ids)
        self.send_ids_all = torch.cat([self.send_ids[p] for p in peers_s]) if peers_s else torch.zeros(0, dtype=torch.int32, device=self.device)
        self.recv_ids_all = torch.cat([self.recv_ids[p] for p in peers_r]) if peers_r else torch.zeros(0, dtype=torch.int32, device=self.device)
        self._bufs = {}
        self._buffers(3)
        self.send_all, self.recv_all, self.send_buf, self.recv_buf = self._bufs[3][:4]

    def _buffers(self, ncomp: int):
        """one contiguous send and one contiguous receive buffer per row width (a slice per peer), so that a
        ghost update costs one pack and one unpack launch whatever the number of neighbours"""
        if ncomp not in self._bufs:
            peers_s, peers_r = sorted(self.send_ids), sorted(self.recv_ids)
            ns = sum(int(self.send_ids[p].numel()) for p in peers_s)
            nr = sum(int(self.recv_ids[p].numel()) for p in peers_r)
            send_all = torch.empty((ns, ncomp), dtype=torch.float64, device=self.device)
            recv_all = torch.empty((nr, ncomp), dtype=torch.float64, device=self.device)
            send_buf, recv_buf = {}, {}
            o = 0
            for p in peers_s:
                n = int(self.send_ids[p].numel())
                send_buf[p] = send_all[o:o + n]
                o += n
            o = 0
            for p in peers_r:
                n = int(self.recv_ids[p].numel())
                recv_buf[p] = recv_all[o:o + n]
                o += n
            self._bufs[ncomp] = [send_all, recv_all, send_buf, recv_buf, None]   # last: the cached P2P op list
        return self._bufs[ncomp]

    @property
    def bytes_sent_per_exchange(self) -> int:
        return int(self.send_all.numel()) * 8 if self.world > 1 else 0

    # -- one ghost update on the current stream ----------------------------
    def exchange(self, u_local: torch.Tensor):
        """ghost rows of a [num_cells, ncomp] array (the solution: ncomp = 3; the second-order
        gradients: ncomp = 6, CommunicateCellGradients) <- the owners' rows"""
        if self.world == 1 or (not self.send_ids and not self.recv_ids):
            return
        if self._halo is not None and u_local.is_cuda:
            from . import _lib
            ncomp = int(u_local.shape[-1]) if u_local.dim() == 2 else 3
            _lib.check(_lib.load().rdyhip_halo_exchange(self._halo, int(u_local.data_ptr()), ncomp,
                                                        int(torch.cuda.current_stream(self.device).cuda_stream)))
            return
        cuda = u_local.is_cuda
        ncomp = int(u_local.shape[-1]) if u_local.dim() == 2 else 3
        rows = u_local.view(-1, ncomp)
        bufs = self._buffers(ncomp)
        send_all, recv_all, send_buf, recv_buf = bufs[:4]
        # device buffers go straight to RCCL; under gloo (single-GPU rehearsal of
        # several ranks, CPU tests) device buffers are staged through the host
        via_host = cuda and dist.get_backend(self.group) == "gloo"
        if cuda:
            from .operator import pack_rows
            if self.send_ids_all.numel():
                pack_rows(rows, self.send_ids_all, send_all)
        elif self.send_ids_all.numel():
            send_all.copy_(rows[self.send_ids_all.long()])
        if via_host:
            wire_send = {p: send_buf[p].cpu() for p in self.send_ids}
            wire_recv = {p: torch.empty_like(recv_buf[p], device="cpu") for p in self.recv_ids}
            ops = self._make_ops(wire_send, wire_recv)
        else:
            if bufs[4] is None:           # the buffers are persistent: build the op list once
                bufs[4] = self._make_ops(send_buf, recv_buf)
            ops = bufs[4]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if via_host:
            for p in self.recv_ids:
                recv_buf[p].copy_(wire_recv[p])
        if cuda:
            from .operator import unpack_rows
            if self.recv_ids_all.numel():
                unpack_rows(rows, self.recv_ids_all, recv_all)
        elif self.recv_ids_all.numel():
            rows[self.recv_ids_all.long()] = recv_all

    def _fused(self, op) -> bool:
        if not hasattr(op, "_second_order_fused"):
            op._second_order_fused = bool(op.layout_info()["second_order_fused"])
        return op._second_order_fused

    def _make_ops(self, send, recv):
        ops = []
        for peer in sorted(set(self.send_ids) | set(self.recv_ids)):
            if peer in self.send_ids:
                ops.append(dist.P2POp(dist.isend, send[peer], peer, group=self.group))
            if peer in self.recv_ids:
                ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group=self.group))
        return ops

    # -- RHS (or a whole Euler step) with the exchange hidden behind the interior cells
    def rhs_overlapped(self, op, dt: float, u_local: torch.Tensor, f_global: torch.Tensor):
        """OperatorRHSFunction (src/rdysetup.c:1120-1172) on one rank: ghost
        update on a side stream while the cells without ghost neighbours are
        evaluated, then the remaining (halo-adjacent) cells."""
        self._overlapped(op, dt, u_local, f_global, None)

    def step_overlapped(self, op, dt: float, u_local: torch.Tensor, u_out: torch.Tensor):
        """u_out[owned] = u_local[owned] + dt RHS(u_local) (Operator.euler_step) with the ghost update of u_local
        overlapped the same way; the ghost rows of u_out are filled by the next call's exchange."""
        self._overlapped(op, dt, u_local, None, u_out)

    def _overlapped(self, op, dt, u_local, f_global, u_out):
        def part(phase, reset=False, ready=False):
            if u_out is not None:
                op.euler_step(dt, u_local, u_out, None, phase=phase, reset_diagnostics=reset, gradients_ready=ready)
            else:
                op.apply_phase(phase, True, dt, u_local, f_global, reset_diagnostics=reset, gradients_ready=ready)

        if self.world == 1:
            part(0, reset=True)
            return
        if self._halo is not None:
            from . import _lib
            lib = _lib.load()
            st = int(torch.cuda.current_stream(self.device).cuda_stream)
            if u_out is not None:
                _lib.check(lib.rdyhip_euler_step_overlapped(op._h, self._halo, float(dt), int(u_local.data_ptr()), int(u_out.data_ptr()), None, st))
            else:
                _lib.check(lib.rdyhip_rhs_overlapped(op._h, self._halo, float(dt), int(u_local.data_ptr()), int(f_global.data_ptr()), st))
            return
        main = torch.cuda.current_stream(self.device)
        self.comm_stream.wait_stream(main)           # u_local's owned part is final
        with torch.cuda.stream(self.comm_stream):
            self.exchange(u_local)
        if op.config.second_order:
            # ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213) needs two exchanges: the state, then the
            # gradients (CommunicateCellGradients).  There is no reverse exchange: every rank evaluates all
            # edges of its cells.
            if self._fused(op):
                # fused kernel: tiles whose cells and first ring touch no ghost need nothing from other ranks and
                # hide both exchanges; only the ghost-adjacent cells' gradients go through memory, on the exchange
                # stream (the interior tiles neither read nor write that array) -- as csrc/halo_exchange.h does
                part(1, reset=True, ready=True)
                with torch.cuda.stream(self.comm_stream):
                    op.compute_gradients(u_local, phase=2)
                    self.exchange(op.gradients)
                main.wait_stream(self.comm_stream)
                part(2, ready=True)
                return
            # split kernels.  Hidden behind the exchanges: the gradients of the cells without ghost neighbours,
            # then the fluxes of the tiles without ghost-adjacent cells (which read owned gradient rows only).
            op.compute_gradients(u_local, phase=1)
            main.wait_stream(self.comm_stream)
            op.compute_gradients(u_local, phase=2)
            grads = op.gradients
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                self.exchange(grads)
            part(1, reset=True, ready=True)
            main.wait_stream(self.comm_stream)
            part(2, ready=True)
            return
        part(1, reset=True)                          # RDYHIP_PHASE_INTERIOR (+ diagnostics reset)
        main.wait_stream(self.comm_stream)
        part(2)                                      # RDYHIP_PHASE_HALO
