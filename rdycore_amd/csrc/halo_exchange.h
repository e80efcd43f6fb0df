// Ghost-cell update of a rank's local vectors and its overlap with the interior tiles, behind the C ABI
// (include/rdyhip.h, "the ghost update and its overlap").  Stands in for DMGlobalToLocalBegin/End in
// OperatorRHSFunction (src/rdysetup.c:1133-1134) and, for the second-order path, for CommunicateCellGradients
// (src/operator_fluxes_ceed.c:1058-1107).
//
// One exchange = one pack launch (every peer's cells into one contiguous buffer, a slice per peer), one
// ncclGroupStart .. ncclSend/ncclRecv per peer .. ncclGroupEnd over xGMI, one unpack launch -- all on an internal
// stream that is forked from and joined back into the caller's stream with HIP events, so the interior
// tiles (rdyhip_apply_phase(INTERIOR)) run on the caller's stream meanwhile.  Included by rdyhip_api.hip only.
#pragma once
#include <rccl/rccl.h>

struct RDyHipHalo_s {
  RDyHipOperator op = nullptr;
  ncclComm_t     comm = nullptr;
  RDyHipTransportFn transport = nullptr;
  void             *transport_ctx = nullptr;
  std::vector<int32_t> peers, send_off, recv_off;  // offsets in cells, [npeers + 1]
  DevBuf<int32_t> d_send_ids, d_recv_ids;
  DevBuf<double>  d_send, d_recv;  // [cells][max_comp]
  int32_t         max_comp = 3;
  // every ghost this rank receives is one of a run of consecutive local rows, in arrival order (peer by peer, the agreed
  // order inside a peer: rdyhip_local_cell_order): the transfer lands in the caller's array itself, no unpack launch
  int32_t         recv_base = -1;  // first row of that run, or -1: receive into d_recv and unpack
  // rdyhip_halo_fuse_pack: the Euler-step kernels store the rows of their send-flagged cells into d_send as they store
  // u_out (per-tile send lists, swe_kernels.h), so the next step's exchange needs no pack launch either
  bool            fused_pack = false;
  const double   *packed_state = nullptr;  // the state array whose send rows d_send holds ([cells][3]), or nullptr
  // The signalled form of a fused-pack Euler step (RCCL halos on devices with hipStreamWaitValue64): the launch that stores
  // the send rows tells the exchange stream when the last of them is in memory (wave_signal_send_rows, swe_kernels.h), so the
  // NEXT step's transfer runs while that launch is still busy with the tiles no other rank needs
  uint64_t        *signal = nullptr;       // signal memory (hipMallocSignalMemory): launches that have stored all their send rows
  uint64_t         packed_epoch = 0;       // the value of *signal that says packed_state's rows are in d_send
  int32_t          n_send_tiles = 0;
  DevBuf<uint32_t> d_send_done;            // [1] the running launch's count of send waves
  DevBuf<uint64_t> d_send_epoch;           // [1] the device's copy of *signal
  bool             grad_pack_fused = false;  // second order: the gradient launch over the halo cell list stores into d_send too
  DevBuf<int32_t>  d_gsend_off, d_gsend_rows;
  DevBuf<int32_t>  d_send_tile_off;        // [ntiles + 1]
  DevBuf<uint32_t> d_send_ent;             // cell-in-tile | send row << 8, sorted by tile
  bool            overlap = true;  // exchange hidden behind the interior tiles (large parts) or everything in order (small parts)
  bool            overlap_forced = false;  // RDYHIP_OVERLAP was set at create
  hipStream_t     cs = nullptr;  // exchange stream
  // fork / join events: a small ring, one pair per step, so that steps still in flight never share an event (the host
  // runs several steps ahead of the device)
  static constexpr int NEV = 8;
  hipEvent_t      ev_fork_ring[NEV] = {}, ev_join_ring[NEV] = {};
  hipEvent_t      ev_fork = nullptr, ev_join = nullptr;  // the pair of the current step
  unsigned        step = 0;
  void next_events() {
    ev_fork = ev_fork_ring[step % NEV];
    ev_join = ev_join_ring[step % NEV];
    ++step;
  }
  ~RDyHipHalo_s() {
    d_send_ids.release(); d_recv_ids.release(); d_send.release(); d_recv.release(); d_send_tile_off.release(); d_send_ent.release();
    d_send_done.release(); d_send_epoch.release(); d_gsend_off.release(); d_gsend_rows.release();
    if (signal) (void)hipFree(signal);
    for (int i = 0; i < NEV; ++i) {
      if (ev_fork_ring[i]) (void)hipEventDestroy(ev_fork_ring[i]);
      if (ev_join_ring[i]) (void)hipEventDestroy(ev_join_ring[i]);
    }
    if (cs) (void)hipStreamDestroy(cs);
  }
};

namespace {

void halo_forget_packed_state(RDyHipHalo_s *h) { h->packed_state = nullptr; }
// the operator is being destroyed before its halo: the send lists die with it
void halo_operator_gone(RDyHipHalo_s *h) {
  h->fused_pack      = false;
  h->grad_pack_fused = false;
  h->packed_state    = nullptr;
}

#define NCCL_TRY(expr)                                                                                 \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) return fail(RDYHIP_ERR_LIB, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)

// the three pieces of one ghost update of `rows` ([num_cells][ncomp]) on stream `s`: pack, transfer, unpack
int halo_check(RDyHipHalo h, const double *rows, int32_t ncomp) {
  if (ncomp < 1 || ncomp > h->max_comp) return fail(RDYHIP_ERR_ARG_SIZ, "halo exchange of %d components per cell (the halo was sized for %d)", ncomp, h->max_comp);
  if (!rows && (h->send_off.back() > 0 || h->recv_off.back() > 0)) return fail(RDYHIP_ERR_USER, "null array");
  return 0;
}
int halo_pack(RDyHipHalo h, const double *rows, int32_t ncomp, hipStream_t s) {
  const int32_t ns = h->send_off.back();
  if (ns == 0) return 0;
  const int64_t tot = (int64_t)ns * ncomp;
  h->packed_state = nullptr;  // d_send is overwritten
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ns, ncomp, rows, h->d_send_ids.p, h->d_send.p);
  HIP_TRY(hipGetLastError());
  return 0;
}
// `rows`: the array being updated ([num_cells][ncomp]); with recv_base >= 0 the ghost rows themselves are the receive buffer
int halo_transfer(RDyHipHalo h, double *rows, int32_t ncomp, hipStream_t s) {
  const int32_t np = (int32_t)h->peers.size();
  if (h->send_off[np] == 0 && h->recv_off[np] == 0) return 0;
  double *recv = h->recv_base >= 0 ? rows + (size_t)h->recv_base * ncomp : h->d_recv.p;
  if (h->transport) {
    const int rc = h->transport(h->transport_ctx, h->d_send.p, recv, ncomp, (void *)s);
    if (rc) return fail(RDYHIP_ERR_LIB, "the halo transport callback returned %d", rc);
    return 0;
  }
  if (!h->comm) return fail(RDYHIP_ERR_USER, "the halo has neither an RCCL communicator nor a transport callback");
  NCCL_TRY(ncclGroupStart());
  // a failure inside the bracket must still close it: an open group would silently queue every later RCCL call of this
  // thread (the caller's own communicators included)
  ncclResult_t r = ncclSuccess;
  const char  *what = "";
  for (int32_t i = 0; i < np && r == ncclSuccess; ++i) {
    const size_t cnt_s = (size_t)(h->send_off[i + 1] - h->send_off[i]) * ncomp, cnt_r = (size_t)(h->recv_off[i + 1] - h->recv_off[i]) * ncomp;
    if (cnt_s) {
      r    = ncclSend(h->d_send.p + (size_t)h->send_off[i] * ncomp, cnt_s, ncclDouble, h->peers[i], h->comm, s);
      what = "ncclSend";
    }
    if (cnt_r && r == ncclSuccess) {
      r    = ncclRecv(recv + (size_t)h->recv_off[i] * ncomp, cnt_r, ncclDouble, h->peers[i], h->comm, s);
      what = "ncclRecv";
    }
  }
  if (r != ncclSuccess) {
    (void)ncclGroupEnd();
    return fail(RDYHIP_ERR_LIB, "%s failed: %s", what, ncclGetErrorString(r));
  }
  NCCL_TRY(ncclGroupEnd());
  return 0;
}
int halo_unpack(RDyHipHalo h, double *rows, int32_t ncomp, hipStream_t s) {
  const int32_t nr = h->recv_off.back();
  if (nr == 0 || h->recv_base >= 0) return 0;  // received in place
  const int64_t tot = (int64_t)nr * ncomp;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, nr, ncomp, rows, h->d_recv_ids.p, h->d_recv.p);
  HIP_TRY(hipGetLastError());
  return 0;
}
int halo_exchange_on(RDyHipHalo h, double *rows, int32_t ncomp, hipStream_t s) {
  int rc = halo_check(h, rows, ncomp);
  if (!rc) rc = halo_pack(h, rows, ncomp, s);
  if (!rc) rc = halo_transfer(h, rows, ncomp, s);
  if (!rc) rc = halo_unpack(h, rows, ncomp, s);
  return rc;
}

// the exchange of the ghost-adjacent cells' gradients right behind launch_gradients(HALO) on the same stream: that launch has
// packed them itself where the fused pack is attached (ColdArgs::gsend_*)
int halo_exchange_gradients(RDyHipOperator op, RDyHipHalo h, hipStream_t s) {
  if (!(h->grad_pack_fused && op->fused_halo == h && op->muscl_fused)) return halo_exchange_on(h, op->d_grad.p, 6, s);
  int rc = halo_check(h, op->d_grad.p, 6);
  h->packed_state = nullptr;  // d_send holds gradient rows now
  if (!rc) rc = halo_transfer(h, op->d_grad.p, 6, s);
  if (!rc) rc = halo_unpack(h, op->d_grad.p, 6, s);
  return rc;
}

// the pack of the STATE at the head of a step: skipped when the previous Euler step's kernel has already stored exactly
// these rows into d_send (rdyhip_halo_fuse_pack)
int halo_pack_state(RDyHipHalo h, const double *u, hipStream_t s) {
  if (h->fused_pack && h->packed_state == u && u) return 0;
  return halo_pack(h, u, 3, s);
}
// what d_send holds once the launches of a step are enqueued: the send rows of u_out if the fused Euler kernel of a
// first-order operator has just stored them (every send cell is ghost-adjacent, i.e. in a tile of the HALO phase, which
// runs after this step's transfer has read d_send), nothing the next step could use otherwise
void halo_note_step(RDyHipOperator op, RDyHipHalo h, const double *u_out) {
  h->packed_state = (h->fused_pack && op->fused_halo == h && u_out && op->use_tiled && (!op->muscl || op->muscl_fused)) ? u_out : nullptr;
  h->packed_epoch = op->send_epoch;  // the launch just enqueued is the one that stores them
}

// OperatorRHSFunction (u_out == nullptr) or one forward-Euler step (u_out != nullptr) with the ghost update of u
// hidden behind the tiles that need no ghost data
int overlapped(RDyHipOperator op, RDyHipHalo h, double dt, double *u, double *f, double *u_out, hipStream_t st) {
  if (!op || !h) return fail(RDYHIP_ERR_USER, "null argument");
  if (h->op != op) return fail(RDYHIP_ERR_USER, "the halo belongs to another operator");
  if (op->n_cells > 0 && !u) return fail(RDYHIP_ERR_USER, "null u_local");
  if (u_out && u_out == u) return fail(RDYHIP_ERR_USER, "rdyhip_euler_step_overlapped needs a second state array (not in place)");
  {
    const int rc0 = halo_check(h, u, 3);
    if (rc0) return rc0;
  }
  op->courant = RDyHipCourant{0.0, -1, -1};
  // The halo tiles go on the exchange stream right behind the unpack, i.e. they run beside the interior tiles' tail
  // instead of in a launch of their own after the join (-10 us per step): the two launches write disjoint rows, and
  // each owns one half of the Courant buckets.  Not with the separate Euler update of the cell-centric kernel, which
  // reads all of F once the halo phase is through, nor with the split second-order form (its own schedule below).
  // RDYHIP_HALO_CONCURRENT=0: measurement knob
  const char *cenv = getenv("RDYHIP_HALO_CONCURRENT");
  const bool  conc = !(cenv && atoi(cenv) == 0) && (!op->muscl || op->muscl_fused) && (!u_out || op->use_tiled);
  auto part = [&](int32_t phase, int reset, bool ready) -> int {
    if (conc && phase == RDYHIP_PHASE_HALO) return launch_rhs(op, phase, 1, 1, dt, u, f, h->cs, ready, u_out, 2);
    if (reset && phase == RDYHIP_PHASE_HALO && (op->use_tiled ? op->n_halo_tiles == 0 : op->n_halo == 0)) {
      const int rc = rdyhip_reset_diagnostics(op, (void *)st);
      if (rc) return rc;
    }
    return launch_rhs(op, phase, 1, reset, dt, u, f, st, ready, u_out, conc && phase == RDYHIP_PHASE_INTERIOR ? 1 : 0);
  };
  int rc;
  // Fused-pack Euler steps over RCCL, where a stream can wait for a word in memory: the signalled form.
  //   exchange stream:  [wait until the launch of step n - 1 has stored its last send row]  transfer (, unpack)  -> event
  //   caller's stream:  [wait for that event]  ONE launch over all tiles, the send-flagged tiles of every XCD chunk first
  // The transfer of step n thus runs beside the rest of step n - 1's launch and the caller's stream finds its event signalled
  // (profiles/r04_wait_value_probe.txt: 3 us from the store to the waiting stream's next kernel).  Opt-in: RDYHIP_SIGNALLED=1.
  // The first step of a run (d_send does not hold u's rows yet) packs with a launch, ordered after the caller's stream.
  if (h->signal && h->fused_pack && op->fused_halo == h && u_out && !op->muscl && op->use_tiled && !h->transport && op->send_signalling) {
    h->next_events();
    if (h->packed_state == u) {
      HIP_TRY(hipStreamWaitValue64(h->cs, h->signal, h->packed_epoch, hipStreamWaitValueGte, ~0ull));
    } else {
      HIP_TRY(hipEventRecord(h->ev_fork, st));
      HIP_TRY(hipStreamWaitEvent(h->cs, h->ev_fork, 0));
    }
    rc = halo_pack_state(h, u, h->cs);  // nothing to do when d_send already holds u's rows
    if (!rc) rc = halo_transfer(h, u, 3, h->cs);
    if (!rc) rc = halo_unpack(h, u, 3, h->cs);
    // (an error leaves the exchange stream joined back all the same: later work on the caller's stream stays ordered behind it)
    hipError_t e = hipEventRecord(h->ev_join, h->cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, h->ev_join, 0);
    if (!rc && e != hipSuccess) rc = fail(RDYHIP_ERR_LIB, "joining the exchange stream failed: %s", hipGetErrorString(e));
    if (!rc) rc = launch_rhs(op, RDYHIP_PHASE_ALL, 1, 1, dt, u, f, st, false, u_out, 0, true);
    halo_note_step(op, h, rc ? nullptr : u_out);
    return rc;
  }
  // A fused-pack Euler step over RCCL is the transfer and one launch: in order that is launch + 8-10 us at every size measured
  // (0.36 - 10 M cells per rank), the two-stream form launch + 14 us or more (profiles/r04_small_parts.txt, _strip_10M) -- there is no pack
  // to hide any more, and the second launch of the ghost-adjacent tiles finds no free workgroup slot until the first one ends.
  // (A transport callback may block the host: it keeps the two-stream form, whose interior launch is enqueued first.)
  const bool fused_euler = h->fused_pack && op->fused_halo == h && u_out && !op->muscl && op->use_tiled && !h->transport;
  if (!h->overlap || (fused_euler && !h->overlap_forced)) {
    // Small parts: the tiles that need no ghost data run for less time than the exchange chain (pack, transfer, unpack, halo
    // tiles) takes, and the two cross-stream dependencies of the overlapped form cost more than they hide -- measured on
    // a 360 000-cell rank (profiles/r03_step_breakdown_360k.json): 51.8 us per overlapped step against 14.0 us for the
    // exchange plus 18.6 us for ONE launch over all tiles.  So: everything in order on the caller's stream.
    rc = halo_pack_state(h, u, st);
    if (!rc) rc = halo_transfer(h, u, 3, st);
    if (!rc) rc = halo_unpack(h, u, 3, st);
    if (!rc && op->muscl) {
      // second order: the ghost-adjacent cells' gradients (fused form) or all of them (split form), then their exchange
      rc = launch_gradients(op, op->muscl_fused ? RDYHIP_PHASE_HALO : RDYHIP_PHASE_ALL, u, st);
      if (!rc) rc = halo_exchange_gradients(op, h, st);
    }
    if (!rc) rc = launch_rhs(op, RDYHIP_PHASE_ALL, 1, 1, dt, u, f, st, op->muscl, u_out, 0);
    halo_note_step(op, h, rc ? nullptr : u_out);
    return rc;
  }
  h->next_events();
  // fork: the exchange starts once everything already enqueued on the caller's stream (the update that produced u) is done
  HIP_TRY(hipEventRecord(h->ev_fork, st));
  HIP_TRY(hipStreamWaitEvent(h->cs, h->ev_fork, 0));
  // an error after the fork still joins the exchange stream back, so that the caller's stream stays ordered behind
  // whatever was enqueued there
  auto bail = [&](int code) -> int {
    (void)hipEventRecord(h->ev_join, h->cs);
    (void)hipStreamWaitEvent(st, h->ev_join, 0);
    return code;
  };
  // the join itself; if it fails, bail() tries once more and the error is reported
  auto join = [&]() -> int {
    hipError_t e = hipEventRecord(h->ev_join, h->cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, h->ev_join, 0);
    return e == hipSuccess ? 0 : bail(fail(RDYHIP_ERR_LIB, "joining the exchange stream failed: %s", hipGetErrorString(e)));
  };
  auto fork = [&]() -> int {
    hipError_t e = hipEventRecord(h->ev_fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->cs, h->ev_fork, 0);
    return e == hipSuccess ? 0 : bail(fail(RDYHIP_ERR_LIB, "forking the exchange stream failed: %s", hipGetErrorString(e)));
  };
  if (!op->muscl) {
    // RCCL (asynchronous): the whole exchange is enqueued first, the interior tiles right behind it on the other stream.
    // A transport callback may block the host: there the interior tiles are enqueued before it is called, so that it
    // blocks while the device already works.
    rc = halo_pack_state(h, u, h->cs);
    if (!rc && h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, false);
    if (!rc) rc = halo_transfer(h, u, 3, h->cs);
    if (!rc) rc = halo_unpack(h, u, 3, h->cs);
    if (!rc && !h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, false);
    if (!rc && conc) rc = part(RDYHIP_PHASE_HALO, 0, false);
    if (rc) return bail(rc);
    rc = join();
    if (!rc && !conc) rc = part(RDYHIP_PHASE_HALO, 0, false);
    halo_note_step(op, h, rc ? nullptr : u_out);
    return rc;
  }
  // ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213) needs two exchanges: the state, then the gradients of the ghost
  // cells (CommunicateCellGradients).  No reverse exchange: every rank evaluates all edges of its owned cells.
  if (op->muscl_fused) {
    // tiles whose cells and first ring touch no ghost need nothing from other ranks and hide BOTH exchanges: the state,
    // then -- still on the exchange stream -- the gradients of the ghost-adjacent owned cells (the only ones that go
    // through memory; the interior tiles neither read nor write that array) and their exchange
    rc = halo_pack_state(h, u, h->cs);
    if (!rc && h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, true);
    if (!rc) rc = halo_transfer(h, u, 3, h->cs);
    if (!rc) rc = halo_unpack(h, u, 3, h->cs);
    if (!rc && !h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, true);
    if (!rc) rc = launch_gradients(op, RDYHIP_PHASE_HALO, u, h->cs);
    if (!rc) rc = halo_exchange_gradients(op, h, h->cs);
    if (!rc && conc) rc = part(RDYHIP_PHASE_HALO, 0, true);
    if (rc) return bail(rc);
    rc = join();
    if (!rc && !conc) rc = part(RDYHIP_PHASE_HALO, 0, true);
    // (the gradient exchange went through the send buffer; the ghost-adjacent tiles' launch, behind it, has stored the new state's rows)
    halo_note_step(op, h, rc ? nullptr : u_out);
    return rc;
  }
  // split kernels: the gradients of the cells without ghost neighbours hide the state exchange, the fluxes of the tiles
  // without ghost-adjacent cells (which read owned gradient rows only) hide the gradient exchange
  rc = halo_pack_state(h, u, h->cs);
  if (!rc && h->transport) rc = launch_gradients(op, RDYHIP_PHASE_INTERIOR, u, st);
  if (!rc) rc = halo_transfer(h, u, 3, h->cs);
  if (!rc) rc = halo_unpack(h, u, 3, h->cs);
  if (!rc && !h->transport) rc = launch_gradients(op, RDYHIP_PHASE_INTERIOR, u, st);
  if (rc) return bail(rc);
  rc = join();
  if (rc) return rc;
  rc = launch_gradients(op, RDYHIP_PHASE_HALO, u, st);
  if (rc) return bail(rc);
  h->next_events();
  rc = fork();
  if (rc) return rc;
  rc = halo_pack(h, op->d_grad.p, 6, h->cs);
  if (!rc && h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, true);
  if (!rc) rc = halo_transfer(h, op->d_grad.p, 6, h->cs);
  if (!rc) rc = halo_unpack(h, op->d_grad.p, 6, h->cs);
  if (!rc && !h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, true);
  if (rc) return bail(rc);
  rc = join();
  if (rc) return rc;
  return part(RDYHIP_PHASE_HALO, 0, true);
}

}  // namespace

extern "C" {

int rdyhip_halo_create(RDyHipOperator op, void *nccl_comm, int32_t npeers, const int32_t *peers, const int32_t *send_counts,
                       const int32_t *send_cell_ids, const int32_t *recv_counts, const int32_t *recv_cell_ids, RDyHipHalo *halo) {
  if (!op || !halo) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_halo_create");
  *halo = nullptr;
  if (npeers < 0 || (npeers > 0 && (!peers || !send_counts || !recv_counts))) return fail(RDYHIP_ERR_USER, "bad peer list");
  RDyHipHalo h = new (std::nothrow) RDyHipHalo_s;
  if (!h) return fail(RDYHIP_ERR_MEM, "out of host memory");
  h->op   = op;
  h->comm = (ncclComm_t)nccl_comm;
  h->peers.assign(peers, peers + npeers);
  h->send_off.assign((size_t)npeers + 1, 0);
  h->recv_off.assign((size_t)npeers + 1, 0);
  for (int32_t i = 0; i < npeers; ++i) {
    if (peers[i] < 0 || send_counts[i] < 0 || recv_counts[i] < 0) {
      delete h;
      return fail(RDYHIP_ERR_USER, "bad entry for peer %d", i);
    }
    h->send_off[i + 1] = h->send_off[i] + send_counts[i];
    h->recv_off[i + 1] = h->recv_off[i] + recv_counts[i];
  }
  const int32_t ns = h->send_off[npeers], nr = h->recv_off[npeers];
  if ((ns > 0 && !send_cell_ids) || (nr > 0 && !recv_cell_ids)) {
    delete h;
    return fail(RDYHIP_ERR_USER, "null cell id list");
  }
  // what is sent are owned cells, what is received are ghost cells (DMPlex's point SF roots and leaves)
  for (int32_t i = 0; i < ns; ++i)
    if (send_cell_ids[i] < 0 || send_cell_ids[i] >= op->n_cells) {
      delete h;
      return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "send cell id %d out of range", send_cell_ids[i]);
    }
  for (int32_t i = 0; i < nr; ++i)
    if (recv_cell_ids[i] < 0 || recv_cell_ids[i] >= op->n_cells) {
      delete h;
      return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "receive cell id %d out of range", recv_cell_ids[i]);
    }
  if (h->comm) {
    int nranks = 0;
    if (ncclCommCount(h->comm, &nranks) != ncclSuccess) {
      delete h;
      return fail(RDYHIP_ERR_LIB, "ncclCommCount failed on the communicator handed to rdyhip_halo_create");
    }
    for (int32_t i = 0; i < npeers; ++i)
      if (peers[i] >= nranks) {
        delete h;
        return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "peer rank %d outside the communicator's %d ranks", peers[i], nranks);
      }
  }
  h->max_comp = op->muscl ? 6 : 3;
  {
    // ghosts numbered peer by peer in arrival order (rdyhip_local_cell_order): receive in place.  RDYHIP_DIRECT_RECV=0: measurement knob
    bool run = nr > 0;
    for (int32_t i = 1; i < nr && run; ++i) run = recv_cell_ids[i] == recv_cell_ids[0] + i;
    const char *e = getenv("RDYHIP_DIRECT_RECV");
    if (run && !(e && atoi(e) == 0)) h->recv_base = recv_cell_ids[0];
  }
  {
    // Overlap only where there is something to hide behind: at least RDYHIP_OVERLAP_MIN_ROUNDS rounds of the persistent grid's
    // worth of interior tiles -- default 12: ~2.4 M cells first order (768 workgroups x 256 cells), ~3.1 M second order on
    // triangles (1 024 workgroups), ~2.4 M second order on quads (768).  What is known, all of it from ONE device with the
    // exchange looped back through a one-rank RCCL communicator:
    //   * the overlapped form costs the kernel + 16-24 us per step at every size measured (two cross-stream event dependencies,
    //     ~3x the host time): RCB parts of 1.4 M cells 81-88 us for a 59 us kernel, 2.9 M cells 127-130 for 108
    //     (profiles/r04_small_parts.txt, with RDYHIP_OVERLAP_MIN_ROUNDS=6); strips: 50 vs 17 us at 0.36 M, 83 vs 73 at 2 M,
    //     113 vs 104 at 3 M, 329 vs 318 at 10 M (profiles/r03_overlap_threshold.txt);
    //   * the in-order form costs the kernel + the exchange itself: 10-12 us looped back with the direct receive (one pack
    //     launch + RCCL), ~7.5 us with the fused pack as well.
    // So the overlapped form pays only where a real xGMI exchange takes longer than ~20 us, and either choice moves a part of
    // >= 2.4 M cells (kernel >= 90 us) by a few per cent at most.  Round 4 tried the switch at 6 rounds (the advisor's point: a
    // real exchange is longer than the loop-back's) and measured 1.4 M-cell parts 15 % slower for it; it is back at 12 until a
    // sweep on two real GPUs exists.  RDYHIP_OVERLAP=0 / 1 forces a form, RDYHIP_OVERLAP_MIN_ROUNDS moves the switch.
    const int pgrid    = std::max(8, op->muscl ? op->pgrid_muscl : op->pgrid);
    int       min_rounds = 12;
    if (const char *e = getenv("RDYHIP_OVERLAP_MIN_ROUNDS")) min_rounds = std::max(0, atoi(e));
    h->overlap = op->use_tiled ? (int64_t)(op->ntiles - op->n_halo_tiles) >= (int64_t)min_rounds * pgrid : op->n_owned >= 1500000;
    if (const char *e = getenv("RDYHIP_OVERLAP")) {
      h->overlap        = atoi(e) != 0;
      h->overlap_forced = true;
    }
  }
  int rc      = h->d_send_ids.upload(std::vector<int32_t>(send_cell_ids, send_cell_ids + ns));
  if (!rc) rc = h->d_recv_ids.upload(std::vector<int32_t>(recv_cell_ids, recv_cell_ids + nr));
  if (!rc) rc = h->d_send.zeros((size_t)ns * h->max_comp);
  if (!rc) rc = h->d_recv.zeros((size_t)nr * h->max_comp);
  if (rc) {
    delete h;
    return rc;
  }
  int lo = 0, hi = 0;  // hi = numerically lowest = highest priority
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
  // Priority of the exchange stream: the DEFAULT one (0), like the streams a caller launches the RHS on.  Measured on one
  // MI355X with the exchange looped back through a one-rank RCCL communicator (tools/step_breakdown.py, 10 M cells): at
  // priority 0 the overlapped step takes 0.334 ms (the two compute phases alone: 0.332 ms), with a high- (or low-)
  // priority exchange stream 0.49 ms and 0.25 ms of host time per step -- ordering work between streams of different
  // priorities is expensive on this runtime.  The interior launch leaves 1/32 of the workgroup slots free, which is what
  // lets the exchange's small kernels run beside it without any priority.
  int prio = 0;
  (void)hi;
  if (const char *e = getenv("RDYHIP_EXCHANGE_PRIORITY")) {  // measurement knob
    prio = atoi(e);
    fprintf(stderr, "rdyhip: stream priority range [least %d, greatest %d], exchange stream at %d\n", lo, hi, prio);
  }
  bool ok = hipStreamCreateWithPriority(&h->cs, hipStreamNonBlocking, prio) == hipSuccess;
  for (int i = 0; ok && i < RDyHipHalo_s::NEV; ++i)
    ok = hipEventCreateWithFlags(&h->ev_fork_ring[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->ev_join_ring[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    delete h;
    return fail(RDYHIP_ERR_LIB, "cannot create the exchange stream / events");
  }

  *halo = h;
  return 0;
}

// attaches (or detaches) the halo's per-tile send lists to its operator: the tile descriptors' send flag and the three
// ColdArgs fields the Euler-step kernels read them through
static int halo_attach_send_lists(RDyHipHalo h, bool on) {
  RDyHipOperator op = h->op;
  HIP_TRY(hipDeviceSynchronize());  // launches in flight read the descriptors
  std::vector<TileDesc> tiles((size_t)op->ntiles + 1);
  HIP_TRY(hipMemcpy(tiles.data(), op->d_tiles.p, tiles.size() * sizeof(TileDesc), hipMemcpyDeviceToHost));
  for (auto &t : tiles) t.cnt &= ~TILE_SEND_FLAG;
  bool signalling = false;
  if (on) {
    const int32_t ns = h->send_off.back();
    std::vector<int32_t> ids((size_t)ns);
    if (ns) HIP_TRY(hipMemcpy(ids.data(), h->d_send_ids.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
    if ((int64_t)ns >= (1 << 24)) return fail(RDYHIP_ERR_ARG_SIZ, "%d send cells do not fit the 24-bit row of a send entry", ns);
    std::vector<std::pair<int32_t, uint32_t>> ent((size_t)ns);  // (tile, cell-in-tile | row << 8)
    for (int32_t i = 0; i < ns; ++i) {
      const int32_t c = ids[i];
      const int32_t o = op->prefix ? c : (c < (int32_t)op->h_l2o.size() ? op->h_l2o[c] : -1);
      if (o < 0 || o >= op->n_owned) return fail(RDYHIP_ERR_USER, "send cell %d is not an owned cell", c);
      ent[i] = std::make_pair(o / TILE, (uint32_t)(o % TILE) | ((uint32_t)i << 8));
    }
    std::sort(ent.begin(), ent.end());
    std::vector<int32_t>  off((size_t)op->ntiles + 1, 0);
    std::vector<uint32_t> packed((size_t)ns);
    for (int32_t i = 0; i < ns; ++i) {
      off[(size_t)ent[i].first + 1]++;
      packed[i] = ent[i].second;
      tiles[(size_t)ent[i].first].cnt |= TILE_SEND_FLAG;
    }
    for (int32_t t = 0; t < op->ntiles; ++t) off[(size_t)t + 1] += off[t];
    h->d_send_tile_off.release();
    h->d_send_ent.release();
    int rc = h->d_send_tile_off.upload(off);
    if (!rc) rc = h->d_send_ent.upload(packed);
    if (rc) return rc;
    // the signalled form (overlapped(), above): RCCL halos on a device whose streams can wait for a word in memory
    // (RDYHIP_SIGNALLED=0: the forms without it, for A/B timing)
    h->n_send_tiles = 0;
    for (int32_t t = 0; t < op->ntiles; ++t) h->n_send_tiles += (tiles[(size_t)t].cnt & TILE_SEND_FLAG) ? 1 : 0;
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, op->device);
    const char *senv = getenv("RDYHIP_SIGNALLED");
    // Opt-in (RDYHIP_SIGNALLED=1): on one device, with the transfer looped back, it is within +-3 % of the in-order form at
    // 1.4 - 2.9 M cells per rank (better on the unstructured part, worse on quads) and loses below ~0.8 M, where the launch is
    // over before the chain it is meant to hide (profiles/r04_small_parts.txt); what a real xGMI hop -- a longer transfer to
    // hide -- makes of it cannot be measured on this pool.  Never under rocprofv3's counter collection (it exports
    // ROCPROF_COUNTER_COLLECTION=1 to the profiled process): its dispatch serialiser does not let the wait packet through --
    // every PMC pass of the looped-back multi-rank step hung at its first signalled step, while plain kernel tracing runs it
    // fine (profiles/RESULTS_LOG.md section 11).
    const char *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
    const bool  counters = pmc && atoi(pmc) != 0;
    signalling = can && h->comm && h->n_send_tiles > 0 && senv && atoi(senv) != 0 && !counters && !op->muscl;  // (the first-order / HR kernels signal)
    if (signalling) {
      // the early transfer writes the receive rows of an array the running launch is still storing owned rows of: they must
      // be ghost rows, which no launch writes (always so for a real partition; a synthetic pattern keeps the other forms)
      const int32_t nr = h->recv_off.back();
      std::vector<int32_t> rid((size_t)nr);
      if (nr) HIP_TRY(hipMemcpy(rid.data(), h->d_recv_ids.p, sizeof(int32_t) * (size_t)nr, hipMemcpyDeviceToHost));
      for (int32_t i = 0; i < nr && signalling; ++i) {
        const int32_t c = rid[(size_t)i];
        if (op->prefix ? c < op->n_owned : (c < (int32_t)op->h_l2o.size() && op->h_l2o[(size_t)c] >= 0)) signalling = false;
      }
    }
    if (const char *e = getenv("RDYHIP_SIGNALLED_SHRINK")) op->signalled_shrink = std::max(0, atoi(e));  // measurement knob
    if (signalling) {
      if (!h->signal) {
        HIP_TRY(hipExtMallocWithFlags((void **)&h->signal, sizeof(uint64_t), hipMallocSignalMemory));
      }
      *h->signal = 0;  // signal memory is host-visible; the device is idle (synchronised above)
      rc = h->d_send_done.zeros(1);
      if (!rc) rc = h->d_send_epoch.zeros(1);
      // the launch's tile list: the XCD chunks of the plain order (launch_rhs), inside each the send-flagged tiles first
      std::vector<int32_t> order;
      order.reserve((size_t)op->ntiles);
      const int32_t chunk = op->tiled_xcd_chunks > 0 ? op->tiled_xcd_chunks : op->ntiles;
      for (int32_t lo = 0; lo < op->ntiles; lo += chunk) {
        const int32_t hi = std::min(op->ntiles, lo + chunk);
        for (int32_t t = lo; t < hi; ++t)
          if (tiles[(size_t)t].cnt & TILE_SEND_FLAG) order.push_back(t);
        for (int32_t t = lo; t < hi; ++t)
          if (!(tiles[(size_t)t].cnt & TILE_SEND_FLAG)) order.push_back(t);
      }
      op->d_tiles_send_first.release();
      if (!rc) rc = op->d_tiles_send_first.upload(order);
      if (rc) return rc;
    }
  }
  // second order (fused form): the rows each ghost-adjacent cell's gradient travels in, by position in the halo cell list
  bool gfused = false;
  if (on && op->muscl && op->muscl_fused && op->n_halo > 0 && !(getenv("RDYHIP_GRAD_PACK_FUSED") && atoi(getenv("RDYHIP_GRAD_PACK_FUSED")) == 0)) {
    const int32_t ns = h->send_off.back();
    std::vector<int32_t> ids((size_t)ns), hl((size_t)op->n_halo), pos((size_t)op->n_owned, -1);
    if (ns) HIP_TRY(hipMemcpy(ids.data(), h->d_send_ids.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hl.data(), op->d_halo_list.p, sizeof(int32_t) * (size_t)op->n_halo, hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < op->n_halo; ++i)
      if (hl[(size_t)i] >= 0 && hl[(size_t)i] < op->n_owned) pos[(size_t)hl[(size_t)i]] = i;
    std::vector<int32_t> off((size_t)op->n_halo + 1, 0), rows((size_t)ns);
    gfused = true;
    for (int32_t i = 0; i < ns && gfused; ++i) {
      const int32_t c = ids[(size_t)i];
      const int32_t o = op->prefix ? c : (c >= 0 && c < (int32_t)op->h_l2o.size() ? op->h_l2o[(size_t)c] : -1);
      if (o < 0 || o >= op->n_owned || pos[(size_t)o] < 0) gfused = false;  // a send cell the gradient launch does not visit: keep the pack launch
      else off[(size_t)pos[(size_t)o] + 1]++;
    }
    if (gfused) {
      for (int32_t i = 0; i < op->n_halo; ++i) off[(size_t)i + 1] += off[(size_t)i];
      std::vector<int32_t> fill(off.begin(), off.end() - 1);
      for (int32_t i = 0; i < ns; ++i) {
        const int32_t c = ids[(size_t)i];
        const int32_t o = op->prefix ? c : op->h_l2o[(size_t)c];
        rows[(size_t)fill[(size_t)pos[(size_t)o]]++] = i;
      }
      h->d_gsend_off.release();
      h->d_gsend_rows.release();
      int rc = h->d_gsend_off.upload(off);
      if (!rc) rc = h->d_gsend_rows.upload(rows);
      if (rc) return rc;
    }
  }
  h->grad_pack_fused = gfused;
  ColdArgs c;
  HIP_TRY(hipMemcpy(&c, op->d_cold.p, sizeof(c), hipMemcpyDeviceToHost));
  c.gsend_off   = gfused ? h->d_gsend_off.p : nullptr;
  c.gsend_rows  = gfused ? h->d_gsend_rows.p : nullptr;
  c.gsend_buf   = gfused ? h->d_send.p : nullptr;
  c.send_off    = on ? h->d_send_tile_off.p : nullptr;
  c.send_ent    = on ? h->d_send_ent.p : nullptr;
  c.send_buf    = on ? h->d_send.p : nullptr;
  c.send_done   = signalling ? h->d_send_done.p : nullptr;
  c.send_epoch  = signalling ? h->d_send_epoch.p : nullptr;
  c.send_signal = signalling ? h->signal : nullptr;
  c.send_waves  = signalling ? (uint32_t)h->n_send_tiles * (uint32_t)(TILE / 64) : 0u;
  op->send_signalling = signalling;
  op->send_epoch      = 0;
  h->packed_epoch     = 0;
  HIP_TRY(hipMemcpy(op->d_cold.p, &c, sizeof(c), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(op->d_tiles.p, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
  op->fused_halo = on ? h : nullptr;
  return 0;
}

int rdyhip_halo_fuse_pack(RDyHipHalo halo, int32_t enable) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null halo");
  RDyHipOperator op = halo->op;
  halo->packed_state = nullptr;
  if (!enable) {
    if (halo->fused_pack && op->fused_halo == halo) {
      const int rc = halo_attach_send_lists(halo, false);
      if (rc) return rc;
    }
    halo->fused_pack = false;
    return 0;
  }
  if (halo->fused_pack) return 0;
  if (!op->use_tiled || (op->muscl && !op->muscl_fused))
    return fail(RDYHIP_ERR_USER, "the fused pack rides on the tiled Euler-step kernels (not RDYHIP_KERNEL=cell, not the split second_order form)");
  if (op->fused_halo && op->fused_halo != halo) return fail(RDYHIP_ERR_USER, "another halo of this operator already has the fused pack");
  const int rc = halo_attach_send_lists(halo, true);
  if (rc) return rc;
  halo->fused_pack = true;
  return 0;
}

int rdyhip_halo_invalidate(RDyHipHalo halo) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null halo");
  halo->packed_state = nullptr;
  return 0;
}

int32_t rdyhip_halo_direct_receive(RDyHipHalo halo) { return halo && halo->recv_base >= 0 ? 1 : 0; }
int32_t rdyhip_halo_pack_fused(RDyHipHalo halo) { return halo && halo->fused_pack ? 1 : 0; }
int32_t rdyhip_halo_signalled(RDyHipHalo halo) {
  return halo && halo->fused_pack && halo->signal && halo->op->fused_halo == halo && halo->op->send_signalling && !halo->transport ? 1 : 0;
}

int rdyhip_halo_destroy(RDyHipHalo *halo) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_halo_destroy");
  if (*halo) {
    (void)hipDeviceSynchronize();
    if ((*halo)->fused_pack && (*halo)->op->fused_halo == *halo) (void)halo_attach_send_lists(*halo, false);
    delete *halo;
    *halo = nullptr;
  }
  return 0;
}

int32_t rdyhip_halo_overlaps(RDyHipHalo halo) { return halo && halo->overlap ? 1 : 0; }

int rdyhip_halo_set_transport(RDyHipHalo halo, RDyHipTransportFn fn, void *ctx) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null halo");
  halo->transport     = fn;
  halo->transport_ctx = ctx;
  return 0;
}

int rdyhip_halo_exchange(RDyHipHalo halo, double *rows, int32_t ncomp, void *stream) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null halo");
  return halo_exchange_on(halo, rows, ncomp, (hipStream_t)stream);
}

int rdyhip_rhs_overlapped(RDyHipOperator op, RDyHipHalo halo, double dt, double *u_local, double *f_global, void *stream) {
  return overlapped(op, halo, dt, u_local, f_global, nullptr, (hipStream_t)stream);
}

int rdyhip_euler_step_overlapped(RDyHipOperator op, RDyHipHalo halo, double dt, double *u_local, double *u_local_out, double *f_global,
                                 void *stream) {
  if (op && op->n_owned > 0 && !u_local_out) return fail(RDYHIP_ERR_USER, "rdyhip_euler_step_overlapped needs a second state array (not in place)");
  return overlapped(op, halo, dt, u_local, f_global, u_local_out, (hipStream_t)stream);
}

int rdyhip_comm_unique_id(char id[RDYHIP_COMM_ID_BYTES]) {
  static_assert(RDYHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  if (!id) return fail(RDYHIP_ERR_USER, "null argument");
  ncclUniqueId uid;
  NCCL_TRY(ncclGetUniqueId(&uid));
  memcpy(id, uid.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int rdyhip_comm_init_rank(int32_t nranks, int32_t rank, const char id[RDYHIP_COMM_ID_BYTES], void **nccl_comm) {
  if (!id || !nccl_comm) return fail(RDYHIP_ERR_USER, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(RDYHIP_ERR_USER, "bad rank %d of %d", rank, nranks);
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  NCCL_TRY(ncclCommInitRank(&comm, nranks, uid, rank));
  *nccl_comm = (void *)comm;
  return 0;
}

int rdyhip_comm_destroy(void *nccl_comm) {
  if (nccl_comm) NCCL_TRY(ncclCommDestroy((ncclComm_t)nccl_comm));
  return 0;
}

int rdyhip_comm_count(void *nccl_comm, int32_t *nranks) {
  if (!nccl_comm || !nranks) return fail(RDYHIP_ERR_USER, "null argument");
  int n = 0;
  NCCL_TRY(ncclCommCount((ncclComm_t)nccl_comm, &n));
  *nranks = (int32_t)n;
  return 0;
}

int32_t rdyhip_rccl_version(void) {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? (int32_t)v : -1;
}

}  // extern "C"
