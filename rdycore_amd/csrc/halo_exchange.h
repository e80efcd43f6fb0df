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
  bool             grad_pack_fused = false;  // second order: the gradient launch over the halo cell list stores into d_send too
  DevBuf<int32_t>  d_gsend_off, d_gsend_rows;
  DevBuf<int32_t>  d_send_tile_off;        // [ntiles + 1]
  DevBuf<uint32_t> d_send_ent;             // cell-in-tile | send row << 8, sorted by tile
  // every send cell sits in a tile of the HALO phase (true for the edge-adjacent overlap; a vertex-adjacent overlap --
  // DMPlexDistributeOverlap, src/rdydm.c:150 -- can put one in a tile no ghost touches): only then may a two-stream step
  // let its launches store send rows while the exchange stream reads the send buffer
  bool            send_cells_in_halo_tiles = true;
  // The form of a step, per kind of step (0: rdyhip_rhs_overlapped, 1: rdyhip_euler_step_overlapped): everything in order
  // on the caller's stream, or the exchange on the library's stream beside the tiles that need no ghost data.  Both forms post
  // the same send / receive group, so every rank chooses for itself: the first 2 x TRIAL steps of a kind alternate between
  // the forms, each timed on the device (events on the caller's stream), and the faster one stays -- on the communicator
  // the run really has, not on a constant tuned elsewhere.  RDYHIP_OVERLAP=0 / 1 or RDYHIP_OVERLAP_MIN_ROUNDS (the size rule
  // of rounds 3-4), read once at create, force a form.
  struct FormChoice {
    static constexpr int TRIAL = 8, SKIP = 2;   // steps per form in the trial; the first SKIP of each are not counted
    int        form = 0;                         // 0: in order, 1: two streams
    int        source = RDYHIP_HALO_FORM_TRIAL_RUNNING;
    int        steps = 0;                        // trial steps taken
    double     ms[2] = {0.0, 0.0};
    int        n[2]  = {0, 0};
    hipEvent_t ev0[2 * TRIAL] = {}, ev1[2 * TRIAL] = {};
  } choice[2];
  bool            halo_concurrent = true;   // RDYHIP_HALO_CONCURRENT (measurement knob), read at create
  bool            grad_pack_allowed = true; // RDYHIP_GRAD_PACK_FUSED (measurement knob), read at create
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
    d_gsend_off.release(); d_gsend_rows.release();
    for (auto &c : choice)
      for (int i = 0; i < 2 * FormChoice::TRIAL; ++i) {
        if (c.ev0[i]) (void)hipEventDestroy(c.ev0[i]);
        if (c.ev1[i]) (void)hipEventDestroy(c.ev1[i]);
      }
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
  if (!(h->grad_pack_fused && op->fused_halo == h)) return halo_exchange_on(h, op->d_grad.p, 6, s);
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
// what d_send holds once the launches of a step are enqueued: the send rows of u_out if the fused Euler kernels have just
// stored them, nothing the next step could use otherwise
void halo_note_step(RDyHipOperator op, RDyHipHalo h, const double *u_out) {
  h->packed_state = (h->fused_pack && op->fused_halo == h && u_out && op->use_tiled) ? u_out : nullptr;
}

// ---- the two forms of a step ----------------------------------------------------------------------------------------------
// In order: everything on the caller's stream -- exchange, (second order: the ghost-adjacent cells' gradients and their
// exchange,) ONE launch over all tiles.  What a small part wants: its tiles without ghost data run for less time than the
// exchange chain takes, and two cross-stream dependencies cost more than they hide (a 360 000-cell rank, one device, the
// exchange looped back: 30 us against 50, profiles/r03_step_breakdown_360k.json).
int step_in_order(RDyHipOperator op, RDyHipHalo h, double dt, double *u, double *f, double *u_out, hipStream_t st) {
  int rc = halo_pack_state(h, u, st);
  if (!rc) rc = halo_transfer(h, u, 3, st);
  if (!rc) rc = halo_unpack(h, u, 3, st);
  if (!rc && op->muscl) {
    rc = launch_gradients(op, RDYHIP_PHASE_HALO, u, st);
    if (!rc) rc = halo_exchange_gradients(op, h, st);
  }
  if (!rc) rc = launch_rhs(op, RDYHIP_PHASE_ALL, 1, 1, dt, u, f, st, op->muscl, u_out, 0);
  halo_note_step(op, h, rc ? nullptr : u_out);
  return rc;
}

// Two streams: the exchange on the library's stream, forked from and joined back into the caller's with events, the tiles
// that need no ghost data on the caller's stream meanwhile; the ghost-adjacent tiles follow the unpack on the library's
// stream, beside the tail of the others (the two launches write disjoint rows and own one half of the Courant buckets
// each).  What a large part wants when the transfer is long.
int step_two_streams(RDyHipOperator op, RDyHipHalo h, double dt, double *u, double *f, double *u_out, hipStream_t st) {
  const bool conc = h->halo_concurrent && (!u_out || op->use_tiled);  // (the cell kernel's separate Euler update reads all of F)
  auto part = [&](int32_t phase, int reset, bool ready) -> int {
    if (conc && phase == RDYHIP_PHASE_HALO) return launch_rhs(op, phase, 1, 1, dt, u, f, h->cs, ready, u_out, 2);
    if (reset && phase == RDYHIP_PHASE_HALO && (op->use_tiled ? op->n_halo_tiles == 0 : op->n_halo == 0)) {
      const int rc = rdyhip_reset_diagnostics(op, (void *)st);
      if (rc) return rc;
    }
    return launch_rhs(op, phase, 1, reset, dt, u, f, st, ready, u_out, conc && phase == RDYHIP_PHASE_INTERIOR ? 1 : 0);
  };
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
  auto join = [&]() -> int {
    hipError_t e = hipEventRecord(h->ev_join, h->cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, h->ev_join, 0);
    return e == hipSuccess ? 0 : bail(fail(RDYHIP_ERR_LIB, "joining the exchange stream failed: %s", hipGetErrorString(e)));
  };
  // A launch of this form that stores send rows (the fused pack) must not run beside the transfer that reads the send
  // buffer: only the HALO launch, behind the transfer on the same stream, may.  With a send cell in a tile of the INTERIOR
  // launch (vertex-adjacent overlaps) rdyhip_halo_fuse_pack has locked the Euler step to the in-order form, so this form
  // only ever sees fused packs whose send cells all sit in HALO tiles.
  const bool ready = op->muscl;
  // RCCL (asynchronous): the whole exchange is enqueued first, the interior tiles right behind it on the other stream.
  // A transport callback may block the host: there the interior tiles are enqueued before it is called, so that it
  // blocks while the device already works.
  int rc = halo_pack_state(h, u, h->cs);
  if (!rc && h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, ready);
  if (!rc) rc = halo_transfer(h, u, 3, h->cs);
  if (!rc) rc = halo_unpack(h, u, 3, h->cs);
  if (!rc && !h->transport) rc = part(RDYHIP_PHASE_INTERIOR, 1, ready);
  if (!rc && op->muscl) {
    // ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213) needs two exchanges: the state, then the gradients of the ghost cells
    // (CommunicateCellGradients).  Tiles whose cells and first ring touch no ghost need nothing from other ranks and hide
    // BOTH: still on the exchange stream, the gradients of the ghost-adjacent owned cells (the only ones that go through
    // memory; the interior tiles neither read nor write that array) and their exchange.  No reverse exchange: every rank
    // evaluates all edges of its owned cells.
    rc = launch_gradients(op, RDYHIP_PHASE_HALO, u, h->cs);
    if (!rc) rc = halo_exchange_gradients(op, h, h->cs);
  }
  if (!rc && conc) rc = part(RDYHIP_PHASE_HALO, 0, ready);
  if (rc) return bail(rc);
  rc = join();
  if (!rc && !conc) rc = part(RDYHIP_PHASE_HALO, 0, ready);
  // (second order: the gradient exchange went through the send buffer; the ghost-adjacent tiles' launch, behind it, has stored
  // the new state's rows)
  halo_note_step(op, h, rc ? nullptr : u_out);
  return rc;
}

// the trial of the two forms (FormChoice): which form this step takes, and the bookkeeping around it
int form_begin(RDyHipHalo h, int kind, hipStream_t st, int *slot) {
  RDyHipHalo_s::FormChoice &c = h->choice[kind];
  *slot = -1;
  if (c.source != RDYHIP_HALO_FORM_TRIAL_RUNNING) return c.form;
  constexpr int T = RDyHipHalo_s::FormChoice::TRIAL;
  if (c.steps == 2 * T) {
    // the trial is over: read the timings (one host wait for the last timed step, once in the life of the halo)
    if (hipEventSynchronize(c.ev1[2 * T - 1]) == hipSuccess) {
      for (int i = 0; i < 2 * T; ++i) {
        float ms = 0.0f;
        if (i / 2 < RDyHipHalo_s::FormChoice::SKIP || hipEventElapsedTime(&ms, c.ev0[i], c.ev1[i]) != hipSuccess) continue;
        c.ms[i & 1] += ms;
        c.n[i & 1]++;
      }
    }
    if (c.n[0] > 0 && c.n[1] > 0) {
      c.ms[0] /= c.n[0];
      c.ms[1] /= c.n[1];
      c.form   = c.ms[1] < c.ms[0] ? 1 : 0;
      c.source = RDYHIP_HALO_FORM_MEASURED;
    } else {
      c.form   = 0;
      c.source = RDYHIP_HALO_FORM_DEFAULT;  // events unavailable: the form without cross-stream dependencies
    }
    return c.form;
  }
  const int i = c.steps++;
  if (!c.ev0[i]) {
    if (hipEventCreate(&c.ev0[i]) != hipSuccess || hipEventCreate(&c.ev1[i]) != hipSuccess) {
      c.form   = 0;
      c.source = RDYHIP_HALO_FORM_DEFAULT;
      return 0;
    }
  }
  (void)hipEventRecord(c.ev0[i], st);
  *slot = i;
  return i & 1;
}
void form_end(RDyHipHalo h, int kind, hipStream_t st, int slot) {
  if (slot >= 0) (void)hipEventRecord(h->choice[kind].ev1[slot], st);
}

// OperatorRHSFunction (u_out == nullptr) or one forward-Euler step (u_out != nullptr) with the ghost update of u
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
  const int kind = u_out ? 1 : 0;
  int       slot = -1;
  int       form = form_begin(h, kind, st, &slot);
  // a fused pack with a send cell outside the ghost-adjacent tiles is only safe in order (step_two_streams)
  if (kind == 1 && h->fused_pack && op->fused_halo == h && !h->send_cells_in_halo_tiles) form = 0;
  const int rc = form ? step_two_streams(op, h, dt, u, f, u_out, st) : step_in_order(op, h, dt, u, f, u_out, st);
  form_end(h, kind, st, slot);
  return rc;
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
    // The form of a step is chosen by a trial on the communicator this halo really has (FormChoice), not by a constant
    // tuned on one device with the exchange looped back (rounds 3-4 switched at twelve rounds of interior tiles).  What
    // forces it, read once here: RDYHIP_OVERLAP=0 / 1; RDYHIP_OVERLAP_MIN_ROUNDS=n (the old size rule: two streams when the
    // interior tiles fill at least n rounds of the persistent grid).  A rank without peers has nothing to choose.
    int forced = -1, source = RDYHIP_HALO_FORM_FORCED;
    if (const char *e = getenv("RDYHIP_OVERLAP_MIN_ROUNDS")) {
      const int pgrid = std::max(8, op->muscl ? op->pgrid_muscl : op->pgrid);
      forced = op->use_tiled ? ((int64_t)(op->ntiles - op->n_halo_tiles) >= (int64_t)std::max(0, atoi(e)) * pgrid ? 1 : 0) : (op->n_owned >= 1500000 ? 1 : 0);
    }
    if (const char *e = getenv("RDYHIP_OVERLAP")) forced = atoi(e) != 0 ? 1 : 0;
    if (forced < 0 && ns == 0 && nr == 0) {
      forced = 0;
      source = RDYHIP_HALO_FORM_DEFAULT;
    }
    // A pattern that receives into OWNED rows (no real partition does: DMPlex's point SF has ghost leaves only; synthetic test
    // patterns do) cannot run its transfer beside the launch that reads those rows: in order, whatever was asked for
    for (int32_t i = 0; i < nr; ++i) {
      const int32_t c = recv_cell_ids[i];
      if (op->prefix ? c < op->n_owned : (c < (int32_t)op->h_l2o.size() && op->h_l2o[(size_t)c] >= 0 && op->h_l2o[(size_t)c] < op->n_owned)) {
        forced = 0;
        source = RDYHIP_HALO_FORM_LOCKED_IN_ORDER;
        break;
      }
    }
    for (auto &c : h->choice)
      if (forced >= 0) {
        c.form   = forced;
        c.source = source;
      }
    if (const char *e = getenv("RDYHIP_HALO_CONCURRENT")) h->halo_concurrent = atoi(e) != 0;   // measurement knob
    if (const char *e = getenv("RDYHIP_GRAD_PACK_FUSED")) h->grad_pack_allowed = atoi(e) != 0;  // measurement knob
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
  h->send_cells_in_halo_tiles = true;
  if (on) {
    const int32_t ns = h->send_off.back();
    std::vector<int32_t> ids((size_t)ns);
    if (ns) HIP_TRY(hipMemcpy(ids.data(), h->d_send_ids.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
    if ((int64_t)ns >= (1 << 24)) return fail(RDYHIP_ERR_ARG_SIZ, "%d send cells do not fit the 24-bit row of a send entry", ns);
    std::vector<std::pair<int32_t, uint32_t>> ent((size_t)ns);  // (tile, cell-in-tile | row << 8)
    const auto &c0 = op->h_tile_c0;
    for (int32_t i = 0; i < ns; ++i) {
      const int32_t c = ids[i];
      const int32_t o = op->prefix ? c : (c < (int32_t)op->h_l2o.size() ? op->h_l2o[c] : -1);
      if (o < 0 || o >= op->n_owned) return fail(RDYHIP_ERR_USER, "send cell %d is not an owned cell", c);
      const int32_t t = (int32_t)(std::upper_bound(c0.begin(), c0.end(), o) - c0.begin()) - 1;  // the tile that holds owned cell o
      ent[i] = std::make_pair(t, (uint32_t)(o - c0[(size_t)t]) | ((uint32_t)i << 8));
    }
    std::sort(ent.begin(), ent.end());
    std::vector<int32_t>  off((size_t)op->ntiles + 1, 0);
    std::vector<uint32_t> packed((size_t)ns);
    for (int32_t i = 0; i < ns; ++i) {
      off[(size_t)ent[i].first + 1]++;
      packed[i] = ent[i].second;
      tiles[(size_t)ent[i].first].cnt |= TILE_SEND_FLAG;
      // A send cell in a tile no ghost touches (the DM's overlap is vertex-adjacent, the tiles' halo flag edge-adjacent): the
      // INTERIOR launch of a two-stream step would store its row while the exchange stream reads the send buffer -- such a
      // halo keeps its fused-pack Euler steps in order (overlapped())
      if (!(tiles[(size_t)ent[i].first].cnt & TILE_HALO_FLAG)) h->send_cells_in_halo_tiles = false;
    }
    for (int32_t t = 0; t < op->ntiles; ++t) off[(size_t)t + 1] += off[t];
    h->d_send_tile_off.release();
    h->d_send_ent.release();
    int rc = h->d_send_tile_off.upload(off);
    if (!rc) rc = h->d_send_ent.upload(packed);
    if (rc) return rc;
  }
  // second order (fused form): the rows each ghost-adjacent cell's gradient travels in, by position in the halo cell list
  bool gfused = false;
  if (on && op->muscl && op->n_halo > 0 && h->grad_pack_allowed) {
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
  if (!op->use_tiled) return fail(RDYHIP_ERR_USER, "the fused pack rides on the tiled Euler-step kernels (not RDYHIP_KERNEL=cell)");
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

int32_t rdyhip_halo_overlaps(RDyHipHalo halo) { return halo && halo->choice[0].form ? 1 : 0; }

int rdyhip_halo_form_info(RDyHipHalo halo, int32_t kind, RDyHipHaloFormInfo *info) {
  if (!halo || !info) return fail(RDYHIP_ERR_USER, "null argument");
  if (kind != RDYHIP_HALO_STEP_RHS && kind != RDYHIP_HALO_STEP_EULER) return fail(RDYHIP_ERR_USER, "unknown kind of step %d", kind);
  const RDyHipHalo_s::FormChoice &c = halo->choice[kind];
  info->form          = c.form;
  info->source        = c.source;
  info->trial_steps   = c.steps;
  info->in_order_ms   = c.ms[0];
  info->two_stream_ms = c.ms[1];
  // the lock of overlapped(): a fused pack with a send cell outside the ghost-adjacent tiles steps in order
  if (kind == RDYHIP_HALO_STEP_EULER && halo->fused_pack && halo->op->fused_halo == halo && !halo->send_cells_in_halo_tiles) {
    info->form   = 0;
    info->source = RDYHIP_HALO_FORM_LOCKED_IN_ORDER;
  }
  return 0;
}

int rdyhip_halo_set_form(RDyHipHalo halo, int32_t kind, int32_t form) {
  if (!halo) return fail(RDYHIP_ERR_USER, "null halo");
  if (kind != RDYHIP_HALO_STEP_RHS && kind != RDYHIP_HALO_STEP_EULER) return fail(RDYHIP_ERR_USER, "unknown kind of step %d", kind);
  RDyHipHalo_s::FormChoice &c = halo->choice[kind];
  if (form < 0) {  // run the trial (again); the events are kept
    c.form = 0;
    c.source = RDYHIP_HALO_FORM_TRIAL_RUNNING;
    c.steps = 0;
    c.ms[0] = c.ms[1] = 0.0;
    c.n[0] = c.n[1] = 0;
    return 0;
  }
  c.form   = form ? 1 : 0;
  c.source = RDYHIP_HALO_FORM_FORCED;
  return 0;
}

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
