// MI355X-native SWE right-hand-side operator: kernels, mesh repack and the
// C ABI declared in include/rdyhip.h.  gfx950 only.
//
// Layout: one thread owns one owned cell.  The cell's
// edges are stored as up to S "slots" (S = 3 for triangle meshes, 4 when
// quads are present), struct-of-arrays by slot so that every per-slot array
// is read with unit stride across a wavefront:
//     nbr [s][o]  int32   neighbour's local cell id (| NBR_GHOST), or -1-k for
//                         boundary edge k, or NBR_EMPTY
//     cn,sn[s][o] double  the edge's unit normal in its canonical left->right
//                         orientation (edges.cn/sn)
//     coef[s][o]  double  -len/area_self if this cell is the edge's left cell,
//                         +len/area_self if it is the right cell -- the factor
//                         the reference multiplies the edge flux by
//                         (src/swe/swe_petsc.c:301-305); its sign is the
//                         orientation flag
// Slots are ordered by the position of the edge in the reference's loops
// (internal edges in internal_edge_ids order, then boundary 0's edges, ...),
// so a cell's contributions are summed in the reference's order.  Each edge
// flux is evaluated in the canonical orientation from both of its cells, so
// the two evaluations are bitwise equal and the scheme stays conservative
// without atomics or a scatter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "rdyhip.h"
#include "swe_device.h"
#include "swe_kernels.h"
#include "forcing_kernels.h"
#include "muscl_kernels.h"

using namespace rdyhip;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char    buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return fail(RDYHIP_ERR_LIB, "%s failed: %s", #expr, hipGetErrorString(e_));       \
  } while (0)

template <typename T>
struct DevBuf {
  T     *p = nullptr;
  size_t n = 0;
  int    alloc(size_t count) {
    n = count;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) return fail(RDYHIP_ERR_MEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    return 0;
  }
  int upload(const std::vector<T> &h) {
    int rc = alloc(h.size());
    if (rc) return rc;
    if (!h.empty()) HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
  int zeros(size_t count) {
    int rc = alloc(count);
    if (rc) return rc;
    HIP_TRY(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
    return 0;
  }
  void   release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

// Staging of the stream-ordered setters (rdyhip_set_*_on): a small ring of slots, each a pinned host buffer + a device
// buffer + an event.  The caller's array is copied into the pinned buffer before the call returns (the caller may reuse it
// at once, as with the reference's setters, which copy into a Vec); the host -> device copy then runs on the operator's own
// copy stream, beside whatever the caller's stream is executing, and only the final scatter into the operator's field is
// ordered on the caller's stream.  A slot is reused once the work that read it has finished (its event).
struct StageRing {
  static constexpr int N = 4;
  struct Slot {
    void      *h = nullptr, *d = nullptr;
    size_t     cap = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool       used = false;
  } slot[N];
  int         next = 0;
  hipStream_t copy = nullptr;
  void release() {
    for (auto &s : slot) {
      if (s.h) (void)hipHostFree(s.h);
      if (s.d) (void)hipFree(s.d);
      if (s.copied) (void)hipEventDestroy(s.copied);
      if (s.done) (void)hipEventDestroy(s.done);
      s = Slot{};
    }
    if (copy) (void)hipStreamDestroy(copy);
    copy = nullptr;
  }
};

struct RDyHipHalo_s;
struct RDyHipOperator_s {
  StageRing stage;
  RDyHipConfig config;
  RDyHipHalo_s *fused_halo = nullptr;  // the halo whose send lists are attached to the tile descriptors (rdyhip_halo_fuse_pack)
  std::vector<int32_t> h_l2o;          // local -> owned cell id, kept only when the owned cells are not a prefix
  int          device = 0;
  int32_t      n_cells = 0, n_owned = 0, S = 3, K = 0, n_internal = 0;
  int64_t      stride = 0;
  bool         prefix = true;
  int          grid = 0, xcd_chunks = 0;        // cell kernel
  int          pgrid = 0, tiled_xcd_chunks = 0; // tiled (persistent) kernel
  int          interior_shrink = 32;            // the INTERIOR phase leaves 1/interior_shrink of the workgroup slots free (0: none)
  bool         balance_rounds = false;          // size the persistent grid so that all workgroups walk the same number of tiles
  bool         keep_fdiv = false;

  DevBuf<int32_t> d_o2l, d_nbr, d_pos, d_halo_list, d_btype, d_bleft, d_bghost_list;
  DevBuf<double>  d_cn, d_sn, d_coef, d_dzdx, d_dzdy, d_mannings, d_extsrc;
  DevBuf<double>  d_bvalues, d_bflux, d_baccum, d_bcn, d_bsn, d_pv, d_fdiv, d_blk_max;
  DevBuf<int32_t> d_blk_pos;
  DevBuf<DeviceCourant> d_courant;
  DevBuf<ColdArgs>      d_cold;   // the kernels' rarely read pointers (swe_kernels.h), written once at create
  int32_t n_halo = 0, n_bghost = 0;
  // tiled kernel (swe_kernels.h)
  bool             use_tiled = true;
  bool             hr = false;   // hydrostatic reconstruction
  bool             uout_cached = false;  // Euler-step kernels store u_out with the default cache policy: the state fits the Infinity Cache
  DevBuf<double>   d_zc_local;
  int32_t          ntiles = 0, n_halo_tiles = 0, emax = 0;
  int64_t          nrec = 0;
  int32_t          hmax = 0;
  int64_t          nhalo_entries = 0;
  size_t           lds_bytes = 0;
  DevBuf<TileDesc> d_tiles;
  DevBuf<uint32_t> d_e_lr;
  DevBuf<int32_t>  d_hcells, d_tile_bk, d_tile_boff, d_halo_tiles, d_e_pos, d_x_lr;
  DevBuf<uint32_t> d_x_flags;
  DevBuf<double>   d_x_cs, d_x_cfac, d_x_mid;
  int32_t          n_xedges = 0;
  std::vector<int32_t> h_tile_c0;  // [ntiles + 1] first owned cell of each tile (the fused pack's send lists are built per tile)
  DevBuf<double>   d_e_cs;
  DevBuf<uint16_t> d_slot_ref;   // S == 4
  DevBuf<uint32_t> d_slot_ref3;  // S == 3
  // second order (muscl_kernels.h)
  bool             muscl = false;
  DevBuf<double>   d_grad, d_e_mid, d_cxy;
  DevBuf<int32_t>  d_hcells2, d_c_off;
  DevBuf<uint16_t> d_bn_idx;
  int32_t          hmax2 = 0;
  size_t           lds_muscl = 0;
  int              pgrid_muscl = 0;

  // host copies needed to resolve the Courant position into ids
  std::vector<int32_t> h_internal_edge, h_edge_cells, h_bedge, h_boff;
  std::vector<int64_t> h_cell_gid, h_edge_gid;
  std::vector<double>  h_area;
  RDyHipCourant        courant{0.0, -1, -1};
  // staging buffers for the setters
  DevBuf<double>  d_scratch_f;   // F of rdyhip_euler_step when the caller wants none and the kernel cannot skip it
  DevBuf<double>  d_stage_vals;
  DevBuf<int32_t> d_stage_ids;

  int64_t device_bytes = 0;

  ~RDyHipOperator_s() {
    d_o2l.release(); d_nbr.release(); d_pos.release(); d_halo_list.release(); d_btype.release(); d_bleft.release();
    d_bghost_list.release(); d_cn.release(); d_sn.release(); d_coef.release(); d_dzdx.release(); d_dzdy.release();
    d_mannings.release(); d_extsrc.release(); d_bvalues.release(); d_bflux.release();
    d_baccum.release(); d_bcn.release(); d_bsn.release(); d_pv.release(); d_fdiv.release(); d_blk_max.release();
    d_blk_pos.release(); d_courant.release(); d_cold.release(); d_stage_vals.release(); d_stage_ids.release(); d_scratch_f.release();
    d_tiles.release(); d_e_lr.release(); d_e_pos.release(); d_x_lr.release(); d_x_flags.release(); d_x_cs.release(); d_x_cfac.release(); d_x_mid.release();
    d_hcells.release(); d_tile_bk.release(); d_tile_boff.release(); d_halo_tiles.release();
    d_e_cs.release(); d_slot_ref.release(); d_slot_ref3.release(); d_zc_local.release();
    d_grad.release(); d_e_mid.release(); d_cxy.release(); d_hcells2.release(); d_c_off.release();
    d_bn_idx.release();
    stage.release();
  }
};

namespace {

using TiledKernelFn = void (*)(const KernelArgs, const double, const double *, double *);

void halo_forget_packed_state(RDyHipHalo_s *h);   // halo_exchange.h
void halo_operator_gone(RDyHipHalo_s *h);

// the instantiation of the tiled kernel for (slots per cell, source method, overwrite, HR, F stored with / without the hint)
template <bool HR, bool FNT>
TiledKernelFn tiled_kernel_fn_hr(int S, int src, bool ovw) {
  if (S == 3) {
    if (src) return ovw ? swe_rhs_tiled_kernel<3, 1, true, HR, false, FNT> : swe_rhs_tiled_kernel<3, 1, false, HR, false, FNT>;
    return ovw ? swe_rhs_tiled_kernel<3, 0, true, HR, false, FNT> : swe_rhs_tiled_kernel<3, 0, false, HR, false, FNT>;
  }
  if (src) return ovw ? swe_rhs_tiled_kernel<4, 1, true, HR, false, FNT> : swe_rhs_tiled_kernel<4, 1, false, HR, false, FNT>;
  return ovw ? swe_rhs_tiled_kernel<4, 0, true, HR, false, FNT> : swe_rhs_tiled_kernel<4, 0, false, HR, false, FNT>;
}
TiledKernelFn tiled_kernel_fn(int S, int src, bool ovw, bool hr, bool cached_f = false) {
  if (cached_f) return hr ? tiled_kernel_fn_hr<true, false>(S, src, ovw) : tiled_kernel_fn_hr<false, false>(S, src, ovw);
  return hr ? tiled_kernel_fn_hr<true, true>(S, src, ovw) : tiled_kernel_fn_hr<false, true>(S, src, ovw);
}
// the instantiation with the forward-Euler update fused into the stores (rdyhip_euler_step); FNT = false: plain (cached)
// stores of u_out, for states that fit the Infinity Cache
template <bool HR, bool FNT>
TiledKernelFn tiled_euler_fn_hr(int S, int src) {
  if (S == 3) return src ? swe_rhs_tiled_kernel<3, 1, true, HR, true, FNT> : swe_rhs_tiled_kernel<3, 0, true, HR, true, FNT>;
  return src ? swe_rhs_tiled_kernel<4, 1, true, HR, true, FNT> : swe_rhs_tiled_kernel<4, 0, true, HR, true, FNT>;
}
TiledKernelFn tiled_euler_fn(int S, int src, bool hr, bool uout_cached = false) {
  if (uout_cached) return hr ? tiled_euler_fn_hr<true, false>(S, src) : tiled_euler_fn_hr<false, false>(S, src);
  return hr ? tiled_euler_fn_hr<true, true>(S, src) : tiled_euler_fn_hr<false, true>(S, src);
}

using MusclKernelFn = void (*)(const KernelArgs, const MusclArgs, const double, const double *, double *);

template <int S, int LIM>
MusclKernelFn muscl_fn_lim(int src, bool ovw, bool euler) {
  if (euler) return src ? swe_rhs_muscl_fused_kernel<S, 1, true, LIM, true> : swe_rhs_muscl_fused_kernel<S, 0, true, LIM, true>;
  if (src) return ovw ? swe_rhs_muscl_fused_kernel<S, 1, true, LIM, false> : swe_rhs_muscl_fused_kernel<S, 1, false, LIM, false>;
  return ovw ? swe_rhs_muscl_fused_kernel<S, 0, true, LIM, false> : swe_rhs_muscl_fused_kernel<S, 0, false, LIM, false>;
}
template <int S>
MusclKernelFn muscl_fn_s(int src, bool ovw, bool euler, int limiter) {
  switch (limiter) {
    case RDYHIP_LIMITER_NONE: return muscl_fn_lim<S, LIMITER_NONE>(src, ovw, euler);
    case RDYHIP_LIMITER_VANLEER: return muscl_fn_lim<S, LIMITER_VANLEER>(src, ovw, euler);
    default: return muscl_fn_lim<S, LIMITER_MINMOD>(src, ovw, euler);
  }
}
MusclKernelFn muscl_kernel_fn(int S, int src, bool ovw, bool euler, int limiter) {
  return S == 3 ? muscl_fn_s<3>(src, ovw, euler, limiter) : muscl_fn_s<4>(src, ovw, euler, limiter);
}

MusclArgs muscl_args(RDyHipOperator op) {
  MusclArgs g{};
  g.grad  = op->d_grad.p;
  g.e_mid = op->d_e_mid.p;
  g.cxy   = op->d_cxy.p;
  g.hcells2 = op->d_hcells2.p;
  g.r2_off  = op->d_c_off.p;
  g.bn_idx  = op->d_bn_idx.p;
  return g;
}

// ComputeLeastSquaresGradients for the owned cells selected by `phase`
int launch_gradients(RDyHipOperator op, int32_t phase, const double *u, hipStream_t st) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (!op->muscl) return fail(RDYHIP_ERR_USER, "the operator was not created with second_order");
  if (phase != RDYHIP_PHASE_ALL && phase != RDYHIP_PHASE_INTERIOR && phase != RDYHIP_PHASE_HALO) return fail(RDYHIP_ERR_USER, "unknown phase %d", phase);
  if (op->n_owned == 0) return 0;
  if (!u) return fail(RDYHIP_ERR_USER, "null u_local");
  KernelArgs a{};
  a.n_owned = op->n_owned;
  a.stride  = op->stride;
  a.o2l     = op->prefix ? nullptr : op->d_o2l.p;
  a.cold    = op->d_cold.p;
  a.phase   = phase;
  int grid;
  if (phase == RDYHIP_PHASE_HALO) {
    if (op->n_halo == 0) return 0;
    // with a fused pack attached this launch stores gradient rows into the halo's send buffer (ColdArgs::gsend_*):
    // whatever state rows it mirrored are gone
    if (op->fused_halo) halo_forget_packed_state(op->fused_halo);
    a.list       = op->d_halo_list.p;
    a.n_work     = op->n_halo;
    a.xcd_chunks = 0;
    a.phase      = RDYHIP_PHASE_ALL;
    grid         = (op->n_halo + BLOCK - 1) / BLOCK;
  } else {
    a.list       = nullptr;
    a.n_work     = op->n_owned;
    a.xcd_chunks = op->xcd_chunks;
    grid         = op->grid;
  }
  const MusclArgs g = muscl_args(op);
  if (op->S == 3) hipLaunchKernelGGL((muscl_gradient_kernel<3>), dim3(grid), dim3(BLOCK), 0, st, a, g, u);
  else hipLaunchKernelGGL((muscl_gradient_kernel<4>), dim3(grid), dim3(BLOCK), 0, st, a, g, u);
  HIP_TRY(hipGetLastError());
  return 0;
}

// bucket_half: which of the Courant buckets (one per workgroup, merged on demand) the launch owns -- 0: all of them (every
// ordinary launch); 1 / 2: the first / second half, for the interior and the halo launch of rdyhip_rhs_overlapped, which
// run side by side on two streams and must not share a bucket
int launch_rhs(RDyHipOperator op, int32_t phase, int32_t overwrite, int reset_diag, double dt, const double *u, double *f, hipStream_t st,
               bool gradients_ready = false, double *u_out = nullptr, int bucket_half = 0) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  // an Euler-step launch rewrites the attached halo's send buffer (tiles flagged TILE_SEND_FLAG): whatever it held is gone
  // (rdyhip_euler_step_overlapped notes the new content itself once its launches are enqueued)
  if (u_out && op->fused_halo) halo_forget_packed_state(op->fused_halo);
  // rdyhip_euler_step: the tiled kernels have the update fused into their stores (F optional); the cell-centric
  // kernel evaluates F (into a scratch vector if the caller wants none) and a separate update follows
  const bool euler_fused = u_out && op->use_tiled;
  if (u_out && !euler_fused && !f && op->n_owned > 0) {
    if (!op->d_scratch_f.p) {
      int rc = op->d_scratch_f.alloc((size_t)3 * op->n_owned);
      if (rc) return rc;
    }
    f = op->d_scratch_f.p;
  }
  if (op->muscl && !gradients_ready && op->n_owned > 0) {
    // only ghost cells' gradients are read from memory, and they have to come from the exchange (CommunicateCellGradients):
    // rdyhip_compute_gradients(RDYHIP_PHASE_HALO), exchange of the ghost rows, then RDYHIP_PHASE_GRADIENTS_READY
    if (op->n_cells > op->n_owned || phase != RDYHIP_PHASE_ALL)
      return fail(RDYHIP_ERR_USER, "second_order with ghost cells or a phased apply needs RDYHIP_PHASE_GRADIENTS_READY (see rdyhip_compute_gradients)");
  }
  if (op->n_owned == 0) return reset_diag ? rdyhip_reset_diagnostics(op, (void *)st) : 0;  // a rank may own nothing (f_global is then empty)
  if (!u || (!f && !euler_fused)) return fail(RDYHIP_ERR_USER, "null u_local / f_global");
  KernelArgs a{};
  a.u_out      = u_out;
  a.n_owned    = op->n_owned;
  a.stride     = op->stride;
  a.o2l        = op->prefix ? nullptr : op->d_o2l.p;
  a.cold       = op->d_cold.p;
  a.coef       = op->d_coef.p;
  a.dzdx       = op->d_dzdx.p;
  a.dzdy       = op->d_dzdy.p;
  a.mannings   = op->d_mannings.p;
  a.extsrc     = op->d_extsrc.p;
  a.pv         = op->d_pv.p;
  a.fdiv       = op->keep_fdiv ? op->d_fdiv.p : nullptr;
  a.n_buckets  = (int32_t)op->d_blk_max.n;
  a.bucket_off = 0;
  if (bucket_half) {
    const int32_t half = a.n_buckets / 2;
    a.n_buckets = half;
    if (bucket_half == 2) a.bucket_off = half;
  }
  a.reset_diag = reset_diag ? 1 : 0;
  a.tiny_h     = op->config.tiny_h;
  a.h_anuga_sq = op->config.h_anuga_regular * op->config.h_anuga_regular;
  a.xq_thresh  = op->config.xq2018_threshold;
  a.overwrite  = overwrite ? 1 : 0;
  a.phase      = phase;

  a.tiles    = op->d_tiles.p;
  a.e_lr     = op->d_e_lr.p;
  a.e_cs     = op->d_e_cs.p;
  a.hcells   = op->d_hcells.p;
  a.slot_ref = op->S == 3 ? (const void *)op->d_slot_ref3.p : (const void *)op->d_slot_ref.p;
  a.zc_local = op->d_zc_local.p;

  int        grid = 0;
  const bool xq = op->config.source_method == RDYHIP_SOURCE_IMPLICIT_XQ2018;
  // a HALO phase on a rank without ghost-adjacent cells has no flux launch, but the tail below (boundary edges of
  // ghost cells, the separate Euler update of the non-fused kernels) still belongs to it
  const bool nothing_to_launch = phase == RDYHIP_PHASE_HALO && (op->use_tiled ? op->n_halo_tiles == 0 : op->n_halo == 0);
  if (nothing_to_launch) {
  } else if (op->use_tiled) {
    const int pgrid = op->muscl ? op->pgrid_muscl : op->pgrid;
    // persistent workgroups: at most as many as the device holds at once
    if (phase == RDYHIP_PHASE_HALO) {
      a.list       = op->d_halo_tiles.p;
      a.n_work     = op->n_halo_tiles;
      a.xcd_chunks = 0;
      a.phase      = RDYHIP_PHASE_ALL;  // the list already holds exactly the halo tiles
      grid         = std::min(pgrid, op->n_halo_tiles);
    } else {
      a.list   = nullptr;
      a.n_work = op->ntiles;
      // While the interior phase runs, the halo exchange's pack / RCCL / unpack kernels need somewhere
      // to run: the persistent grid would otherwise fill every SIMD's register file for the whole launch.
      const int shrink = phase == RDYHIP_PHASE_INTERIOR ? op->interior_shrink : 0;
      const int pg     = shrink > 0 ? std::max(8, pgrid - std::max(8, pgrid / shrink)) : pgrid;
      if (op->tiled_xcd_chunks > 0) {
        a.xcd_chunks = op->tiled_xcd_chunks;
        int per_xcd  = std::max(1, std::min(pg >> 3, op->tiled_xcd_chunks));
        if (op->balance_rounds) {
          // every workgroup of an XCD walks the same number of tiles (+-1): a mesh of 3 907 tiles on 768 resident workgroups
          // would otherwise run five full rounds and a sixth with 9 % of the device busy
          const int rounds = (op->tiled_xcd_chunks + per_xcd - 1) / per_xcd;
          per_xcd          = (op->tiled_xcd_chunks + rounds - 1) / rounds;
        }
        grid = per_xcd * 8;
      } else {
        a.xcd_chunks = 0;
        grid         = std::min(pg, op->ntiles);
      }
    }
    if (op->muscl) {
      MusclKernelFn kfn = muscl_kernel_fn(op->S, xq ? 1 : 0, overwrite != 0, euler_fused, op->config.limiter);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(kfn), dim3(grid), dim3(TILE), op->lds_muscl, st, a, muscl_args(op), dt, u, f);
    } else if (euler_fused) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(tiled_euler_fn(op->S, xq ? 1 : 0, op->hr, op->uout_cached)), dim3(grid), dim3(TILE), op->lds_bytes, st, a, dt, u, f);
    } else {
      const size_t lds = op->lds_bytes;
      const bool cached_f = (op->config.flags & RDYHIP_CONFIG_CACHED_F_STORES) != 0;
      hipLaunchKernelGGL(HIP_KERNEL_NAME(tiled_kernel_fn(op->S, xq ? 1 : 0, overwrite != 0, op->hr, cached_f)), dim3(grid), dim3(TILE), lds, st, a, dt, u, f);
    }
  } else {
    if (phase == RDYHIP_PHASE_HALO) {
      a.list       = op->d_halo_list.p;
      a.n_work     = op->n_halo;
      a.xcd_chunks = 0;
      a.phase      = RDYHIP_PHASE_ALL;  // the list already holds exactly the halo cells
      grid         = (op->n_halo + BLOCK - 1) / BLOCK;
    } else {
      a.list       = nullptr;
      a.n_work     = op->n_owned;
      a.xcd_chunks = op->xcd_chunks;
      grid         = op->grid;
    }
    if (op->S == 3) {
      if (xq) hipLaunchKernelGGL((swe_rhs_kernel<3, 1>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
      else hipLaunchKernelGGL((swe_rhs_kernel<3, 0>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
    } else {
      if (xq) hipLaunchKernelGGL((swe_rhs_kernel<4, 1>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
      else hipLaunchKernelGGL((swe_rhs_kernel<4, 0>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
    }
  }
  HIP_TRY(hipGetLastError());
  // boundary edges hanging off ghost cells (diagnostic vectors only); once per full apply
  if (op->n_bghost > 0 && phase != RDYHIP_PHASE_INTERIOR) {
    hipLaunchKernelGGL(boundary_ghost_kernel, dim3((op->n_bghost + 63) / 64), dim3(64), 0, st, op->n_bghost, op->d_bghost_list.p, op->d_bleft.p,
                       op->d_btype.p, op->d_bcn.p, op->d_bsn.p, op->d_bvalues.p, op->d_bflux.p, op->d_baccum.p, u, dt, a.tiny_h, a.h_anuga_sq);
    HIP_TRY(hipGetLastError());
  }
  if (u_out && !euler_fused && phase != RDYHIP_PHASE_INTERIOR) {
    // F is complete once the halo (or the only) phase has run
    const int64_t n3 = 3 * (int64_t)op->n_owned;
    hipLaunchKernelGGL(euler_out_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, op->n_owned, op->prefix ? nullptr : op->d_o2l.p, dt, f, u, u_out);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

}  // namespace

// Everything rdyhip_create derives from the mesh on the host, before any device call: validation, the slot tables in the
// reference's loop order, the tiles with their edge records / halo lists / boundary lists, the second-order stencils.
struct HostLayout {
  int32_t nc = 0, no = 0, ne = 0, ni = 0, K = 0, S = 3, ntiles = 0, emax = 0, hmax = 0, hmax2 = 0;
  int64_t stride = 0;
  bool    prefix = true, hr_on = false, muscl_on = false;
  size_t  lds_bytes = 0, lds_muscl = 0;
  std::vector<int32_t>  o2l, boff, nbr, pos, btype, bleft, bedge, bghost, halo, hcells, tile_bk, tile_boff, tile_c0, halo_tiles, hcells2, c_off, e_pos;
  std::vector<double>   cn, sn, coef, bcn, bsn, e_cs, e_mid, dzdx, dzdy;
  std::vector<TileDesc> tiles;
  std::vector<uint32_t> e_lr;
  // the extra Courant edges (ColdArgs::x_*): internal edges the owned cells' slots do not cover as the reference's loop does
  std::vector<int32_t>  x_lr, x_pos;
  std::vector<uint32_t> x_flags;
  std::vector<double>   x_cs, x_cfac, x_mid;
  std::vector<uint16_t> slot_ref, bn_idx;
};

// argument checks of rdyhip_create (the reference's PetscCheck messages where it has them)
static int layout_check_arguments(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries) {
  if (!config || !mesh) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_create");
  if (num_boundaries < 0 || (num_boundaries > 0 && !boundaries)) return fail(RDYHIP_ERR_USER, "bad boundary list");
  if (config->riemann != RDYHIP_RIEMANN_ROE) return fail(RDYHIP_ERR_USER, "Unsupported Riemann solver");  // swe_petsc.c:269
  if (config->source_method != RDYHIP_SOURCE_SEMI_IMPLICIT && config->source_method != RDYHIP_SOURCE_IMPLICIT_XQ2018)
    return fail(RDYHIP_ERR_USER, "Only semi_implicit and implicit_xq2018 are supported");  // swe_petsc.c:973
  if (config->well_balancing != RDYHIP_WELL_BALANCING_NONE && config->well_balancing != RDYHIP_WELL_BALANCING_HR)
    return fail(RDYHIP_ERR_USER, "Only well_balancing = none or hydrostatic_reconstruction is supported in the PETSc version");  // operator.c:388
  if (config->well_balancing == RDYHIP_WELL_BALANCING_HR && !mesh->cell_zc)
    return fail(RDYHIP_ERR_USER, "hydrostatic reconstruction needs the per-cell bed elevation (RDyHipMesh.cell_zc)");
  const bool muscl_on = config->second_order != 0;
  if (muscl_on && config->well_balancing == RDYHIP_WELL_BALANCING_HR)
    return fail(RDYHIP_ERR_USER, "-second_order cannot be used with well_balancing = HR simultaneously (not yet implemented)");  // operator.c:388-389
  if (muscl_on && config->limiter != RDYHIP_LIMITER_MINMOD && config->limiter != RDYHIP_LIMITER_NONE && config->limiter != RDYHIP_LIMITER_VANLEER)
    return fail(RDYHIP_ERR_USER, "unknown slope limiter %d", config->limiter);
  if (muscl_on && mesh->num_edges > 0 && (!mesh->cell_centroids || !mesh->edge_vertex_ids || !mesh->vertex_points))
    return fail(RDYHIP_ERR_USER, "second_order needs RDyHipMesh.cell_centroids, edge_vertex_ids and vertex_points");
  const int32_t nc = mesh->num_cells, no = mesh->num_owned_cells, ne = mesh->num_edges, ni = mesh->num_internal_edges;
  if (nc < 0 || no < 0 || no > nc || ne < 0 || ni < 0 || ni > ne) return fail(RDYHIP_ERR_ARG_SIZ, "inconsistent mesh sizes");
  if (nc >= NBR_GHOST) return fail(RDYHIP_ERR_ARG_SIZ, "too many local cells (%d) for the 30-bit neighbour encoding", nc);
  if (nc > 0 && (!mesh->cell_is_owned || !mesh->cell_local_to_owned || !mesh->cell_areas || !mesh->cell_dz_dx || !mesh->cell_dz_dy))
    return fail(RDYHIP_ERR_USER, "null cell array");
  if (ne > 0 && (!mesh->edge_cell_ids || !mesh->edge_lengths || !mesh->edge_cn || !mesh->edge_sn)) return fail(RDYHIP_ERR_USER, "null edge array");
  if (ni > 0 && !mesh->edge_internal_ids) return fail(RDYHIP_ERR_USER, "null internal edge list");

  return 0;
}

// owned <-> local maps, the boundary-edge table and the per-cell slot tables in the reference's loop order
// (with the least-squares gradient coefficients of the second-order path)
static int layout_build_slots(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                              HostLayout &L) {
  const bool    muscl_on = config->second_order != 0;
  const int32_t nc = mesh->num_cells, no = mesh->num_owned_cells, ne = mesh->num_edges, ni = mesh->num_internal_edges;
  // ---- owned <-> local maps ---------------------------------------------
  std::vector<int32_t> o2l((size_t)no, -1);
  int32_t              owned_seen = 0;
  for (int32_t c = 0; c < nc; ++c) {
    if (mesh->cell_is_owned[c]) {
      const int32_t o = mesh->cell_local_to_owned[c];
      if (o < 0 || o >= no || o2l[o] != -1) return fail(RDYHIP_ERR_USER, "cells.local_to_owned is not a bijection onto the owned cells");
      o2l[o] = c;
      ++owned_seen;
    }
  }
  if (owned_seen != no) return fail(RDYHIP_ERR_ARG_SIZ, "num_owned_cells (%d) does not match cells.is_owned (%d)", no, owned_seen);
  bool prefix = true;
  for (int32_t o = 0; o < no; ++o) prefix = prefix && (o2l[o] == o);

  // ---- boundary-edge table ------------------------------------------------
  std::vector<int32_t> boff((size_t)num_boundaries + 1, 0);
  for (int32_t b = 0; b < num_boundaries; ++b) {
    if (boundaries[b].num_edges < 0 || (boundaries[b].num_edges > 0 && !boundaries[b].edge_ids)) return fail(RDYHIP_ERR_USER, "bad boundary %d", b);
    const int32_t t = boundaries[b].condition_type;
    if (t != RDYHIP_CONDITION_DIRICHLET && t != RDYHIP_CONDITION_REFLECTING && t != RDYHIP_CONDITION_CRITICAL_OUTFLOW)
      return fail(RDYHIP_ERR_USER, "Invalid boundary condition encountered for boundary %d", b);  // swe_petsc.c:568
    boff[b + 1] = boff[b] + boundaries[b].num_edges;
  }
  const int32_t K = boff[num_boundaries];

  // ---- count slots per owned cell -----------------------------------------
  std::vector<int32_t> cnt((size_t)no, 0);
  for (int32_t p = 0; p < ni; ++p) {
    const int32_t e = mesh->edge_internal_ids[p];
    if (e < 0 || e >= ne) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "internal edge id %d out of range", e);
    const int32_t l = mesh->edge_cell_ids[2 * e], r = mesh->edge_cell_ids[2 * e + 1];
    if (r == -1) continue;  // swe_petsc.c:249
    if (l < 0 || l >= nc || r < 0 || r >= nc) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "edge %d has cell ids (%d,%d) out of range", e, l, r);
    if (mesh->cell_is_owned[l]) cnt[mesh->cell_local_to_owned[l]]++;
    if (mesh->cell_is_owned[r]) cnt[mesh->cell_local_to_owned[r]]++;
  }
  for (int32_t b = 0; b < num_boundaries; ++b) {
    for (int32_t i = 0; i < boundaries[b].num_edges; ++i) {
      const int32_t e = boundaries[b].edge_ids[i];
      if (e < 0 || e >= ne) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "boundary %d edge id %d out of range", b, e);
      const int32_t l = mesh->edge_cell_ids[2 * e];
      if (l < 0 || l >= nc) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "boundary edge %d has no left cell", e);
      if (mesh->cell_is_owned[l]) cnt[mesh->cell_local_to_owned[l]]++;
    }
  }
  int32_t maxcnt = 0;
  for (int32_t o = 0; o < no; ++o) maxcnt = std::max(maxcnt, cnt[o]);
  if (maxcnt > 4) return fail(RDYHIP_ERR_USER, "a cell has %d edges (must be 3 or 4)", maxcnt);  // rdymesh.c:809
  const int32_t S      = maxcnt <= 3 ? 3 : 4;
  const int64_t stride = ((int64_t)no + 63) / 64 * 64;

  // ---- fill slots in the reference's loop order ---------------------------
  std::vector<int32_t> nbr((size_t)(S * stride), NBR_EMPTY), pos((size_t)(S * stride), -1);
  std::vector<double>  cn((size_t)(S * stride), 0.0), sn((size_t)(S * stride), 0.0), coef((size_t)(S * stride), 0.0);
  std::fill(cnt.begin(), cnt.end(), 0);
  if (muscl_on) {
    // every internal edge needs both cells for the second-order stencils (src/operator_fluxes_ceed.c:897-901)
    for (int32_t p = 0; p < ni; ++p) {
      const int32_t e = mesh->edge_internal_ids[p];
      if (mesh->edge_cell_ids[2 * e + 1] == -1) return fail(RDYHIP_ERR_USER, "second_order: internal edge %d has no right cell", e);
    }
  }
  auto put = [&](int32_t o, int32_t id, double c, double s, double k, int32_t p) {
    const int64_t idx = (int64_t)cnt[o]++ * stride + o;
    nbr[idx]          = id;
    cn[idx]           = c;
    sn[idx]           = s;
    coef[idx]         = k;
    pos[idx]          = p;
  };
  for (int32_t p = 0; p < ni; ++p) {
    const int32_t e = mesh->edge_internal_ids[p];
    const int32_t l = mesh->edge_cell_ids[2 * e], r = mesh->edge_cell_ids[2 * e + 1];
    if (r == -1) continue;
    const double len = mesh->edge_lengths[e];
    if (mesh->cell_is_owned[l]) put(mesh->cell_local_to_owned[l], r | (mesh->cell_is_owned[r] ? 0 : NBR_GHOST), mesh->edge_cn[e], mesh->edge_sn[e], -len / mesh->cell_areas[l], p);
    if (mesh->cell_is_owned[r]) put(mesh->cell_local_to_owned[r], l | (mesh->cell_is_owned[l] ? 0 : NBR_GHOST), mesh->edge_cn[e], mesh->edge_sn[e], len / mesh->cell_areas[r], p);
  }
  std::vector<int32_t> btype((size_t)K), bleft((size_t)K), bedge((size_t)K), bghost;
  std::vector<double>  bcn((size_t)K), bsn((size_t)K);
  for (int32_t b = 0; b < num_boundaries; ++b) {
    for (int32_t i = 0; i < boundaries[b].num_edges; ++i) {
      const int32_t k = boff[b] + i;
      const int32_t e = boundaries[b].edge_ids[i];
      const int32_t l = mesh->edge_cell_ids[2 * e];
      btype[k]        = boundaries[b].condition_type;
      bleft[k]        = l;
      bedge[k]        = e;
      bcn[k]          = mesh->edge_cn[e];
      bsn[k]          = mesh->edge_sn[e];
      if (mesh->cell_is_owned[l]) put(mesh->cell_local_to_owned[l], -1 - k, mesh->edge_cn[e], mesh->edge_sn[e], -mesh->edge_lengths[e] / mesh->cell_areas[l], ni + k);
      else bghost.push_back(k);
    }
  }
  // owned cells with a ghost neighbour
  std::vector<int32_t> halo;
  for (int32_t o = 0; o < no; ++o) {
    bool g = false;
    for (int32_t s = 0; s < S; ++s) {
      const int32_t id = nbr[(int64_t)s * stride + o];
      g                = g || (id >= 0 && (id & NBR_GHOST));
    }
    if (g) halo.push_back(o);
  }

  L.nc = nc; L.no = no; L.ne = ne; L.ni = ni; L.K = K; L.S = S; L.stride = stride; L.prefix = prefix; L.muscl_on = muscl_on;
  L.o2l = std::move(o2l); L.boff = std::move(boff); L.nbr = std::move(nbr); L.pos = std::move(pos); L.btype = std::move(btype);
  L.bleft = std::move(bleft); L.bedge = std::move(bedge); L.bghost = std::move(bghost); L.halo = std::move(halo);
  L.cn = std::move(cn); L.sn = std::move(sn); L.coef = std::move(coef);
  L.bcn = std::move(bcn); L.bsn = std::move(bsn);
  return 0;
}

// Where the tiles end.  A tile is a run of consecutive owned cells, at most `max_cells` (<= TILE) of them, grown 16 cells
// at a time (whole 128-byte lines of the [cell][3] outputs) while it fits the kernels' fixed capacities: TILE_MAX_REC edge
// records (two register rounds of the edge phase), the halo-cell planes in LDS and, for second order, the two rings.  Nothing
// else ends a tile (round 5 first also ended one where 16 cells shared no edge with it -- "the numbering jumps" -- which cut
// the nested numbering of a refined mesh into 64-cell tiles: 2.3 x slower, profiles/RESULTS_LOG.md section 12).  Any numbering
// is accepted; one without locality gets small tiles (a random one: ~30 cells) and runs slowly, correctly.
// Returns the first owned cell of every tile, [ntiles + 1].
static std::vector<int32_t> layout_cut_tiles(const RDyHipMesh *mesh, const HostLayout &L, int32_t max_cells) {
  const int32_t nc = L.nc, no = L.no, S = L.S, npos = L.ni + L.K;
  const int64_t stride = L.stride;
  const auto   &nbr = L.nbr;
  const auto   &pos = L.pos;
  const int32_t max_rec = TILE_MAX_REC, max_h1 = S == 3 ? TILE_MAX_HALO_TRI : TILE_MAX_HALO_QUAD;
  const int32_t max_ring = S == 3 ? MUSCL_MAX_RING_TRI : MUSCL_MAX_RING_QUAD;
  std::vector<int32_t> c0;
  c0.reserve((size_t)no / 200 + 2);
  // stamps: edge positions in the tile / in the candidate granule; cells in the first ring / candidates / second ring
  std::vector<int32_t> emark((size_t)npos, -1), etry((size_t)npos, -1), r1mark((size_t)nc, -1), r1try((size_t)nc, -1), r2try((size_t)nc, -1);
  std::vector<int32_t> r1, r1new;
  int32_t stamp = 0, attempt = 0;
  int32_t base = 0;
  while (base < no) {
    c0.push_back(base);
    ++stamp;
    r1.clear();
    int32_t cells = 0, rec = 0;
    int32_t gran = 16;
    while (cells < max_cells && base + cells < no) {
      const int32_t g = std::min(std::min(gran, max_cells - cells), no - (base + cells));
      const int32_t lo = base, hi = base + cells + g;  // the tile with the granule: owned cells [lo, hi)
      ++attempt;
      int32_t newrec = 0;
      r1new.clear();
      for (int32_t o = base + cells; o < hi; ++o) {
        for (int32_t sl = 0; sl < S; ++sl) {
          const int64_t idx = (int64_t)sl * stride + o;
          const int32_t id  = nbr[idx];
          if (id == NBR_EMPTY) continue;
          const int32_t p = pos[idx];
          if (emark[p] != stamp && etry[p] != attempt) {
            etry[p] = attempt;
            ++newrec;
          }
          if (id < 0) continue;  // boundary edge: no cell beyond it
          const int32_t n  = id & NBR_MASK;
          const int32_t on = mesh->cell_is_owned[n] ? mesh->cell_local_to_owned[n] : -1;
          if (on >= lo && on < hi) continue;  // inside the tile
          if (r1mark[n] != stamp && r1try[n] != attempt) {
            r1try[n] = attempt;
            r1new.push_back(n);
          }
        }
      }
      // first-ring cells the granule swallows
      int32_t left = 0;
      for (int32_t o = base + cells; o < hi; ++o)
        if (r1mark[L.o2l[o]] == stamp) ++left;
      const int32_t nh = (int32_t)r1.size() - left + (int32_t)r1new.size();
      bool bad = rec + newrec > max_rec || nh > max_h1;
      if (!bad && L.muscl_on) {
        // second ring: the other neighbours of the (owned) first-ring cells
        int32_t n2 = 0;
        auto ring2_of = [&](int32_t cell) {
          if (!mesh->cell_is_owned[cell]) return;  // a ghost's stencil is on another rank
          const int32_t ob = mesh->cell_local_to_owned[cell];
          if (ob >= lo && ob < hi) return;          // swallowed by the granule
          for (int32_t sl = 0; sl < S; ++sl) {
            const int32_t id = nbr[(int64_t)sl * stride + ob];
            if (id < 0) continue;
            const int32_t n  = id & NBR_MASK;
            const int32_t on = mesh->cell_is_owned[n] ? mesh->cell_local_to_owned[n] : -1;
            if (on >= lo && on < hi) continue;
            if ((r1mark[n] == stamp) || r1try[n] == attempt || r2try[n] == attempt) continue;
            r2try[n] = attempt;
            ++n2;
          }
        };
        for (int32_t cell : r1) ring2_of(cell);
        for (int32_t cell : r1new) ring2_of(cell);
        // (a first-ring cell the granule swallows may still be counted as somebody's second-ring neighbour: it is inside
        // the tile then, and the test above skips it)
        bad = nh + n2 > max_ring;
      }
      if (bad) {
        if (cells > 0) break;        // the granule starts the next tile
        if (gran > 1) {              // not even one granule fits: cell by cell
          gran = 1;
          continue;
        }
        // a single cell always fits (S records, S halo cells, 3 S ring cells)
      }
      // accept
      for (int32_t o = base + cells; o < hi; ++o)
        for (int32_t sl = 0; sl < S; ++sl) {
          const int64_t idx = (int64_t)sl * stride + o;
          if (nbr[idx] != NBR_EMPTY) emark[pos[idx]] = stamp;
        }
      if (left > 0) {
        size_t w = 0;
        for (size_t i = 0; i < r1.size(); ++i) {
          const int32_t cell = r1[i];
          const int32_t on   = mesh->cell_is_owned[cell] ? mesh->cell_local_to_owned[cell] : -1;
          if (on >= lo && on < hi) r1mark[cell] = -1;
          else r1[w++] = cell;
        }
        r1.resize(w);
      }
      for (int32_t cell : r1new) {
        r1mark[cell] = stamp;
        r1.push_back(cell);
      }
      cells += g;
      rec += newrec;
    }
    base += cells;
  }
  c0.push_back(no);
  return c0;
}

// the tiles: edge records, halo-cell lists, boundary lists, slot references, and for the second-order kernel the
// first-ring stencils and the second ring
static int layout_build_tiles(const RDyHipConfig *config, const RDyHipMesh *mesh, HostLayout &L) {
  const int32_t nc = L.nc, no = L.no, ni = L.ni, S = L.S;
  const int64_t stride   = L.stride;
  const bool    muscl_on = L.muscl_on;
  const auto &nbr = L.nbr; const auto &pos = L.pos; const auto &bedge = L.bedge; const auto &bleft = L.bleft;
  (void)config;
  int32_t max_cells = TILE;
  if (const char *e = getenv("RDYHIP_TILE_CELLS")) {  // measurement knob: the largest tile
    if (atoi(e) > 0) max_cells = std::min<int32_t>(atoi(e), TILE);
  }
  std::vector<int32_t>  tile_c0 = layout_cut_tiles(mesh, L, max_cells);
  const int32_t         ntiles = (int32_t)tile_c0.size() - 1;
  std::vector<TileDesc> tiles((size_t)ntiles + 1);
  std::vector<uint32_t> e_lr;
  std::vector<int32_t>  hcells, tile_bk, tile_boff((size_t)ntiles + 1, 0), halo_tiles, e_pos;
  std::vector<double>   e_cs, e_mid;
  std::vector<int32_t>  hslot2, touched2, hcells2, c_off;
  std::vector<uint16_t> bn_idx;
  int32_t               hmax2 = 0;
  std::vector<uint16_t> slot_ref((size_t)no * 4, SLOT_EMPTY);
  int32_t               emax = 0, hmax = 0;
  {
    e_lr.reserve((size_t)no * 2);
    e_cs.reserve((size_t)no * 2);
    std::vector<std::pair<int32_t, int32_t>> items;  // (loop position of the edge, owned cell * 4 + slot)
    items.reserve(4 * TILE);
    std::vector<int32_t> hslot((size_t)nc, -1);      // local cell -> halo slot in the current tile
    std::vector<int32_t> touched;
    if (muscl_on) {
      hslot2.assign((size_t)nc, -1);
      c_off.assign((size_t)ntiles + 1, 0);
    }
    for (int32_t t = 0; t < ntiles; ++t) {
      const int32_t base = tile_c0[t], cntc = tile_c0[t + 1] - base;
      items.clear();
      bool halo_tile = false;
      for (int32_t j = 0; j < cntc; ++j) {
        for (int32_t sl = 0; sl < S; ++sl) {
          const int64_t idx = (int64_t)sl * stride + base + j;
          if (nbr[idx] == NBR_EMPTY) continue;
          items.emplace_back(pos[idx], (base + j) * 4 + sl);
          if (nbr[idx] >= 0 && (nbr[idx] & NBR_GHOST)) halo_tile = true;
        }
      }
      std::sort(items.begin(), items.end());
      tiles[t].e_off = (int32_t)e_lr.size();
      tiles[t].h_off = (int32_t)hcells.size();
      tiles[t].c_off = base;
      tile_boff[t]   = (int32_t)tile_bk.size();
      tiles[t].cnt   = halo_tile ? TILE_HALO_FLAG : 0u;   // the counts are filled in below
      if (halo_tile) halo_tiles.push_back(t);
      touched.clear();
      int32_t nh = 0, nbk = 0;
      // LDS slot of a cell: 0..255 inside the tile, 256.. for the tile's halo cells
      auto slot_of = [&](int32_t cell) -> uint32_t {
        if (mesh->cell_is_owned[cell]) {
          const int32_t oo = mesh->cell_local_to_owned[cell];
          if (oo >= base && oo < base + cntc) return (uint32_t)(oo - base);
        }
        if (hslot[cell] < 0) {
          hslot[cell] = nh++;
          hcells.push_back(cell);
          touched.push_back(cell);
        }
        return (uint32_t)(TILE + hslot[cell]);
      };
      int32_t last = -1, local = -1;
      for (const auto &it : items) {
        if (it.first != last) {
          last = it.first;
          ++local;
          int32_t  e;
          uint32_t lr;
          if (last < ni) {
            e  = mesh->edge_internal_ids[last];
            lr = slot_of(mesh->edge_cell_ids[2 * e]) | (slot_of(mesh->edge_cell_ids[2 * e + 1]) << EDGE_R_SHIFT);
            if (muscl_on && mesh->edge_is_owned && !mesh->edge_is_owned[e]) lr |= EDGE_NOT_OWNED;
            if (muscl_on) {
              // the edge midpoint of ReconstructFaceValues (src/operator_fluxes_ceed.c:1169-1172); the kernel subtracts the
              // two cell centroids itself (1175-1178)
              const int32_t v0 = mesh->edge_vertex_ids[2 * e], v1 = mesh->edge_vertex_ids[2 * e + 1];
              if (v0 < 0 || v1 < 0 || v0 >= mesh->num_vertices || v1 >= mesh->num_vertices)
                return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "edge %d has vertex ids (%d,%d) out of range", e, v0, v1);
              e_mid.push_back(0.5 * (mesh->vertex_points[3 * (size_t)v0 + 0] + mesh->vertex_points[3 * (size_t)v1 + 0]));
              e_mid.push_back(0.5 * (mesh->vertex_points[3 * (size_t)v0 + 1] + mesh->vertex_points[3 * (size_t)v1 + 1]));
            }
          } else {
            if (muscl_on) e_mid.insert(e_mid.end(), 2, 0.0);
            const int32_t k = last - ni;
            e               = bedge[k];
            lr              = slot_of(bleft[k]) | ((uint32_t)nbk << EDGE_R_SHIFT) | EDGE_BOUNDARY;
            tile_bk.push_back(k);
            ++nbk;
          }
          // unit normal as one double: the smaller-magnitude component, the other is +-sqrt(1 - cs^2)
          const double ecn = mesh->edge_cn[e], esn = mesh->edge_sn[e];
          if (std::fabs(ecn) <= std::fabs(esn)) {
            lr |= EDGE_CS_IS_CN | (std::signbit(esn) ? EDGE_OTHER_NEG : 0u);
            e_cs.push_back(ecn);
          } else {
            lr |= (std::signbit(ecn) ? EDGE_OTHER_NEG : 0u);
            e_cs.push_back(esn);
          }
          e_lr.push_back(lr);
          e_pos.push_back(last);  // records of a tile are in loop order; across tiles only this table orders them
        }
        slot_ref[(size_t)(it.second >> 2) * 4 + (it.second & 3)] = (uint16_t)local;
      }
      int32_t nc2 = 0;
      if (muscl_on) {
        // fused second-order kernel: the stencil of every first-ring cell (LDS slots of its neighbours) and the tile's
        // second ring (neighbours of first-ring cells outside tile + ring 1)
        c_off[t]    = (int32_t)hcells2.size();
        touched2.clear();
        for (int32_t b = 0; b < nh; ++b) {
          const int32_t cell  = hcells[(size_t)tiles[t].h_off + b];
          uint16_t      ix[4] = {BN_NONE, BN_NONE, BN_NONE, BN_NONE};
          if (!mesh->cell_is_owned[cell]) {
            ix[0] = ix[1] = ix[2] = ix[3] = BN_GLOBAL;  // a ghost: its stencil is on another rank
          } else {
            const int32_t ob = mesh->cell_local_to_owned[cell];
            for (int32_t sl = 0; sl < S; ++sl) {
              const int64_t idx = (int64_t)sl * stride + ob;
              const int32_t id  = nbr[idx];
              if (id < 0) continue;
              const int32_t n = id & NBR_MASK;
              if (id & NBR_GHOST) halo_tile = true;  // the tile needs a ghost's state
              int32_t slot;
              const int32_t on = mesh->cell_is_owned[n] ? mesh->cell_local_to_owned[n] : -1;
              if (on >= base && on < base + cntc) slot = on - base;
              else if (hslot[n] >= 0) slot = TILE + hslot[n];
              else {
                if (hslot2[n] < 0) {
                  hslot2[n] = nc2++;
                  hcells2.push_back(n);
                  touched2.push_back(n);
                }
                slot = TILE + nh + hslot2[n];
              }
              ix[sl] = (uint16_t)slot;
            }
          }
          bn_idx.insert(bn_idx.end(), ix, ix + 4);
        }
        for (int32_t cell : touched2) hslot2[cell] = -1;
        hmax2 = std::max(hmax2, nh + nc2);
        tiles[t].cnt = halo_tile ? TILE_HALO_FLAG : 0u;
        if (halo_tile && (halo_tiles.empty() || halo_tiles.back() != t)) halo_tiles.push_back(t);
      }
      for (int32_t cell : touched) hslot[cell] = -1;
      // what layout_cut_tiles promised (the kernels' LDS planes and register rounds have exactly this room)
      const int32_t cap_h1 = S == 3 ? TILE_MAX_HALO_TRI : TILE_MAX_HALO_QUAD, cap_ring = S == 3 ? MUSCL_MAX_RING_TRI : MUSCL_MAX_RING_QUAD;
      if (cntc < 1 || cntc > TILE || local + 1 > TILE_MAX_REC || nh > cap_h1 || (muscl_on && nh + nc2 > cap_ring))
        return fail(RDYHIP_ERR_LIB, "internal error: tile %d (%d cells, %d edge records, %d + %d ring cells) exceeds the kernels' capacities", t, cntc, local + 1, nh, nc2);
      tiles[t].cnt |= (uint32_t)(local + 1) | ((uint32_t)nh << 11) | ((uint32_t)(cntc - 1) << 22);
      emax = std::max(emax, local + 1);
      hmax = std::max(hmax, nh);
      if ((int64_t)e_lr.size() > (int64_t)INT32_MAX - 4 * TILE) return fail(RDYHIP_ERR_ARG_SIZ, "too many tile edge records");
    }
    tiles[ntiles].e_off = (int32_t)e_lr.size();
    tiles[ntiles].h_off = (int32_t)hcells.size();
    tiles[ntiles].c_off = no;
    tile_boff[ntiles]   = (int32_t)tile_bk.size();
    tiles[ntiles].cnt   = 0;
    if (muscl_on) c_off[ntiles] = (int32_t)hcells2.size();
  }
  L.ntiles = ntiles; L.emax = emax; L.hmax = hmax; L.hmax2 = hmax2;
  L.hcells = std::move(hcells); L.tile_bk = std::move(tile_bk); L.tile_boff = std::move(tile_boff); L.halo_tiles = std::move(halo_tiles);
  L.hcells2 = std::move(hcells2); L.tile_c0 = std::move(tile_c0);
  L.c_off = std::move(c_off); L.e_cs = std::move(e_cs); L.e_mid = std::move(e_mid); L.tiles = std::move(tiles);
  // ---- extra Courant edges: the reference's interior loop covers every local internal edge with len / min(area_l, area_r)
  // (swe_petsc.c:275-296); the slots of the owned cells cover an edge from the owned side(s) only
  for (int32_t p = 0; p < ni; ++p) {
    const int32_t e = mesh->edge_internal_ids[p];
    const int32_t l = mesh->edge_cell_ids[2 * e], r = mesh->edge_cell_ids[2 * e + 1];
    if (r == -1) continue;
    const bool   lo = mesh->cell_is_owned[l] != 0, ro = mesh->cell_is_owned[r] != 0;
    const double al = mesh->cell_areas[l], ar = mesh->cell_areas[r];
    if (lo && ro) continue;
    if (lo != ro && !((lo ? ar : al) < (lo ? al : ar))) continue;  // the owned cell is the smaller one (or equal): its slot has len / min already
    // second order: only the rank that owns an edge reports it (swe_petsc.c:172-190) -- never an edge between two ghosts
    if (muscl_on && mesh->edge_is_owned && !mesh->edge_is_owned[e]) continue;
    if (muscl_on && !mesh->edge_is_owned && !lo && !ro) continue;
    L.x_lr.push_back(l);
    L.x_lr.push_back(r);
    L.x_pos.push_back(p);
    const double ecn = mesh->edge_cn[e], esn = mesh->edge_sn[e];
    if (std::fabs(ecn) <= std::fabs(esn)) {
      L.x_flags.push_back(EDGE_CS_IS_CN | (std::signbit(esn) ? EDGE_OTHER_NEG : 0u));
      L.x_cs.push_back(ecn);
    } else {
      L.x_flags.push_back(std::signbit(ecn) ? EDGE_OTHER_NEG : 0u);
      L.x_cs.push_back(esn);
    }
    L.x_cfac.push_back(mesh->edge_lengths[e] / std::min(al, ar));
    if (muscl_on) {
      const int32_t v0 = mesh->edge_vertex_ids[2 * e], v1 = mesh->edge_vertex_ids[2 * e + 1];
      if (v0 < 0 || v1 < 0 || v0 >= mesh->num_vertices || v1 >= mesh->num_vertices)
        return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "edge %d has vertex ids (%d,%d) out of range", e, v0, v1);
      L.x_mid.push_back(0.5 * (mesh->vertex_points[3 * (size_t)v0 + 0] + mesh->vertex_points[3 * (size_t)v1 + 0]));
      L.x_mid.push_back(0.5 * (mesh->vertex_points[3 * (size_t)v0 + 1] + mesh->vertex_points[3 * (size_t)v1 + 1]));
    }
  }
  L.e_pos = std::move(e_pos);
  L.e_pos.insert(L.e_pos.end(), L.x_pos.begin(), L.x_pos.end());  // extra edge i is "record" nrec + i
  L.e_lr = std::move(e_lr); L.slot_ref = std::move(slot_ref); L.bn_idx = std::move(bn_idx);
  return 0;
}

static int build_host_layout(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                             HostLayout &L) {
  int rc = layout_check_arguments(config, mesh, num_boundaries, boundaries);
  if (!rc) rc = layout_build_slots(config, mesh, num_boundaries, boundaries, L);
  if (!rc) rc = layout_build_tiles(config, mesh, L);
  if (rc) return rc;
  const int32_t no = L.no;
  const auto   &o2l      = L.o2l;
  const bool   hr_on     = config->well_balancing == RDYHIP_WELL_BALANCING_HR;
  const size_t lds_bytes = tiled_lds_bytes(L.S, hr_on);
  const size_t lds_muscl = !L.muscl_on ? 0 : (L.S == 3 ? MusclSoATri::lds_bytes : MusclSoAQuad::lds_bytes);
  static_assert(tiled_lds_bytes(4, true) <= 64 * 1024 && MusclSoATri::lds_bytes <= 64 * 1024 && MusclSoAQuad::lds_bytes <= 64 * 1024,
                "the kernels' LDS fits the default dynamic allocation");

  // ---- per-owned-cell geometry --------------------------------------------
  std::vector<double> dzdx((size_t)no), dzdy((size_t)no);
  for (int32_t o = 0; o < no; ++o) {
    dzdx[o] = mesh->cell_dz_dx[o2l[o]];
    dzdy[o] = mesh->cell_dz_dy[o2l[o]];
  }

  L.hr_on = hr_on; L.lds_bytes = lds_bytes; L.lds_muscl = lds_muscl;
  L.dzdx = std::move(dzdx); L.dzdy = std::move(dzdy);
  return 0;
}

extern "C" {

const char *rdyhip_last_error(void) { return g_err.c_str(); }
int32_t     rdyhip_version(void) { return RDYHIP_VERSION; }

int rdyhip_create(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                  RDyHipOperator *op_out) {
  if (!config || !mesh || !op_out) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_create");
  *op_out = nullptr;
  HostLayout L;
  {
    const int rc0 = build_host_layout(config, mesh, num_boundaries, boundaries, L);
    if (rc0) return rc0;
  }
  const int32_t nc = L.nc, no = L.no, ne = L.ne, ni = L.ni, K = L.K, S = L.S, ntiles = L.ntiles, emax = L.emax, hmax = L.hmax, hmax2 = L.hmax2;
  const int64_t stride = L.stride;
  const bool    prefix = L.prefix, hr_on = L.hr_on, muscl_on = L.muscl_on;
  const size_t  lds_bytes = L.lds_bytes, lds_muscl = L.lds_muscl;
  auto &o2l = L.o2l; auto &boff = L.boff; auto &nbr = L.nbr; auto &pos = L.pos; auto &btype = L.btype; auto &bleft = L.bleft; auto &bedge = L.bedge;
  auto &bghost = L.bghost; auto &halo = L.halo; auto &hcells = L.hcells; auto &tile_bk = L.tile_bk; auto &halo_tiles = L.halo_tiles;
  auto &hcells2 = L.hcells2; auto &c_off = L.c_off; auto &cn = L.cn; auto &sn = L.sn; auto &coef = L.coef;
  auto &bcn = L.bcn; auto &bsn = L.bsn; auto &e_cs = L.e_cs; auto &e_mid = L.e_mid; auto &dzdx = L.dzdx; auto &dzdy = L.dzdy;
  auto &tiles = L.tiles; auto &e_lr = L.e_lr; auto &slot_ref = L.slot_ref; auto &bn_idx = L.bn_idx;
  (void)ne;

  // ---- build the operator --------------------------------------------------
  RDyHipOperator op = new (std::nothrow) RDyHipOperator_s;
  if (!op) return fail(RDYHIP_ERR_MEM, "out of host memory");
  op->config     = *config;
  op->n_cells    = nc;
  op->n_owned    = no;
  op->S          = S;
  op->K          = K;
  op->n_internal = ni;
  op->stride     = stride;
  op->prefix     = prefix;
  op->n_halo     = (int32_t)halo.size();
  op->n_bghost   = (int32_t)bghost.size();
  op->ntiles       = ntiles;
  op->n_halo_tiles = (int32_t)halo_tiles.size();
  op->emax         = emax;
  op->hmax         = hmax;
  op->lds_bytes    = lds_bytes;
  {
    // The Euler-step kernels' u_out is what the next step reads: stored without the non-temporal hint it is still in the
    // Infinity Cache (256 MB) then -- when it fits beside what else lives there.  A sweep over 0.36 M .. 10 M cells in a time loop
    // (profiles/r05_uout_policy_sweep.txt): plain stores win 3-5 % from 1.2 M to 6 M cells (29 .. 150 MB of state), lose 0.3-3 %
    // below 1 M (the launch is over before a second pass could profit) and 2 % from 8 M cells.  So: plain stores for
    // 28 MB <= state <= RDYHIP_UOUT_CACHED_MAX_MB (default 144); RDYHIP_UOUT_CACHED=0 / 1 forces.
    double max_mb = 144.0;
    if (const char *e = getenv("RDYHIP_UOUT_CACHED_MAX_MB")) max_mb = atof(e);
    const double state_bytes = 24.0 * (double)op->n_cells;
    op->uout_cached = state_bytes >= 28.0 * 1048576.0 && state_bytes <= max_mb * 1048576.0;
    if (const char *e = getenv("RDYHIP_UOUT_CACHED")) op->uout_cached = atoi(e) != 0;
  }
  op->muscl       = muscl_on;
  op->hmax2       = hmax2;
  op->lds_muscl   = lds_muscl;
  op->nrec         = (int64_t)e_lr.size();
  op->nhalo_entries = (int64_t)hcells.size();
  {
    const char *kenv = getenv("RDYHIP_KERNEL");
    op->use_tiled    = !(kenv && strcmp(kenv, "cell") == 0);
    op->hr           = hr_on;
    if ((hr_on || muscl_on) && !op->use_tiled) {
      delete op;
      return fail(RDYHIP_ERR_USER, "hydrostatic reconstruction and second_order are implemented by the tiled kernels only (unset RDYHIP_KERNEL=cell)");
    }
  }
  int rc         = 0;
  if (hipGetDevice(&op->device) != hipSuccess) {
    delete op;
    return fail(RDYHIP_ERR_LIB, "hipGetDevice failed: no usable HIP device");
  }

  const int tiles_n = (no + BLOCK - 1) / BLOCK;
  const char *env   = getenv("RDYHIP_XCD_SWIZZLE");
  const bool  swz   = env ? atoi(env) != 0 : true;
  op->xcd_chunks    = (swz && tiles_n >= 64) ? (tiles_n + 7) / 8 : 0;
  op->grid          = op->xcd_chunks > 0 ? op->xcd_chunks * 8 : tiles_n;
  {
    // size of the persistent grid: resident workgroups per CU (occupancy query) x CUs
    int cus = 256, per_cu = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, op->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int q = 0;
    const void *kfn = (const void *)tiled_kernel_fn(S, config->source_method == RDYHIP_SOURCE_IMPLICIT_XQ2018 ? 1 : 0, true, hr_on);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kfn, TILE, lds_bytes) == hipSuccess && q > 0) per_cu = q;
    if (const char *e2 = getenv("RDYHIP_BLOCKS_PER_CU")) {
      if (atoi(e2) > 0) per_cu = atoi(e2);
    }
    op->pgrid            = std::max(8, cus * per_cu);
    if (muscl_on) {
      int qm = 0, per_cu_m = 2;
      const void *mfn = (const void *)muscl_kernel_fn(S, config->source_method == RDYHIP_SOURCE_IMPLICIT_XQ2018 ? 1 : 0, true, false, config->limiter);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&qm, mfn, TILE, lds_muscl) == hipSuccess && qm > 0) per_cu_m = qm;
      if (const char *e2 = getenv("RDYHIP_BLOCKS_PER_CU")) {
        if (atoi(e2) > 0) per_cu_m = atoi(e2);
      }
      op->pgrid_muscl = std::max(8, cus * per_cu_m);
    }
    op->tiled_xcd_chunks = (swz && ntiles >= 64) ? (ntiles + 7) / 8 : 0;
    if (const char *e3 = getenv("RDYHIP_INTERIOR_SHRINK")) op->interior_shrink = std::max(0, atoi(e3));  // measurement knob
    if (const char *e4 = getenv("RDYHIP_PGRID")) {  // measurement knob: the persistent grid itself
      if (atoi(e4) >= 8) op->pgrid = op->pgrid_muscl = atoi(e4) & ~7;
    }
    if (const char *e5 = getenv("RDYHIP_BALANCE_ROUNDS")) op->balance_rounds = atoi(e5) != 0;
  }
  const int maxgrid = std::max(std::max(std::max(op->grid, op->pgrid), op->pgrid_muscl), 1);

#define TRY_RC(x)     \
  do {                \
    rc = (x);         \
    if (rc) {         \
      delete op;      \
      return rc;      \
    }                 \
  } while (0)
  if (!prefix) TRY_RC(op->d_o2l.upload(o2l));
  TRY_RC(op->d_nbr.upload(nbr));
  TRY_RC(op->d_pos.upload(pos));
  TRY_RC(op->d_cn.upload(cn));
  TRY_RC(op->d_sn.upload(sn));
  TRY_RC(op->d_coef.upload(coef));
  TRY_RC(op->d_dzdx.upload(dzdx));
  TRY_RC(op->d_dzdy.upload(dzdy));
  TRY_RC(op->d_halo_list.upload(halo));
  TRY_RC(op->d_btype.upload(btype));
  TRY_RC(op->d_bleft.upload(bleft));
  TRY_RC(op->d_bghost_list.upload(bghost));
  TRY_RC(op->d_bcn.upload(bcn));
  TRY_RC(op->d_bsn.upload(bsn));
  {
    std::vector<double> area(mesh->cell_areas, mesh->cell_areas + nc);
    op->h_area.swap(area);
  }
  TRY_RC(op->d_tiles.upload(tiles));
  TRY_RC(op->d_halo_tiles.upload(halo_tiles));
  TRY_RC(op->d_e_lr.upload(e_lr));
  TRY_RC(op->d_e_pos.upload(L.e_pos));
  op->n_xedges = (int32_t)L.x_pos.size();
  TRY_RC(op->d_x_lr.upload(L.x_lr));
  TRY_RC(op->d_x_flags.upload(L.x_flags));
  TRY_RC(op->d_x_cs.upload(L.x_cs));
  TRY_RC(op->d_x_cfac.upload(L.x_cfac));
  TRY_RC(op->d_x_mid.upload(L.x_mid));
  TRY_RC(op->d_hcells.upload(hcells));
  TRY_RC(op->d_tile_bk.upload(tile_bk));
  TRY_RC(op->d_tile_boff.upload(L.tile_boff));
  op->h_tile_c0 = L.tile_c0;
  TRY_RC(op->d_e_cs.upload(e_cs));
  if (S == 3) {
    std::vector<uint32_t> ref3((size_t)no);
    for (int32_t o = 0; o < no; ++o) {
      uint32_t w = 0;
      for (int sl = 0; sl < 3; ++sl) {
        const uint16_t r = slot_ref[(size_t)o * 4 + sl];
        w |= (r == SLOT_EMPTY ? REF3_EMPTY : (uint32_t)r) << (10 * sl);
      }
      ref3[o] = w;
    }
    TRY_RC(op->d_slot_ref3.upload(ref3));
  } else {
    TRY_RC(op->d_slot_ref.upload(slot_ref));
  }
  if (hr_on) {
    std::vector<double> zc(mesh->cell_zc, mesh->cell_zc + nc);
    TRY_RC(op->d_zc_local.upload(zc));
  }
  if (muscl_on) {
    TRY_RC(op->d_grad.zeros((size_t)6 * nc));
    TRY_RC(op->d_e_mid.upload(e_mid));
    {
      std::vector<double> cxy((size_t)2 * nc);
      for (int32_t c = 0; c < nc; ++c) {
        cxy[2 * (size_t)c]     = mesh->cell_centroids[3 * (size_t)c + 0];
        cxy[2 * (size_t)c + 1] = mesh->cell_centroids[3 * (size_t)c + 1];
      }
      TRY_RC(op->d_cxy.upload(cxy));
    }
    TRY_RC(op->d_hcells2.upload(hcells2));
    TRY_RC(op->d_c_off.upload(c_off));
    TRY_RC(op->d_bn_idx.upload(bn_idx));
  }
  TRY_RC(op->d_mannings.zeros((size_t)no));
  TRY_RC(op->d_extsrc.zeros((size_t)3 * no));
  TRY_RC(op->d_bvalues.zeros((size_t)3 * K));
  TRY_RC(op->d_bflux.zeros((size_t)3 * K));
  TRY_RC(op->d_baccum.zeros((size_t)3 * K));
  TRY_RC(op->d_pv.zeros((size_t)3 * no));
  TRY_RC(op->d_blk_max.zeros((size_t)2 * maxgrid));  // two halves: see launch_rhs(bucket_half)
  TRY_RC(op->d_blk_pos.zeros((size_t)2 * maxgrid));
  TRY_RC(op->d_courant.zeros(1));
  {
    ColdArgs c{};
    c.nbr = op->d_nbr.p; c.cn = op->d_cn.p; c.sn = op->d_sn.p; c.pos = op->d_pos.p; c.e_pos = op->d_e_pos.p;
    c.btype = op->d_btype.p; c.bvalues = op->d_bvalues.p; c.bflux = op->d_bflux.p; c.baccum = op->d_baccum.p;
    c.tile_bk = op->d_tile_bk.p; c.tile_boff = op->d_tile_boff.p;
    c.n_xedges = op->n_xedges; c.x_rec0 = (int32_t)e_lr.size(); c.x_lr = op->d_x_lr.p; c.x_flags = op->d_x_flags.p; c.x_cs = op->d_x_cs.p;
    c.x_cfac = op->d_x_cfac.p; c.x_mid = op->d_x_mid.p;
    c.blk_max = op->d_blk_max.p; c.blk_pos = op->d_blk_pos.p;
    TRY_RC(op->d_cold.upload(std::vector<ColdArgs>(1, c)));
  }
#undef TRY_RC
  hipLaunchKernelGGL(courant_reset_kernel, dim3(4), dim3(1024), 0, 0, (int)op->d_blk_max.n, op->d_blk_max.p, op->d_blk_pos.p);
  if (hipDeviceSynchronize() != hipSuccess) {
    delete op;
    return fail(RDYHIP_ERR_LIB, "device synchronisation failed after create");
  }

  op->h_internal_edge.assign(mesh->edge_internal_ids, mesh->edge_internal_ids + ni);
  op->h_edge_cells.assign(mesh->edge_cell_ids, mesh->edge_cell_ids + 2 * (size_t)ne);
  op->h_bedge = bedge;
  op->h_boff  = boff;
  if (!prefix) op->h_l2o.assign(mesh->cell_local_to_owned, mesh->cell_local_to_owned + nc);
  if (mesh->cell_global_ids) op->h_cell_gid.assign(mesh->cell_global_ids, mesh->cell_global_ids + nc);
  if (mesh->edge_global_ids) op->h_edge_gid.assign(mesh->edge_global_ids, mesh->edge_global_ids + ne);

  op->device_bytes = op->d_o2l.bytes() + op->d_nbr.bytes() + op->d_pos.bytes() + op->d_cn.bytes() + op->d_sn.bytes() + op->d_coef.bytes() +
                     op->d_dzdx.bytes() + op->d_dzdy.bytes() + op->d_mannings.bytes() + op->d_extsrc.bytes() +
                     op->d_pv.bytes() + op->d_bvalues.bytes() + op->d_bflux.bytes() + op->d_baccum.bytes() + op->d_blk_max.bytes() +
                     op->d_blk_pos.bytes() + op->d_tiles.bytes() + op->d_e_lr.bytes() + op->d_e_pos.bytes() + op->d_hcells.bytes() + op->d_tile_bk.bytes() + op->d_tile_boff.bytes() +
                     op->d_e_cs.bytes() + op->d_slot_ref.bytes() + op->d_slot_ref3.bytes() + op->d_grad.bytes() + op->d_e_mid.bytes() +
                     op->d_cxy.bytes() + op->d_hcells2.bytes() + op->d_c_off.bytes() + op->d_bn_idx.bytes();
  *op_out = op;
  return 0;
}

int rdyhip_destroy(RDyHipOperator *op) {
  if (!op) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_destroy");
  if (*op) {
    (void)hipDeviceSynchronize();
    if ((*op)->fused_halo) halo_operator_gone((*op)->fused_halo);
    delete *op;
    *op = nullptr;
  }
  return 0;
}

int rdyhip_apply(RDyHipOperator op, double dt, const double *u_local, double *f_global, void *stream) {
  return launch_rhs(op, RDYHIP_PHASE_ALL, 0, 0, dt, u_local, f_global, (hipStream_t)stream);
}

int rdyhip_rhs_function(RDyHipOperator op, double dt, const double *u_local, double *f_global, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  op->courant = RDyHipCourant{0.0, -1, -1};
  // the diagnostic reset rides on the Courant finalize kernel (no extra launch)
  return launch_rhs(op, RDYHIP_PHASE_ALL, 1, 1, dt, u_local, f_global, (hipStream_t)stream);
}

int rdyhip_apply_phase(RDyHipOperator op, int32_t phase, int32_t flags, double dt, const double *u_local, double *f_global, void *stream) {
  if (phase != RDYHIP_PHASE_ALL && phase != RDYHIP_PHASE_INTERIOR && phase != RDYHIP_PHASE_HALO) return fail(RDYHIP_ERR_USER, "bad phase %d", phase);
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  const int reset = (flags & RDYHIP_PHASE_RESET_DIAGNOSTICS) ? 1 : 0;
  if (reset) op->courant = RDyHipCourant{0.0, -1, -1};
  if (reset && phase == RDYHIP_PHASE_HALO && (op->use_tiled ? op->n_halo_tiles == 0 : op->n_halo == 0)) {
    int rc = rdyhip_reset_diagnostics(op, stream);  // nothing to launch in this phase: reset on its own
    if (rc) return rc;
  }
  return launch_rhs(op, phase, (flags & RDYHIP_PHASE_OVERWRITE) ? 1 : 0, reset, dt, u_local, f_global, (hipStream_t)stream,
                    (flags & RDYHIP_PHASE_GRADIENTS_READY) != 0);
}

int rdyhip_euler_step(RDyHipOperator op, int32_t phase, int32_t flags, double dt, const double *u_local, double *u_local_out, double *f_global,
                      void *stream) {
  if (phase != RDYHIP_PHASE_ALL && phase != RDYHIP_PHASE_INTERIOR && phase != RDYHIP_PHASE_HALO) return fail(RDYHIP_ERR_USER, "bad phase %d", phase);
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (op->n_owned > 0 && (!u_local_out || u_local_out == u_local)) return fail(RDYHIP_ERR_USER, "rdyhip_euler_step needs a second state array (not in place)");
  const int reset = (flags & RDYHIP_PHASE_RESET_DIAGNOSTICS) ? 1 : 0;
  if (reset) op->courant = RDyHipCourant{0.0, -1, -1};
  if (reset && phase == RDYHIP_PHASE_HALO && (op->use_tiled ? op->n_halo_tiles == 0 : op->n_halo == 0)) {
    int rc = rdyhip_reset_diagnostics(op, stream);
    if (rc) return rc;
  }
  return launch_rhs(op, phase, 1, reset, dt, u_local, f_global, (hipStream_t)stream, (flags & RDYHIP_PHASE_GRADIENTS_READY) != 0, u_local_out);
}

int rdyhip_set_boundary_values(RDyHipOperator op, int32_t boundary, int32_t comp_offset, int32_t num_comp, int32_t num_edges, const double *values) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (boundary < 0 || boundary + 1 >= (int32_t)op->h_boff.size()) return fail(RDYHIP_ERR_USER, "Invalid boundary index %d", boundary);  // operator.c CheckOperatorBoundary
  const int32_t n = op->h_boff[boundary + 1] - op->h_boff[boundary];
  if (n != num_edges) return fail(RDYHIP_ERR_USER, "num_edges (%d) does not match boundary.num_edges (%d)", num_edges, n);  // operator.c:1052
  if (comp_offset < 0 || num_comp < 0 || comp_offset + num_comp > 3) return fail(RDYHIP_ERR_USER, "bad component range [%d,%d)", comp_offset, comp_offset + num_comp);
  if (n == 0 || num_comp == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
  // the apply calls run on the caller's streams (PETSc's and torch's are non-blocking): an RHS still in flight may be
  // reading the values this call replaces
  HIP_TRY(hipDeviceSynchronize());
  double *dst = op->d_bvalues.p + 3 * (size_t)op->h_boff[boundary];
  if (comp_offset == 0 && num_comp == 3) {
    HIP_TRY(hipMemcpy(dst, values, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy2D(dst + comp_offset, 3 * sizeof(double), values, num_comp * sizeof(double), num_comp * sizeof(double), n, hipMemcpyHostToDevice));
  }
  return 0;
}

int rdyhip_get_boundary_fluxes(RDyHipOperator op, int32_t boundary, int32_t accumulated, int32_t num_edges, double *fluxes) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (boundary < 0 || boundary + 1 >= (int32_t)op->h_boff.size()) return fail(RDYHIP_ERR_USER, "Invalid boundary index %d", boundary);
  const int32_t n = op->h_boff[boundary + 1] - op->h_boff[boundary];
  if (n != num_edges) return fail(RDYHIP_ERR_USER, "num_edges (%d) does not match boundary.num_edges (%d)", num_edges, n);
  if (n == 0) return 0;
  if (!fluxes) return fail(RDYHIP_ERR_USER, "null output");
  const double *src = (accumulated ? op->d_baccum.p : op->d_bflux.p) + 3 * (size_t)op->h_boff[boundary];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(fluxes, src, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

int rdyhip_reset_boundary_fluxes_accum(RDyHipOperator op) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(op->d_baccum.p, 0, std::max<size_t>(op->d_baccum.bytes(), sizeof(double))));
  return 0;
}

static int scatter_component(RDyHipOperator op, double *dst, int ncomp, int comp, int32_t n, const int32_t *ids, const double *values) {
  if (n < 0 || n > op->n_owned) return fail(RDYHIP_ERR_ARG_SIZ, "n (%d) exceeds the number of owned cells (%d)", n, op->n_owned);
  if (n == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
  if (ids) {
    for (int32_t i = 0; i < n; ++i)
      if (ids[i] < 0 || ids[i] >= op->n_owned) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "owned cell id %d out of range", ids[i]);
  }
  HIP_TRY(hipDeviceSynchronize());  // an RHS in flight on a non-blocking stream may still read the array (and the staging buffers)
  if (op->d_stage_vals.n < (size_t)n) {
    op->d_stage_vals.release();
    int rc = op->d_stage_vals.alloc((size_t)n);
    if (rc) return rc;
  }
  HIP_TRY(hipMemcpy(op->d_stage_vals.p, values, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  const int32_t *dids = nullptr;
  if (ids) {
    if (op->d_stage_ids.n < (size_t)n) {
      op->d_stage_ids.release();
      int rc = op->d_stage_ids.alloc((size_t)n);
      if (rc) return rc;
    }
    HIP_TRY(hipMemcpy(op->d_stage_ids.p, ids, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    dids = op->d_stage_ids.p;
  }
  hipLaunchKernelGGL(scatter_component_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, dids, op->d_stage_vals.p, dst, ncomp, comp);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

int rdyhip_set_external_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *owned_cell_ids, const double *values) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (comp < 0 || comp > 2) return fail(RDYHIP_ERR_USER, "bad source component %d", comp);
  return scatter_component(op, op->d_extsrc.p, 3, comp, n, owned_cell_ids, values);
}

int rdyhip_set_mannings(RDyHipOperator op, int32_t n, const int32_t *owned_cell_ids, const double *values) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  return scatter_component(op, op->d_mannings.p, 1, 0, n, owned_cell_ids, values);
}

// ---- the stream-ordered forms: no device-wide synchronisation, no blocking copy ---------------------------------------
// A slot of the staging ring with room for `bytes`, free to be written by the host (the work that read it last is through).
static int stage_acquire(RDyHipOperator op, size_t bytes, StageRing::Slot **out) {
  StageRing &r = op->stage;
  if (!r.copy) HIP_TRY(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
  StageRing::Slot &s = r.slot[r.next];
  r.next = (r.next + 1) % StageRing::N;
  if (!s.copied) {
    HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  if (s.used) HIP_TRY(hipEventSynchronize(s.done));
  s.used = false;
  if (s.cap < bytes) {
    if (s.h) HIP_TRY(hipHostFree(s.h));
    if (s.d) HIP_TRY(hipFree(s.d));
    s.h = s.d = nullptr;
    s.cap = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
    HIP_TRY(hipHostMalloc(&s.h, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s.d, cap));
    s.cap = cap;
  }
  *out = &s;
  return 0;
}
// host -> pinned -> device, `bytes` from `src` to offset `off` of the slot.  Large arrays go in 8-MB chunks through a few
// threads (one core copies ~10 GB/s: 8 B per cell take it as long as 25 RHS evaluations of that many cells), and the upload of
// a chunk starts as soon as it sits in pinned memory, beside the host copies of the chunks behind it: a caller that has just
// read the Courant struct back (adaptive dt: the device is idle) waits for max(host copy, DMA), not for their sum.
static int stage_push(RDyHipOperator op, StageRing::Slot *s, size_t off, const void *src, size_t bytes) {
  constexpr size_t CHUNK = 8u << 20;
  char            *h = (char *)s->h + off, *d = (char *)s->d + off;
  const size_t     nchunks = (bytes + CHUNK - 1) / CHUNK;
  const int        nt = (int)std::min<size_t>(4, nchunks);
  if (nt <= 1) {
    memcpy(h, src, bytes);
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, op->stage.copy));
    return 0;
  }
  std::vector<std::atomic<int>> ready(nchunks);
  for (auto &r : ready) r.store(0, std::memory_order_relaxed);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=, &ready] {
      for (size_t c = (size_t)t; c < nchunks; c += (size_t)nt) {
        const size_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
        memcpy(h + o, (const char *)src + o, len);
        ready[c].store(1, std::memory_order_release);
      }
    });
  hipError_t err = hipSuccess;
  for (size_t c = 0; c < nchunks; ++c) {
    while (!ready[c].load(std::memory_order_acquire)) std::this_thread::yield();
    const size_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
    if (err == hipSuccess) err = hipMemcpyAsync(d + o, h + o, len, hipMemcpyHostToDevice, op->stage.copy);
  }
  for (auto &t : th) t.join();
  HIP_TRY(err);
  return 0;
}
// everything pushed into the slot is on its way: `st` waits for the last upload
static int stage_ready(RDyHipOperator op, StageRing::Slot *s, hipStream_t st) {
  HIP_TRY(hipEventRecord(s->copied, op->stage.copy));
  HIP_TRY(hipStreamWaitEvent(st, s->copied, 0));
  return 0;
}
static int stage_release(StageRing::Slot *s, hipStream_t st) {
  HIP_TRY(hipEventRecord(s->done, st));
  s->used = true;
  return 0;
}

static int scatter_component_on(RDyHipOperator op, double *dst, int ncomp, int comp, int32_t n, const int32_t *ids, const double *values, hipStream_t st) {
  if (n < 0 || n > op->n_owned) return fail(RDYHIP_ERR_ARG_SIZ, "n (%d) exceeds the number of owned cells (%d)", n, op->n_owned);
  if (n == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
  if (ids) {
    for (int32_t i = 0; i < n; ++i)
      if (ids[i] < 0 || ids[i] >= op->n_owned) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "owned cell id %d out of range", ids[i]);
  }
  const size_t vbytes = sizeof(double) * (size_t)n, ibytes = ids ? sizeof(int32_t) * (size_t)n : 0;
  StageRing::Slot *s;
  int rc = stage_acquire(op, vbytes + ibytes, &s);
  if (rc) return rc;
  rc = stage_push(op, s, 0, values, vbytes);
  if (!rc && ids) rc = stage_push(op, s, vbytes, ids, ibytes);
  if (!rc) rc = stage_ready(op, s, st);
  if (rc) return rc;
  const int32_t *dids = ids ? (const int32_t *)((const char *)s->d + vbytes) : nullptr;
  hipLaunchKernelGGL(scatter_component_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, dids, (const double *)s->d, dst, ncomp, comp);
  HIP_TRY(hipGetLastError());
  return stage_release(s, st);
}

int rdyhip_set_external_source_on(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *owned_cell_ids, const double *values, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (comp < 0 || comp > 2) return fail(RDYHIP_ERR_USER, "bad source component %d", comp);
  return scatter_component_on(op, op->d_extsrc.p, 3, comp, n, owned_cell_ids, values, (hipStream_t)stream);
}

int rdyhip_set_mannings_on(RDyHipOperator op, int32_t n, const int32_t *owned_cell_ids, const double *values, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  return scatter_component_on(op, op->d_mannings.p, 1, 0, n, owned_cell_ids, values, (hipStream_t)stream);
}

int rdyhip_set_boundary_values_on(RDyHipOperator op, int32_t boundary, int32_t comp_offset, int32_t num_comp, int32_t num_edges, const double *values,
                                  void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (boundary < 0 || boundary + 1 >= (int32_t)op->h_boff.size()) return fail(RDYHIP_ERR_USER, "Invalid boundary index %d", boundary);
  const int32_t n = op->h_boff[boundary + 1] - op->h_boff[boundary];
  if (n != num_edges) return fail(RDYHIP_ERR_USER, "num_edges (%d) does not match boundary.num_edges (%d)", num_edges, n);
  if (comp_offset < 0 || num_comp < 0 || comp_offset + num_comp > 3) return fail(RDYHIP_ERR_USER, "bad component range [%d,%d)", comp_offset, comp_offset + num_comp);
  if (n == 0 || num_comp == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
  hipStream_t      st = (hipStream_t)stream;
  const size_t     bytes = sizeof(double) * (size_t)num_comp * (size_t)n;
  StageRing::Slot *s;
  int rc = stage_acquire(op, bytes, &s);
  if (rc) return rc;
  rc = stage_push(op, s, 0, values, bytes);
  if (!rc) rc = stage_ready(op, s, st);
  if (rc) return rc;
  double *dst = op->d_bvalues.p + 3 * (size_t)op->h_boff[boundary];
  if (comp_offset == 0 && num_comp == 3) {
    HIP_TRY(hipMemcpyAsync(dst, s->d, bytes, hipMemcpyDeviceToDevice, st));
  } else {
    HIP_TRY(hipMemcpy2DAsync(dst + comp_offset, 3 * sizeof(double), s->d, num_comp * sizeof(double), num_comp * sizeof(double), n, hipMemcpyDeviceToDevice, st));
  }
  return stage_release(s, st);
}

int rdyhip_refresh_field(RDyHipOperator op, RDyHipField field, const double *values, int64_t num_values, int32_t values_on_device, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  double *dst = nullptr;
  int64_t n   = 0;
  if (field != RDYHIP_FIELD_EXTERNAL_SOURCES && field != RDYHIP_FIELD_MANNINGS)
    return fail(RDYHIP_ERR_USER, "rdyhip_refresh_field: only the operator's input fields (external sources, Manning n) can be written");
  int rc = rdyhip_field_ptr(op, field, &dst, &n);
  if (rc) return rc;
  if (num_values != n) return fail(RDYHIP_ERR_ARG_SIZ, "%lld values for a device field of %lld", (long long)num_values, (long long)n);
  if (n == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
  hipStream_t  st = (hipStream_t)stream;
  const size_t bytes = sizeof(double) * (size_t)n;
  if (values_on_device) {
    HIP_TRY(hipMemcpyAsync(dst, values, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  StageRing::Slot *s;
  rc = stage_acquire(op, bytes, &s);
  if (rc) return rc;
  rc = stage_push(op, s, 0, values, bytes);
  if (!rc) rc = stage_ready(op, s, st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst, s->d, bytes, hipMemcpyDeviceToDevice, st));
  return stage_release(s, st);
}

// ---- device-side forcing ingestion (forcing_kernels.h) -------------------------------------------------
static int forcing_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }

static int forcing_source_args(RDyHipOperator op, int32_t comp, int32_t n) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (comp < 0 || comp > 2) return fail(RDYHIP_ERR_USER, "bad source component %d", comp);
  if (n < 0 || n > op->n_owned) return fail(RDYHIP_ERR_ARG_SIZ, "n (%d) exceeds the number of owned cells (%d)", n, op->n_owned);
  return 0;
}

int rdyhip_forcing_fill_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *d_owned_cell_ids, double value, void *stream) {
  int rc = forcing_source_args(op, comp, n);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(forcing_fill_kernel, dim3(forcing_grid(n)), dim3(256), 0, (hipStream_t)stream, n, d_owned_cell_ids, value, op->d_extsrc.p, 3, comp);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_forcing_gather_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *d_owned_cell_ids, const double *d_data,
                                 const int32_t *d_data2mesh_idx, int64_t stride, int64_t offset, double scale, void *stream) {
  int rc = forcing_source_args(op, comp, n);
  if (rc || n == 0) return rc;
  if (!d_data || !d_data2mesh_idx) return fail(RDYHIP_ERR_USER, "null dataset or map");
  if (stride < 1 || offset < 0) return fail(RDYHIP_ERR_USER, "bad stride/offset (%lld, %lld)", (long long)stride, (long long)offset);
  if (scale == 1.0)
    hipLaunchKernelGGL(forcing_gather_kernel<false>, dim3(forcing_grid(n)), dim3(256), 0, (hipStream_t)stream, n, d_owned_cell_ids, d_data,
                       d_data2mesh_idx, stride, offset, scale, op->d_extsrc.p, 3, comp);
  else
    hipLaunchKernelGGL(forcing_gather_kernel<true>, dim3(forcing_grid(n)), dim3(256), 0, (hipStream_t)stream, n, d_owned_cell_ids, d_data,
                       d_data2mesh_idx, stride, offset, scale, op->d_extsrc.p, 3, comp);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int forcing_boundary_args(RDyHipOperator op, int32_t boundary, int32_t num_edges, int32_t *n) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (boundary < 0 || boundary + 1 >= (int32_t)op->h_boff.size()) return fail(RDYHIP_ERR_USER, "Invalid boundary index %d", boundary);
  *n = op->h_boff[boundary + 1] - op->h_boff[boundary];
  if (*n != num_edges) return fail(RDYHIP_ERR_USER, "num_edges (%d) does not match boundary.num_edges (%d)", num_edges, *n);
  return 0;
}

int rdyhip_forcing_fill_boundary(RDyHipOperator op, int32_t boundary, int32_t num_edges, double h, void *stream) {
  int32_t n  = 0;
  int     rc = forcing_boundary_args(op, boundary, num_edges, &n);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(forcing_fill_boundary_kernel, dim3(forcing_grid(3 * (int64_t)n)), dim3(256), 0, (hipStream_t)stream, n, h,
                     op->d_bvalues.p + 3 * (size_t)op->h_boff[boundary]);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_forcing_gather_boundary(RDyHipOperator op, int32_t boundary, int32_t num_edges, const double *d_data, const int32_t *d_data2mesh_idx,
                                   int64_t stride, int64_t offset, void *stream) {
  int32_t n  = 0;
  int     rc = forcing_boundary_args(op, boundary, num_edges, &n);
  if (rc || n == 0) return rc;
  if (!d_data || !d_data2mesh_idx) return fail(RDYHIP_ERR_USER, "null dataset or map");
  if (stride < 3 || offset < 0) return fail(RDYHIP_ERR_USER, "bad stride/offset (%lld, %lld)", (long long)stride, (long long)offset);
  hipLaunchKernelGGL(forcing_gather_boundary_kernel, dim3(forcing_grid(3 * (int64_t)n)), dim3(256), 0, (hipStream_t)stream, n, d_data,
                     d_data2mesh_idx, stride, offset, op->d_bvalues.p + 3 * (size_t)op->h_boff[boundary]);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_forcing_nearest_map(int32_t n, const double *d_xc, const double *d_yc, int32_t ndata, const double *d_data_xc, const double *d_data_yc,
                               double min_dist0, int32_t *d_map, void *stream) {
  if (n < 0 || ndata < 0) return fail(RDYHIP_ERR_ARG_SIZ, "negative count");
  if (n == 0 || ndata == 0) return 0;
  if (!d_xc || !d_yc || !d_data_xc || !d_data_yc || !d_map) return fail(RDYHIP_ERR_USER, "null argument");
  hipLaunchKernelGGL(forcing_nearest_kernel, dim3((n + NN_BLOCK - 1) / NN_BLOCK), dim3(NN_BLOCK), 0, (hipStream_t)stream, n, d_xc, d_yc, ndata,
                     d_data_xc, d_data_yc, min_dist0, d_map);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_field_ptr(RDyHipOperator op, RDyHipField field, double **device_ptr, int64_t *num_values) {
  if (!op || !device_ptr) return fail(RDYHIP_ERR_USER, "null argument");
  int64_t n = 0;
  switch (field) {
    case RDYHIP_FIELD_PRIMITIVE_VARIABLES: *device_ptr = op->d_pv.p; n = 3 * (int64_t)op->n_owned; break;
    case RDYHIP_FIELD_EXTERNAL_SOURCES: *device_ptr = op->d_extsrc.p; n = 3 * (int64_t)op->n_owned; break;
    case RDYHIP_FIELD_MANNINGS: *device_ptr = op->d_mannings.p; n = op->n_owned; break;
    case RDYHIP_FIELD_FLUX_DIVERGENCE:
      if (!op->keep_fdiv) return fail(RDYHIP_ERR_USER, "flux divergence is not enabled (rdyhip_enable_flux_divergence)");
      *device_ptr = op->d_fdiv.p;
      n           = 3 * (int64_t)op->n_owned;
      break;
    case RDYHIP_FIELD_GRADIENTS:
      if (!op->muscl) return fail(RDYHIP_ERR_USER, "the operator was not created with second_order");
      *device_ptr = op->d_grad.p;
      n           = 6 * (int64_t)op->n_cells;
      break;
    default: return fail(RDYHIP_ERR_USER, "unknown field %d", (int)field);
  }
  if (num_values) *num_values = n;
  return 0;
}

int rdyhip_enable_flux_divergence(RDyHipOperator op, int32_t enable) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (enable && !op->d_fdiv.p) {
    int rc = op->d_fdiv.zeros((size_t)3 * op->n_owned);
    if (rc) return rc;
  }
  op->keep_fdiv = enable != 0;
  return 0;
}

int rdyhip_reset_diagnostics(RDyHipOperator op, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  hipLaunchKernelGGL(courant_reset_kernel, dim3(4), dim3(1024), 0, (hipStream_t)stream, (int)op->d_blk_max.n, op->d_blk_max.p, op->d_blk_pos.p);
  HIP_TRY(hipGetLastError());
  op->courant = RDyHipCourant{0.0, -1, -1};
  return 0;
}

int rdyhip_update_diagnostics(RDyHipOperator op, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  // the RHS launches keep one running (max, first position) bucket per workgroup slot; they are merged only here
  hipLaunchKernelGGL(courant_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)op->d_blk_max.n, op->d_blk_max.p, op->d_blk_pos.p,
                     op->d_courant.p);
  HIP_TRY(hipGetLastError());
  DeviceCourant dc;
  HIP_TRY(hipMemcpyAsync(&dc, op->d_courant.p, sizeof(dc), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  RDyHipCourant out{dc.max_courant, -1, -1};
  if (dc.pos >= 0) {
    int32_t edge, cell;
    if (dc.pos < op->n_internal) {
      edge             = op->h_internal_edge[dc.pos];
      const int32_t l  = op->h_edge_cells[2 * (size_t)edge], r = op->h_edge_cells[2 * (size_t)edge + 1];
      cell             = (op->h_area[l] < op->h_area[r]) ? l : r;  // swe_petsc.c:294-295
    } else {
      const int32_t k = dc.pos - op->n_internal;
      edge            = op->h_bedge[k];
      cell            = op->h_edge_cells[2 * (size_t)edge];
    }
    out.global_edge_id = op->h_edge_gid.empty() ? edge : op->h_edge_gid[edge];
    out.global_cell_id = op->h_cell_gid.empty() ? cell : op->h_cell_gid[cell];
  }
  op->courant = out;
  return 0;
}

int rdyhip_get_diagnostics(RDyHipOperator op, RDyHipCourant *courant) {
  if (!op || !courant) return fail(RDYHIP_ERR_USER, "null argument");
  *courant = op->courant;
  return 0;
}

int rdyhip_pack_cells(const double *u_local, const int32_t *cell_ids, int32_t n, double *buf, void *stream) {
  if (n < 0) return fail(RDYHIP_ERR_ARG_SIZ, "negative count");
  if (n == 0) return 0;
  if (!u_local || !cell_ids || !buf) return fail(RDYHIP_ERR_USER, "null argument");
  hipLaunchKernelGGL(pack_cells_kernel, dim3((unsigned)((3 * (int64_t)n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, u_local, cell_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_unpack_cells(double *u_local, const int32_t *cell_ids, int32_t n, const double *buf, void *stream) {
  if (n < 0) return fail(RDYHIP_ERR_ARG_SIZ, "negative count");
  if (n == 0) return 0;
  if (!u_local || !cell_ids || !buf) return fail(RDYHIP_ERR_USER, "null argument");
  hipLaunchKernelGGL(unpack_cells_kernel, dim3((unsigned)((3 * (int64_t)n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, u_local, cell_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_pack_rows(const double *src, int32_t ncomp, const int32_t *row_ids, int32_t n, double *buf, void *stream) {
  if (n < 0 || ncomp < 1) return fail(RDYHIP_ERR_ARG_SIZ, "bad count");
  if (n == 0) return 0;
  if (!src || !row_ids || !buf) return fail(RDYHIP_ERR_USER, "null argument");
  const int64_t tot = (int64_t)n * ncomp;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, ncomp, src, row_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_unpack_rows(double *dst, int32_t ncomp, const int32_t *row_ids, int32_t n, const double *buf, void *stream) {
  if (n < 0 || ncomp < 1) return fail(RDYHIP_ERR_ARG_SIZ, "bad count");
  if (n == 0) return 0;
  if (!dst || !row_ids || !buf) return fail(RDYHIP_ERR_USER, "null argument");
  const int64_t tot = (int64_t)n * ncomp;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, ncomp, dst, row_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_compute_gradients(RDyHipOperator op, int32_t phase, const double *u_local, void *stream) {
  return launch_gradients(op, phase, u_local, (hipStream_t)stream);
}

int rdyhip_axpy_owned(RDyHipOperator op, double dt, const double *f_global, double *u_local, void *stream) {
  if (!op || !f_global || !u_local) return fail(RDYHIP_ERR_USER, "null argument");
  if (op->n_owned == 0) return 0;
  const int64_t n3 = 3 * (int64_t)op->n_owned;
  hipLaunchKernelGGL(axpy_owned_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op->n_owned, op->prefix ? nullptr : op->d_o2l.p, dt,
                     f_global, u_local);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_copy_owned_rows(RDyHipOperator op, const double *u_global, double *u_local, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (op->n_owned == 0) return 0;
  if (!u_global || !u_local) return fail(RDYHIP_ERR_USER, "null argument");
  if (op->prefix) {  // the owned cells are the first rows of the local vector: one contiguous copy
    if (u_global != u_local)
      HIP_TRY(hipMemcpyAsync(u_local, u_global, sizeof(double) * 3 * (size_t)op->n_owned, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
  }
  if (u_global == u_local) return fail(RDYHIP_ERR_USER, "rdyhip_copy_owned_rows in place needs owned cells numbered first");
  const int64_t n3 = 3 * (int64_t)op->n_owned;
  hipLaunchKernelGGL(copy_owned_rows_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op->n_owned, op->d_o2l.p, u_global, u_local);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_probe_layout(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                        RDyHipLayoutInfo *info) {
  if (!info) return fail(RDYHIP_ERR_USER, "null argument");
  HostLayout L;
  const int  rc = build_host_layout(config, mesh, num_boundaries, boundaries, L);
  if (rc) return rc;
  RDyHipOperator_s tmp;  // scalars only: nothing is allocated on a device
  tmp.n_cells = L.nc; tmp.n_owned = L.no; tmp.S = L.S; tmp.K = L.K; tmp.n_halo = (int32_t)L.halo.size(); tmp.use_tiled = true;
  tmp.ntiles = L.ntiles; tmp.n_halo_tiles = (int32_t)L.halo_tiles.size(); tmp.emax = L.emax; tmp.hmax = L.hmax;
  tmp.nhalo_entries = (int64_t)L.hcells.size(); tmp.nrec = (int64_t)L.e_lr.size(); tmp.prefix = L.prefix; tmp.muscl = L.muscl_on;
  tmp.hmax2 = L.hmax2; tmp.d_hcells2.n = L.hcells2.size(); tmp.lds_bytes = L.lds_bytes; tmp.lds_muscl = L.lds_muscl;
  const int rc2 = rdyhip_layout_info(&tmp, info);
  tmp.d_hcells2.n = 0;
  return rc2;
}

int rdyhip_layout_info(RDyHipOperator op, RDyHipLayoutInfo *info) {
  if (!op || !info) return fail(RDYHIP_ERR_USER, "null argument");
  info->num_owned_cells    = op->n_owned;
  info->num_cells          = op->n_cells;
  info->slots_per_cell     = op->S;
  info->num_boundary_edges = op->K;
  info->num_halo_cells     = op->n_halo;
  info->tiled_kernel       = op->use_tiled ? 1 : 0;
  info->num_tiles          = op->ntiles;
  info->num_halo_tiles     = op->n_halo_tiles;
  info->max_tile_edges     = op->emax;
  info->max_tile_halo_cells = op->hmax;
  info->num_halo_entries   = op->nhalo_entries;
  info->num_edge_records   = op->nrec;
  info->owned_is_prefix    = op->prefix ? 1 : 0;
  info->device_bytes       = op->device_bytes;
  info->second_order_fused   = op->muscl ? 1 : 0;
  info->max_tile_ring2_cells = op->hmax2;
  info->persistent_grid      = op->muscl ? op->pgrid_muscl : op->pgrid;
  info->lds_bytes            = (int32_t)(op->muscl ? op->lds_muscl : op->lds_bytes);
  info->lds_fixed_layout     = op->use_tiled ? 1 : 0;  // every tile is cut to the kernels' fixed LDS capacities
  // u (own cell) 24 + slots S*(4+8+8+8) + dz 16 + n 8 + ext src 24 + F 24 + pv 24 (+4 for o2l)
  if (op->use_tiled) {
    // u 24 + slot refs 4 (8 for quads) + coef S*8 + dz 16 + n 8 + ext src 24 + F 24 + pv 24 (+4 for o2l) per cell;
    // 12 B per edge record; 4 B id per halo-cell entry (the halo states themselves are mostly L2 hits)
    info->bytes_per_apply = (int64_t)op->n_owned * (24 + (op->S == 3 ? 4 : 8) + op->S * 8 + 16 + 8 + 24 + 24 + 24 + (op->prefix ? 0 : 4)) +
                            op->nrec * 12 + op->nhalo_entries * 4 + (int64_t)op->ntiles * 16;
  } else {
    info->bytes_per_apply = (int64_t)op->n_owned * (24 + op->S * 28 + 16 + 8 + 24 + 24 + 24 + (op->prefix ? 0 : 4));
  }
  if (op->muscl) {
    // + the cell centroid (16 B per cell) and the edge midpoint (16 B per edge record); fused: second-ring ids and the
    // first-ring stencils (8 B per halo entry); split: the gradient array written and read (96) + the state, the centroid and
    // the neighbour ids read a second time (24 + 16 + S*4)
    info->bytes_per_apply += (int64_t)op->n_owned * 16 + op->nrec * 16;
    info->bytes_per_apply += (int64_t)op->d_hcells2.n * 4 + op->nhalo_entries * 8;
  }
  return 0;
}

}  // extern "C"

#include "halo_exchange.h"
#include "halo_plan.h"
