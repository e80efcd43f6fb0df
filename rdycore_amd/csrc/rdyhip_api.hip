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
#include <vector>

#include "rdyhip.h"
#include "swe_device.h"

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

constexpr int BLOCK = 256;

// persistent Courant diagnostic on the device
struct DeviceCourant {
  double  max_courant;
  int32_t pos;  // position of the edge in the reference's loop order, -1 = none
  int32_t pad;
};

// everything a kernel needs, passed by value
struct KernelArgs {
  int32_t        n_owned;    // owned cells
  int32_t        n_work;     // threads with work: n_owned, or the length of `list`
  int64_t        stride;     // distance between slot planes
  const int32_t *list;       // owned-cell ids to process, or nullptr for 0..n_owned-1
  const int32_t *o2l;        // owned -> local cell id, or nullptr if the identity
  const int32_t *nbr;        // [S][stride]
  const double  *cn, *sn, *coef;
  const int32_t *pos;        // [S][stride] loop position of each slot's edge (Courant tie-break only)
  const double  *dzdx, *dzdy;  // [n_owned]
  const double  *mannings;   // [n_owned]
  const double  *extsrc;     // [n_owned][3]
  const double  *area_local; // [num_cells]
  const int32_t *btype;      // [K] condition type of boundary edge k
  const double  *bvalues;    // [K][3]
  double        *bflux;      // [K][3]
  double        *baccum;     // [K][3]
  double        *pv;         // [n_owned][3]
  double        *fdiv;       // [n_owned][3] or nullptr
  double        *blk_max;    // [grid]
  int32_t       *blk_pos;    // [grid]
  double         tiny_h, h_anuga_sq, xq_thresh;
  int32_t        phase;      // RDYHIP_PHASE_*
  int32_t        overwrite;  // 1: f = rhs, 0: f += rhs
  int32_t        xcd_chunks; // >0: blocks are dealt to XCDs in contiguous chunks of this many tiles
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

// One thread = one owned cell: all edge fluxes of the cell (ApplyInteriorFlux +
// ApplyBoundaryFlux, src/swe/swe_petsc.c:215-316, 506-630), then the source
// term on the in-register flux sum (ApplySource*, 704-932), then the block's
// share of the Courant-number max (289-296, 595-600).
template <int S, int SRC>
__global__ __launch_bounds__(BLOCK) void swe_rhs_kernel(const KernelArgs a, const double dt, const double *__restrict__ u,
                                                        double *__restrict__ f) {
  // XCD-aware tile mapping: consecutive block ids are dealt round-robin to the
  // 8 XCDs, so give each XCD a contiguous range of tiles (neighbour gathers
  // then hit that XCD's own L2).
  int tile = blockIdx.x;
  if (a.xcd_chunks > 0) tile = (blockIdx.x & 7) * a.xcd_chunks + (blockIdx.x >> 3);
  const int i = tile * BLOCK + threadIdx.x;

  double best      = 0.0;  // largest Courant number seen by this thread (> 0 only)
  int    best_slot = -1;
  int    o         = 0;

  bool active = i < a.n_work;
  int32_t id[S];
  if (active) {
    o = a.list ? a.list[i] : i;
    bool has_ghost = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      id[s] = a.nbr[s * a.stride + o];
      has_ghost |= (id[s] >= 0) && (id[s] & NBR_GHOST);
    }
    if (a.phase == RDYHIP_PHASE_INTERIOR && has_ghost) active = false;
    if (a.phase == RDYHIP_PHASE_HALO && !has_ghost) active = false;
  }

  if (active) {
    const int    c  = a.o2l ? a.o2l[o] : o;
    const double h  = u[3 * (int64_t)c + 0];
    const double hu = u[3 * (int64_t)c + 1];
    const double hv = u[3 * (int64_t)c + 2];
    const RiemannSide self = riemann_side(h, hu, hv, a.tiny_h, a.h_anuga_sq);

    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    if (!a.overwrite) {
      acc0 = f[3 * (int64_t)o + 0];
      acc1 = f[3 * (int64_t)o + 1];
      acc2 = f[3 * (int64_t)o + 2];
    }

#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int32_t nid = id[s];
      if (S > 3 && nid == NBR_EMPTY) continue;
      const double cn   = a.cn[s * a.stride + o];
      const double sn   = a.sn[s * a.stride + o];
      const double coef = a.coef[s * a.stride + o];
      RoeFlux      fl;
      bool         wet;
      double       cfac = fabs(coef);  // len / area_self
      if (nid >= 0) {
        const int         n     = nid & NBR_MASK;
        const double      hn    = u[3 * (int64_t)n + 0];
        const double      hun   = u[3 * (int64_t)n + 1];
        const double      hvn   = u[3 * (int64_t)n + 2];
        const RiemannSide other = riemann_side(hn, hun, hvn, a.tiny_h, a.h_anuga_sq);
        const bool        self_left = coef < 0.0;
        RiemannSide       L, R;
        L.h = self_left ? self.h : other.h;  R.h = self_left ? other.h : self.h;
        L.u = self_left ? self.u : other.u;  R.u = self_left ? other.u : self.u;
        L.v = self_left ? self.v : other.v;  R.v = self_left ? other.v : self.v;
        L.sqh = self_left ? self.sqh : other.sqh;  R.sqh = self_left ? other.sqh : self.sqh;
        L.c = self_left ? self.c : other.c;  R.c = self_left ? other.c : self.c;
        fl  = roe_flux(L, R, sn, cn);
        wet = !(R.h < a.tiny_h && L.h < a.tiny_h);
        if (nid & NBR_GHOST) {
          // the ghost side is not visited on this rank: use len / min(area_l, area_r)
          const double as = a.area_local[c], an = a.area_local[n];
          if (an < as) cfac = cfac * (as / an);
        }
      } else {
        const int    k  = -1 - nid;
        BoundaryFlux bf = boundary_flux(a.btype[k], true, self, a.bvalues + 3 * (int64_t)k, sn, cn, a.tiny_h, a.h_anuga_sq);
        fl              = bf.flux;
        wet             = bf.wet;
        // boundary_fluxes[b] and VecAXPY(boundary_fluxes_accum, dt, boundary_fluxes), swe_petsc.c:574, 623
        a.bflux[3 * (int64_t)k + 0] = fl.f0;
        a.bflux[3 * (int64_t)k + 1] = fl.f1;
        a.bflux[3 * (int64_t)k + 2] = fl.f2;
        a.baccum[3 * (int64_t)k + 0] += dt * fl.f0;
        a.baccum[3 * (int64_t)k + 1] += dt * fl.f1;
        a.baccum[3 * (int64_t)k + 2] += dt * fl.f2;
      }
      if (wet) {
        acc0 += fl.f0 * coef;
        acc1 += fl.f1 * coef;
        acc2 += fl.f2 * coef;
        const double cnum = fl.amax * cfac * dt;
        if (cnum > best) {
          best      = cnum;
          best_slot = s;
        }
      }
    }

    // ---- source term on the in-register flux sum (operator.c:663: the source reads the pre-source F)
    const double bedx = a.dzdx[o] * GRAVITY * h;
    const double bedy = a.dzdy[o] * GRAVITY * h;
    double       tbx = 0.0, tby = 0.0;
    if (h >= a.tiny_h) {
      const double n = a.mannings[o];
      if (SRC == RDYHIP_SOURCE_SEMI_IMPLICIT) friction_semi_implicit(h, hu, hv, n, dt, acc1, acc2, bedx, bedy, tbx, tby);
      else friction_xq2018(h, hu, hv, n, dt, a.xq_thresh, acc1, acc2, bedx, bedy, tbx, tby);
    }
    if (a.fdiv) {
      a.fdiv[3 * (int64_t)o + 0] = acc0;
      a.fdiv[3 * (int64_t)o + 1] = acc1;
      a.fdiv[3 * (int64_t)o + 2] = acc2;
    }
    const double s0 = a.extsrc[3 * (int64_t)o + 0];
    const double s1 = a.extsrc[3 * (int64_t)o + 1];
    const double s2 = a.extsrc[3 * (int64_t)o + 2];
    f[3 * (int64_t)o + 0] = acc0 + s0;
    f[3 * (int64_t)o + 1] = acc1 + (-bedx - tbx + s1);
    f[3 * (int64_t)o + 2] = acc2 + (-bedy - tby + s2);

    // primitive variables (swe_petsc.c:788-791): the same regularised velocities
    // the Riemann states use, zero below tiny_h
    a.pv[3 * (int64_t)o + 0] = h;
    a.pv[3 * (int64_t)o + 1] = self.u;
    a.pv[3 * (int64_t)o + 2] = self.v;
  }

  // ---- block reduction of the Courant number: max value, then the smallest
  // loop position among the lanes that hold it (the reference keeps the first
  // edge that reaches the max, swe_petsc.c:291).
  __shared__ double s_max[BLOCK / 64];
  __shared__ int    s_pos[BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double    wmax = wave_max(best);
  if (lane == 0) s_max[wave] = wmax;
  __syncthreads();
  double bmax = s_max[0];
#pragma unroll
  for (int w = 1; w < BLOCK / 64; ++w) bmax = fmax(bmax, s_max[w]);
  int p = INT32_MAX;
  if (best_slot >= 0 && best == bmax) p = a.pos[best_slot * a.stride + o];
  p = wave_min(p);
  if (lane == 0) s_pos[wave] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    int bp = s_pos[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) bp = min(bp, s_pos[w]);
    a.blk_max[blockIdx.x] = bmax;
    a.blk_pos[blockIdx.x] = (bmax > 0.0) ? bp : -1;
  }
}

// merges the per-block partials into the persistent diagnostic (reset != 0:
// the diagnostic is first reset, ResetOperatorDiagnostics src/operator.c:772-784)
__global__ __launch_bounds__(1024) void courant_finalize_kernel(int nblk, const double *__restrict__ blk_max, const int32_t *__restrict__ blk_pos,
                                                               DeviceCourant *diag, int reset) {
  double m = 0.0;
  int    p = INT32_MAX;
  constexpr int U = 8;  // independent loads in flight per thread
  for (int base = threadIdx.x; base < nblk; base += 1024 * U) {
    double v[U];
    int    q[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int i = base + j * 1024;
      v[j]        = i < nblk ? blk_max[i] : 0.0;
      q[j]        = i < nblk ? blk_pos[i] : INT32_MAX;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (v[j] > m || (v[j] == m && v[j] > 0.0 && q[j] < p)) {
        m = v[j];
        p = q[j];
      }
    }
  }
  __shared__ double s_max[16];
  __shared__ int    s_pos[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double wm = wave_max(m);
  int          wp = (m == wm && m > 0.0) ? p : INT32_MAX;
  wp              = wave_min(wp);
  if (lane == 0) {
    s_max[wave] = wm;
    s_pos[wave] = wp;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double bm = 0.0;
    int    bp = INT32_MAX;
    for (int w = 0; w < 16; ++w) {
      if (s_max[w] > bm || (s_max[w] == bm && bm > 0.0 && s_pos[w] < bp)) {
        bm = s_max[w];
        bp = s_pos[w];
      }
    }
    double cur_max = reset ? 0.0 : diag->max_courant;
    int    cur_pos = reset ? -1 : diag->pos;
    if (bm > cur_max || (bm == cur_max && bm > 0.0 && bp < cur_pos)) {
      cur_max = bm;
      cur_pos = bp;
    }
    diag->max_courant = cur_max;
    diag->pos         = cur_pos;
  }
}

__global__ void courant_reset_kernel(DeviceCourant *diag) {
  diag->max_courant = 0.0;
  diag->pos         = -1;
}

// boundary edges whose left cell is a ghost: the reference still evaluates
// their Riemann problem into boundary_fluxes[b] (swe_petsc.c:574) although
// nothing is accumulated into F (588).  Diagnostic output only.
__global__ void boundary_ghost_kernel(int n, const int32_t *__restrict__ klist, const int32_t *__restrict__ bleft, const int32_t *__restrict__ btype,
                                      const double *__restrict__ bcn, const double *__restrict__ bsn, const double *__restrict__ bvalues,
                                      double *__restrict__ bflux, double *__restrict__ baccum, const double *__restrict__ u, double dt, double tiny_h,
                                      double h_anuga_sq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int    k = klist[i];
  const int    c = bleft[k];
  const double h = u[3 * (int64_t)c + 0], hu = u[3 * (int64_t)c + 1], hv = u[3 * (int64_t)c + 2];
  const RiemannSide L  = riemann_side(h, hu, hv, tiny_h, h_anuga_sq);
  BoundaryFlux      bf = boundary_flux(btype[k], false, L, bvalues + 3 * (int64_t)k, bsn[k], bcn[k], tiny_h, h_anuga_sq);
  bflux[3 * (int64_t)k + 0] = bf.flux.f0;
  bflux[3 * (int64_t)k + 1] = bf.flux.f1;
  bflux[3 * (int64_t)k + 2] = bf.flux.f2;
  baccum[3 * (int64_t)k + 0] += dt * bf.flux.f0;
  baccum[3 * (int64_t)k + 1] += dt * bf.flux.f1;
  baccum[3 * (int64_t)k + 2] += dt * bf.flux.f2;
}

__global__ void pack_cells_kernel(int n, const double *__restrict__ u, const int32_t *__restrict__ ids, double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n) return;
  const int cell = i / 3, comp = i - 3 * cell;
  buf[i] = u[3 * (int64_t)ids[cell] + comp];
}
__global__ void unpack_cells_kernel(int n, double *__restrict__ u, const int32_t *__restrict__ ids, const double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n) return;
  const int cell = i / 3, comp = i - 3 * cell;
  u[3 * (int64_t)ids[cell] + comp] = buf[i];
}
__global__ void axpy_owned_kernel(int n_owned, const int32_t *__restrict__ o2l, double dt, const double *__restrict__ f, double *__restrict__ u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n_owned) return;
  if (o2l) {
    const int o = i / 3, comp = i - 3 * o;
    u[3 * (int64_t)o2l[o] + comp] += dt * f[i];
  } else {
    u[i] += dt * f[i];
  }
}
__global__ void scatter_component_kernel(int n, const int32_t *__restrict__ ids, const double *__restrict__ vals, double *__restrict__ dst, int ncomp,
                                         int comp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = ids ? ids[i] : i;
  dst[(int64_t)o * ncomp + comp] = vals[i];
}

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

struct RDyHipOperator_s {
  RDyHipConfig config;
  int          device = 0;
  int32_t      n_cells = 0, n_owned = 0, S = 3, K = 0, n_internal = 0;
  int64_t      stride = 0;
  bool         prefix = true;
  int          grid = 0, xcd_chunks = 0;
  bool         keep_fdiv = false;

  DevBuf<int32_t> d_o2l, d_nbr, d_pos, d_halo_list, d_btype, d_bleft, d_bghost_list;
  DevBuf<double>  d_cn, d_sn, d_coef, d_dzdx, d_dzdy, d_mannings, d_extsrc, d_area_local;
  DevBuf<double>  d_bvalues, d_bflux, d_baccum, d_bcn, d_bsn, d_pv, d_fdiv, d_blk_max;
  DevBuf<int32_t> d_blk_pos;
  DevBuf<DeviceCourant> d_courant;
  int32_t n_halo = 0, n_bghost = 0;

  // host copies needed to resolve the Courant position into ids
  std::vector<int32_t> h_internal_edge, h_edge_cells, h_bedge, h_boff;
  std::vector<int64_t> h_cell_gid, h_edge_gid;
  std::vector<double>  h_area;
  RDyHipCourant        courant{0.0, -1, -1};
  // staging buffers for the setters
  DevBuf<double>  d_stage_vals;
  DevBuf<int32_t> d_stage_ids;

  int64_t device_bytes = 0;

  ~RDyHipOperator_s() {
    d_o2l.release(); d_nbr.release(); d_pos.release(); d_halo_list.release(); d_btype.release(); d_bleft.release();
    d_bghost_list.release(); d_cn.release(); d_sn.release(); d_coef.release(); d_dzdx.release(); d_dzdy.release();
    d_mannings.release(); d_extsrc.release(); d_area_local.release(); d_bvalues.release(); d_bflux.release();
    d_baccum.release(); d_bcn.release(); d_bsn.release(); d_pv.release(); d_fdiv.release(); d_blk_max.release();
    d_blk_pos.release(); d_courant.release(); d_stage_vals.release(); d_stage_ids.release();
  }
};

namespace {

int launch_rhs(RDyHipOperator op, int32_t phase, int32_t overwrite, int reset_diag, double dt, const double *u, double *f, hipStream_t st) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (!u || !f) return fail(RDYHIP_ERR_USER, "null u_local / f_global");
  if (op->n_owned == 0) return reset_diag ? rdyhip_reset_diagnostics(op, (void *)st) : 0;
  KernelArgs a{};
  a.n_owned    = op->n_owned;
  a.stride     = op->stride;
  a.o2l        = op->prefix ? nullptr : op->d_o2l.p;
  a.nbr        = op->d_nbr.p;
  a.cn         = op->d_cn.p;
  a.sn         = op->d_sn.p;
  a.coef       = op->d_coef.p;
  a.pos        = op->d_pos.p;
  a.dzdx       = op->d_dzdx.p;
  a.dzdy       = op->d_dzdy.p;
  a.mannings   = op->d_mannings.p;
  a.extsrc     = op->d_extsrc.p;
  a.area_local = op->d_area_local.p;
  a.btype      = op->d_btype.p;
  a.bvalues    = op->d_bvalues.p;
  a.bflux      = op->d_bflux.p;
  a.baccum     = op->d_baccum.p;
  a.pv         = op->d_pv.p;
  a.fdiv       = op->keep_fdiv ? op->d_fdiv.p : nullptr;
  a.blk_max    = op->d_blk_max.p;
  a.blk_pos    = op->d_blk_pos.p;
  a.tiny_h     = op->config.tiny_h;
  a.h_anuga_sq = op->config.h_anuga_regular * op->config.h_anuga_regular;
  a.xq_thresh  = op->config.xq2018_threshold;
  a.overwrite  = overwrite ? 1 : 0;
  a.phase      = phase;

  int grid;
  if (phase == RDYHIP_PHASE_HALO) {
    if (op->n_halo == 0) return 0;  // (never combined with reset_diag)
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
  const bool xq = op->config.source_method == RDYHIP_SOURCE_IMPLICIT_XQ2018;
  if (op->S == 3) {
    if (xq) hipLaunchKernelGGL((swe_rhs_kernel<3, 1>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
    else hipLaunchKernelGGL((swe_rhs_kernel<3, 0>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
  } else {
    if (xq) hipLaunchKernelGGL((swe_rhs_kernel<4, 1>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
    else hipLaunchKernelGGL((swe_rhs_kernel<4, 0>), dim3(grid), dim3(BLOCK), 0, st, a, dt, u, f);
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(courant_finalize_kernel, dim3(1), dim3(1024), 0, st, grid, op->d_blk_max.p, op->d_blk_pos.p, op->d_courant.p, reset_diag);
  HIP_TRY(hipGetLastError());
  // boundary edges hanging off ghost cells (diagnostic vectors only); once per full apply
  if (op->n_bghost > 0 && phase != RDYHIP_PHASE_INTERIOR) {
    hipLaunchKernelGGL(boundary_ghost_kernel, dim3((op->n_bghost + 63) / 64), dim3(64), 0, st, op->n_bghost, op->d_bghost_list.p, op->d_bleft.p,
                       op->d_btype.p, op->d_bcn.p, op->d_bsn.p, op->d_bvalues.p, op->d_bflux.p, op->d_baccum.p, u, dt, a.tiny_h, a.h_anuga_sq);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

}  // namespace

extern "C" {

const char *rdyhip_last_error(void) { return g_err.c_str(); }
int32_t     rdyhip_version(void) { return RDYHIP_VERSION; }

int rdyhip_create(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                  RDyHipOperator *op_out) {
  if (!config || !mesh || !op_out) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_create");
  *op_out = nullptr;
  if (num_boundaries < 0 || (num_boundaries > 0 && !boundaries)) return fail(RDYHIP_ERR_USER, "bad boundary list");
  if (config->riemann != RDYHIP_RIEMANN_ROE) return fail(RDYHIP_ERR_USER, "Unsupported Riemann solver");  // swe_petsc.c:269
  if (config->source_method != RDYHIP_SOURCE_SEMI_IMPLICIT && config->source_method != RDYHIP_SOURCE_IMPLICIT_XQ2018)
    return fail(RDYHIP_ERR_USER, "Only semi_implicit and implicit_xq2018 are supported");  // swe_petsc.c:973
  const int32_t nc = mesh->num_cells, no = mesh->num_owned_cells, ne = mesh->num_edges, ni = mesh->num_internal_edges;
  if (nc < 0 || no < 0 || no > nc || ne < 0 || ni < 0 || ni > ne) return fail(RDYHIP_ERR_ARG_SIZ, "inconsistent mesh sizes");
  if (nc >= NBR_GHOST) return fail(RDYHIP_ERR_ARG_SIZ, "too many local cells (%d) for the 30-bit neighbour encoding", nc);
  if (nc > 0 && (!mesh->cell_is_owned || !mesh->cell_local_to_owned || !mesh->cell_areas || !mesh->cell_dz_dx || !mesh->cell_dz_dy))
    return fail(RDYHIP_ERR_USER, "null cell array");
  if (ne > 0 && (!mesh->edge_cell_ids || !mesh->edge_lengths || !mesh->edge_cn || !mesh->edge_sn)) return fail(RDYHIP_ERR_USER, "null edge array");
  if (ni > 0 && !mesh->edge_internal_ids) return fail(RDYHIP_ERR_USER, "null internal edge list");

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

  // ---- per-owned-cell geometry --------------------------------------------
  std::vector<double> dzdx((size_t)no), dzdy((size_t)no);
  for (int32_t o = 0; o < no; ++o) {
    dzdx[o] = mesh->cell_dz_dx[o2l[o]];
    dzdy[o] = mesh->cell_dz_dy[o2l[o]];
  }

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
  int rc         = 0;
  if (hipGetDevice(&op->device) != hipSuccess) {
    delete op;
    return fail(RDYHIP_ERR_LIB, "hipGetDevice failed: no usable HIP device");
  }

  const int tiles   = (no + BLOCK - 1) / BLOCK;
  const char *env   = getenv("RDYHIP_XCD_SWIZZLE");
  const bool  swz   = env ? atoi(env) != 0 : true;
  op->xcd_chunks    = (swz && tiles >= 64) ? (tiles + 7) / 8 : 0;
  op->grid          = op->xcd_chunks > 0 ? op->xcd_chunks * 8 : tiles;
  const int maxgrid = std::max(op->grid, 1);

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
    TRY_RC(op->d_area_local.upload(area));
    op->h_area.swap(area);
  }
  TRY_RC(op->d_mannings.zeros((size_t)no));
  TRY_RC(op->d_extsrc.zeros((size_t)3 * no));
  TRY_RC(op->d_bvalues.zeros((size_t)3 * K));
  TRY_RC(op->d_bflux.zeros((size_t)3 * K));
  TRY_RC(op->d_baccum.zeros((size_t)3 * K));
  TRY_RC(op->d_pv.zeros((size_t)3 * no));
  TRY_RC(op->d_blk_max.zeros((size_t)maxgrid));
  TRY_RC(op->d_blk_pos.zeros((size_t)maxgrid));
  TRY_RC(op->d_courant.zeros(1));
#undef TRY_RC
  hipLaunchKernelGGL(courant_reset_kernel, dim3(1), dim3(1), 0, 0, op->d_courant.p);
  if (hipDeviceSynchronize() != hipSuccess) {
    delete op;
    return fail(RDYHIP_ERR_LIB, "device synchronisation failed after create");
  }

  op->h_internal_edge.assign(mesh->edge_internal_ids, mesh->edge_internal_ids + ni);
  op->h_edge_cells.assign(mesh->edge_cell_ids, mesh->edge_cell_ids + 2 * (size_t)ne);
  op->h_bedge = bedge;
  op->h_boff  = boff;
  if (mesh->cell_global_ids) op->h_cell_gid.assign(mesh->cell_global_ids, mesh->cell_global_ids + nc);
  if (mesh->edge_global_ids) op->h_edge_gid.assign(mesh->edge_global_ids, mesh->edge_global_ids + ne);

  op->device_bytes = op->d_o2l.bytes() + op->d_nbr.bytes() + op->d_pos.bytes() + op->d_cn.bytes() + op->d_sn.bytes() + op->d_coef.bytes() +
                     op->d_dzdx.bytes() + op->d_dzdy.bytes() + op->d_mannings.bytes() + op->d_extsrc.bytes() + op->d_area_local.bytes() +
                     op->d_pv.bytes() + op->d_bvalues.bytes() + op->d_bflux.bytes() + op->d_baccum.bytes() + op->d_blk_max.bytes() +
                     op->d_blk_pos.bytes();
  *op_out = op;
  return 0;
}

int rdyhip_destroy(RDyHipOperator *op) {
  if (!op) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_destroy");
  if (*op) {
    (void)hipDeviceSynchronize();
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

int rdyhip_apply_phase(RDyHipOperator op, int32_t phase, int32_t overwrite, double dt, const double *u_local, double *f_global, void *stream) {
  if (phase != RDYHIP_PHASE_ALL && phase != RDYHIP_PHASE_INTERIOR && phase != RDYHIP_PHASE_HALO) return fail(RDYHIP_ERR_USER, "bad phase %d", phase);
  return launch_rhs(op, phase, overwrite, 0, dt, u_local, f_global, (hipStream_t)stream);
}

int rdyhip_set_boundary_values(RDyHipOperator op, int32_t boundary, int32_t comp_offset, int32_t num_comp, int32_t num_edges, const double *values) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
  if (boundary < 0 || boundary + 1 >= (int32_t)op->h_boff.size()) return fail(RDYHIP_ERR_USER, "Invalid boundary index %d", boundary);  // operator.c CheckOperatorBoundary
  const int32_t n = op->h_boff[boundary + 1] - op->h_boff[boundary];
  if (n != num_edges) return fail(RDYHIP_ERR_USER, "num_edges (%d) does not match boundary.num_edges (%d)", num_edges, n);  // operator.c:1052
  if (comp_offset < 0 || num_comp < 0 || comp_offset + num_comp > 3) return fail(RDYHIP_ERR_USER, "bad component range [%d,%d)", comp_offset, comp_offset + num_comp);
  if (n == 0 || num_comp == 0) return 0;
  if (!values) return fail(RDYHIP_ERR_USER, "null values");
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
  hipLaunchKernelGGL(courant_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, op->d_courant.p);
  HIP_TRY(hipGetLastError());
  op->courant = RDyHipCourant{0.0, -1, -1};
  return 0;
}

int rdyhip_update_diagnostics(RDyHipOperator op, void *stream) {
  if (!op) return fail(RDYHIP_ERR_USER, "null operator");
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
  hipLaunchKernelGGL(pack_cells_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, u_local, cell_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_unpack_cells(double *u_local, const int32_t *cell_ids, int32_t n, const double *buf, void *stream) {
  if (n < 0) return fail(RDYHIP_ERR_ARG_SIZ, "negative count");
  if (n == 0) return 0;
  if (!u_local || !cell_ids || !buf) return fail(RDYHIP_ERR_USER, "null argument");
  hipLaunchKernelGGL(unpack_cells_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, u_local, cell_ids, buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_axpy_owned(RDyHipOperator op, double dt, const double *f_global, double *u_local, void *stream) {
  if (!op || !f_global || !u_local) return fail(RDYHIP_ERR_USER, "null argument");
  if (op->n_owned == 0) return 0;
  const int n3 = 3 * op->n_owned;
  hipLaunchKernelGGL(axpy_owned_kernel, dim3((n3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, op->n_owned, op->prefix ? nullptr : op->d_o2l.p, dt,
                     f_global, u_local);
  HIP_TRY(hipGetLastError());
  return 0;
}

int rdyhip_layout_info(RDyHipOperator op, RDyHipLayoutInfo *info) {
  if (!op || !info) return fail(RDYHIP_ERR_USER, "null argument");
  info->num_owned_cells    = op->n_owned;
  info->num_cells          = op->n_cells;
  info->slots_per_cell     = op->S;
  info->num_boundary_edges = op->K;
  info->num_halo_cells     = op->n_halo;
  info->owned_is_prefix    = op->prefix ? 1 : 0;
  info->device_bytes       = op->device_bytes;
  // u (own cell) 24 + slots S*(4+8+8+8) + dz 16 + n 8 + ext src 24 + F 24 + pv 24 (+4 for o2l)
  info->bytes_per_apply = (int64_t)op->n_owned * (24 + op->S * 28 + 16 + 8 + 24 + 24 + 24 + (op->prefix ? 0 : 4));
  return 0;
}

}  // extern "C"
