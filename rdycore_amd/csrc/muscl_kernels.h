// Second-order (MUSCL) variant of the SWE right-hand side for gfx950:
// ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213) with its helpers
// ComputeLeastSquaresGradients and ReconstructFaceValues
// (src/operator_fluxes_ceed.c:998-1042, 1155-1206).
//
//  * muscl_gradient_kernel: one thread per owned cell; the weighted
//    least-squares gradient of (h, hu, hv) from the cell's <= S neighbours,
//    accumulated in the reference's internal-edge order; writes grad[local cell][6].
//    The least-squares coefficients are NOT streamed from memory: the reference
//    precomputes c_edge = M^-1 w d per (cell, edge) (PrecomputeLSGradCoeffs, 48 B per
//    triangle) -- here every gradient is formed from the cell centroids (16 B per cell)
//    as M^-1 sum_n w_n d_n (q_n - q_c), the same expression by linearity (ls_add /
//    ls_solve below; flops are free on this memory-bound path).  Likewise an edge's
//    centroid -> midpoint displacements (32 B per edge in ReconstructFaceValues) are
//    formed from ONE stored midpoint (16 B) and the centroids already on the chip.
//  * swe_rhs_muscl_fused_kernel: the tiled structure of swe_kernels.h with the
//    gradients formed on the chip.  Phase 0 stages the conserved state and the
//    centroid of the tile's own cells and of its two rings in LDS; phase G forms
//    the gradients of own + first-ring cells in LDS; phase 1 reconstructs the two
//    limited face states of every tile edge from LDS, derives the Riemann side
//    data per edge side and evaluates the Roe flux once per edge; phase 2 is the
//    first-order kernel's (segmented per-cell sum in the reference's order, source
//    terms, stores).  Boundary edges stay first order (ApplyBoundaryFlux is
//    unchanged by numerics.second_order).  (A split form -- gradients through
//    memory, a flux kernel that reads them -- was the A/B partner of rounds 1-4:
//    tools/probes/split_muscl_form.patch.)
//
// Across ranks the reference solves each cut edge on the rank that owns it and
// adds the ghost side back with DMLocalToGlobal(ADD_VALUES).  Here every rank
// evaluates all edges of its owned cells (the cut ones redundantly, from
// bitwise-identical operands once the ghost gradients have been exchanged), so
// no reverse exchange exists; see rdycore_amd/halo.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "swe_kernels.h"

namespace rdyhip {

// non-temporal hints for the fused kernel's streamed-once data (see swe_kernels.h)
#define RDY_MLD(ptr) __builtin_nontemporal_load(ptr)
#define RDY_MST(ptr, val) __builtin_nontemporal_store((val), (ptr))

typedef double   rdy_d2v __attribute__((ext_vector_type(2)));
typedef uint32_t rdy_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_d2(const double *p) {  // 16-byte aligned pair
  const rdy_d2v v = RDY_MLD(reinterpret_cast<const rdy_d2v *>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ uint2 load_u2(const void *p) {
  const rdy_u2v v = RDY_MLD(reinterpret_cast<const rdy_u2v *>(p));
  return make_uint2(v.x, v.y);
}

struct MusclArgs {
  double       *grad;   // [num_cells][6]: dh/dx, dh/dy, dhu/dx, dhu/dy, dhv/dx, dhv/dy (local cell index)
  const double *e_mid;  // [nrec][2]: edge midpoint (x, y) of each tile edge record
  const double *cxy;    // [num_cells][2]: cell centroid (x, y), local cell index
  // fused kernel only: the second ring of each tile and the stencils of its first-ring cells
  const int32_t  *hcells2;  // second-ring cells of each tile (local cell ids): neighbours of first-ring cells outside the tile
  const int32_t  *r2_off;   // [ntiles+1] first hcells2 entry of each tile
  const uint16_t *bn_idx;   // [halo entries][4] LDS slot of each first-ring cell's s-th neighbour; BN_NONE: no neighbour in that
                            //                   slot; BN_GLOBAL: a ghost cell, its gradient comes from `grad` (exchanged)
};
constexpr uint16_t BN_NONE = 0xFFFF, BN_GLOBAL = 0xFFFE;

// LDS layout of the second-order kernel: one plane per value with COMPILE-TIME plane strides -- the plane offsets are
// instruction immediates (no register, no address arithmetic) and every read is a full-rate ds_read_b64.  Every tile is cut
// at create so that its rings and edge records fit the capacities below (layout_build_tiles).  (Rounds 1-3 also carried a
// record layout with run-time sizes for meshes numbered without locality: every plane offset cost an SGPR there -- 161
// v_readlane_b32 of spilled offsets in the edge phase alone -- and ds_read2_b64 pairs served at half rate.)
template <int NQ, int NG, int NE>
struct MusclSoA {
  static constexpr int  nq = NQ, ng = NG, ne = NE;  // capacities: state records (own + both rings), gradient records (own + first ring), edges
  static __device__ __forceinline__ int qidx(int k, int j) { return k * NQ + j; }
  static __device__ __forceinline__ int gidx(int k, int j) { return k * NG + j; }
  static __device__ __forceinline__ int eidx(int c, int e_) { return c * NE + e_; }
  static constexpr size_t lds_bytes = sizeof(double) * (6 * (size_t)NG + 5 * (size_t)NQ) + sizeof(uint32_t) * (size_t)NE;
};
// triangles: 256 own + <= 104 first-ring cells, <= 264 ring cells in all, <= 512 edge records (two register rounds);
// 40 128 B per workgroup: four workgroups per CU.  The edge fluxes (4 planes) overlay the gradients (6 planes): a thread
// keeps the fluxes of its (at most two) edges in registers across one extra barrier and then writes them over the
// gradients, dead by then.  The plane strides are deliberately NOT multiples of 64 doubles: hipcc would otherwise fuse the
// reads of two planes into ds_read2st64_b64, which is served like ds_read2_b64.
// (every ring cell is staged by one thread: at most TILE of them; the state planes keep 8 slots of slack so that their stride
// is no multiple of 64 doubles)
constexpr int MUSCL_MAX_RING1_TRI = TILE_MAX_HALO_TRI, MUSCL_MAX_RING_TRI = TILE;
using MusclSoATri = MusclSoA<TILE + MUSCL_MAX_RING_TRI + 8, TILE + MUSCL_MAX_RING1_TRI, TILE_MAX_REC + 8>;
static_assert(MUSCL_MAX_RING_TRI <= TILE, "one thread stages one ring cell");
// quads / mixed meshes: <= 112 first-ring cells, <= 168 ring cells in all; 37 664 B per workgroup
constexpr int MUSCL_MAX_RING1_QUAD = TILE_MAX_HALO_QUAD, MUSCL_MAX_RING_QUAD = 168;
using MusclSoAQuad = MusclSoA<TILE + MUSCL_MAX_RING_QUAD, TILE + MUSCL_MAX_RING1_QUAD, TILE_MAX_REC + 8>;
static_assert(MUSCL_MAX_RING_QUAD <= TILE && TILE_MAX_HALO_QUAD <= TILE && TILE_MAX_HALO_TRI <= TILE, "one thread stages one ring / halo cell");
#define MSQ(k, j) sq[LAY::qidx((k), (j))]
#define MSG(k, j) sg[LAY::gidx((k), (j))]
#define MEF(c, e) ef[LAY::eidx((c), (e))]

// Weighted least-squares gradient of a cell (PrecomputeLSGradCoeffs + ComputeLeastSquaresGradients,
// src/operator_fluxes_ceed.c:884-1042) accumulated neighbour by neighbour: with d = centroid_n - centroid_c,
// w = 1/|d|, M += w d d^T and b_k += w d (q_n,k - q_c,k); the gradient is M^-1 b_k (zero for a degenerate stencil,
// |det M| < 1e-15, as in the reference).  Explicit fma throughout and ONE code path shared by every kernel that
// forms a gradient: a cell's gradient is computed by its own tile, by every tile that has it in its first ring and
// (for the exchange between ranks) by muscl_gradient_kernel, and a cut edge is conservative only if all of them
// get the same bits.
struct LsAcc {
  double m00 = 0.0, m01 = 0.0, m11 = 0.0;
  double b[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // (bx, by) of h, hu, hv
};
__device__ __forceinline__ void ls_add(LsAcc &a, double dx, double dy, double d0, double d1, double d2) {
  const double r2 = fma(dx, dx, dy * dy);
  // w = 1/sqrt(r2) by v_rsq_f64 + Newton (a few ulp: every copy of a gradient goes through this same code, so they all
  // agree bit for bit, and the parity bar is 1e-10); coincident centroids give w = 0 as in the reference
  double w = __builtin_amdgcn_rsq(r2);
  w        = w * fma(-0.5 * r2 * w, w, 1.5);
  if (!(r2 > 0.0)) w = 0.0;
  const double wdx = w * dx, wdy = w * dy;
  a.m00  = fma(wdx, dx, a.m00);
  a.m01  = fma(wdx, dy, a.m01);
  a.m11  = fma(wdy, dy, a.m11);
  a.b[0] = fma(wdx, d0, a.b[0]);
  a.b[1] = fma(wdy, d0, a.b[1]);
  a.b[2] = fma(wdx, d1, a.b[2]);
  a.b[3] = fma(wdy, d1, a.b[3]);
  a.b[4] = fma(wdx, d2, a.b[4]);
  a.b[5] = fma(wdy, d2, a.b[5]);
}
__device__ __forceinline__ void ls_solve(const LsAcc &a, double (&g)[6]) {
  const double p   = a.m01 * a.m01;
  const double det = fma(a.m00, a.m11, -p);
  if (fabs(det) < 1e-15) {
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = 0.0;
    return;
  }
  const double inv_det = rdy_rcp(det);
  const double i00 = a.m11 * inv_det, i01 = -(a.m01 * inv_det), i11 = a.m00 * inv_det;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t0 = i01 * a.b[2 * k + 1], t1 = i11 * a.b[2 * k + 1];
    g[2 * k]     = fma(i00, a.b[2 * k], t0);
    g[2 * k + 1] = fma(i01, a.b[2 * k], t1);
  }
}

// RDyLimiterType, include/private/rdyconfigimpl.h:67-71
constexpr int LIMITER_MINMOD = 0, LIMITER_NONE = 1, LIMITER_VANLEER = 2;

// Minmod / VanLeer / LimitSlope, src/operator_fluxes_ceed.c:1110-1138
template <int LIM>
__device__ __forceinline__ double limit_slope(double extrap, double half_dq) {
  if (LIM == LIMITER_NONE) return extrap;
  // minmod(a, b) = "0 if the signs differ, else the one of smaller magnitude" is the median of (a, b, 0): four min / max
  // operations instead of a multiply, two compares and two 64-bit selects
  if (LIM == LIMITER_MINMOD) return fmax(fmin(extrap, half_dq), fmin(fmax(extrap, half_dq), 0.0));
  if (extrap * half_dq <= 0.0) return 0.0;
  if (LIM == LIMITER_VANLEER) return 2.0 * extrap * half_dq * rdy_rcp(extrap + half_dq);
  return fabs(extrap) < fabs(half_dq) ? extrap : half_dq;
}

// One tile edge of the second-order path: ReconstructFaceValues (src/operator_fluxes_ceed.c:1180-1203) for an
// interior edge -- limited extrapolation of both cells' states to the edge midpoint, depth clamped from below --
// then ComputeRiemannVelocities + the Roe flux on the reconstructed states (src/swe/swe_petsc.c:139-161); a boundary
// edge stays first order.  `sq` / `sg`: LDS planes of the state (stride nq; planes 3 and 4: the centroid x, y) and the
// gradient (stride ng); mid: the edge midpoint, from which the two centroid -> midpoint displacements are formed
// (src/operator_fluxes_ceed.c:1169-1178).  Returns the flux; the caller parks it in LDS for phase 2 (store_edge_flux).
struct EdgeFlux {
  double f0, f1, f2, am;  // am: largest wave speed, -1 for a dry-dry edge (skipped, swe_petsc.c:184)
};
template <class LAY>
__device__ __forceinline__ void store_edge_flux(const KernelArgs &a, double *ef, int e, const EdgeFlux &r) {
  MEF(0, e) = r.f0;
  MEF(1, e) = r.f1;
  MEF(2, e) = r.f2;
  MEF(3, e) = r.am;
}
template <int LIM, class LAY>
__device__ __forceinline__ EdgeFlux muscl_edge(const KernelArgs &a, int tile, double dt, uint32_t lr, double cs, double2 mid,
                                               const double *sq, const double *sg) {
  double cn, sn;
  edge_normal(lr, cs, cn, sn);
  const int jl = lr & EDGE_SLOT_MASK;
  RoeFlux   fl;
  bool      wet;
  if (!(lr & EDGE_BOUNDARY)) {
    const int jr = (lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK;
    double    ql[3], qr[3];
    double2   dl, dr;
    dl.x = mid.x - MSQ(3, jl);
    dl.y = mid.y - MSQ(4, jl);
    dr.x = mid.x - MSQ(3, jr);
    dr.y = mid.y - MSQ(4, jr);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double cl_ = MSQ(k, jl), cr_ = MSQ(k, jr);
      const double extrap_l = MSG(2 * k, jl) * dl.x + MSG(2 * k + 1, jl) * dl.y;
      const double extrap_r = MSG(2 * k, jr) * dr.x + MSG(2 * k + 1, jr) * dr.y;
      const double dq       = cr_ - cl_;
      ql[k]                 = cl_ + limit_slope<LIM>(extrap_l, 0.5 * dq);
      qr[k]                 = cr_ + limit_slope<LIM>(extrap_r, -0.5 * dq);
    }
    ql[0] = fmax(0.0, ql[0]);  // 1201-1203, swe_petsc.c:143-146
    qr[0] = fmax(0.0, qr[0]);
    const RiemannSide L = riemann_side(ql[0], ql[1], ql[2], a.tiny_h, a.h_anuga_sq);
    const RiemannSide R = riemann_side(qr[0], qr[1], qr[2], a.tiny_h, a.h_anuga_sq);
    fl                  = roe_flux(L, R, sn, cn);
    wet                 = !(R.h < a.tiny_h && L.h < a.tiny_h);  // swe_petsc.c:184
  } else {
    const RiemannSide L  = riemann_side(MSQ(0, jl), MSQ(1, jl), MSQ(2, jl), a.tiny_h, a.h_anuga_sq);
    const int         k  = RDY_COLD(a, tile_bk)[load_uniform(RDY_COLD(a, tile_boff), tile) + ((lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK)];
    BoundaryFlux      bf = boundary_flux(RDY_COLD(a, btype)[k], true, L, RDY_COLD(a, bvalues) + 3 * (int64_t)k, sn, cn, a.tiny_h, a.h_anuga_sq);
    fl                   = bf.flux;
    wet                  = bf.wet;
    store_boundary_flux(a, k, fl, dt);
  }
  EdgeFlux r;
  r.f0 = fl.f0;
  r.f1 = fl.f1;
  r.f2 = fl.f2;
  // (-2: a wet edge of another rank -- any value other than -1 adds the flux; a negative one never reaches the Courant maximum)
  r.am = wet ? ((lr & EDGE_NOT_OWNED) ? -2.0 : fl.amax) : -1.0;
  return r;
}

// index of slot s's edge in the tile's edge list, or -1 for an unused slot
template <int S>
__device__ __forceinline__ int slot_edge(uint32_t r0, uint32_t r1, int s) {
  if (S == 3) {
    const uint32_t ref = (r0 >> (10 * s)) & 0x3FF;
    return ref == REF3_EMPTY ? -1 : (int)ref;
  }
  const uint32_t w   = (s < 2) ? r0 : r1;
  const uint32_t ref = (s & 1) ? (w >> 16) : (w & 0xFFFFu);
  return ref == SLOT_EMPTY ? -1 : (int)ref;
}

// Phase 2 of both second-order kernels: a cell's flux sum in the reference's edge order + the Courant number
// (src/swe/swe_petsc.c:184-201); kf[s] = -+len/area of slot s.
template <int S, class LAY>
__device__ __forceinline__ void muscl_cell_sum(const KernelArgs &a, uint32_t r0, uint32_t r1, const double (&kf)[S], const double *ef, double dt, int e_off,
                                               int pos_lo, int pos_q, double &acc0, double &acc1, double &acc2, CourantTrack &trk) {
  bool      tie    = false;  // a slot of this cell met the thread's running Courant maximum to the last bit (CourantTrack, swe_kernels.h)
  const int rec_in = trk.rec;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int ref = slot_edge<S>(r0, r1, s);
    if (ref < 0) continue;
    const double am = MEF(3, ref);
    if (am != -1.0) {
      const double k = kf[s];
      acc0 += MEF(0, ref) * k;
      acc1 += MEF(1, ref) * k;
      acc2 += MEF(2, ref) * k;
      const double cnum = am * fabs(k) * dt;  // len/area_self: the max over the two cells is len / min(area_l, area_r)
      tie |= cnum == trk.best;
      if (cnum > trk.best) {
        trk.best = cnum;
        trk.rec  = e_off + ref;
      }
    }
  }
  if (trk.rec != rec_in) trk.pos = -1;
  // cold: which of the equal edges comes first in the reference's loop; dismissed at once where the incumbent's position is
  // known and smaller than every position of this tile
  if (tie && !(trk.pos >= 0 && trk.pos < pos_lo)) {
    int first = -1;  // the first slot of this cell at the running maximum: the only one that can come before the incumbent
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int ref = slot_edge<S>(r0, r1, s);
      if (ref < 0) continue;
      const double am = MEF(3, ref);
      if (first < 0 && am != -1.0 && am * fabs(kf[s]) * dt == trk.best) first = ref;
    }
    if (first >= 0) courant_resolve_tie(a, trk, e_off + first, first >= COURANT_Q ? pos_q : pos_lo);
  }
}

// ComputeLeastSquaresGradients, src/operator_fluxes_ceed.c:998-1042, gathered per cell: the cell's internal edges in
// the reference's loop order (slot order), neighbour minus self for both the left and the right cell of an edge (the
// reference multiplies c_LR and c_RL by q_R - q_L; the signs cancel in w d (q_n - q_c)).
template <int S>
__global__ __launch_bounds__(BLOCK) void muscl_gradient_kernel(const KernelArgs a, const MusclArgs g, const double *__restrict__ u) {
  int tile = blockIdx.x;
  if (a.xcd_chunks > 0) tile = (blockIdx.x & 7) * a.xcd_chunks + (blockIdx.x >> 3);
  const int i = tile * BLOCK + threadIdx.x;
  if (i >= a.n_work) return;
  const int o = a.list ? a.list[i] : i;
  int32_t   id[S];
  bool      has_ghost = false;
  const int32_t *nbr  = RDY_COLD(a, nbr);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    id[s] = nbr[s * a.stride + o];
    has_ghost |= (id[s] >= 0) && (id[s] & NBR_GHOST);
  }
  if (a.phase == RDYHIP_PHASE_INTERIOR && has_ghost) return;
  if (a.phase == RDYHIP_PHASE_HALO && !has_ghost) return;
  const int    c  = a.o2l ? a.o2l[o] : o;
  const double q0 = u[3 * (int64_t)c + 0], q1 = u[3 * (int64_t)c + 1], q2 = u[3 * (int64_t)c + 2];
  const double x0 = g.cxy[2 * (int64_t)c], y0 = g.cxy[2 * (int64_t)c + 1];
  LsAcc        acc;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (id[s] < 0) continue;  // boundary edge or unused slot: not part of the stencil
    const int n = id[s] & NBR_MASK;
    ls_add(acc, g.cxy[2 * (int64_t)n] - x0, g.cxy[2 * (int64_t)n + 1] - y0, u[3 * (int64_t)n + 0] - q0, u[3 * (int64_t)n + 1] - q1,
           u[3 * (int64_t)n + 2] - q2);
  }
  double gr[6];
  ls_solve(acc, gr);
  double2 *dst = reinterpret_cast<double2 *>(g.grad + 6 * (int64_t)c);
  dst[0]       = make_double2(gr[0], gr[1]);
  dst[1]       = make_double2(gr[2], gr[3]);
  dst[2]       = make_double2(gr[4], gr[5]);
  // the launch over the halo cell list (a.list) also packs: the rows of the send buffer this cell's gradient travels in
  // (rdyhip_halo_fuse_pack on a second-order operator; ColdArgs::gsend_*)
  const int32_t *goff = RDY_COLD(a, gsend_off);
  if (goff && a.list) {
    const int32_t *rows = RDY_COLD(a, gsend_rows);
    double        *sbuf = RDY_COLD(a, gsend_buf);
    for (int r = goff[i]; r < goff[i + 1]; ++r) {
      double2 *sd = reinterpret_cast<double2 *>(sbuf + 6 * (int64_t)rows[r]);
      sd[0]       = make_double2(gr[0], gr[1]);
      sd[1]       = make_double2(gr[2], gr[3]);
      sd[2]       = make_double2(gr[4], gr[5]);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused form (default): the gradients never leave the chip.  A tile stages the
// state and the centroid of its own cells, of its first ring (cells sharing an
// edge with a tile cell) and of its second ring (the remaining neighbours of
// first-ring cells), forms the least-squares gradients of own + first-ring
// cells in LDS, and goes on as above.  Against a form with the gradients in memory this saves the
// gradient array's write + read, the second read of the state and the streamed
// least-squares coefficients / displacements.  First-ring cells that are ghosts
// take their gradient from `grad`, filled by the caller's exchange (their
// stencil is not local).  Four workgroups per CU, each tile's loads in one batch
// at its top.  (Rounds 3-4 ran quads at three workgroups with a cross-tile
// software pipeline, worth 5.6 % while a quad tile had a third flux round;
// with tiles of two rounds the fourth workgroup wins: 2-6 %,
// profiles/r05_quads_ab.txt.)
// ---------------------------------------------------------------------------
// The extra Courant edges (ColdArgs::x_*, swe_kernels.h) of the second-order path: both cells' states, gradients and centroids
// from memory (the gradients of ghost cells and of ghost-adjacent owned cells are there: the exchange and the launch over the
// halo cell list have written them), the reconstruction of muscl_edge, the largest wave speed.
template <int LIM>
__device__ __forceinline__ void courant_extra_edges_muscl(const KernelArgs &a, const MusclArgs &g, double dt, const double *__restrict__ u, CourantTrack &t) {
  const int nx = RDY_COLD(a, n_xedges);
  if (nx == 0 || a.phase == RDYHIP_PHASE_INTERIOR) return;  // uniform
  const int32_t  *xlr = RDY_COLD(a, x_lr);
  const uint32_t *xfl = RDY_COLD(a, x_flags);
  const double   *xcs = RDY_COLD(a, x_cs), *xcf = RDY_COLD(a, x_cfac), *xmid = RDY_COLD(a, x_mid);
  const int       rec0 = RDY_COLD(a, x_rec0);
  for (int i = blockIdx.x * TILE + threadIdx.x; i < nx; i += gridDim.x * TILE) {
    const int64_t l = xlr[2 * i], r = xlr[2 * i + 1];
    const double  mx = xmid[2 * i], my = xmid[2 * i + 1];
    const double  dlx = mx - g.cxy[2 * l], dly = my - g.cxy[2 * l + 1], drx = mx - g.cxy[2 * r], dry = my - g.cxy[2 * r + 1];
    double        ql[3], qr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double cl_ = u[3 * l + k], cr_ = u[3 * r + k];
      const double extrap_l = g.grad[6 * l + 2 * k] * dlx + g.grad[6 * l + 2 * k + 1] * dly;
      const double extrap_r = g.grad[6 * r + 2 * k] * drx + g.grad[6 * r + 2 * k + 1] * dry;
      const double dq       = cr_ - cl_;
      ql[k]                 = cl_ + limit_slope<LIM>(extrap_l, 0.5 * dq);
      qr[k]                 = cr_ + limit_slope<LIM>(extrap_r, -0.5 * dq);
    }
    ql[0] = fmax(0.0, ql[0]);
    qr[0] = fmax(0.0, qr[0]);
    const RiemannSide L = riemann_side(ql[0], ql[1], ql[2], a.tiny_h, a.h_anuga_sq);
    const RiemannSide R = riemann_side(qr[0], qr[1], qr[2], a.tiny_h, a.h_anuga_sq);
    double            cn, sn;
    edge_normal(xfl[i], xcs[i], cn, sn);
    if (!(R.h < a.tiny_h && L.h < a.tiny_h)) {
      const double cnum = roe_flux(L, R, sn, cn).amax * xcf[i] * dt;
      if (cnum > t.best) {
        t.best = cnum;
        t.rec  = rec0 + i;
        t.pos  = -1;
      } else if (cnum == t.best) {
        courant_resolve_tie(a, t, rec0 + i, 0);
      }
    }
  }
}

// The per-cell streams become visible to phase 2 HERE and on every path: without this hipcc hoists their first uses (a
// multiply by a constant) into the block that requests them -- the wave then waits for them before the barriers that were
// meant to cover their latency -- and, the waits sitting inside `if (active)`, treats the registers as still pending after the
// merge: a vmcnt(0) between the tile's first store and its second, i.e. a wait for the store itself.
#define RDY_STREAMS_ARRIVE()                                                                                                              \
  do {                                                                                                                                    \
    asm volatile("" ::"v"(dzx), "v"(dzy), "v"(nman), "v"(s0), "v"(s1), "v"(s2), "v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[S - 1]) : "memory"); \
  } while (0)
template <int S, int SRC, bool OVW, int LIM, bool EULER = false>
__global__ __launch_bounds__(TILE) void swe_rhs_muscl_fused_kernel(const KernelArgs a, const MusclArgs g, const double dt, const double *__restrict__ u,
                                                                    double *__restrict__ f) {
  // LDS: gradients of own + first-ring cells (6 planes) | state + centroid of own cells and both rings (5 planes) | the
  // tile's edge records, read by the gradient phase only (the edge phase has its records in registers).  The edge fluxes
  // take over the gradients' storage once every edge has been evaluated: ~37-40 KB per workgroup, so FOUR workgroups of the
  // triangle kernel share a CU's 160 KB.
  using LAY = typename std::conditional<S == 3, MusclSoATri, MusclSoAQuad>::type;
  extern __shared__ double lds[];
  double   *sg  = lds;                                             // 6 planes of LAY::ng
  double   *sq  = lds + 6 * LAY::ng;                               // h, hu, hv, centroid x, centroid y: 5 planes of LAY::nq
  double   *ef  = sg;                                              // the edge fluxes: 4 planes of LAY::ne
  uint32_t *slr = reinterpret_cast<uint32_t *>(sq + 5 * LAY::nq);  // [LAY::ne] the tile's edge records
  static_assert(4 * LAY::ne <= 6 * LAY::ng, "the flux planes overlay the gradient planes");
  const int tid = threadIdx.x;

  int idx, step, hi;
  if (a.xcd_chunks > 0) {
    const int x = blockIdx.x & 7;
    step        = gridDim.x >> 3;
    idx         = x * a.xcd_chunks + (blockIdx.x >> 3);
    hi          = min((x + 1) * a.xcd_chunks, a.n_work);
  } else {
    idx  = blockIdx.x;
    step = gridDim.x;
    hi   = a.n_work;
  }
  auto tile_at = [&](int i) -> int { return __builtin_amdgcn_readfirstlane(a.list ? load_uniform(a.list, i) : i); };
  auto tile_desc = [&](int t) -> TileDesc {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = load_uniform(reinterpret_cast<const v4i *>(a.tiles), t);
    TileDesc  d;
    d.e_off = v.x; d.h_off = v.y; d.c_off = v.z; d.cnt = (uint32_t)v.w;
    return d;
  };
  // id of the ring cell (first or second ring) this thread stages for a tile
  auto ring_id = [&](const TileDesc &td_, int nh_, int c0_, int nc2_) -> int {
    int id = -1;
    if (tid < nh_) id = a.hcells[td_.h_off + tid];
    else if (tid < nh_ + nc2_) id = g.hcells2[c0_ + tid - nh_];
    return id;
  };
  // least-squares gradient of the cell in LDS slot `self` from the cells in slots nb[0..S-1] (-1: none), slot order
  auto lds_gradient = [&](int self, const int (&nb)[S], double (&gr)[6]) {
    const double q0 = MSQ(0, self), q1 = MSQ(1, self), q2 = MSQ(2, self), x0 = MSQ(3, self), y0 = MSQ(4, self);
    LsAcc        acc;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int n = nb[s];
      if (n < 0) continue;
      ls_add(acc, MSQ(3, n) - x0, MSQ(4, n) - y0, MSQ(0, n) - q0, MSQ(1, n) - q1, MSQ(2, n) - q2);
    }
    ls_solve(acc, gr);
  };

  CourantTrack trk;
  {
  // Triangles: four workgroups per CU.  A tile's loads are ONE batch at its top -- except the CELLS group (state + centroid
  // of the own cell and of this thread's ring cell), which for tile T+1 is requested right after phase 0's barrier of tile T,
  // into the registers phase 0 has just emptied: 127 VGPRs instead of 107, still four waves, and every workgroup has
  // requests in flight while it computes (-1.2 % on C3, -2.6 % on the refined Houston mesh: profiles/
  // r03_ab_muscl_tri_prefetch.txt).  The XQ2018 source variant needs four registers more (131: it would lose the fourth
  // workgroup) and requests its cells group with the rest of the batch.  The ids the group depends on (ring cell, own
  // cell) run one tile further ahead.  What keeps loads in flight across phases is what is NOT between the request and the first
  // use: (i) no global load on any path every wave takes -- hipcc answers a conditional load whose result is used after the
  // merge with s_waitcnt vmcnt(0) AT THE MERGE, for every wave (the ghost gradients below wait inside their branch for that
  // reason); (ii) no first use hoisted into the requesting block (opaque predicates); (iii) no register that is "pending" at
  // the loop header (the ids are waited for explicitly).
  constexpr bool PF = (SRC == 0);
  auto next_valid = [&](int i) -> int {
    if (a.phase == RDYHIP_PHASE_INTERIOR) {
      while (i < hi && tile_desc(tile_at(i)).halo()) i += step;
    }
    return i;
  };
  double  q[3] = {0.0, 0.0, 0.0}, hq[3] = {0.0, 0.0, 0.0};  // their first use is a plain LDS store: nothing to fold into the loading block
  double2 cxy = make_double2(0.0, 0.0), hcxy = make_double2(0.0, 0.0);
  auto tile_ids = [&](int i, int &c_, int &hid_) {
    const int      t_  = tile_at(i);
    const TileDesc d_  = tile_desc(t_);
    const int      c0_ = load_uniform(g.r2_off, t_);
    hid_               = ring_id(d_, d_.nh(), c0_, load_uniform(g.r2_off, t_ + 1) - c0_);
    const int o_       = d_.c_off + tid;
    c_                 = (a.o2l && tid < d_.nc()) ? a.o2l[o_] : o_;
  };
  auto issue_cells = [&](const TileDesc &d_, int c_, int hid_) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = 0.0;
    cxy = make_double2(0.0, 0.0);
    int nown = d_.nc();
    asm volatile("" : "+s"(nown));
    if (tid < nown) {
#pragma unroll
      for (int k = 0; k < 3; ++k) q[k] = u[3 * (int64_t)c_ + k];
      cxy = *reinterpret_cast<const double2 *>(g.cxy + 2 * (int64_t)c_);
    }
    if (hid_ >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) hq[k] = u[3 * (int64_t)hid_ + k];
      hcxy = *reinterpret_cast<const double2 *>(g.cxy + 2 * (int64_t)hid_);
    }
  };
  idx = next_valid(idx);
  int idx1 = hi, c1 = 0, hid1 = -1;
  if (idx < hi) {  // c1 / hid1: the ids of the tile whose cells group is requested next (PF: the next tile's, else this one's)
    tile_ids(idx, c1, hid1);
    asm volatile("" ::"v"(hid1), "v"(c1));
    idx1 = next_valid(idx + step);
    if (PF) {
      issue_cells(tile_desc(tile_at(idx)), c1, hid1);
      c1   = 0;
      hid1 = -1;
      if (idx1 < hi) tile_ids(idx1, c1, hid1);
      asm volatile("" ::"v"(hid1), "v"(c1));
    }
  }
  while (idx < hi) {
    const int      tile = tile_at(idx);
    const TileDesc td = tile_desc(tile);
    const int  ne = td.ne(), nh = td.nh();
    const int  c0 = load_uniform(g.r2_off, tile), nc2 = load_uniform(g.r2_off, tile + 1) - c0;
    const int  pos_lo = load_uniform(RDY_COLD(a, e_pos), td.e_off);  // smallest loop position of the tile's records (Courant tie path)
    const int  pos_q  = load_uniform(RDY_COLD(a, e_pos), td.e_off + min(COURANT_Q, ne - 1));
    const int  o      = td.c_off + tid;
    const bool active = tid < td.nc();
    const int  hid    = (tid < nh + nc2) ? 0 : -1;  // does this thread stage a ring cell
    int idx2 = hi, c2 = 0, hid2 = -1;

    __builtin_amdgcn_s_setprio(3);
    if (!PF) {
      issue_cells(td, c1, hid1);
      if (idx1 < hi) {
        tile_ids(idx1, c2, hid2);
        idx2 = next_valid(idx1 + step);
      }
    }
    uint32_t  r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu;  // triangles: three 10-bit slot references in r0; quads: four 16-bit ones in r0, r1
    double    kf[S];
    double    dzx = 0.0, dzy = 0.0, nman = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) kf[s] = 0.0;
    // unconditional loads with clamped indices (a tile has edges; lanes past the end read the last record, unused): a
    // lane-conditional load costs register copies of the loaded value at its merge -- and a wait in the middle of the batch
    if (S == 3) {
      r0 = RDY_MLD(&reinterpret_cast<const uint32_t *>(a.slot_ref)[active ? o : td.c_off]);
    } else {
      const uint2 w = load_u2(reinterpret_cast<const uint2 *>(a.slot_ref) + (active ? o : td.c_off));
      r0            = w.x;
      r1            = w.y;
    }
    uint32_t ones = 0xFFFFFFFFu;
    asm volatile("" : "+v"(ones));
    uint2 bw = make_uint2(ones, ones);
    if (tid < nh) bw = load_u2(g.bn_idx + 4 * ((int64_t)td.h_off + tid));
    const int      e0 = td.e_off + min(tid, ne - 1), e1 = td.e_off + min(tid + TILE, ne - 1);
    const uint32_t lr0 = RDY_MLD(&a.e_lr[e0]);
    const double   cs0 = RDY_MLD(&a.e_cs[e0]);
    const double2  md0 = load_d2(g.e_mid + 2 * (int64_t)e0);
    const uint32_t lr1 = RDY_MLD(&a.e_lr[e1]);
    const double   cs1 = RDY_MLD(&a.e_cs[e1]);
    const double2  md1 = load_d2(g.e_mid + 2 * (int64_t)e1);
    __builtin_amdgcn_s_setprio(0);
    // ---- phase 0: state + centroid of own cells, first ring, second ring; the tile's edge records -> LDS
#pragma unroll
    for (int k = 0; k < 3; ++k) MSQ(k, tid) = q[k];
    MSQ(3, tid) = cxy.x;
    MSQ(4, tid) = cxy.y;
    if (hid >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) MSQ(k, TILE + tid) = hq[k];
      MSQ(3, TILE + tid) = hcxy.x;
      MSQ(4, TILE + tid) = hcxy.y;
    }
    if (tid < ne) slr[tid] = lr0;
    if (tid + TILE < ne) slr[tid + TILE] = lr1;
    // the top batch has arrived on EVERY path before the next tile's cells are requested (the waits above sit inside
    // lane-conditional branches; a register still "pending" on a skipped path costs a vmcnt(0) at its next use)
    asm volatile("" ::"v"(r0), "v"(bw.x), "v"(bw.y), "v"(lr0), "v"(lr1), "v"(cs0), "v"(cs1), "v"(md0.x), "v"(md0.y), "v"(md1.x), "v"(md1.y));
    __syncthreads();
    if (PF && idx1 < hi) {
      __builtin_amdgcn_s_setprio(3);
      idx2 = next_valid(idx1 + step);
      if (idx2 < hi) tile_ids(idx2, c2, hid2);
      issue_cells(tile_desc(tile_at(idx1)), c1, hid1);
      __builtin_amdgcn_s_setprio(0);
    }

    // ---- phase G: least-squares gradients of own and first-ring cells -> LDS
    {
      double gr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (active) {
        int nb[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          nb[s]         = -1;
          const int ref = slot_edge<S>(r0, r1, s);
          if (ref < 0) continue;
          const uint32_t lr = slr[ref];
          if (lr & EDGE_BOUNDARY) continue;
          const int jl = lr & EDGE_SLOT_MASK, jr = (lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK;
          nb[s]        = (jl == tid) ? jr : jl;
        }
        lds_gradient(tid, nb, gr);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) MSG(k, tid) = gr[k];
      auto ring_gradient = [&](int j, uint2 w) {
        const uint32_t ix[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
        double         hg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (ix[0] == BN_GLOBAL) {  // a ghost cell: its gradient was computed by its owner
          const int hc = a.hcells[td.h_off + j];
#pragma unroll
          for (int k = 0; k < 6; ++k) hg[k] = g.grad[6 * (int64_t)hc + k];
          asm volatile("" ::"v"(hg[0]), "v"(hg[1]), "v"(hg[2]), "v"(hg[3]), "v"(hg[4]), "v"(hg[5]));  // waited for inside the branch: at the merge hipcc would wait with vmcnt(0) in every wave, ghost or not
        } else {
          int nb[S];
#pragma unroll
          for (int s = 0; s < S; ++s) nb[s] = (ix[s] == BN_NONE) ? -1 : (int)ix[s];
          lds_gradient(TILE + j, nb, hg);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) MSG(k, TILE + j) = hg[k];
      };
      if (tid < nh) ring_gradient(tid, bw);
    }
    __syncthreads();

    // ---- phase 1: every edge of the tile once
    auto do_edge = [&](uint32_t lr, double cs, double2 mid) -> EdgeFlux {
      return muscl_edge<LIM, LAY>(a, tile, dt, lr, cs, mid, sq, sg);
    };
    // the two register-resident rounds one after the other (no per-value selects between the rounds' registers); the
    // scheduling barrier keeps the compiler from interleaving them, which would double the live registers
    auto request_streams = [&]() {
      // unconditional (cells past the end read the last owned cell's, unused): a branch here costs register copies of
      // loaded values at its merge, and with them a wait for the streams right where they are requested
      const int oc = active ? o : td.c_off;
#pragma unroll
      for (int s = 0; s < S; ++s) kf[s] = RDY_MLD(&a.coef[s * a.stride + oc]);
      dzx  = RDY_MLD(&a.dzdx[oc]);
      dzy  = RDY_MLD(&a.dzdy[oc]);
      nman = RDY_MLD(&a.mannings[oc]);
      s0   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 0]);
      s1   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 1]);
      s2   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 2]);
    };
    EdgeFlux x0 = {0.0, 0.0, 0.0, -1.0}, x1 = x0;
    if (tid < ne) x0 = do_edge(lr0, cs0, md0);
    __builtin_amdgcn_sched_barrier(0);
    if (tid + TILE < ne) x1 = do_edge(lr1, cs1, md1);
    // the per-cell streams of phase 2 are requested only here: held from the tile's top they would cost 18 registers
    // through the edge phase; the barriers and the flux stores below cover part of their latency, the other resident
    // workgroups the rest (requested between the two edge rounds instead: 123 VGPRs, no gain --
    // profiles/r03_ab_muscl_mid_streams.txt)
    request_streams();
    __syncthreads();  // every edge has read its gradients: the fluxes may overwrite them
    if (tid < ne) store_edge_flux<LAY>(a, ef, tid, x0);
    if (tid + TILE < ne) store_edge_flux<LAY>(a, ef, tid + TILE, x1);
    __syncthreads();

    // ---- phase 2: per-cell sum in the reference's edge order, source terms, stores
    RDY_STREAMS_ARRIVE();
    asm volatile("" ::"v"(hid2), "v"(c2));
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, res[3] = {0.0, 0.0, 0.0}, pu = 0.0, pv_ = 0.0;
    const double h = MSQ(0, tid), hu = MSQ(1, tid), hv = MSQ(2, tid);
    if (active) {
      if (!OVW) {
        acc0 = f[3 * (int64_t)o + 0];
        acc1 = f[3 * (int64_t)o + 1];
        acc2 = f[3 * (int64_t)o + 2];
      }
      muscl_cell_sum<S, LAY>(a, r0, r1, kf, ef, dt, td.e_off, pos_lo, pos_q, acc0, acc1, acc2, trk);
      const RiemannSide self = riemann_side(h, hu, hv, a.tiny_h, a.h_anuga_sq);
      pu                     = self.u;
      pv_                    = self.v;
      cell_results<SRC>(a, dt, h, hu, hv, acc0, acc1, acc2, dzx, dzy, nman, s0, s1, s2, res);
    }
    {  // whole-line stores of the [cell][3] rows (wave_store_rows3, swe_kernels.h); all 64 lanes take part
      const int     lane  = tid & 63;
      const int64_t base  = 3 * ((int64_t)o - lane);
      const int     ncell = td.nc() - (tid - lane);  // the wave's cells of this tile
      if (a.fdiv) wave_store_rows3(a.fdiv, base, lane, ncell, acc0, acc1, acc2);
      if (!EULER || f) wave_store_rows3(f, base, lane, ncell, res[0], res[1], res[2]);
      wave_store_rows3(a.pv, base, lane, ncell, h, pu, pv_);
      if (EULER) {  // rdyhip_euler_step: the forward-Euler update rides on the stores, F only if asked for
        const double n0 = h + dt * res[0], n1 = hu + dt * res[1], n2 = hv + dt * res[2];
        if (!a.o2l) {
          wave_store_rows3(a.u_out, base, lane, ncell, n0, n1, n2);
        } else if (active) {
          const int64_t c = a.o2l[o];
          RDY_MST(&a.u_out[3 * c + 0], n0);
          RDY_MST(&a.u_out[3 * c + 1], n1);
          RDY_MST(&a.u_out[3 * c + 2], n2);
        }
        if (td.send()) wave_store_send_rows(a, tile, tid, n0, n1, n2);  // the fused pack of the next state exchange (swe_kernels.h)
      }
    }
    idx  = idx1;
    idx1 = idx2;
    c1   = c2;
    hid1 = hid2;
    __syncthreads();  // the LDS records are rewritten by the next tile (dropping this barrier where the layout allows it gains nothing)
  }

  }
  courant_extra_edges_muscl<LIM>(a, g, dt, u, trk);
  block_courant_reduce<TILE>(a, trk.best, RDY_COLD(a, e_pos), trk.rec);
}

}  // namespace rdyhip
