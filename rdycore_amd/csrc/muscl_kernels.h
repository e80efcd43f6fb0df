// Second-order (MUSCL) variant of the SWE right-hand side for gfx950:
// ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213) with its helpers
// ComputeLeastSquaresGradients and ReconstructFaceValues
// (src/operator_fluxes_ceed.c:998-1042, 1155-1206).
//
//  * muscl_gradient_kernel: one thread per owned cell; the weighted
//    least-squares gradient of (h, hu, hv) from the cell's <= S neighbours with
//    coefficients precomputed at create (PrecomputeLSGradCoeffs), summed in the
//    reference's internal-edge order; writes grad[local cell][6].
//  * swe_rhs_muscl_kernel: the tiled three-phase structure of swe_kernels.h.
//    Phase 0 stages the conserved state and the gradient of the tile's own and
//    halo cells in LDS; phase 1 reconstructs the two limited face states of every
//    tile edge from LDS, derives the Riemann side data per edge side and evaluates
//    the Roe flux once per edge; phase 2 is the first-order kernel's (segmented
//    per-cell sum in the reference's order, source terms, stores).
//    Boundary edges stay first order (ApplyBoundaryFlux is unchanged by
//    numerics.second_order).
//
// Across ranks the reference solves each cut edge on the rank that owns it and
// adds the ghost side back with DMLocalToGlobal(ADD_VALUES).  Here every rank
// evaluates all edges of its owned cells (the cut ones redundantly, from
// bitwise-identical operands once the ghost gradients have been exchanged), so
// no reverse exchange exists; see rdycore_amd/halo.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swe_kernels.h"

namespace rdyhip {

struct MusclArgs {
  double       *grad;   // [num_cells][6]: dh/dx, dh/dy, dhu/dx, dhu/dy, dhv/dx, dhv/dy (local cell index)
  const double *e_geo;  // [nrec][4]: edge midpoint minus left centroid (x, y), minus right centroid (x, y)
  const double *gcx;    // [S][stride] least-squares coefficient of each slot's neighbour difference (q_nbr - q_self)
  const double *gcy;
};

// RDyLimiterType, include/private/rdyconfigimpl.h:67-71
constexpr int LIMITER_MINMOD = 0, LIMITER_NONE = 1, LIMITER_VANLEER = 2;

// Minmod / VanLeer / LimitSlope, src/operator_fluxes_ceed.c:1110-1138
template <int LIM>
__device__ __forceinline__ double limit_slope(double extrap, double half_dq) {
  if (LIM == LIMITER_NONE) return extrap;
  if (extrap * half_dq <= 0.0) return 0.0;
  if (LIM == LIMITER_VANLEER) return 2.0 * extrap * half_dq * rdy_rcp(extrap + half_dq);
  return fabs(extrap) < fabs(half_dq) ? extrap : half_dq;
}

// ComputeLeastSquaresGradients, src/operator_fluxes_ceed.c:998-1042, gathered per cell:
// grad(cell) = sum over its internal edges of c_edge * (q_nbr - q_cell), where c_edge is
// (cx_LR, cy_LR) if the cell is the edge's left cell and -(cx_RL, cy_RL) if it is the right
// one (the reference multiplies by q_R - q_L for both).
template <int S>
__global__ __launch_bounds__(BLOCK) void muscl_gradient_kernel(const KernelArgs a, const MusclArgs g, const double *__restrict__ u) {
  int tile = blockIdx.x;
  if (a.xcd_chunks > 0) tile = (blockIdx.x & 7) * a.xcd_chunks + (blockIdx.x >> 3);
  const int i = tile * BLOCK + threadIdx.x;
  if (i >= a.n_work) return;
  const int o = a.list ? a.list[i] : i;
  int32_t   id[S];
  bool      has_ghost = false;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    id[s] = a.nbr[s * a.stride + o];
    has_ghost |= (id[s] >= 0) && (id[s] & NBR_GHOST);
  }
  if (a.phase == RDYHIP_PHASE_INTERIOR && has_ghost) return;
  if (a.phase == RDYHIP_PHASE_HALO && !has_ghost) return;
  const int    c  = a.o2l ? a.o2l[o] : o;
  const double q0 = u[3 * (int64_t)c + 0], q1 = u[3 * (int64_t)c + 1], q2 = u[3 * (int64_t)c + 2];
  double       gr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (id[s] < 0) continue;  // boundary edge or unused slot: not part of the stencil
    const int    n  = id[s] & NBR_MASK;
    const double cx = g.gcx[s * a.stride + o], cy = g.gcy[s * a.stride + o];
    const double d0 = u[3 * (int64_t)n + 0] - q0, d1 = u[3 * (int64_t)n + 1] - q1, d2 = u[3 * (int64_t)n + 2] - q2;
    gr[0] += cx * d0;
    gr[1] += cy * d0;
    gr[2] += cx * d1;
    gr[3] += cy * d1;
    gr[4] += cx * d2;
    gr[5] += cy * d2;
  }
  double2 *dst = reinterpret_cast<double2 *>(g.grad + 6 * (int64_t)c);
  dst[0]       = make_double2(gr[0], gr[1]);
  dst[1]       = make_double2(gr[2], gr[3]);
  dst[2]       = make_double2(gr[4], gr[5]);
}

template <int S, int SRC, bool OVW, int LIM>
__global__ __launch_bounds__(TILE) void swe_rhs_muscl_kernel(const KernelArgs a, const MusclArgs g, const double dt, const double *__restrict__ u,
                                                              double *__restrict__ f) {
  extern __shared__ double lds[];
  const int nside = TILE + a.hmax;
  double   *sq    = lds;              // 3 planes of nside: h, hu, hv
  double   *sg    = lds + 3 * nside;  // 6 planes of nside: the gradient
  double   *ef0 = lds + 9 * nside, *ef1 = ef0 + a.emax, *ef2 = ef1 + a.emax, *eam = ef2 + a.emax;
  const int tid = threadIdx.x;

  // the tile sequence of this (persistent) workgroup: as in swe_rhs_tiled_kernel
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

  double best      = 0.0;
  int    best_slot = -1, best_o = 0;

  for (; idx < hi; idx += step) {
    const int      tile = __builtin_amdgcn_readfirstlane(a.list ? load_uniform(a.list, idx) : idx);
    const TileDesc td = a.tiles[tile], tn = a.tiles[tile + 1];
    if (a.phase == RDYHIP_PHASE_INTERIOR && td.halo) continue;  // wave-uniform
    const int  ne = tn.e_off - td.e_off, nh = tn.h_off - td.h_off;
    const int  o      = tile * TILE + tid;
    const bool active = o < a.n_owned;

    // ---- phase 0: conserved state and gradient of the tile's own and halo cells -> LDS
    {
      double q[3] = {0.0, 0.0, 0.0}, gr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (active) {
        const int c = a.o2l ? a.o2l[o] : o;
#pragma unroll
        for (int k = 0; k < 3; ++k) q[k] = u[3 * (int64_t)c + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) gr[k] = g.grad[6 * (int64_t)c + k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) sq[k * nside + tid] = q[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) sg[k * nside + tid] = gr[k];
      for (int j = tid; j < nh; j += TILE) {
        const int hc = a.hcells[td.h_off + j];
#pragma unroll
        for (int k = 0; k < 3; ++k) sq[k * nside + TILE + j] = u[3 * (int64_t)hc + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) sg[k * nside + TILE + j] = g.grad[6 * (int64_t)hc + k];
      }
    }
    __syncthreads();

    // ---- phase 1: every edge of the tile once
    for (int e = tid; e < ne; e += TILE) {
      const uint32_t lr = a.e_lr[td.e_off + e];
      double         cn, sn;
      edge_normal(lr, a.e_cs[td.e_off + e], cn, sn);
      const int jl = lr & EDGE_SLOT_MASK;
      RoeFlux   fl;
      bool      wet;
      if (!(lr & EDGE_BOUNDARY)) {
        const int      jr  = (lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK;
        const double2 *geo = reinterpret_cast<const double2 *>(g.e_geo + 4 * ((int64_t)td.e_off + e));
        const double2  dl = geo[0], dr = geo[1];
        double         ql[3], qr[3];
        // ReconstructFaceValues, src/operator_fluxes_ceed.c:1180-1200
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double cl_ = sq[k * nside + jl], cr_ = sq[k * nside + jr];
          const double extrap_l = sg[(2 * k) * nside + jl] * dl.x + sg[(2 * k + 1) * nside + jl] * dl.y;
          const double extrap_r = sg[(2 * k) * nside + jr] * dr.x + sg[(2 * k + 1) * nside + jr] * dr.y;
          const double dq       = cr_ - cl_;
          ql[k]                 = cl_ + limit_slope<LIM>(extrap_l, 0.5 * dq);
          qr[k]                 = cr_ + limit_slope<LIM>(extrap_r, -0.5 * dq);
        }
        ql[0] = fmax(0.0, ql[0]);  // depth clamped from below (1201-1203, swe_petsc.c:143-146)
        qr[0] = fmax(0.0, qr[0]);
        const RiemannSide L = riemann_side(ql[0], ql[1], ql[2], a.tiny_h, a.h_anuga_sq);
        const RiemannSide R = riemann_side(qr[0], qr[1], qr[2], a.tiny_h, a.h_anuga_sq);
        fl                  = roe_flux(L, R, sn, cn);
        wet                 = !(R.h < a.tiny_h && L.h < a.tiny_h);  // swe_petsc.c:184
      } else {
        const RiemannSide L  = riemann_side(sq[jl], sq[nside + jl], sq[2 * nside + jl], a.tiny_h, a.h_anuga_sq);
        const int         k  = a.tile_bk[td.b_off + ((lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK)];
        BoundaryFlux      bf = boundary_flux(a.btype[k], true, L, a.bvalues + 3 * (int64_t)k, sn, cn, a.tiny_h, a.h_anuga_sq);
        fl                   = bf.flux;
        wet                  = bf.wet;
        store_boundary_flux(a, k, fl, dt);
      }
      ef0[e] = fl.f0;
      ef1[e] = fl.f1;
      ef2[e] = fl.f2;
      eam[e] = wet ? fl.amax : -1.0;
    }
    __syncthreads();

    // ---- phase 2: per-cell sum in the reference's edge order, source terms, stores
    if (active) {
      uint32_t r0, r1 = 0;
      if (S == 3) {
        r0 = reinterpret_cast<const uint32_t *>(a.slot_ref)[o];
      } else {
        const uint2 w = reinterpret_cast<const uint2 *>(a.slot_ref)[o];
        r0            = w.x;
        r1            = w.y;
      }
      double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
      if (!OVW) {
        acc0 = f[3 * (int64_t)o + 0];
        acc1 = f[3 * (int64_t)o + 1];
        acc2 = f[3 * (int64_t)o + 2];
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        uint32_t ref;
        if (S == 3) {
          ref = (r0 >> (10 * s)) & 0x3FF;
          if (ref == REF3_EMPTY) continue;
        } else {
          const uint32_t w = (s < 2) ? r0 : r1;
          ref              = (s & 1) ? (w >> 16) : (w & 0xFFFFu);
          if (ref == SLOT_EMPTY) continue;
        }
        const double am = eam[ref];
        if (am != -1.0) {
          const double k = a.coef[s * a.stride + o];
          acc0 += ef0[ref] * k;
          acc1 += ef1[ref] * k;
          acc2 += ef2[ref] * k;
          const double cnum = am * fabs(k) * dt;
          if (cnum > best) {
            best      = cnum;
            best_slot = s;
            best_o    = o;
          }
        }
      }
      const double      h = sq[tid], hu = sq[nside + tid], hv = sq[2 * nside + tid];
      const RiemannSide self = riemann_side(h, hu, hv, a.tiny_h, a.h_anuga_sq);
      cell_epilogue<SRC>(a, o, dt, h, hu, hv, self.u, self.v, acc0, acc1, acc2, a.dzdx[o], a.dzdy[o], a.mannings[o], a.extsrc[3 * (int64_t)o + 0],
                         a.extsrc[3 * (int64_t)o + 1], a.extsrc[3 * (int64_t)o + 2], f);
    }
    __syncthreads();  // the LDS planes are rewritten by the next tile
  }
  block_courant_reduce<TILE>(a, best, best_slot, best_o);
}

}  // namespace rdyhip
