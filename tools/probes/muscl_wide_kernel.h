// EXPERIMENT, NOT PART OF THE BUILD (round 4; rejected: 11-12 % slower than the 256-thread kernels on C3 triangles, the dam-break
// quads and the refined Houston mesh, profiles/r04_muscl_wide_ab.txt; parity suite green).  Kept as the record of what was tried.
// It reached its resource goal -- 127 VGPRs, two 512-thread workgroups = 16 waves per CU with the full cross-tile pipeline -- but
// half of its waves idle through phase 2 and most of the ring waves through phase G: 3 rounds x 512 threads of wave time per tile
// against ~3.8 x 256 for the 256-thread kernel, and only two tiles in flight per CU instead of three or four.
//
// Second order, 512 threads per tile: the fused MUSCL kernel of muscl_kernels.h with its work split over EIGHT waves instead of
// four, so that a tile's phases run twice as wide and a thread carries half the pipeline registers.
//
//   waves 0-3 ("cell waves", thread ct = tid owns tile cell ct)   phase 0: own state + centroid -> LDS; phase G: own gradient;
//                                                                  phase 2: flux sum, sources, stores
//   waves 4-7 ("ring waves", thread ct = tid - 256)                phase 0: state + centroid of ring cell ct (first ring, then
//                                                                  second ring) -> LDS; phase G: gradient of first-ring cell ct
//   all eight                                                      phase 1: edge `tid` of the tile (and edge tid + 512 for the few
//                                                                  tiles / lanes that have one: a 16 x 16 quad block has 544)
//
// Why: the 256-thread kernel holds, per thread, its own cell AND a ring cell AND two or three rounds of edge records of the tile
// in flight for the NEXT tile (the cross-tile software pipeline) -- 160-165 VGPRs on quads, three workgroups = 12 waves per CU;
// triangles keep four workgroups (16 waves) only by giving up most of the pipeline.  Here a thread carries one cell OR one ring
// cell and one (rarely two) edge records: the full pipeline fits 128 VGPRs, so two workgroups = 16 waves share a CU with every
// load of tile T+1 in flight while T is computed, the gradient phase runs own and ring cells side by side, and the edge phase is
// one round instead of 1.6 (triangles) / 2.1 (quads).  LDS: planes with compile-time strides as MusclSoA, edge fluxes in planes
// of their own (no overlay on the gradients: one barrier less per tile); two workgroups of ~54 KB.
//
// The arithmetic is muscl_kernels.h's (muscl_edge, ls_add / ls_solve, muscl_cell_sum, cell_results): same bits.  Selected at create
// when the tiles fit the fixed capacities (RDYHIP_MUSCL_WIDE=0 / 1 forces); meshes without locality keep the 256-thread kernels.
#pragma once
#include "muscl_kernels.h"

namespace rdyhip {

constexpr int WIDE = 2 * TILE;  // threads per workgroup

template <int NQ, int NG, int NE>
struct MusclWide {
  static constexpr bool fixed = true;
  static constexpr int  nq = NQ, ng = NG, ne = NE, n3 = 0;
  static __device__ __forceinline__ int qidx(int k, int j) { return k * NQ + j; }
  static __device__ __forceinline__ int gidx(int k, int j) { return k * NG + j; }
  static __device__ __forceinline__ int eidx(int c, int e_) { return c * NE + e_; }
  static constexpr size_t lds_bytes = sizeof(double) * (6 * (size_t)NG + 5 * (size_t)NQ + 4 * (size_t)NE) + sizeof(uint32_t) * (size_t)NE;
};
// capacities as MusclSoATri / MusclSoAQuad (the strides are not multiples of 64 doubles: no ds_read2st64 fusion)
using MusclWideTri  = MusclWide<520, 360, 520>;
using MusclWideQuad = MusclWide<424, 368, 552>;

// least-squares gradient of the cell in LDS slot `self` from the cells in slots nb[0..S-1] (-1: none), slot order
template <int S, class LAY>
__device__ __forceinline__ void wide_lds_gradient(const double *sq, int self, const int (&nb)[S], double (&gr)[6]) {
  const double q0 = MSQ(0, self), q1 = MSQ(1, self), q2 = MSQ(2, self), x0 = MSQ(3, self), y0 = MSQ(4, self);
  LsAcc        acc;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int n = nb[s];
    if (n < 0) continue;
    ls_add(acc, MSQ(3, n) - x0, MSQ(4, n) - y0, MSQ(0, n) - q0, MSQ(1, n) - q1, MSQ(2, n) - q2);
  }
  ls_solve(acc, gr);
}

template <int S, int SRC, bool OVW, int LIM, bool EULER, class LAY>
__global__ __launch_bounds__(WIDE) void swe_rhs_muscl_wide_kernel(const KernelArgs a, const MusclArgs g, const double dt, const double *__restrict__ u,
                                                                  double *__restrict__ f) {
  extern __shared__ double lds[];
  double   *sg  = lds;                      // 6 planes of LAY::ng: gradients of own + first-ring cells
  double   *sq  = lds + 6 * LAY::ng;        // 5 planes of LAY::nq: state + centroid of own cells, first ring, second ring
  double   *ef  = sq + 5 * LAY::nq;         // 4 planes of LAY::ne: the edge fluxes
  uint32_t *slr = reinterpret_cast<uint32_t *>(ef + 4 * LAY::ne);  // [LAY::ne] the tile's edge records (read by the gradient phase)
  const int  tid   = threadIdx.x;
  const int  ct    = tid & (TILE - 1);
  const bool cellw = tid < TILE;  // wave-uniform: waves 0-3 own the cells, waves 4-7 the ring

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
    d.e_off = v.x; d.h_off = v.y; d.b_off = v.z; d.cnt = (uint32_t)v.w;
    return d;
  };
  auto next_valid = [&](int i) -> int {  // INTERIOR phase skips tiles with ghost-adjacent cells (wave-uniform)
    if (a.phase == RDYHIP_PHASE_INTERIOR) {
      while (i < hi && tile_desc(tile_at(i)).halo()) i += step;
    }
    return i;
  };
  // Values a conditional load may leave untouched start from OPAQUE registers (muscl_kernels.h, the quads' pipeline: with a
  // constant on the other side hipcc folds the first use into the loading block and the wave waits where it requests)
  double   zero = 0.0;
  uint32_t ones = 0xFFFFFFFFu;
  asm volatile("" : "+v"(zero), "+v"(ones));
  // the CELL of this thread for the tile being started: own cell (cell waves) or ring cell (ring waves) -- one set of registers
  double   q[3] = {zero, zero, zero};
  double2  cxy  = make_double2(zero, zero);
  uint32_t w0 = ones, w1 = ones;  // cell waves: the slot references; ring waves: the first-ring stencil (bn_idx)
  struct EdgeRegs {
    uint32_t lr0 = 0, lr1 = 0;
    double   cs0 = 0.0, cs1 = 0.0;
    double2  md0 = make_double2(0.0, 0.0), md1 = make_double2(0.0, 0.0);
  };
  EdgeRegs E;
  // local id of this thread's cell for the tile at position i: own cell, or ring cell ct (-1: none)
  auto tile_id = [&](int i) -> int {
    const int      t_ = tile_at(i);
    const TileDesc d_ = tile_desc(t_);
    if (cellw) {
      const int o_ = t_ * TILE + ct;
      return (a.o2l && o_ < a.n_owned) ? a.o2l[o_] : (o_ < a.n_owned ? o_ : -1);
    }
    const int c0_ = load_uniform(g.c_off, t_), nh_ = d_.nh(), nc2_ = load_uniform(g.c_off, t_ + 1) - c0_;
    int       id = -1;
    if (ct < nh_) id = a.hcells[d_.h_off + ct];
    else if (ct < nh_ + nc2_) id = g.hcells2[c0_ + ct - nh_];
    return id;
  };
  auto issue_cell = [&](int t_, int id_, const TileDesc &d_) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = zero;
    cxy = make_double2(zero, zero);
    w0 = w1 = ones;
    int idv = id_;
    asm volatile("" : "+v"(idv));  // opaque predicate (see issue_cells in muscl_kernels.h)
    if (idv >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) q[k] = u[3 * (int64_t)idv + k];
      cxy = *reinterpret_cast<const double2 *>(g.cxy + 2 * (int64_t)idv);
    }
    if (cellw) {
      const int o_ = t_ * TILE + ct;
      int nown = a.n_owned;
      asm volatile("" : "+s"(nown));
      if (o_ < nown) {
        if (S == 3) {
          w0 = RDY_MLD(&reinterpret_cast<const uint32_t *>(a.slot_ref)[o_]);
        } else {
          const uint2 w = load_u2(reinterpret_cast<const uint2 *>(a.slot_ref) + o_);
          w0            = w.x;
          w1            = w.y;
        }
      }
    } else {
      int nh_ = d_.nh();
      asm volatile("" : "+s"(nh_));
      if (ct < nh_) {
        const uint2 w = load_u2(g.bn_idx + 4 * ((int64_t)d_.h_off + ct));
        w0            = w.x;
        w1            = w.y;
      }
    }
  };
  auto issue_edges = [&](const TileDesc &d_, EdgeRegs &R) {
    int ne_ = d_.ne();
    asm volatile("" : "+s"(ne_));
    if (tid < ne_) {
      R.lr0 = RDY_MLD(&a.e_lr[d_.e_off + tid]);
      R.cs0 = RDY_MLD(&a.e_cs[d_.e_off + tid]);
      R.md0 = load_d2(g.e_mid + 2 * ((int64_t)d_.e_off + tid));
    }
    if (tid + WIDE < ne_) {
      R.lr1 = RDY_MLD(&a.e_lr[d_.e_off + WIDE + tid]);
      R.cs1 = RDY_MLD(&a.e_cs[d_.e_off + WIDE + tid]);
      R.md1 = load_d2(g.e_mid + 2 * ((int64_t)d_.e_off + WIDE + tid));
    }
  };

  double best      = 0.0;
  int    best_slot = -1, best_o = 0;

  idx = next_valid(idx);
  if (idx < hi) {
    int idx1 = next_valid(idx + step);
    int id1  = -1;
    {  // prologue: both groups of the first tile, the id of the second tile's cell
      const int      t_ = tile_at(idx);
      const TileDesc d_ = tile_desc(t_);
      const int      id0 = tile_id(idx);
      issue_cell(t_, id0, d_);
      issue_edges(d_, E);
      if (idx1 < hi) id1 = tile_id(idx1);
      asm volatile("" ::"v"(id1));
    }
    while (true) {
      const int      tile = tile_at(idx);
      const TileDesc td   = tile_desc(tile);
      const int      ne = td.ne(), nh = td.nh();
      const int      c0 = load_uniform(g.c_off, tile), nc2 = load_uniform(g.c_off, tile + 1) - c0;
      const int      o      = tile * TILE + ct;
      const bool     active = cellw && o < a.n_owned;
      int            idx2 = hi, id2 = -1;
      const uint32_t r0 = w0, r1 = w1;  // slot references (cell waves) / first-ring stencil (ring waves) of THIS tile

      // ---- phase 0: state + centroid of own cells (cell waves) and ring cells (ring waves); the tile's edge records -> LDS
      {
        const int slot = cellw ? ct : TILE + ct;
        if (cellw || ct < nh + nc2) {
#pragma unroll
          for (int k = 0; k < 3; ++k) MSQ(k, slot) = q[k];
          MSQ(3, slot) = cxy.x;
          MSQ(4, slot) = cxy.y;
        }
        if (tid < ne) slr[tid] = E.lr0;
        if (tid + WIDE < ne) slr[tid + WIDE] = E.lr1;
      }
      __syncthreads();
      // ---- the next tile's loads, all of them, and the id of the cell after that
      __builtin_amdgcn_s_setprio(3);
      EdgeRegs N;
      if (idx1 < hi) {
        idx2 = next_valid(idx1 + step);
        if (idx2 < hi) id2 = tile_id(idx2);
        const int      t1 = tile_at(idx1);
        const TileDesc d1 = tile_desc(t1);
        issue_cell(t1, id1, d1);
        issue_edges(d1, N);
      }
      __builtin_amdgcn_s_setprio(0);

      // ---- phase G: least-squares gradients -- own cells (cell waves) beside first-ring cells (ring waves) -> LDS
      if (cellw) {
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
            nb[s]        = (jl == ct) ? jr : jl;
          }
          wide_lds_gradient<S, LAY>(sq, ct, nb, gr);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) MSG(k, ct) = gr[k];
      } else if (ct < nh) {
        const uint32_t ix[4] = {r0 & 0xFFFFu, r0 >> 16, r1 & 0xFFFFu, r1 >> 16};
        double         hg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (ix[0] == BN_GLOBAL) {  // a ghost cell: its gradient was computed by its owner
          const int hc = a.hcells[td.h_off + ct];
#pragma unroll
          for (int k = 0; k < 6; ++k) hg[k] = g.grad[6 * (int64_t)hc + k];
          // waited for inside the branch (at the merge hipcc would wait with vmcnt(0) in every wave, and with that for the next tile's groups)
          asm volatile("" ::"v"(hg[0]), "v"(hg[1]), "v"(hg[2]), "v"(hg[3]), "v"(hg[4]), "v"(hg[5]));
        } else {
          int nb[S];
#pragma unroll
          for (int s = 0; s < S; ++s) nb[s] = (ix[s] == BN_NONE) ? -1 : (int)ix[s];
          wide_lds_gradient<S, LAY>(sq, TILE + ct, nb, hg);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) MSG(k, TILE + ct) = hg[k];
      }
      __syncthreads();

      // ---- phase 1: every edge of the tile once, one per thread (a second one for the few records past 512)
      if (tid < ne) store_edge_flux<LAY>(a, ef, tid, muscl_edge<LIM, LAY>(a, td, dt, E.lr0, E.cs0, E.md0, sq, sg));
      __builtin_amdgcn_sched_barrier(0);
      if (tid + WIDE < ne) store_edge_flux<LAY>(a, ef, tid + WIDE, muscl_edge<LIM, LAY>(a, td, dt, E.lr1, E.cs1, E.md1, sq, sg));
      // the per-cell streams of phase 2 (cell waves), requested only here
      double kf[S];
      double dzx = 0.0, dzy = 0.0, nman = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int s = 0; s < S; ++s) kf[s] = 0.0;
      if (cellw) {
        __builtin_amdgcn_s_setprio(3);
        const int oc = active ? o : a.n_owned - 1;
#pragma unroll
        for (int s = 0; s < S; ++s) kf[s] = RDY_MLD(&a.coef[s * a.stride + oc]);
        dzx  = RDY_MLD(&a.dzdx[oc]);
        dzy  = RDY_MLD(&a.dzdy[oc]);
        nman = RDY_MLD(&a.mannings[oc]);
        s0   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 0]);
        s1   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 1]);
        s2   = RDY_MLD(&a.extsrc[3 * (int64_t)oc + 2]);
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();

      // ---- phase 2 (cell waves): per-cell sum in the reference's edge order, source terms, stores
      RDY_STREAMS_ARRIVE();
      double       acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, res[3] = {0.0, 0.0, 0.0}, pu = 0.0, pv_ = 0.0;
      double       h = 0.0, hu = 0.0, hv = 0.0;
      if (cellw) {
        h  = MSQ(0, ct);
        hu = MSQ(1, ct);
        hv = MSQ(2, ct);
      }
      if (active) {
        if (!OVW) {
          acc0 = f[3 * (int64_t)o + 0];
          acc1 = f[3 * (int64_t)o + 1];
          acc2 = f[3 * (int64_t)o + 2];
        }
        muscl_cell_sum<S, LAY>(a, r0, r1, kf, ef, dt, o, acc0, acc1, acc2, best, best_slot, best_o);
        const RiemannSide self = riemann_side(h, hu, hv, a.tiny_h, a.h_anuga_sq);
        pu                     = self.u;
        pv_                    = self.v;
        cell_results<SRC>(a, dt, h, hu, hv, acc0, acc1, acc2, dzx, dzy, nman, s0, s1, s2, res);
      }
      asm volatile("" ::"v"(id2));  // never "pending" at the loop header
      // the tile's wait on the next tile's groups comes BEFORE its own stores are issued (vmcnt counts stores too)
      asm volatile("" ::"v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(cxy.x), "v"(cxy.y), "v"(w0), "v"(w1));
      asm volatile("" ::"v"(N.lr0), "v"(N.lr1), "v"(N.cs0), "v"(N.cs1), "v"(N.md0.x), "v"(N.md0.y), "v"(N.md1.x), "v"(N.md1.y));
      E = N;
      __builtin_amdgcn_sched_barrier(0);
      if (cellw) {  // whole-line stores of the [cell][3] rows (wave_store_rows3, swe_kernels.h); all 64 lanes of a cell wave take part
        const int     lane  = tid & 63;
        const int64_t base  = 3 * ((int64_t)o - lane);
        const int     ncell = a.n_owned - (o - lane);
        if (a.fdiv) wave_store_rows3(a.fdiv, base, lane, ncell, acc0, acc1, acc2);
        if (!EULER || f) wave_store_rows3(f, base, lane, ncell, res[0], res[1], res[2], a.f_cached != 0);
        wave_store_rows3(a.pv, base, lane, ncell, h, pu, pv_);
        if (EULER) {
          const double n0 = h + dt * res[0], n1 = hu + dt * res[1], n2 = hv + dt * res[2];
          if (!a.o2l) {
            wave_store_rows3(a.u_out, base, lane, ncell, n0, n1, n2);
          } else if (active) {
            const int64_t c = a.o2l[o];
            RDY_MST(&a.u_out[3 * c + 0], n0);
            RDY_MST(&a.u_out[3 * c + 1], n1);
            RDY_MST(&a.u_out[3 * c + 2], n2);
          }
        }
      }
      if (idx1 >= hi) break;
      idx  = idx1;
      idx1 = idx2;
      id1  = id2;
      __syncthreads();  // the LDS planes are rewritten by the next tile
    }
  }
  block_courant_reduce<WIDE>(a, best, best_slot, best_o);
}

}  // namespace rdyhip
