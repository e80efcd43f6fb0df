// HIP kernels of the SWE right-hand side for gfx950 (wave64, 160 KB LDS per CU).
//
// Two implementations of the same operator share the epilogue and the
// Courant reduction:
//
//  * swe_rhs_tiled_kernel (default): one workgroup = one tile of up to 256
//    consecutive owned cells (cut at create so that a tile's edge records and
//    halo cells fit fixed capacities: rdyhip_api.hip, layout_build_tiles).
//      phase 0  every thread loads its cell's state, derives the Riemann side
//               data (velocities, sqrt(h), sqrt(g h)) once and stages it in LDS;
//      phase 1  threads sweep the tile's edge list (every edge that touches a
//               tile cell, built at setup): left/right states come from LDS
//               (or from global memory for the few cells outside the tile),
//               each Roe flux is evaluated ONCE and parked in LDS;
//      phase 2  every thread sums its cell's <= S edge fluxes from LDS in the
//               reference's loop order (a segmented reduction with no
//               atomics), applies the source terms and writes F and the
//               primitive variables.
//    Edges cut by a tile boundary are evaluated by both tiles (bitwise
//    identically), so no inter-workgroup communication is needed.
//
//  * swe_rhs_kernel (RDYHIP_KERNEL=cell): one thread = one cell, every edge of
//    the cell evaluated by that thread (each interior edge twice overall).
//    Kept as the simple reference point for A/B measurements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rdyhip.h"
#include "swe_device.h"

namespace rdyhip {

constexpr int BLOCK = 256;  // threads per workgroup of the cell-centric kernel
constexpr int TILE = 256;   // threads per workgroup of the tiled kernels = the largest number of cells in a tile (a multiple of 64)
// Capacities every tile is cut to at create (layout_build_tiles): edge records = two register-resident rounds of the edge
// phase; halo cells (cells outside the tile that share an edge with it) per slots-per-cell flavour.  The LDS planes have
// these compile-time lengths, so every plane offset is an instruction immediate.
constexpr int TILE_MAX_REC = 2 * TILE;
constexpr int TILE_MAX_HALO_TRI = 104, TILE_MAX_HALO_QUAD = 112;

constexpr uint16_t SLOT_EMPTY = 0xFFFF;

// Data that is streamed through exactly once per launch -- the per-cell streams (flux coefficients, bed slopes,
// Manning n, external source), the tile edge records and the outputs F / primitive variables -- is loaded and
// stored with the non-temporal hint, which leaves the L2 to the state vector (the only data with reuse: halo cells
// are read by neighbouring tiles).  Measured A/B on one box: 0.342 ms vs 0.359 ms per 10 M-cell RHS; marking the
// state loads as well costs 5 % (0.375 ms), the stores alone cost 6 %.
#define RDY_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define RDY_LD(ptr) __builtin_nontemporal_load(ptr)

// persistent Courant diagnostic on the device
struct DeviceCourant {
  double  max_courant;
  int32_t pos;  // position of the edge in the reference's loop order, -1 = none
  int32_t pad;
};

// Kernel arguments.  A wave has 102 SGPRs; the persistent tiled kernels keep three tile descriptors, the loop state and
// every pointer their hot path reads in them for the whole launch.  Whatever only rare paths read -- the boundary-edge
// tables (a few boundary edges per tile at most), the Courant tie-break positions and the per-workgroup buckets (once per
// workgroup), the cell-centric kernel's neighbour tables -- lives in a device-resident ColdArgs block, written once at
// create and read through ONE pointer with scalar loads at the point of use (RDY_COLD), so that it occupies no register
// on the hot path.  (Round 2 passed all ~40 pointers by value: 51-83 SGPRs spilled to VGPR lanes per instantiation.)
struct TileDesc;

struct ColdArgs {
  const int32_t *nbr;        // [S][stride] neighbour ids (cell kernel, gradient kernel)
  const double  *cn, *sn;    // [S][stride] (cell kernel)
  const int32_t *pos;        // [S][stride] loop position of each slot's edge (Courant tie-break of the cell kernel)
  const int32_t *e_pos;      // [nrec + n_xedges] loop position of each tile edge record (Courant tie-break of the tiled kernels),
                             //                   then of each extra Courant edge
  // Internal edges of the local mesh that no owned cell's slot covers the way the reference's loop does (swe_petsc.c:275-296
  // runs over ALL local internal edges and divides by min(area_l, area_r) whoever owns the cells): edges between two ghost
  // cells, and owned / ghost edges whose ghost cell is the smaller one.  O(cut edges); evaluated -- the largest wave speed only --
  // by the threads of a launch that has ghost data (courant_extra_edges), so that a rank's Courant diagnostic BEFORE the
  // cross-rank reduction is the reference's, ties and all.
  int32_t        n_xedges;
  int32_t        x_rec0;     // = nrec: extra edge i is "record" x_rec0 + i of e_pos
  const int32_t *x_lr;       // [n_xedges][2] local ids of the left / right cell
  const uint32_t *x_flags;   // [n_xedges] EDGE_CS_IS_CN | EDGE_OTHER_NEG of the packed normal
  const double  *x_cs;       // [n_xedges] its stored component
  const double  *x_cfac;     // [n_xedges] len / min(area_l, area_r)
  const double  *x_mid;      // [n_xedges][2] edge midpoint (second order)
  const int32_t *btype;      // [K] condition type of boundary edge k
  const double  *bvalues;    // [K][3]
  double        *bflux;      // [K][3]
  double        *baccum;     // [K][3]
  const int32_t *tile_bk;    // boundary-edge ids k of each tile's boundary edges
  const int32_t *tile_boff;  // [ntiles + 1] first tile_bk entry of each tile
  double        *blk_max;    // [2 * maxgrid] the Courant diagnostic, one running (max, first position) bucket per workgroup slot;
  int32_t       *blk_pos;    //               merged by courant_finalize_kernel only when the host asks (rdyhip_update_diagnostics)
  // the pack of the next ghost update fused into the Euler-step kernels' stores (rdyhip_halo_fuse_pack): tiles flagged
  // TILE_SEND_FLAG also store the new state of their send cells into the halo's send buffer
  const int32_t  *send_off;  // [ntiles + 1] first send entry of each tile
  const uint32_t *send_ent;  // cell-in-tile (bits 0-7) | row of the send buffer (bits 8-31), sorted by tile
  double         *send_buf;  // [send cells][3]
  // second order: the ghost-adjacent cells' gradient launch (muscl_gradient_kernel over the halo cell list) also stores each
  // gradient into the rows of the send buffer that carry it to other ranks: no pack launch for the gradient exchange
  const int32_t *gsend_off;   // [n_halo + 1] first send row of the cell at each position of the halo cell list
  const int32_t *gsend_rows;  // rows of the send buffer ([send cells][6])
  double        *gsend_buf;
};

struct KernelArgs {
  int32_t        n_owned;    // owned cells
  int32_t        n_work;     // cell kernel: threads with work; tiled kernel: number of tiles to run
  int64_t        stride;     // distance between slot planes
  const int32_t *list;       // cell kernel: owned-cell ids; tiled kernel: tile ids; or nullptr for 0..n_work-1
  const int32_t *o2l;        // owned -> local cell id, or nullptr if the identity
  const double  *coef;       // [S][stride]
  const double  *dzdx, *dzdy;  // [n_owned]
  const double  *mannings;   // [n_owned]
  const double  *extsrc;     // [n_owned][3]
  double        *pv;         // [n_owned][3]
  double        *fdiv;       // [n_owned][3] or nullptr
  double        *u_out;      // EULER kernels: [num_cells][3] state after the step, owned rows written (u_out = u + dt F)
  const ColdArgs *cold;      // device memory: see above
  int32_t        bucket_off; // first Courant bucket of this launch (launch_rhs(bucket_half))
  int32_t        n_buckets;
  int32_t        reset_diag; // 1: this launch starts a new diagnostic (ResetOperatorDiagnostics): buckets are overwritten
  double         tiny_h, h_anuga_sq, xq_thresh;
  int32_t        phase;      // RDYHIP_PHASE_*
  int32_t        overwrite;  // 1: f = rhs, 0: f += rhs
  int32_t        xcd_chunks; // >0: blocks are dealt to XCDs in contiguous chunks of this many tiles
  // ---- tiled kernel only
  const struct TileDesc *tiles;  // [ntiles+1]
  const uint32_t *e_lr;      // [nrec] packed LDS slots of the edge's cells
  const double   *e_cs;      // [nrec] smaller-magnitude component of the edge normal
  const int32_t  *hcells;    // halo cells of each tile (local cell ids)
  const void     *slot_ref;  // index of each slot's edge in the tile's edge list: S == 3: uint32[n_owned] (3 x 10 bits),
                             // S == 4: uint16[n_owned][4]
  const double   *zc_local;  // [num_cells] vertex-averaged bed elevation (hydrostatic reconstruction only)
};

// Wave-uniform loads of read-only index data through the constant address space,
// so that hipcc emits scalar loads (s_load, counted by lgkmcnt) instead of vector
// loads whose s_waitcnt vmcnt(0) would also wait for every prefetch in flight.
template <typename T>
__device__ __forceinline__ T load_uniform(const T *p, int i) {
  typedef const __attribute__((address_space(4))) T *ConstPtr;
  return ((ConstPtr)(uintptr_t)p)[i];
}

// a field of the device-resident ColdArgs block: one scalar load at the point of use
#define RDY_COLD(a, field) (load_uniform(&(a).cold->field, 0))

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

// Source terms on the in-register flux sum, then the stores of F and the
// primitive variables.  ApplySourceSemiImplicit / ApplySourceImplicitXQ2018
// (src/swe/swe_petsc.c:704-804, 816-932); the source reads the pre-source F
// (src/operator.c:663).
template <int SRC>
__device__ __forceinline__ void cell_epilogue(const KernelArgs &a, int o, double dt, double h, double hu, double hv, double pu, double pv_, double acc0,
                                              double acc1, double acc2, double dzdx, double dzdy, double n, double s0, double s1, double s2,
                                              double *__restrict__ f) {
  const double bedx = dzdx * GRAVITY * h;
  const double bedy = dzdy * GRAVITY * h;
  double       tbx = 0.0, tby = 0.0;
  if (h >= a.tiny_h) {
    if (SRC == RDYHIP_SOURCE_SEMI_IMPLICIT) friction_semi_implicit(h, hu, hv, n, dt, acc1, acc2, bedx, bedy, tbx, tby);
    else friction_xq2018(h, hu, hv, n, dt, a.xq_thresh, acc1, acc2, bedx, bedy, tbx, tby);
  }
  if (a.fdiv) {
    a.fdiv[3 * (int64_t)o + 0] = acc0;
    a.fdiv[3 * (int64_t)o + 1] = acc1;
    a.fdiv[3 * (int64_t)o + 2] = acc2;
  }
  f[3 * (int64_t)o + 0] = acc0 + s0;
  f[3 * (int64_t)o + 1] = acc1 + (-bedx - tbx + s1);
  f[3 * (int64_t)o + 2] = acc2 + (-bedy - tby + s2);
  // primitive variables (swe_petsc.c:788-791): the regularised velocities of the Riemann states, zero below tiny_h
  a.pv[3 * (int64_t)o + 0] = h;
  a.pv[3 * (int64_t)o + 1] = pu;
  a.pv[3 * (int64_t)o + 2] = pv_;
}

// cell_epilogue split in two for the pipelined kernel: the arithmetic ...
template <int SRC>
__device__ __forceinline__ void cell_results(const KernelArgs &a, double dt, double h, double hu, double hv, double acc0, double acc1, double acc2,
                                             double dzdx, double dzdy, double n, double s0, double s1, double s2, double *out) {
  const double bedx = dzdx * GRAVITY * h;
  const double bedy = dzdy * GRAVITY * h;
  double       tbx = 0.0, tby = 0.0;
  if (h >= a.tiny_h) {
    if (SRC == RDYHIP_SOURCE_SEMI_IMPLICIT) friction_semi_implicit(h, hu, hv, n, dt, acc1, acc2, bedx, bedy, tbx, tby);
    else friction_xq2018(h, hu, hv, n, dt, a.xq_thresh, acc1, acc2, bedx, bedy, tbx, tby);
  }
  out[0] = acc0 + s0;
  out[1] = acc1 + (-bedx - tbx + s1);
  out[2] = acc2 + (-bedy - tby + s2);
}
// Stores one [cell][3] row per lane of a full wave, transposed so that the wave writes its 192 consecutive doubles
// with three unit-stride instructions.  `base`: index of the wave's first double; lanes >= ncell hold no cell.
// NT = false: store with the default cache policy instead of the non-temporal hint -- F when a separate update kernel reads it
// straight back (TSEULER's VecAXPY: RDYHIP_CONFIG_CACHED_F_STORES).  A COMPILE-TIME choice on purpose: round 4 first made it a
// wave-uniform run-time branch around the store, and the optimiser sank the two arms (same value, same address) into ONE store
// whose metadata is the intersection of the two -- the hint was dropped from every F / pv / u_out store of every kernel, first
// order -6 %, and nothing but a same-box A/B against the previous round's library showed it (tests/test_isa_cpu.py now counts
// the hinted stores of every RHS kernel in the built library).
template <bool NT = true>
__device__ __forceinline__ void wave_store_rows3(double *__restrict__ arr, int64_t base, int lane, int ncell, double v0, double v1, double v2) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int    e = 64 * k + lane, src = e / 3, comp = e - 3 * src;
    const double s0 = __shfl(v0, src, 64), s1 = __shfl(v1, src, 64), s2 = __shfl(v2, src, 64);
    if (src < ncell) {
      if (NT) RDY_ST(&arr[base + e], comp == 0 ? s0 : (comp == 1 ? s1 : s2));
      else arr[base + e] = comp == 0 ? s0 : (comp == 1 ? s1 : s2);
    }
  }
}

// A thread's running Courant maximum over the tiles it walks (swe_petsc.c:289-296: the reference keeps the FIRST edge in
// loop order that reaches the maximum).  Inside one cell the slots are in loop order, so "strictly greater" keeps the
// first; across the cells a thread visits nothing orders the loop positions (DMPlex numbers edges on its own,
// src/rdymesh.c:693-710).  The hot path only NOTES that a slot met the running maximum to the last bit (one compare per
// slot, the flag lives in a scalar mask); a cell that did goes through courant_resolve_tie -- cold code, taken by states
// that are uniform over what a thread has seen so far (a lake at rest, the flat pools of the reference's dam-break
// benchmark at t = 0) -- which compares loop positions.  That path must be cheap too (the dam break's first steps run
// through it in every tile): the incumbent's position is looked up once and kept; a candidate whose position is known
// to be larger is dismissed without touching memory -- the records of a tile are sorted by position, so the position of
// record 0 bounds every candidate of the tile from below and that of record COURANT_Q bounds the candidates from
// record COURANT_Q on (two scalars per tile, fetched a tile ahead; where the edge numbering follows the cells the first
// few dozen records of a tile are the edges it shares with earlier tiles, everything behind them is newer than any
// incumbent); otherwise one 4-byte load per tying cell, waited for inside the branch.
constexpr int COURANT_Q = 48;
struct CourantTrack {
  double best = 0.0;  // largest Courant number seen by this thread (> 0 only)
  int    rec  = 0;    // its tile edge record (index into e_lr / e_cs / e_pos)
  int    pos  = -1;   // the loop position of `rec` once a tie has made the thread look it up
};
// rec_new: the record of the FIRST slot of the current cell that equals t.best; tile_pos_lo: the smallest loop position among the
// current tile's records
__device__ __forceinline__ void courant_resolve_tie(const KernelArgs &a, CourantTrack &t, int rec_new, int tile_pos_lo) {
  if (rec_new == t.rec) return;  // the incumbent is that very slot
  const int32_t *e_pos = RDY_COLD(a, e_pos);
  if (t.pos < 0) {
    int po = e_pos[t.rec];
    // waited for HERE: a load whose result is first used after the merge costs every wave an s_waitcnt vmcnt(0) at the
    // merge, and with it a wait for the next tile's prefetch batch
    asm volatile("" : "+v"(po));
    t.pos = po;
  }
  if (t.pos < tile_pos_lo) return;
  int pn = e_pos[rec_new];
  asm volatile("" : "+v"(pn));
  if (pn < t.pos) {
    t.rec = rec_new;
    t.pos = pn;
  }
}

// `table[index]`: the loop position of the thread's candidate (read only by the lanes that hold the block's maximum)
template <int NT>
__device__ __forceinline__ void block_courant_reduce(const KernelArgs &a, double best, const int32_t *table, int64_t index) {
  __shared__ double s_max[NT / 64];
  __shared__ int    s_pos[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double    wmax = wave_max(best);
  if (lane == 0) s_max[wave] = wmax;
  __syncthreads();
  double bmax = s_max[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) bmax = fmax(bmax, s_max[w]);
  int p = INT32_MAX;
  if (best > 0.0 && best == bmax) p = table[index];
  p = wave_min(p);
  if (lane == 0) s_pos[wave] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    int bp = s_pos[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) bp = min(bp, s_pos[w]);
    double m = bmax;
    int    q = (bmax > 0.0) ? bp : -1;
    const int b       = blockIdx.x;
    double   *blk_max = RDY_COLD(a, blk_max) + a.bucket_off;
    int32_t  *blk_pos = RDY_COLD(a, blk_pos) + a.bucket_off;
    if (!a.reset_diag) {  // merge into the bucket: larger value, then the earlier edge (swe_petsc.c:291 keeps the first)
      const double pm = blk_max[b];
      const int    pq = blk_pos[b];
      if (pm > m || (pm == m && pm > 0.0 && pq < q)) {
        m = pm;
        q = pq;
      }
    }
    blk_max[b] = m;
    blk_pos[b] = q;
    if (a.reset_diag) {  // buckets no workgroup of this launch owns
      for (int k = b + gridDim.x; k < a.n_buckets; k += gridDim.x) {
        blk_max[k] = 0.0;
        blk_pos[k] = -1;
      }
    }
  }
}

__device__ __forceinline__ void store_boundary_flux(const KernelArgs &a, int k, const RoeFlux &fl, double dt) {
  // boundary_fluxes[b] and VecAXPY(boundary_fluxes_accum, dt, boundary_fluxes), swe_petsc.c:574, 623
  double *bflux = RDY_COLD(a, bflux), *baccum = RDY_COLD(a, baccum);
  bflux[3 * (int64_t)k + 0] = fl.f0;
  bflux[3 * (int64_t)k + 1] = fl.f1;
  bflux[3 * (int64_t)k + 2] = fl.f2;
  baccum[3 * (int64_t)k + 0] += dt * fl.f0;
  baccum[3 * (int64_t)k + 1] += dt * fl.f1;
  baccum[3 * (int64_t)k + 2] += dt * fl.f2;
}

// ---------------------------------------------------------------------------
// tiled kernel
// ---------------------------------------------------------------------------
// packed end points of a tile edge: LDS slot of the left cell | slot of the
// right cell << 11 | EDGE_BOUNDARY.  Slots 0..255 are the tile's own cells,
// 256.. the tile's halo cells (cells outside the tile that share an edge with
// it).  For a boundary edge the right field is the index into the tile's
// boundary-edge list.
constexpr uint32_t EDGE_SLOT_MASK = 0x7FF;
constexpr int      EDGE_R_SHIFT   = 11;
constexpr uint32_t EDGE_BOUNDARY  = 1u << 22;
// The unit normal (cn, sn) is stored as ONE double: the component of smaller
// magnitude; the other is +-sqrt(1 - cs^2) (well conditioned: cs^2 <= 1/2).
constexpr uint32_t EDGE_CS_IS_CN     = 1u << 23;  // the stored component is cn (else sn)
constexpr uint32_t EDGE_OTHER_NEG    = 1u << 24;  // the reconstructed component is negative
// second order: the edge belongs to another rank (edges.is_owned = false): ApplyInteriorFlux2R evaluates its Courant number
// there (swe_petsc.c:172-190); here its flux still feeds the owned cell, its wave speed is kept out of the diagnostic
constexpr uint32_t EDGE_NOT_OWNED    = 1u << 25;
// slot references of a triangle mesh (S == 3): 3 x 10 bits in one uint32, 0x3FF = unused
constexpr uint32_t REF3_EMPTY = 0x3FF;

__device__ __forceinline__ void edge_normal(uint32_t lr, double cs, double &cn, double &sn) {
  double other = rdy_sqrt(fma(-cs, cs, 1.0));
  if (lr & EDGE_OTHER_NEG) other = -other;
  const bool is_cn = lr & EDGE_CS_IS_CN;
  cn               = is_cn ? cs : other;
  sn               = is_cn ? other : cs;
}

struct TileDesc {  // 16 B, one per tile (+1 sentinel): everything about a tile in ONE scalar load of four registers
  int32_t  e_off;  // first edge record
  int32_t  h_off;  // first halo-cell entry
  int32_t  c_off;  // first owned cell (the tile's cells are c_off .. c_off + nc() - 1; a multiple of 16 wherever the numbering allows)
  uint32_t cnt;    // edge records (bits 0-10) | halo cells (bits 11-21) | cells - 1 (bits 22-29) | bit 30: a tile cell is sent to
                   // another rank (set while a halo with the fused pack is attached) | bit 31: a tile cell has a ghost neighbour
  __host__ __device__ int  ne() const { return (int)(cnt & 0x7FFu); }
  __host__ __device__ int  nh() const { return (int)((cnt >> 11) & 0x7FFu); }
  __host__ __device__ int  nc() const { return (int)((cnt >> 22) & 0xFFu) + 1; }
  __host__ __device__ bool halo() const { return (cnt >> 31) != 0; }
  __host__ __device__ bool send() const { return ((cnt >> 30) & 1u) != 0; }
};
constexpr uint32_t TILE_HALO_FLAG = 1u << 31;
constexpr uint32_t TILE_SEND_FLAG = 1u << 30;

// The fused pack: the lanes of a wave hand the new state of the tile's send cells to the send buffer.  Every wave scans the
// tile's (short) entry list 64 entries at a time and takes the entries whose cell it holds; the values come from the owning
// lane through a wave shuffle, so there is no LDS traffic and no barrier.  Runs only in tiles flagged TILE_SEND_FLAG.
__device__ __forceinline__ void wave_store_send_rows(const KernelArgs &a, int tile, int tid, double n0, double n1, double n2) {
  const int32_t  *soff = RDY_COLD(a, send_off);
  const int       s0 = load_uniform(soff, tile), s1 = load_uniform(soff, tile + 1);
  const uint32_t *sent = RDY_COLD(a, send_ent);
  double         *sbuf = RDY_COLD(a, send_buf);
  const int       lane = tid & 63, wave = tid >> 6;
  for (int base = s0; base < s1; base += 64) {
    const int      i   = base + lane;
    const uint32_t ent = i < s1 ? sent[i] : 0u;
    const int      j   = (int)(ent & 0xFFu);
    const double   v0 = __shfl(n0, j & 63, 64), v1 = __shfl(n1, j & 63, 64), v2 = __shfl(n2, j & 63, 64);
    if (i < s1 && (j >> 6) == wave) {
      const int64_t row = (int64_t)(ent >> 8);
      sbuf[3 * row + 0] = v0;
      sbuf[3 * row + 1] = v1;
      sbuf[3 * row + 2] = v2;
    }
  }
}

// per-cell streams consumed in phase 2 (slot references, flux coefficients,
// bed slopes, Manning n, external source)
template <int S>
struct CellStreams {
  uint32_t r0, r1;
  double   coef[S];
  double   dzdx, dzdy, nman, s0, s1, s2;
};

template <int S, bool HR>
__device__ __forceinline__ void load_streams(const KernelArgs &a, int o, bool active, CellStreams<S> &c) {
  c.r0 = c.r1 = 0xFFFFFFFFu;
#pragma unroll
  for (int s = 0; s < S; ++s) c.coef[s] = 0.0;
  c.dzdx = c.dzdy = c.nman = c.s0 = c.s1 = c.s2 = 0.0;
  if (active) {
    if (S == 3) {
      c.r0 = RDY_LD(&reinterpret_cast<const uint32_t *>(a.slot_ref)[o]);
    } else {
      const uint2 w = reinterpret_cast<const uint2 *>(a.slot_ref)[o];
      c.r0          = w.x;
      c.r1          = w.y;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) c.coef[s] = RDY_LD(&a.coef[s * a.stride + o]);
    if (!HR) {  // under HR the pressure correction of the flux carries the bed slope
      c.dzdx = RDY_LD(&a.dzdx[o]);
      c.dzdy = RDY_LD(&a.dzdy[o]);
    }
    c.nman = RDY_LD(&a.mannings[o]);
    c.s0   = RDY_LD(&a.extsrc[3 * (int64_t)o + 0]);
    c.s1   = RDY_LD(&a.extsrc[3 * (int64_t)o + 1]);
    c.s2   = RDY_LD(&a.extsrc[3 * (int64_t)o + 2]);
  }
}

// Persistent, software-pipelined form: a workgroup walks a sequence of tiles.
// All index and geometry data is static and u is read-only during the launch,
// so every global load of tile T+1 is issued while tile T is being computed
// (and the halo-cell ids of T+2 as well): its per-cell streams at the top of
// T, its cell states and edge records after T's first barrier.  A tile never
// waits for a dependent chain of global loads, and every load has about one
// tile time to complete, which is what keeps HBM busy at 3 workgroups per CU.
//
// HR = hydrostatic reconstruction (ApplyInteriorFluxHR, src/swe/swe_petsc.c:1000-1161):
// each interior edge's two depths are reconstructed against max(zc_l, zc_r) before
// the Roe solver, a pressure correction 0.5 g (h^2 - h_rec^2) (cn, sn) is added per
// side, and the source term drops the bed slope (CreatePetscSWESourceHROperator, 1229-1263).
//
// EULER = the forward-Euler update fused into the store phase (what TSEULER's VecAXPY does after the RHS,
// rdyhip_euler_step): the owned rows of a second state array receive u + dt F, F itself is stored only if f != nullptr.
// The LDS planes have COMPILE-TIME lengths (side-data slots: TILE own + the tile's halo cells; edge slots) -- every tile is
// cut to those capacities at create -- so every plane offset is an instruction immediate instead of a register and an add.
// The lengths are not multiples of 64 doubles on purpose: hipcc
// would fuse the reads of two planes into ds_read2st64_b64, which the LDS serves at half the rate of two ds_read_b64.
// HR kernels stage the velocities with the HR operator's wet test, "h > tiny_h" (swe_petsc.c:1061-1064), instead of
// ComputeRiemannVelocities' "h < tiny_h => 0" (62): the edge phase then needs no per-edge selects.  The two rules differ only
// at h == tiny_h exactly, where the boundary edges and the primitive variables (which keep the other rule) recompute.
__device__ __forceinline__ void hr_velocity_rule(RiemannSide &s, double tiny_h) {
  const bool wet = s.h > tiny_h;
  s.u            = wet ? s.u : 0.0;
  s.v            = wet ? s.v : 0.0;
}
// The two reconstructed Riemann sides of an interior edge (swe_petsc.c:1046-1071) from the staged ones (velocities already
// under the HR operator's rule, hr_velocity_rule) and the two bed elevations.
__device__ __forceinline__ void hr_reconstruct(const RiemannSide &L, const RiemannSide &R, double zl, double zr, RiemannSide &Lr, RiemannSide &Rr) {
  const double z_max = fmax(zl, zr);
  Lr.h = fmax(0.0, (L.h + zl) - z_max);
  Rr.h = fmax(0.0, (R.h + zr) - z_max);
  Lr.u = L.u; Lr.v = L.v; Rr.u = R.u; Rr.v = R.v;
  // Only the side with the LOWER bed changes its depth; the other one keeps (h + z) - z, i.e. its own depth up to one
  // rounding of the sum, and its staged square root stands in for the root of that value (relative difference
  // <= ulp(h + z) / (4 h): ~1e-13 for 1 cm of water over a bed at 50 m).  One square root
  // per edge instead of two.  Where the reconstructed depth of that side is clamped to zero -- a negative depth
  // (a drying overshoot: its staged root is NaN) or a film thinner than ulp(z) -- the root is the reference's
  // sqrt(0) = 0, not the staged one (swe_petsc.c:1051-1053).
  // (equal beds: both sides keep their depth, both take their staged root -- the rule must not depend on which cell
  // is called left, or two edges that tie in the reference would not tie here)
  const bool   l_high = zl >= zr, r_high = zr >= zl;
  const double sx     = rdy_sqrt(l_high ? Rr.h : Lr.h);
  Lr.sqh = l_high ? (Lr.h > 0.0 ? L.sqh : 0.0) : sx;
  Rr.sqh = r_high ? (Rr.h > 0.0 ? R.sqh : 0.0) : sx;
  Lr.c   = SQRT_GRAVITY * Lr.sqh;
  Rr.c   = SQRT_GRAVITY * Rr.sqh;
}
// The extra Courant edges (ColdArgs::x_*), first order / HR: the largest wave speed of each, from global memory, through the
// SAME device functions as the edge phase (riemann_side, hr_reconstruct, roe_flux: every operation that feeds amax is an
// explicit fma / uncontracted product, so the value has the bits the edge phase would give it -- a tie stays a tie).  Spread
// over the threads of the launch, after its tile loop; a launch of the INTERIOR phase (no ghost data yet) skips it.
template <bool HR>
__device__ __forceinline__ void courant_extra_edges(const KernelArgs &a, double dt, const double *__restrict__ u, CourantTrack &t) {
  const int nx = RDY_COLD(a, n_xedges);
  if (nx == 0 || a.phase == RDYHIP_PHASE_INTERIOR) return;  // uniform
  const int32_t  *xlr = RDY_COLD(a, x_lr);
  const uint32_t *xfl = RDY_COLD(a, x_flags);
  const double   *xcs = RDY_COLD(a, x_cs), *xcf = RDY_COLD(a, x_cfac);
  const int       rec0 = RDY_COLD(a, x_rec0);
  for (int i = blockIdx.x * TILE + threadIdx.x; i < nx; i += gridDim.x * TILE) {
    const int64_t l = xlr[2 * i], r = xlr[2 * i + 1];
    RiemannSide   L = riemann_side(u[3 * l + 0], u[3 * l + 1], u[3 * l + 2], a.tiny_h, a.h_anuga_sq);
    RiemannSide   R = riemann_side(u[3 * r + 0], u[3 * r + 1], u[3 * r + 2], a.tiny_h, a.h_anuga_sq);
    double        cn, sn;
    edge_normal(xfl[i], xcs[i], cn, sn);
    double am;
    bool   wet = !(R.h < a.tiny_h && L.h < a.tiny_h);
    if (!HR) {
      am = roe_flux(L, R, sn, cn).amax;
    } else {
      hr_velocity_rule(L, a.tiny_h);
      hr_velocity_rule(R, a.tiny_h);
      RiemannSide Lr, Rr;
      hr_reconstruct(L, R, a.zc_local[l], a.zc_local[r], Lr, Rr);
      am  = roe_flux(Lr, Rr, sn, cn).amax;
      wet = wet && (Lr.h > a.tiny_h || Rr.h > a.tiny_h);
    }
    if (wet) {
      const double cnum = am * xcf[i] * dt;
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

constexpr int TILED_NS_TRI = TILE + TILE_MAX_HALO_TRI, TILED_NS_QUAD = TILE + TILE_MAX_HALO_QUAD, TILED_NE = TILE_MAX_REC + 8;
constexpr size_t tiled_lds_bytes(int S, bool hr) {
  return sizeof(double) * ((hr ? 6 : 5) * (size_t)(S == 3 ? TILED_NS_TRI : TILED_NS_QUAD) + 2 * (size_t)TILE + (hr ? 6 : 4) * (size_t)TILED_NE);
}
// FNT = false: F (EULER: u_out) is stored without the non-temporal hint (RDYHIP_CONFIG_CACHED_F_STORES / states that fit the Infinity Cache)
template <int S, int SRC, bool OVW, bool HR, bool EULER = false, bool FNT = true>
__global__ __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu(3, 3))) void swe_rhs_tiled_kernel(const KernelArgs a, const double dt, const double *__restrict__ u,
                                                              double *__restrict__ f) {
  // edge-record rounds held in registers: every tile has <= TILE_MAX_REC = 2 x 256 edge records (a 256-cell tile of a
  // well-numbered triangle mesh 1.6 per cell; a quad tile is cut at 240 cells, a 16 x 15 block = 511 records; round 4 ran
  // 256-cell quad tiles, 544 records, with a third round in which 32 of 256 lanes had work)
  constexpr bool HR_STAGED_VEL = HR;
  extern __shared__ double lds[];
  constexpr int nside = S == 3 ? TILED_NS_TRI : TILED_NS_QUAD, nedge = TILED_NE;
  double   *sd_h = lds, *sd_u = lds + nside, *sd_v = lds + 2 * nside, *sd_sq = lds + 3 * nside, *sd_c = lds + 4 * nside;
  double   *sd_hu = lds + 5 * nside, *sd_hv = sd_hu + TILE;
  double   *ef0 = sd_hv + TILE, *ef1 = ef0 + nedge, *ef2 = ef1 + nedge, *eam = ef2 + nedge;
  // HR only: bed elevation per slot; per edge the momentum flux as the RIGHT cell sees it (ef1 / ef2 then hold the left
  // cell's: the Roe flux plus each side's own pressure correction)
  double   *sd_zc = eam + nedge, *efr1 = sd_zc + nside, *efr2 = efr1 + nedge;
  const int tid = threadIdx.x;

  // ---- this workgroup's tile sequence.  Block ids are dealt round-robin to the
  // 8 XCDs: give each XCD a contiguous range of tiles and let its workgroups
  // interleave inside it, so concurrently running tiles are neighbours and
  // their halo cells hit that XCD's own L2.
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
  auto next_valid = [&](int i) -> int {  // INTERIOR phase skips tiles with ghost-adjacent cells (wave-uniform)
    if (a.phase == RDYHIP_PHASE_INTERIOR) {
      while (i < hi && tile_desc(tile_at(i)).halo()) i += step;
    }
    return i;
  };

  CourantTrack trk;

  idx = next_valid(idx);
  if (idx < hi) {
    // ---- prologue: everything tile T0 needs, and the halo ids of T1
    int      tile = tile_at(idx);
    TileDesc td = tile_desc(tile);
    double   pu0 = 0.0, pu1 = 0.0, pu2 = 0.0;  // own cell state of the tile being started
    double   ph0 = 0.0, ph1 = 0.0, ph2 = 0.0;  // state of this thread's halo cell
    double   pz = 0.0, phz = 0.0;              // HR: bed elevation of the own / halo cell
    uint32_t lr0 = 0, lr1 = 0;                 // the two rounds of edge records
    double   cs0 = 0.0, cs1 = 0.0;
    CellStreams<S> cur;
    {
      const int o = td.c_off + tid;
      if (tid < td.nc()) {
        const int c = a.o2l ? a.o2l[o] : o;
        pu0 = u[3 * (int64_t)c + 0]; pu1 = u[3 * (int64_t)c + 1]; pu2 = u[3 * (int64_t)c + 2];
        if (HR) pz = a.zc_local[c];
      }
      if (tid < td.nh()) {
        const int hc = a.hcells[td.h_off + tid];
        ph0 = u[3 * (int64_t)hc + 0]; ph1 = u[3 * (int64_t)hc + 1]; ph2 = u[3 * (int64_t)hc + 2];
        if (HR) phz = a.zc_local[hc];
      }
      const int ne = td.ne();
      if (tid < ne) { lr0 = RDY_LD(&a.e_lr[td.e_off + tid]); cs0 = RDY_LD(&a.e_cs[td.e_off + tid]); }
      if (tid + TILE < ne) { lr1 = RDY_LD(&a.e_lr[td.e_off + TILE + tid]); cs1 = RDY_LD(&a.e_cs[td.e_off + TILE + tid]); }
      load_streams<S, HR>(a, o, tid < td.nc(), cur);
    }
    // the loop positions of the tile's records 0 and COURANT_Q (sorted by position): what lets the Courant tie path dismiss a
    // candidate without touching memory (CourantTrack); fetched a tile ahead like everything else
    int      pos_lo = load_uniform(RDY_COLD(a, e_pos), td.e_off);
    int      pos_q  = load_uniform(RDY_COLD(a, e_pos), td.e_off + min(COURANT_Q, td.ne() - 1));
    int      idx1 = next_valid(idx + step);
    int      tile1 = 0, hid1 = 0, c1 = 0;  // next tile: this thread's halo cell and own cell (local ids)
    TileDesc td1 = td;
    if (idx1 < hi) {
      tile1 = tile_at(idx1);
      td1   = tile_desc(tile1);
      if (tid < td1.nh()) hid1 = a.hcells[td1.h_off + tid];
      const int o1 = td1.c_off + tid;
      c1           = (a.o2l && tid < td1.nc()) ? a.o2l[o1] : o1;
    }

    while (true) {
      const int  ne = td.ne(), nh = td.nh();
      const int  o      = td.c_off + tid;
      const bool active = tid < td.nc();

      // ---- phase 0: Riemann side data of the tile's own and halo cells -> LDS
      {
        RiemannSide self;
        self.h = self.u = self.v = self.sqh = self.c = 0.0;
        if (active) self = riemann_side(pu0, pu1, pu2, a.tiny_h, a.h_anuga_sq);
        if (HR_STAGED_VEL) hr_velocity_rule(self, a.tiny_h);
        sd_h[tid] = self.h; sd_u[tid] = self.u; sd_v[tid] = self.v; sd_sq[tid] = self.sqh; sd_c[tid] = self.c;
        sd_hu[tid] = pu1;
        sd_hv[tid] = pu2;
        if (HR) sd_zc[tid] = pz;
        if (tid < nh) {
          RiemannSide hs = riemann_side(ph0, ph1, ph2, a.tiny_h, a.h_anuga_sq);
          if (HR_STAGED_VEL) hr_velocity_rule(hs, a.tiny_h);
          sd_h[TILE + tid] = hs.h; sd_u[TILE + tid] = hs.u; sd_v[TILE + tid] = hs.v; sd_sq[TILE + tid] = hs.sqh; sd_c[TILE + tid] = hs.c;
          if (HR) sd_zc[TILE + tid] = phz;
        }
      }
      __syncthreads();

      // ---- software pipeline: EVERY global load of the next tile (cell states, edge records, per-cell
      // streams) and the halo ids of the one after are issued here in one batch.  hipcc waits with
      // s_waitcnt vmcnt(0) wherever a loaded register is first used, so the batch must not be followed
      // by any such use until the end of the tile: phases 1 and 2 below touch only registers whose
      // loads completed a tile ago, and the first use of this batch is the register rotation at the
      // very end (one wait per tile, a whole flux phase after the loads were issued).
      // (a) which tile comes after the next, and the ids it needs (the only dependent loads: first)
      __builtin_amdgcn_s_setprio(3);  // waves that reach their load batch issue it ahead of waves that are computing (+0.7..1 %)
      int      idx2 = hi, tile2 = 0, hid2 = 0, c2 = 0;
      TileDesc td2 = td1;
      if (idx1 < hi) {
        idx2 = next_valid(idx1 + step);
        if (idx2 < hi) {
          tile2 = tile_at(idx2);
          td2   = tile_desc(tile2);
          if (tid < td2.nh()) hid2 = a.hcells[td2.h_off + tid];
          const int o2 = td2.c_off + tid;
          c2           = (a.o2l && tid < td2.nc()) ? a.o2l[o2] : o2;
        }
      }
      // (b) the next tile's cell states, edge records and per-cell streams
      uint32_t nlr0 = 0, nlr1 = 0;
      double   ncs0 = 0.0, ncs1 = 0.0;
      int      npos_lo = 0, npos_q = 0;
      CellStreams<S> nxt;
      if (idx1 < hi) {
        npos_lo = load_uniform(RDY_COLD(a, e_pos), td1.e_off);
        npos_q  = load_uniform(RDY_COLD(a, e_pos), td1.e_off + min(COURANT_Q, td1.ne() - 1));
        if (tid < td1.nc()) {
          pu0 = u[3 * (int64_t)c1 + 0]; pu1 = u[3 * (int64_t)c1 + 1]; pu2 = u[3 * (int64_t)c1 + 2];
          if (HR) pz = a.zc_local[c1];
        }
        if (tid < td1.nh()) {
          ph0 = u[3 * (int64_t)hid1 + 0]; ph1 = u[3 * (int64_t)hid1 + 1]; ph2 = u[3 * (int64_t)hid1 + 2];
          if (HR) phz = a.zc_local[hid1];
        }
        const int ne1 = td1.ne();
        if (tid < ne1) { nlr0 = RDY_LD(&a.e_lr[td1.e_off + tid]); ncs0 = RDY_LD(&a.e_cs[td1.e_off + tid]); }
        if (tid + TILE < ne1) { nlr1 = RDY_LD(&a.e_lr[td1.e_off + TILE + tid]); ncs1 = RDY_LD(&a.e_cs[td1.e_off + TILE + tid]); }
      }
      load_streams<S, HR>(a, td1.c_off + tid, idx1 < hi && tid < td1.nc(), nxt);
      __builtin_amdgcn_s_setprio(0);

      // ---- phase 1: every edge of the tile once, operands from LDS only
      // (ApplyInteriorFlux / ApplyBoundaryFlux, swe_petsc.c:215-316, 506-630)
      auto do_edge = [&](int e, uint32_t lr, double cs) {
        double cn, sn;
        edge_normal(lr, cs, cn, sn);
        const int   jl = lr & EDGE_SLOT_MASK;
        RiemannSide L;
        L.h = sd_h[jl]; L.u = sd_u[jl]; L.v = sd_v[jl]; L.sqh = sd_sq[jl]; L.c = sd_c[jl];
        RoeFlux fl;
        bool    wet;
        if (!(lr & EDGE_BOUNDARY)) {
          const int   jr = (lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK;
          RiemannSide R;
          R.h = sd_h[jr]; R.u = sd_u[jr]; R.v = sd_v[jr]; R.sqh = sd_sq[jr]; R.c = sd_c[jr];
          if (!HR) {
            fl  = roe_flux(L, R, sn, cn);
            wet = !(R.h < a.tiny_h && L.h < a.tiny_h);
          } else {
            // hydrostatic reconstruction (swe_petsc.c:1046-1071); velocities are kept, with the strict
            // "h > tiny_h" wet test of the HR operator (1061-1064)
            const double zl = sd_zc[jl], zr = sd_zc[jr];
            RiemannSide  Lr, Rr;
            hr_reconstruct(L, R, zl, zr, Lr, Rr);
            fl     = roe_flux(Lr, Rr, sn, cn);
            const bool outer = !(R.h < a.tiny_h && L.h < a.tiny_h);       // 1094
            wet              = outer && (Lr.h > a.tiny_h || Rr.h > a.tiny_h);  // inner guard, 1112
            // pressure correction g (h^2 - h*^2) / 2 n of each side, applied whenever the outer guard holds (1136-1152), the
            // Roe flux only under the inner one: both folded into one momentum flux per side here, so that phase 2 reads
            // four values per slot instead of seven
            const double pl = outer ? 0.5 * GRAVITY * (L.h * L.h - Lr.h * Lr.h) : 0.0;
            const double pr = outer ? 0.5 * GRAVITY * (R.h * R.h - Rr.h * Rr.h) : 0.0;
            if (!wet) fl.f0 = fl.f1 = fl.f2 = 0.0;
            efr1[e] = fma(pr, cn, fl.f1);
            efr2[e] = fma(pr, sn, fl.f2);
            fl.f1   = fma(pl, cn, fl.f1);
            fl.f2   = fma(pl, sn, fl.f2);
          }
        } else {
          // boundary edges: HR is a no-op (operator_fluxes_petsc.c:57-58); their cell is the left one
          const int    k  = RDY_COLD(a, tile_bk)[load_uniform(RDY_COLD(a, tile_boff), tile) + ((lr >> EDGE_R_SHIFT) & EDGE_SLOT_MASK)];
          if (HR_STAGED_VEL && L.h == a.tiny_h) {
            // the staged velocities carry the HR operator's "h > tiny_h" rule, ApplyBoundaryFlux wants ComputeRiemannVelocities'
            // "h < tiny_h => 0" (swe_petsc.c:62): they differ at equality only (the left cell of a boundary edge is a tile cell)
            const RiemannSide s = riemann_side(L.h, sd_hu[jl], sd_hv[jl], a.tiny_h, a.h_anuga_sq);
            L.u = s.u;
            L.v = s.v;
          }
          BoundaryFlux bf = boundary_flux(RDY_COLD(a, btype)[k], true, L, RDY_COLD(a, bvalues) + 3 * (int64_t)k, sn, cn, a.tiny_h, a.h_anuga_sq);
          fl              = bf.flux;
          wet             = bf.wet;
          store_boundary_flux(a, k, fl, dt);
          if (HR) {
            if (!wet) fl.f0 = fl.f1 = fl.f2 = 0.0;  // phase 2 of the HR variant adds every edge's flux
            // a boundary edge has no right cell, but phase 2 picks the side by the sign of the slot's coefficient: keep the
            // right-hand copy defined too (never stale LDS of the previous tile, whatever the coefficient's sign bit says)
            efr1[e] = fl.f1;
            efr2[e] = fl.f2;
          }
        }
        ef0[e] = fl.f0;
        ef1[e] = fl.f1;
        ef2[e] = fl.f2;
        eam[e] = wet ? fl.amax : -1.0;  // -1 marks a dry-dry edge (skipped, swe_petsc.c:285)
      };
      // Both rounds take their records from registers.  This loop must contain no global load on any
      // path: hipcc would put an s_waitcnt vmcnt(0) at the merge, and every round would then wait for
      // the whole prefetch batch issued above.
#pragma unroll 1
      for (int r = 0; r < 2; ++r) {
        const int e = tid + r * TILE;
        if (e < ne) do_edge(e, r == 0 ? lr0 : lr1, r == 0 ? cs0 : cs1);
      }
      __syncthreads();

      // ---- phase 2: per-cell sum in the reference's edge order, source terms; the stores come last
      double out[6]      = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // F[3], then the primitive variables (h, u, v)
      double acc_fdiv[3] = {0.0, 0.0, 0.0};
      double own_hu = 0.0, own_hv = 0.0;  // EULER only
      if (active) {
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
        if (!OVW) {  // ApplyOperator semantics: add into f (and let the friction term see it)
          acc0 = f[3 * (int64_t)o + 0];
          acc1 = f[3 * (int64_t)o + 1];
          acc2 = f[3 * (int64_t)o + 2];
        }
        bool      tie      = false;  // a slot of this cell met the thread's running Courant maximum to the last bit
        const int rec_in   = trk.rec;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          uint32_t ref;
          if (S == 3) {
            ref = (cur.r0 >> (10 * s)) & 0x3FF;
            if (ref == REF3_EMPTY) continue;
          } else {
            const uint32_t w = (s < 2) ? cur.r0 : cur.r1;
            ref              = (s & 1) ? (w >> 16) : (w & 0xFFFFu);
            if (ref == SLOT_EMPTY) continue;
          }
          const double am = eam[ref];
          const double k  = cur.coef[s];
          if (HR) {
            // a dry-dry edge (am == -1) carries zero Roe flux here but still its pressure correction
            const bool left = __builtin_signbit(k);  // this cell is the edge's left (k = -len/area, -0.0 for a degenerate edge) or right cell
            acc0 += ef0[ref] * k;
            acc1 += (left ? ef1 : efr1)[ref] * k;
            acc2 += (left ? ef2 : efr2)[ref] * k;
          } else if (am != -1.0) {
            acc0 += ef0[ref] * k;
            acc1 += ef1[ref] * k;
            acc2 += ef2[ref] * k;
          }
          if (am != -1.0) {
            // len/area_self: the max over an edge's two cells is len / min(area_l, area_r) (swe_petsc.c:289)
            const double cnum = am * fabs(k) * dt;
            tie |= cnum == trk.best;
            if (cnum > trk.best) {
              trk.best = cnum;
              trk.rec  = td.e_off + (int)ref;
            }
          }
        }
        if (trk.rec != rec_in) trk.pos = -1;
        // cold: which of the equal edges comes first in the reference's loop (CourantTrack).  Dismissed at once where the
        // incumbent's position is known and smaller than every position of this tile.
        if (tie && !(trk.pos >= 0 && trk.pos < pos_lo)) {
          int first = -1;  // the first slot of this cell at the running maximum: the only one that can come before the incumbent
#pragma unroll
          for (int s = 0; s < S; ++s) {
            uint32_t ref;
            if (S == 3) {
              ref = (cur.r0 >> (10 * s)) & 0x3FF;
              if (ref == REF3_EMPTY) continue;
            } else {
              const uint32_t w = (s < 2) ? cur.r0 : cur.r1;
              ref              = (s & 1) ? (w >> 16) : (w & 0xFFFFu);
              if (ref == SLOT_EMPTY) continue;
            }
            const double am = eam[ref];
            if (first < 0 && am != -1.0 && am * fabs(cur.coef[s]) * dt == trk.best) first = (int)ref;
          }
          if (first >= 0) courant_resolve_tie(a, trk, td.e_off + first, first >= COURANT_Q ? pos_q : pos_lo);
        }
        acc_fdiv[0] = acc0;
        acc_fdiv[1] = acc1;
        acc_fdiv[2] = acc2;
        cell_results<SRC>(a, dt, sd_h[tid], sd_hu[tid], sd_hv[tid], acc0, acc1, acc2, cur.dzdx, cur.dzdy, cur.nman, cur.s0, cur.s1, cur.s2, out);
        out[3] = sd_h[tid];
        out[4] = sd_u[tid];
        out[5] = sd_v[tid];
        if (HR_STAGED_VEL && out[3] == a.tiny_h) {  // primitive variables: zero BELOW tiny_h (swe_petsc.c:788-791), see hr_velocity_rule
          const RiemannSide s = riemann_side(out[3], sd_hu[tid], sd_hv[tid], a.tiny_h, a.h_anuga_sq);
          out[4] = s.u;
          out[5] = s.v;
        }
        if (EULER) {
          own_hu = sd_hu[tid];
          own_hv = sd_hv[tid];
        }
      }
      const bool last = idx1 >= hi;
      // The tile's single wait on global loads: the youngest load of the prefetch batch is "used" here, before
      // this tile's stores are issued (vmcnt counts stores too, and the register rotation below is
      // materialised at the very end of the loop body).
      asm volatile("" ::"v"(pu0), "v"(pu1), "v"(pu2), "v"(ph0), "v"(ph1), "v"(ph2), "v"(pz), "v"(phz), "v"(nlr0), "v"(nlr1), "v"(ncs0), "v"(ncs1),
                   "v"(hid2), "v"(c2));
      asm volatile("" ::"v"(nxt.r0), "v"(nxt.r1), "v"(nxt.coef[0]), "v"(nxt.coef[1]), "v"(nxt.coef[2]), "v"(nxt.coef[S - 1]), "v"(nxt.dzdx),
                   "v"(nxt.dzdy), "v"(nxt.nman), "v"(nxt.s0), "v"(nxt.s1), "v"(nxt.s2));
      // rotate the pipeline registers, store
      const bool send_tile = EULER && td.send();  // wave-uniform
      const int  tile_cur  = tile, nc_cur = td.nc();
      idx = idx1; tile = tile1; td = td1;
      idx1 = idx2; tile1 = tile2; td1 = td2; hid1 = hid2; c1 = c2;
      pos_lo = npos_lo;
      pos_q  = npos_q;
      lr0 = nlr0; lr1 = nlr1; cs0 = ncs0; cs1 = ncs1;
      cur = nxt;
      __builtin_amdgcn_sched_barrier(0);
      // F, pv (and fdiv, u_out) are [cell][3]: a wave's 64 cells own 192 consecutive doubles of each.  The rows are
      // transposed through wave shuffles so that every store instruction writes 512 contiguous bytes (whole lines)
      // instead of 64 x 8 bytes at a 24-byte stride -- with the non-temporal hint the strided partial lines reach HBM
      // unmerged (0.58 GB written for 0.48 GB of output).  All 64 lanes take part, cells past the end are masked.
      {
        const int     lane  = tid & 63;
        const int64_t base  = 3 * ((int64_t)o - lane);
        const int     ncell = nc_cur - (tid - lane);  // the wave's cells of this tile
        if (!EULER || f) wave_store_rows3<FNT>(f, base, lane, ncell, out[0], out[1], out[2]);
        wave_store_rows3(a.pv, base, lane, ncell, out[3], out[4], out[5]);
        if (a.fdiv) wave_store_rows3(a.fdiv, base, lane, ncell, acc_fdiv[0], acc_fdiv[1], acc_fdiv[2]);
        if (EULER) {
          const double n0 = out[3] + dt * out[0], n1 = own_hu + dt * out[1], n2 = own_hv + dt * out[2];
          if (!a.o2l) {
            // u_out is what the NEXT step reads.  With the hint it is not in the Infinity Cache then; without it, it is -- if it
            // fits: a 2.8 M-cell part (67 MB of state) steps 8 % faster with plain stores, a 10 M-cell one (240 MB of a 256 MB
            // cache) 3.5 % slower (profiles/r04_uout_store_policy.txt).  FNT = false is the instantiation for states that fit.
            wave_store_rows3<FNT>(a.u_out, base, lane, ncell, n0, n1, n2);
          } else if (active) {  // owned cells are not a prefix of the local numbering: scattered rows
            const int64_t c = a.o2l[o];
            if constexpr (FNT) {
              RDY_ST(&a.u_out[3 * c + 0], n0);
              RDY_ST(&a.u_out[3 * c + 1], n1);
              RDY_ST(&a.u_out[3 * c + 2], n2);
            } else {
              a.u_out[3 * c + 0] = n0;
              a.u_out[3 * c + 1] = n1;
              a.u_out[3 * c + 2] = n2;
            }
          }
          if (send_tile) wave_store_send_rows(a, tile_cur, tid, n0, n1, n2);
        }
      }
      if (last) break;
    }
  }
  courant_extra_edges<HR>(a, dt, u, trk);
  block_courant_reduce<TILE>(a, trk.best, RDY_COLD(a, e_pos), trk.rec);
}

// ---------------------------------------------------------------------------
// cell-centric kernel: one thread = one owned cell, all of its edges
// ---------------------------------------------------------------------------
template <int S, int SRC>
__global__ __launch_bounds__(BLOCK) void swe_rhs_kernel(const KernelArgs a, const double dt, const double *__restrict__ u,
                                                        double *__restrict__ f) {
  int tile = blockIdx.x;
  if (a.xcd_chunks > 0) tile = (blockIdx.x & 7) * a.xcd_chunks + (blockIdx.x >> 3);
  const int i = tile * BLOCK + threadIdx.x;

  double best      = 0.0;  // largest Courant number seen by this thread (> 0 only)
  int    best_slot = -1;
  int    o         = 0;

  bool    active = i < a.n_work;
  int32_t id[S];
  const int32_t *nbr = RDY_COLD(a, nbr), *btype = RDY_COLD(a, btype);  // one thread = one cell: read once per thread
  const double  *cn_ = RDY_COLD(a, cn), *sn_ = RDY_COLD(a, sn), *bvalues = RDY_COLD(a, bvalues);
  if (active) {
    o              = a.list ? a.list[i] : i;
    bool has_ghost = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      id[s] = nbr[s * a.stride + o];
      has_ghost |= (id[s] >= 0) && (id[s] & NBR_GHOST);
    }
    if (a.phase == RDYHIP_PHASE_INTERIOR && has_ghost) active = false;
    if (a.phase == RDYHIP_PHASE_HALO && !has_ghost) active = false;
  }

  if (active) {
    const int         c    = a.o2l ? a.o2l[o] : o;
    const double      h    = u[3 * (int64_t)c + 0];
    const double      hu   = u[3 * (int64_t)c + 1];
    const double      hv   = u[3 * (int64_t)c + 2];
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
      const double cn   = cn_[s * a.stride + o];
      const double sn   = sn_[s * a.stride + o];
      const double coef = a.coef[s * a.stride + o];
      RoeFlux      fl;
      bool         wet;
      const double cfac = fabs(coef);  // len / area_self: the max over an edge's two cells is len / min(area_l, area_r)
      if (nid >= 0) {
        const int         n     = nid & NBR_MASK;
        const RiemannSide other = riemann_side(u[3 * (int64_t)n + 0], u[3 * (int64_t)n + 1], u[3 * (int64_t)n + 2], a.tiny_h, a.h_anuga_sq);
        const bool        self_left = coef < 0.0;
        RiemannSide       L, R;
        L.h = self_left ? self.h : other.h;        R.h = self_left ? other.h : self.h;
        L.u = self_left ? self.u : other.u;        R.u = self_left ? other.u : self.u;
        L.v = self_left ? self.v : other.v;        R.v = self_left ? other.v : self.v;
        L.sqh = self_left ? self.sqh : other.sqh;  R.sqh = self_left ? other.sqh : self.sqh;
        L.c = self_left ? self.c : other.c;        R.c = self_left ? other.c : self.c;
        fl  = roe_flux(L, R, sn, cn);
        wet = !(R.h < a.tiny_h && L.h < a.tiny_h);
      } else {
        const int    k  = -1 - nid;
        BoundaryFlux bf = boundary_flux(btype[k], true, self, bvalues + 3 * (int64_t)k, sn, cn, a.tiny_h, a.h_anuga_sq);
        fl              = bf.flux;
        wet             = bf.wet;
        store_boundary_flux(a, k, fl, dt);
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
    cell_epilogue<SRC>(a, o, dt, h, hu, hv, self.u, self.v, acc0, acc1, acc2, a.dzdx[o], a.dzdy[o], a.mannings[o], a.extsrc[3 * (int64_t)o + 0],
                       a.extsrc[3 * (int64_t)o + 1], a.extsrc[3 * (int64_t)o + 2], f);
  }
  block_courant_reduce<BLOCK>(a, best, RDY_COLD(a, pos), (best_slot < 0 ? 0 : best_slot) * a.stride + o);
}

// merges the per-workgroup buckets into the 16-byte diagnostic the host reads (rdyhip_update_diagnostics)
__global__ __launch_bounds__(1024) void courant_finalize_kernel(int nblk, const double *__restrict__ blk_max, const int32_t *__restrict__ blk_pos,
                                                               DeviceCourant *diag) {
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
  const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    diag->max_courant = bm;
    diag->pos         = bm > 0.0 ? bp : -1;
  }
}

// ResetOperatorDiagnostics (src/operator.c:772-784) on its own: clears every bucket
__global__ void courant_reset_kernel(int nblk, double *__restrict__ blk_max, int32_t *__restrict__ blk_pos) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += gridDim.x * blockDim.x) {
    blk_max[i] = 0.0;
    blk_pos[i] = -1;
  }
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
  const int         k = klist[i];
  const int         c = bleft[k];
  const RiemannSide L = riemann_side(u[3 * (int64_t)c + 0], u[3 * (int64_t)c + 1], u[3 * (int64_t)c + 2], tiny_h, h_anuga_sq);
  BoundaryFlux      bf = boundary_flux(btype[k], false, L, bvalues + 3 * (int64_t)k, bsn[k], bcn[k], tiny_h, h_anuga_sq);
  bflux[3 * (int64_t)k + 0] = bf.flux.f0;
  bflux[3 * (int64_t)k + 1] = bf.flux.f1;
  bflux[3 * (int64_t)k + 2] = bf.flux.f2;
  baccum[3 * (int64_t)k + 0] += dt * bf.flux.f0;
  baccum[3 * (int64_t)k + 1] += dt * bf.flux.f1;
  baccum[3 * (int64_t)k + 2] += dt * bf.flux.f2;
}

__global__ void pack_cells_kernel(int n, const double *__restrict__ u, const int32_t *__restrict__ ids, double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * (int64_t)n) return;
  const int cell = (int)(i / 3), comp = (int)(i - 3 * (int64_t)cell);
  buf[i]         = u[3 * (int64_t)ids[cell] + comp];
}
__global__ void unpack_cells_kernel(int n, double *__restrict__ u, const int32_t *__restrict__ ids, const double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * (int64_t)n) return;
  const int cell = (int)(i / 3), comp = (int)(i - 3 * (int64_t)cell);
  u[3 * (int64_t)ids[cell] + comp] = buf[i];
}
__global__ void pack_rows_kernel(int n, int ncomp, const double *__restrict__ src, const int32_t *__restrict__ ids, double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * ncomp) return;
  const int row = (int)(i / ncomp), comp = (int)(i - (int64_t)row * ncomp);
  buf[i]        = src[(int64_t)ids[row] * ncomp + comp];
}
__global__ void unpack_rows_kernel(int n, int ncomp, double *__restrict__ dst, const int32_t *__restrict__ ids, const double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * ncomp) return;
  const int row = (int)(i / ncomp), comp = (int)(i - (int64_t)row * ncomp);
  dst[(int64_t)ids[row] * ncomp + comp] = buf[i];
}
__global__ void axpy_owned_kernel(int n_owned, const int32_t *__restrict__ o2l, double dt, const double *__restrict__ f, double *__restrict__ u) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * (int64_t)n_owned) return;
  if (o2l) {
    const int o = (int)(i / 3), comp = (int)(i - 3 * (int64_t)o);
    u[3 * (int64_t)o2l[o] + comp] += dt * f[i];
  } else {
    u[i] += dt * f[i];
  }
}
// u_local[owned cell o] = u_global[o]: the local part of DMGlobalToLocal (src/rdysetup.c:1133-1134)
__global__ void copy_owned_rows_kernel(int n_owned, const int32_t *__restrict__ o2l, const double *__restrict__ u_global, double *__restrict__ u_local) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * (int64_t)n_owned) return;
  const int o = (int)(i / 3), comp = (int)(i - 3 * (int64_t)o);
  u_local[3 * (int64_t)o2l[o] + comp] = u_global[i];
}
// u_out[owned cell o] = u_in[o] + dt * f[o]  (fallback of rdyhip_euler_step for the kernels without the fused update)
__global__ void euler_out_kernel(int n_owned, const int32_t *__restrict__ o2l, double dt, const double *__restrict__ f, const double *__restrict__ u_in,
                                 double *__restrict__ u_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * (int64_t)n_owned) return;
  int64_t j = i;
  if (o2l) {
    const int o = (int)(i / 3), comp = (int)(i - 3 * (int64_t)o);
    j           = 3 * (int64_t)o2l[o] + comp;
  }
  u_out[j] = u_in[j] + dt * f[i];
}
__global__ void scatter_component_kernel(int n, const int32_t *__restrict__ ids, const double *__restrict__ vals, double *__restrict__ dst, int ncomp,
                                         int comp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o                    = ids ? ids[i] : i;
  dst[(int64_t)o * ncomp + comp] = vals[i];
}

}  // namespace rdyhip
