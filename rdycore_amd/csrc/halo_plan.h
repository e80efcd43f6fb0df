// Host-side planning of a rank's ghost exchange and of its local cell numbering, behind the C ABI (include/rdyhip.h,
// "planning the exchange").  No device call anywhere in this file: everything here runs -- and is tested -- on a machine
// without a GPU.  Included by rdyhip_api.hip only.
//
// What it replaces on the RDycore side: the DM's point SF does this work inside DMGlobalToLocalBegin/End
// (src/rdysetup.c:1133-1134; the 1-cell overlap comes from DMPlexDistributeOverlap, src/rdydm.c:145-157).  A host that
// wants the ghost update on RCCL needs the same information as flat lists: for every neighbour rank, which of MY owned
// cells it reads (send list) and which of my ghost cells it owns (receive list), both sides in one agreed order.  A ghost
// cell's owner and the owner's name for it are known locally (PetscSFGetGraph: iremote[i].rank / .index; or a global cell
// id); the send side is the transpose of that relation and needs ONE all-to-all of the requests, which the caller performs
// with whatever it has (MPI_Alltoall + MPI_Alltoallv in RDycore, torch.distributed.all_to_all_single in bench.py):
//
//   rdyhip_halo_plan_create    ghosts grouped by owner rank, ascending key inside a group  -> request counts / keys
//   (caller)                   all-to-all of the counts, all-to-all-v of the keys
//   rdyhip_halo_plan_finish    incoming requests resolved to local owned cells             -> send lists
//   rdyhip_halo_plan_get       npeers, peers, send_counts, send_cell_ids, recv_counts, recv_cell_ids = rdyhip_halo_create's arguments
#pragma once

struct RDyHipHaloPlan_s {
  int32_t world = 0, rank = 0;
  bool    finished = false;
  std::vector<int32_t> req_counts;       // [world] cells requested from each rank
  std::vector<int64_t> req_keys;         // [nghost] grouped by owner, ascending key
  std::vector<int32_t> req_cells;        // [nghost] my ghost cell (local id) for each request, same order
  std::vector<int32_t> peers, send_counts, recv_counts, send_cells, recv_cells;
};

namespace {

// Hilbert index of a point of the 2^16 x 2^16 lattice (the classical xy -> d conversion, 16 levels)
inline uint64_t hilbert_d(uint32_t x, uint32_t y) {
  uint64_t d = 0;
  for (uint32_t s = 32768; s > 0; s >>= 1) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += (uint64_t)s * s * ((3u * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) {
        x = 65535u - x;
        y = 65535u - y;
      }
      const uint32_t t = x;
      x = y;
      y = t;
    }
  }
  return d;
}

}  // namespace

extern "C" {

int rdyhip_halo_plan_create(int32_t world, int32_t rank, int32_t num_ghosts, const int32_t *ghost_cell_ids, const int32_t *ghost_owner_ranks,
                            const int64_t *ghost_keys, RDyHipHaloPlan *plan) {
  if (!plan) return fail(RDYHIP_ERR_USER, "null argument to rdyhip_halo_plan_create");
  *plan = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(RDYHIP_ERR_USER, "bad rank %d of %d", rank, world);
  if (num_ghosts < 0 || (num_ghosts > 0 && (!ghost_cell_ids || !ghost_owner_ranks || !ghost_keys))) return fail(RDYHIP_ERR_USER, "bad ghost list");
  for (int32_t i = 0; i < num_ghosts; ++i) {
    if (ghost_owner_ranks[i] < 0 || ghost_owner_ranks[i] >= world)
      return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "ghost cell %d: owner rank %d outside [0, %d)", ghost_cell_ids[i], ghost_owner_ranks[i], world);
    if (ghost_owner_ranks[i] == rank) return fail(RDYHIP_ERR_USER, "ghost cell %d is owned by this rank (%d)", ghost_cell_ids[i], rank);
    if (ghost_cell_ids[i] < 0) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "negative ghost cell id");
  }
  RDyHipHaloPlan p = new (std::nothrow) RDyHipHaloPlan_s;
  if (!p) return fail(RDYHIP_ERR_MEM, "out of host memory");
  p->world = world;
  p->rank  = rank;
  std::vector<int32_t> order((size_t)num_ghosts);
  for (int32_t i = 0; i < num_ghosts; ++i) order[i] = i;
  // the order both sides agree on: by owner, then by the owner's key for the cell (no duplicates allowed)
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    if (ghost_owner_ranks[a] != ghost_owner_ranks[b]) return ghost_owner_ranks[a] < ghost_owner_ranks[b];
    return ghost_keys[a] < ghost_keys[b];
  });
  p->req_counts.assign((size_t)world, 0);
  p->req_keys.resize((size_t)num_ghosts);
  p->req_cells.resize((size_t)num_ghosts);
  for (int32_t j = 0; j < num_ghosts; ++j) {
    const int32_t i = order[j];
    if (j > 0 && ghost_owner_ranks[order[j - 1]] == ghost_owner_ranks[i] && ghost_keys[order[j - 1]] == ghost_keys[i]) {
      const long long k = (long long)ghost_keys[i];
      delete p;
      return fail(RDYHIP_ERR_USER, "two ghost cells name the same cell (key %lld) of rank %d", k, ghost_owner_ranks[i]);
    }
    p->req_counts[ghost_owner_ranks[i]]++;
    p->req_keys[j]  = ghost_keys[i];
    p->req_cells[j] = ghost_cell_ids[i];
  }
  *plan = p;
  return 0;
}

int rdyhip_halo_plan_requests(RDyHipHaloPlan plan, const int32_t **request_counts, const int64_t **request_keys) {
  if (!plan || !request_counts || !request_keys) return fail(RDYHIP_ERR_USER, "null argument");
  *request_counts = plan->req_counts.data();
  *request_keys   = plan->req_keys.data();
  return 0;
}

int rdyhip_halo_plan_finish(RDyHipHaloPlan plan, const int32_t *incoming_counts, const int64_t *incoming_keys, int32_t num_cells,
                            const int32_t *cell_is_owned, const int64_t *cell_keys) {
  if (!plan || !incoming_counts) return fail(RDYHIP_ERR_USER, "null argument");
  plan->finished = false;  // a second call that fails must not leave the first call's lists readable
  if (num_cells < 0 || (num_cells > 0 && !cell_is_owned)) return fail(RDYHIP_ERR_USER, "bad cell list");
  const int32_t world = plan->world;
  int64_t       total = 0;
  for (int32_t r = 0; r < world; ++r) {
    if (incoming_counts[r] < 0) return fail(RDYHIP_ERR_ARG_SIZ, "negative request count from rank %d", r);
    if (r == plan->rank && incoming_counts[r] != 0) return fail(RDYHIP_ERR_USER, "rank %d requests cells of itself", r);
    total += incoming_counts[r];
  }
  if (total > 0 && !incoming_keys) return fail(RDYHIP_ERR_USER, "null request list");
  // key -> local cell: the local cell id itself (PetscSF remote indices), or a lookup in the owned cells' keys (global ids)
  std::vector<std::pair<int64_t, int32_t>> table;
  if (cell_keys) {
    for (int32_t c = 0; c < num_cells; ++c)
      if (cell_is_owned[c]) table.emplace_back(cell_keys[c], c);
    std::sort(table.begin(), table.end());
    for (size_t i = 1; i < table.size(); ++i)
      if (table[i].first == table[i - 1].first) return fail(RDYHIP_ERR_USER, "two owned cells carry the same key %lld", (long long)table[i].first);
  }
  for (int32_t j = 0; j < (int32_t)plan->req_cells.size(); ++j) {
    const int32_t c = plan->req_cells[j];
    if (c >= num_cells) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "ghost cell id %d out of range (%d local cells)", c, num_cells);
    if (cell_is_owned[c]) return fail(RDYHIP_ERR_USER, "cell %d is listed as a ghost but is owned", c);
  }
  plan->peers.clear(); plan->send_counts.clear(); plan->recv_counts.clear(); plan->send_cells.clear(); plan->recv_cells.clear();
  plan->send_cells.reserve((size_t)total);
  int64_t in_off = 0, req_off = 0;
  for (int32_t r = 0; r < world; ++r) {
    const int32_t ns = incoming_counts[r], nr = plan->req_counts[r];
    if (ns > 0 || nr > 0) {
      plan->peers.push_back(r);
      plan->send_counts.push_back(ns);
      plan->recv_counts.push_back(nr);
    }
    for (int32_t i = 0; i < ns; ++i) {
      const int64_t key = incoming_keys[in_off + i];
      int32_t       c   = -1;
      if (cell_keys) {
        auto it = std::lower_bound(table.begin(), table.end(), std::make_pair(key, (int32_t)INT32_MIN));
        if (it != table.end() && it->first == key) c = it->second;
      } else if (key >= 0 && key < num_cells && cell_is_owned[key]) {
        c = (int32_t)key;
      }
      if (c < 0) return fail(RDYHIP_ERR_USER, "rank %d asks for cell %lld, which this rank (%d) does not own", r, (long long)key, plan->rank);
      if (i > 0 && incoming_keys[in_off + i - 1] >= key) return fail(RDYHIP_ERR_USER, "the requests of rank %d are not in ascending key order", r);
      plan->send_cells.push_back(c);
    }
    for (int32_t i = 0; i < nr; ++i) plan->recv_cells.push_back(plan->req_cells[(size_t)req_off + i]);
    in_off += ns;
    req_off += nr;
  }
  plan->finished = true;
  return 0;
}

int rdyhip_halo_plan_get(RDyHipHaloPlan plan, int32_t *npeers, const int32_t **peers, const int32_t **send_counts, const int32_t **send_cell_ids,
                         const int32_t **recv_counts, const int32_t **recv_cell_ids) {
  if (!plan || !npeers || !peers || !send_counts || !send_cell_ids || !recv_counts || !recv_cell_ids) return fail(RDYHIP_ERR_USER, "null argument");
  if (!plan->finished) return fail(RDYHIP_ERR_USER, "rdyhip_halo_plan_finish has not run");
  *npeers        = (int32_t)plan->peers.size();
  *peers         = plan->peers.data();
  *send_counts   = plan->send_counts.data();
  *send_cell_ids = plan->send_cells.data();
  *recv_counts   = plan->recv_counts.data();
  *recv_cell_ids = plan->recv_cells.data();
  return 0;
}

int rdyhip_halo_plan_destroy(RDyHipHaloPlan *plan) {
  if (!plan) return fail(RDYHIP_ERR_USER, "null argument");
  delete *plan;
  *plan = nullptr;
  return 0;
}

// perm[new] = old: owned cells first along a Hilbert curve through their centroids; the ghost cells after them -- along the
// curve as well (cell_owner_rank == NULL), or grouped by owner rank and ascending key inside a group (the order in which
// rdyhip_halo_plan_* lists a peer's cells on both sides: every peer's ghosts are then ONE run of consecutive rows in the order
// they arrive, and rdyhip_halo_create receives straight into the local vector, with no unpack launch)
static int local_cell_order(int32_t num_cells, const double *xy, int32_t stride, const int32_t *cell_is_owned, const int32_t *cell_owner_rank,
                            const int64_t *cell_keys, int32_t *perm) {
  if (num_cells < 0 || stride < 2) return fail(RDYHIP_ERR_ARG_SIZ, "bad size");
  if (num_cells == 0) return 0;
  if (!xy || !perm) return fail(RDYHIP_ERR_USER, "null argument");
  if ((cell_owner_rank != nullptr) != (cell_keys != nullptr)) return fail(RDYHIP_ERR_USER, "cell_owner_rank and cell_keys go together");
  if (cell_owner_rank && !cell_is_owned) return fail(RDYHIP_ERR_USER, "ghost ordering needs cell_is_owned");
  double lo[2] = {xy[0], xy[1]}, hi[2] = {xy[0], xy[1]};
  for (int32_t c = 1; c < num_cells; ++c)
    for (int k = 0; k < 2; ++k) {
      lo[k] = std::min(lo[k], xy[(size_t)c * stride + k]);
      hi[k] = std::max(hi[k], xy[(size_t)c * stride + k]);
    }
  const double ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
  struct Key {
    uint64_t a, b;  // owned: (0, Hilbert index); ghost: (1 + owner rank, key) or (1, Hilbert index)
    int32_t  c;
    bool operator<(const Key &o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); }
  };
  std::vector<Key> key((size_t)num_cells);
  for (int32_t c = 0; c < num_cells; ++c) {
    const bool ghost = cell_is_owned && !cell_is_owned[c];
    if (ghost && cell_owner_rank) {
      if (cell_owner_rank[c] < 0) return fail(RDYHIP_ERR_ARG_OUTOFRANGE, "ghost cell %d: negative owner rank", c);
      key[c] = Key{1ull + (uint64_t)cell_owner_rank[c], (uint64_t)cell_keys[c] ^ (1ull << 63), c};  // signed order of the keys
      continue;
    }
    double fx = (xy[(size_t)c * stride] - lo[0]) / ext * 65535.0, fy = (xy[(size_t)c * stride + 1] - lo[1]) / ext * 65535.0;
    if (!(fx == fx)) fx = 0.0;  // a NaN coordinate must not reach the integer conversion
    if (!(fy == fy)) fy = 0.0;
    const uint32_t x = (uint32_t)std::min(65535.0, std::max(0.0, fx)), y = (uint32_t)std::min(65535.0, std::max(0.0, fy));
    // owned cells first (a contiguous prefix: the owned rows of a local vector are then one block), ghosts after them
    key[c] = Key{ghost ? 1ull : 0ull, hilbert_d(x, y), c};
  }
  std::sort(key.begin(), key.end());  // ties (coincident lattice points) fall back to the old cell id: deterministic
  for (int32_t i = 0; i < num_cells; ++i) perm[i] = key[i].c;
  return 0;
}

int rdyhip_hilbert_cell_order(int32_t num_cells, const double *xy, int32_t stride, const int32_t *cell_is_owned, int32_t *perm) {
  return local_cell_order(num_cells, xy, stride, cell_is_owned, nullptr, nullptr, perm);
}

int rdyhip_local_cell_order(int32_t num_cells, const double *xy, int32_t stride, const int32_t *cell_is_owned, const int32_t *cell_owner_rank,
                            const int64_t *cell_keys, int32_t *perm) {
  return local_cell_order(num_cells, xy, stride, cell_is_owned, cell_owner_rank, cell_keys, perm);
}

}  // extern "C"
