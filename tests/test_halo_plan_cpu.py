"""rdyhip_halo_plan_* / rdyhip_hilbert_cell_order (include/rdyhip.h, "planning the exchange"): the host-side half of the
multi-rank binding -- what the DM's point SF knows about a rank's ghost cells turned into rdyhip_halo_create's arguments.
No device is touched; the all-to-all between the ranks' plans is played by the test itself."""
import ctypes as C

import numpy as np
import pytest

from rdycore_amd import _lib
from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd import partition as P

pi = lambda a: a.ctypes.data_as(_lib.c_int32_p)
pl = lambda a: a.ctypes.data_as(_lib.c_int64_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _plan_all(meshes, keys_are_local_ids):
    """plans of all ranks with the all-to-all done by hand; returns per rank (peers, send_counts, send_cells, recv_counts, recv_cells)"""
    lib = _lib.load()
    world = len(meshes)
    # owner-side local id of every global cell (the PetscSF remote index of a DMPlex host)
    local_of = [dict(zip(m.cell_global_ids[m.cell_owned_to_local].tolist(), m.cell_owned_to_local.tolist())) for m in meshes]
    plans, counts, keys = [], [], []
    for r, m in enumerate(meshes):
        ghost = _i32(np.nonzero(m.cell_is_owned == 0)[0])
        owner = _i32(m.cell_owner_rank[ghost])
        gk = m.cell_global_ids[ghost]
        k = _i64([local_of[o][g] for o, g in zip(owner.tolist(), gk.tolist())]) if keys_are_local_ids else _i64(gk)
        plan = C.c_void_p()
        _lib.check(lib.rdyhip_halo_plan_create(world, r, ghost.size, pi(ghost), pi(owner), pl(k), C.byref(plan)))
        cp, kp = _lib.c_int32_p(), _lib.c_int64_p()
        _lib.check(lib.rdyhip_halo_plan_requests(plan, C.byref(cp), C.byref(kp)))
        counts.append(np.ctypeslib.as_array(cp, shape=(world,)).copy())
        keys.append(np.ctypeslib.as_array(kp, shape=(max(ghost.size, 1),))[:ghost.size].copy())
        plans.append(plan)
    out = []
    for r, m in enumerate(meshes):
        inc_counts = _i32([counts[q][r] for q in range(world)])
        chunks = []
        for q in range(world):
            off = int(counts[q][:r].sum())
            chunks.append(keys[q][off:off + int(counts[q][r])])
        inc_keys = _i64(np.concatenate(chunks)) if chunks else _i64([])
        owned = _i32(m.cell_is_owned)
        ck = None if keys_are_local_ids else pl(_i64(m.cell_global_ids))
        _lib.check(lib.rdyhip_halo_plan_finish(plans[r], pi(inc_counts), pl(inc_keys), m.num_cells, pi(owned), ck))
        n = C.c_int32(0)
        ptrs = [_lib.c_int32_p() for _ in range(5)]
        _lib.check(lib.rdyhip_halo_plan_get(plans[r], C.byref(n), *[C.byref(q) for q in ptrs]))
        take = lambda q, k: np.ctypeslib.as_array(q, shape=(max(k, 1),))[:k].copy()
        npeers = int(n.value)
        peers, sc, rc = take(ptrs[0], npeers), take(ptrs[1], npeers), take(ptrs[3], npeers)
        out.append((peers, sc, take(ptrs[2], int(sc.sum())), rc, take(ptrs[4], int(rc.sum()))))
    for plan in plans:
        _lib.check(lib.rdyhip_halo_plan_destroy(C.byref(plan)))
    return out


def _check_pattern(meshes, res):
    world = len(meshes)
    for r, (peers, sc, send, rc, recv) in enumerate(res):
        m = meshes[r]
        assert np.all(np.diff(peers) > 0) and r not in peers.tolist()
        assert np.all(m.cell_is_owned[send] == 1) and np.all(m.cell_is_owned[recv] == 0)
        # every ghost is received exactly once
        assert np.array_equal(np.sort(recv), np.nonzero(m.cell_is_owned == 0)[0])
        so = ro = 0
        for q, cs, cr in zip(peers.tolist(), sc.tolist(), rc.tolist()):
            pq = res[q]
            j = pq[0].tolist().index(r)
            # what r sends to q is, cell for cell, what q expects from r (compared through global ids)
            q_ro = int(pq[3][:j].sum())
            q_recv = pq[4][q_ro:q_ro + int(pq[3][j])]
            assert cs == q_recv.size
            assert np.array_equal(m.cell_global_ids[send[so:so + cs]], meshes[q].cell_global_ids[q_recv])
            assert np.all(m.cell_owner_rank[recv[ro:ro + cr]] == q)
            so, ro = so + cs, ro + cr
    assert world == len(res)


@pytest.mark.parametrize("keys_are_local_ids", [False, True])
@pytest.mark.parametrize("world", [2, 5, 8])
def test_rcb_parts_of_the_refined_houston_mesh(world, keys_are_local_ids):
    import os
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "houston")
    meshes = [CS.houston_refined_mesh(data, 1, "hilbert", rank=r, world=world)[0] for r in range(world)]
    res = _plan_all(meshes, keys_are_local_ids)
    _check_pattern(meshes, res)
    assert max(len(p[0]) for p in res) >= min(3, world - 1)        # ranks with three and more neighbours


@pytest.mark.parametrize("world", [5, 8])
def test_ghosts_numbered_by_owner_are_received_in_place(world):
    """the contiguous-ghost plan: a mesh whose ghost cells are numbered peer by peer, ascending key inside a peer
    (mesh.extract_local_mesh with owner ranks; rdyhip_local_cell_order for a C host) gets receive lists that are ONE run of
    consecutive rows -- what lets rdyhip_halo_create receive in place (no unpack launch)"""
    import os
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "houston")
    meshes = [CS.houston_refined_mesh(data, 1, "hilbert", rank=r, world=world)[0] for r in range(world)]
    res = _plan_all(meshes, False)
    _check_pattern(meshes, res)
    for m, (peers, sc, send, rc, recv) in zip(meshes, res):
        assert np.array_equal(recv, np.arange(m.num_owned_cells, m.num_cells))
    # the same numbering from the C routine, starting from ghosts in arbitrary order
    lib = _lib.load()
    m = meshes[world // 2]
    rng = np.random.default_rng(world)
    shuffle = np.concatenate([rng.permutation(m.num_owned_cells), m.num_owned_cells + rng.permutation(m.num_cells - m.num_owned_cells)])
    owned, owner, gid = _i32(m.cell_is_owned[shuffle]), _i32(m.cell_owner_rank[shuffle]), _i64(m.cell_global_ids[shuffle])
    xy = np.ascontiguousarray(m.cell_centroids[shuffle])
    perm = np.empty(m.num_cells, dtype=np.int32)
    _lib.check(lib.rdyhip_local_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(owner), pl(gid), pi(perm)))
    assert np.array_equal(np.sort(perm), np.arange(m.num_cells))
    no = m.num_owned_cells
    assert owned[perm[:no]].all() and not owned[perm[no:]].any()
    assert np.array_equal(gid[perm[no:]], m.cell_global_ids[no:])        # ghosts: by owner, then by key -- the mesh's own order
    hil = np.empty(m.num_cells, dtype=np.int32)
    _lib.check(lib.rdyhip_hilbert_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(hil)))
    assert np.array_equal(perm[:no], hil[:no])                            # owned cells: along the curve, as before
    # argument errors: owners and keys go together, and need the owned flags
    assert lib.rdyhip_local_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(owner), None, pi(perm)) == 83
    assert lib.rdyhip_local_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, None, pi(owner), pl(gid), pi(perm)) == 83
    owner[np.nonzero(owned == 0)[0][0]] = -3
    assert lib.rdyhip_local_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(owner), pl(gid), pi(perm)) == 63


def test_strips_and_an_empty_halo_rank():
    meshes = [M.strip_partition_tri_mesh(6, 5, r, 3) for r in range(3)]
    res = _plan_all(meshes, False)
    _check_pattern(meshes, res)
    assert [len(p[0]) for p in res] == [1, 2, 1]
    # one rank, no ghosts: an empty plan
    lib = _lib.load()
    plan = C.c_void_p()
    _lib.check(lib.rdyhip_halo_plan_create(1, 0, 0, None, None, None, C.byref(plan)))
    _lib.check(lib.rdyhip_halo_plan_finish(plan, pi(_i32([0])), None, 4, pi(_i32([1, 1, 1, 1])), None))
    n = C.c_int32(7)
    ptrs = [_lib.c_int32_p() for _ in range(5)]
    _lib.check(lib.rdyhip_halo_plan_get(plan, C.byref(n), *[C.byref(q) for q in ptrs]))
    assert n.value == 0
    _lib.check(lib.rdyhip_halo_plan_destroy(C.byref(plan)))


def test_argument_errors():
    lib = _lib.load()
    plan = C.c_void_p()

    def create(world, rank, cells, owners, keys):
        return lib.rdyhip_halo_plan_create(world, rank, len(cells), pi(_i32(cells)), pi(_i32(owners)), pl(_i64(keys)), C.byref(plan))

    assert create(2, 2, [], [], []) == 83                                  # bad rank
    assert create(2, 0, [4], [0], [9]) == 83                               # a ghost owned by this rank
    assert create(2, 0, [4], [2], [9]) == 63                               # owner outside the communicator
    assert create(3, 0, [4, 5], [1, 1], [9, 9]) == 83                      # two ghosts naming one cell
    assert b"same cell" in lib.rdyhip_last_error()
    assert create(3, 0, [4, 5], [2, 1], [7, 9]) == 0
    cp, kp = _lib.c_int32_p(), _lib.c_int64_p()
    _lib.check(lib.rdyhip_halo_plan_requests(plan, C.byref(cp), C.byref(kp)))
    assert np.ctypeslib.as_array(cp, shape=(3,)).tolist() == [0, 1, 1] and np.ctypeslib.as_array(kp, shape=(2,)).tolist() == [9, 7]
    n = C.c_int32(0)
    ptrs = [_lib.c_int32_p() for _ in range(5)]
    assert lib.rdyhip_halo_plan_get(plan, C.byref(n), *[C.byref(q) for q in ptrs]) == 83      # finish has not run
    owned = _i32([1, 1, 1, 1, 0, 0])
    fin = lambda counts, keys, ck=None: lib.rdyhip_halo_plan_finish(plan, pi(_i32(counts)), pl(_i64(keys)), 6, pi(owned), ck)
    assert fin([1, 0, 0], [0]) == 83                                       # a request from myself
    assert fin([0, 1, 0], [4]) == 83                                       # rank 1 asks for a cell I do not own (a ghost)
    assert fin([0, 1, 0], [17]) == 83                                      # ... or that does not exist
    assert fin([0, 2, 0], [3, 1]) == 83                                    # requests out of order
    gk = _i64([100, 101, 102, 103, 9, 7])
    assert fin([0, 1, 1], [102, 55], pl(gk)) == 83                         # an unknown global id
    assert fin([0, 2, 1], [100, 102, 103], pl(gk)) == 0
    _lib.check(lib.rdyhip_halo_plan_get(plan, C.byref(n), *[C.byref(q) for q in ptrs]))
    assert n.value == 2
    assert np.ctypeslib.as_array(ptrs[2], shape=(3,)).tolist() == [0, 2, 3]          # send cells, peer order
    assert np.ctypeslib.as_array(ptrs[4], shape=(2,)).tolist() == [5, 4]             # recv cells: rank 1's (key 9), rank 2's (key 7)
    _lib.check(lib.rdyhip_halo_plan_destroy(C.byref(plan)))


def test_hilbert_cell_order_matches_the_python_one_and_puts_owned_cells_first():
    lib = _lib.load()
    rng = np.random.default_rng(5)
    xy = np.ascontiguousarray(rng.random((5000, 3)) * [40.0, 25.0, 1.0])
    perm = np.empty(5000, dtype=np.int32)
    _lib.check(lib.rdyhip_hilbert_cell_order(5000, xy.ctypes.data_as(_lib.c_double_p), 3, None, pi(perm)))
    assert np.array_equal(perm, M.hilbert_cell_order(xy))
    owned = _i32(rng.random(5000) < 0.8)
    _lib.check(lib.rdyhip_hilbert_cell_order(5000, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(perm)))
    no = int(owned.sum())
    assert owned[perm[:no]].all() and not owned[perm[no:]].any() and np.array_equal(np.sort(perm), np.arange(5000))
    # inside each group the cells keep the order of the curve through ALL cells
    full = M.hilbert_cell_order(xy)
    assert np.array_equal(perm[:no], full[owned[full] == 1]) and np.array_equal(perm[no:], full[owned[full] == 0])


def test_a_tile_numbering_from_the_c_routine_is_as_compact_as_the_python_one():
    """the numbering an adapter would apply (RDyHipPermuteLocalCells -> rdyhip_hilbert_cell_order) gives the operator
    tiles as compact as rdycore_amd.mesh.hilbert_cell_order's"""
    from rdycore_amd import operator as OP
    lib = _lib.load()
    xyz, conn, _, _ = M.structured_tri_connectivity(96, 64, 1.0, order="rowmajor")
    cent = np.ascontiguousarray((xyz[conn[:, 0]] + xyz[conn[:, 1]] + xyz[conn[:, 2]]) / 3.0)
    perm = np.empty(conn.shape[0], dtype=np.int32)
    _lib.check(lib.rdyhip_hilbert_cell_order(conn.shape[0], cent.ctypes.data_as(_lib.c_double_p), 3, None, pi(perm)))
    rec = {}
    for name, c in (("rowmajor", conn), ("hilbert_c", conn[perm])):
        m = M.build_mesh(xyz, c, boundary_classifier=M.single_boundary())
        info = OP.probe_layout(OP.RDyFlowConfig(), m, [M.CONDITION_REFLECTING])
        rec[name] = info["num_edge_records"] / m.num_cells
    assert rec["hilbert_c"] < 1.72 < 1.85 < rec["rowmajor"]


@pytest.mark.parametrize("world", [5, 8])
def test_two_pass_numbering_of_a_petscsf_host(world):
    """what adapter/rdyhip_petsc.c:RDyHipPermuteLocalCells does, without PETSc: pass 1 numbers every rank's owned cells along the
    curve (ghosts anywhere); the point SF then names each ghost by its owner's NEW local id; pass 2 moves only the ghosts -- by owner
    rank and that id -- and must leave the owned numbers of pass 1 alone (nobody's remote indices move).  The plan with local-id keys
    then lists every peer's ghosts as consecutive rows in arrival order: the exchange can receive in place."""
    import os
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "houston")
    meshes = [CS.houston_refined_mesh(data, 1, "natural", rank=r, world=world)[0] for r in range(world)]
    lib = _lib.load()

    def order(m, owner=None, key=None, start=None):
        """perm[new] = old for mesh m (optionally on top of an earlier numbering `start`: old ids are positions in start)"""
        idx = np.arange(m.num_cells) if start is None else start
        xy = np.ascontiguousarray(m.cell_centroids[idx])
        owned = _i32(m.cell_is_owned[idx])
        perm = np.empty(m.num_cells, dtype=np.int32)
        if owner is None:
            _lib.check(lib.rdyhip_hilbert_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(perm)))
        else:
            _lib.check(lib.rdyhip_local_cell_order(m.num_cells, xy.ctypes.data_as(_lib.c_double_p), 3, pi(owned), pi(_i32(owner[idx])), pl(_i64(key[idx])), pi(perm)))
        return idx[perm]                       # new local id -> original local id

    pass1 = [order(m) for m in meshes]                                     # new -> original
    new_of = []
    for m, p1 in zip(meshes, pass1):
        inv = np.empty(m.num_cells, dtype=np.int64)
        inv[p1] = np.arange(m.num_cells)
        new_of.append(inv)                                                 # original -> id after pass 1
    # the SF after pass 1: a ghost's key is its owner's pass-1 id of that cell (found through the global id)
    owner_new_id = [dict(zip(m.cell_global_ids[m.cell_owned_to_local].tolist(), new_of[r][m.cell_owned_to_local].tolist())) for r, m in enumerate(meshes)]
    pass2, keys = [], []
    for r, m in enumerate(meshes):
        key = np.full(m.num_cells, -1, dtype=np.int64)
        ghost = np.nonzero(m.cell_is_owned == 0)[0]
        key[ghost] = [owner_new_id[o][g] for o, g in zip(m.cell_owner_rank[ghost].tolist(), m.cell_global_ids[ghost].tolist())]
        keys.append(key)
        p2 = order(m, owner=m.cell_owner_rank, key=key, start=pass1[r])
        no = m.num_owned_cells
        assert np.array_equal(p2[:no], pass1[r][:no])                      # the owned cells keep their pass-1 numbers
        pass2.append(p2)
    # the plan on the final numbering, PetscSF-style keys (the owner's local id)
    plans, counts, kk = [], [], []
    for r, m in enumerate(meshes):
        final_of = np.empty(m.num_cells, dtype=np.int64)
        final_of[pass2[r]] = np.arange(m.num_cells)
        ghost = np.nonzero(m.cell_is_owned == 0)[0]
        plan = C.c_void_p()
        _lib.check(lib.rdyhip_halo_plan_create(world, r, ghost.size, pi(_i32(final_of[ghost])), pi(_i32(m.cell_owner_rank[ghost])), pl(_i64(keys[r][ghost])), C.byref(plan)))
        cp, kp = _lib.c_int32_p(), _lib.c_int64_p()
        _lib.check(lib.rdyhip_halo_plan_requests(plan, C.byref(cp), C.byref(kp)))
        counts.append(np.ctypeslib.as_array(cp, shape=(world,)).copy())
        kk.append(np.ctypeslib.as_array(kp, shape=(max(ghost.size, 1),))[:ghost.size].copy())
        plans.append(plan)
    for r, m in enumerate(meshes):
        inc_counts = _i32([counts[q][r] for q in range(world)])
        inc_keys = _i64(np.concatenate([kk[q][int(counts[q][:r].sum()):int(counts[q][:r].sum()) + int(counts[q][r])] for q in range(world)]))
        owned_final = _i32(m.cell_is_owned[pass2[r]])
        _lib.check(lib.rdyhip_halo_plan_finish(plans[r], pi(inc_counts), pl(inc_keys), m.num_cells, pi(owned_final), None))
        n = C.c_int32(0)
        ptrs = [_lib.c_int32_p() for _ in range(5)]
        _lib.check(lib.rdyhip_halo_plan_get(plans[r], C.byref(n), *[C.byref(q) for q in ptrs]))
        nrecv = int(np.ctypeslib.as_array(ptrs[3], shape=(max(n.value, 1),))[:n.value].sum())
        recv = np.ctypeslib.as_array(ptrs[4], shape=(max(nrecv, 1),))[:nrecv]
        assert np.array_equal(recv, np.arange(m.num_owned_cells, m.num_cells))          # in place
        nsend = int(np.ctypeslib.as_array(ptrs[1], shape=(max(n.value, 1),))[:n.value].sum())
        send = np.ctypeslib.as_array(ptrs[2], shape=(max(nsend, 1),))[:nsend]
        assert owned_final[send].all()
    for plan in plans:
        _lib.check(lib.rdyhip_halo_plan_destroy(C.byref(plan)))
