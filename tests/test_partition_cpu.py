"""rdycore_amd.partition: recursive coordinate bisection and the local meshes it cuts (SURVEY.md 8.e "RCB or METIS-style
for C5-like meshes"; the 1-cell overlap of src/rdydm.c:145-157).  CPU: the oracle on the parts = the oracle on the whole."""
import os

import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd import partition as P
from helpers import oracle_from_case, rel_linf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nparts", [1, 2, 3, 5, 8])
def test_rcb_parts_are_balanced_compact_and_deterministic(nparts):
    rng = np.random.default_rng(3)
    pts = rng.random((10007, 2)) * [3.0, 1.0]
    parts = P.rcb_partition(pts, nparts)
    counts = np.bincount(parts, minlength=nparts)
    assert counts.sum() == pts.shape[0] and counts.max() - counts.min() <= max(1, nparts // 2)
    for r in range(nparts):
        assert np.array_equal(P.rcb_owned_mask(pts, nparts, r), parts == r)      # the lean single-branch form agrees
    assert np.array_equal(parts, P.rcb_partition(pts.copy(), nparts))
    if nparts > 1:
        # compact: the parts' bounding boxes cover little more than the domain (no interleaving)
        area = sum(np.prod(pts[parts == r].max(0) - pts[parts == r].min(0)) for r in range(nparts))
        assert area < 1.1 * 3.0


def test_rcb_handles_massive_ties_of_structured_grids():
    qi, qj = np.meshgrid(np.arange(64), np.arange(48), indexing="xy")
    pts = np.stack([qi.ravel() + 0.5, qj.ravel() + 0.5], axis=1)
    parts = P.rcb_partition(pts, 6)
    assert np.array_equal(np.bincount(parts), np.full(6, 64 * 48 // 6))


@pytest.mark.parametrize("world", [3, 5, 6])
def test_rcb_cut_inside_a_grid_column_is_contiguous(world):
    """a cut that lands inside a column of a structured grid splits that column once (ties broken along the other axis):
    the number of cut edges stays at the perimeter of compact parts, not a salt-and-pepper column"""
    nx, ny = 100, 64
    qi, qj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    pts = np.stack([qi.ravel() + 0.5, qj.ravel() + 0.5], axis=1)
    parts = P.rcb_partition(pts, world).reshape(ny, nx)
    cut = int((parts[:, 1:] != parts[:, :-1]).sum() + (parts[1:, :] != parts[:-1, :]).sum())
    # every bisection adds at most one straight cut across the shorter extent of what it splits, plus one step
    assert cut <= (world - 1) * (min(nx, ny) + 2)
    for r in range(world):                     # each part is one 4-connected piece whose columns are contiguous runs
        m = parts == r
        cols = np.nonzero(m.any(axis=0))[0]
        assert cols.max() - cols.min() + 1 == cols.size
        for c in cols:
            rows = np.nonzero(m[:, c])[0]
            assert rows.max() - rows.min() + 1 == rows.size


def _global_and_parts(make_mesh, make_case, world):
    gm = make_mesh(0, 1)
    gc = make_case(gm)
    fg = oracle_from_case(gc).apply(gc.dt, gc.u_local)
    seen = np.zeros(gm.num_cells, dtype=int)
    g2row = {int(g): i for i, g in enumerate(gm.cell_global_ids)}
    worst = 0.0
    for r in range(world):
        m = make_mesh(r, world)
        c = make_case(m)
        # ghost cells carry their owners' values (what the halo update delivers): take them from the global state
        rows = np.array([g2row[int(g)] for g in m.cell_global_ids])
        c.u_local[:] = gc.u_local[rows]
        f = oracle_from_case(c).apply(c.dt, c.u_local)
        own_rows = rows[m.cell_owned_to_local]
        seen[own_rows] += 1
        worst = max(worst, rel_linf(f, fg[own_rows]))
        # the ghost layer is exactly the edge-adjacent cells of other ranks
        assert m.num_cells > m.num_owned_cells
    assert np.all(seen == 1), "every cell is owned by exactly one rank"
    return worst


@pytest.mark.parametrize("world", [3, 4])
def test_dam_break_quads_rcb_parts_reproduce_the_single_rank_rhs(world):
    err = _global_and_parts(lambda r, w: CS.dam_break_quads_mesh(160, 80, r, w), lambda m: _perturbed(CS.dam_break_quads_case(m)), world)
    assert err <= 1e-13


def _perturbed(case):
    xc, yc = case.mesh.cell_centroids[:, 0], case.mesh.cell_centroids[:, 1]
    case.u_local[:, 1] = 0.3 * case.u_local[:, 0] * np.sin(1.7 * xc + 0.9 * yc)
    case.u_local[:, 2] = 0.2 * case.u_local[:, 0] * np.cos(1.1 * xc - 2.3 * yc)
    return case


def test_c5_miniature_rcb_parts_reproduce_the_single_rank_rhs():
    err = _global_and_parts(lambda r, w: CS.c5_mesh(60, 50, r, w), lambda m: CS.c5_case(m, 60.0, 50.0), 3)
    assert err <= 1e-13


def test_c5_case_is_what_baseline_md_describes():
    m = CS.c5_mesh(200, 200)
    c = CS.c5_case(m, 200.0, 200.0)
    dry = float((c.u_local[:, 0] == 0.0).mean())
    assert 0.30 <= dry <= 0.55                                     # ">= 30 % dry"
    names = {b.name: b.num_edges for b in m.boundaries}
    assert names["outlet"] > 0 and names["walls"] > names["outlet"]
    assert c.condition_types[m.boundary_by_name("outlet")] == M.CONDITION_CRITICAL_OUTFLOW
    assert np.all(c.ext_src[:, 0] == 1e-5) and c.config.well_balancing == 2


def test_reference_dam_break_mesh_has_the_published_cell_count():
    """docs/user/example-cases/dam-break/index.md:10-11: 11,534,336 cells on the 5120 x 2560 grid"""
    keep = CS.dam_break_keep(5120, 2560)
    qi, qj = np.meshgrid(np.arange(5120, dtype=np.int32), np.arange(2560, dtype=np.int32), indexing="xy")
    assert int(np.count_nonzero(keep(qi.ravel(), qj.ravel()))) == 11_534_336
    m = CS.dam_break_quads_mesh(320, 160)
    assert m.num_cells == 320 * 160 * 88 // 100 and len(m.boundaries) == 1
    # every boundary edge is reflecting wall: the outer perimeter minus what the dam covers (2 x 64), plus the dam's faces
    # (lower block 64 wide x 64 tall: two sides + top; upper block 64 x 32: two sides + bottom)
    assert m.boundaries[0].num_edges == 2 * (320 + 160) - 2 * 64 + (64 + 64 + 64) + (32 + 32 + 64)


def test_houston_mesh_cut_by_rcb_reproduces_the_single_rank_rhs():
    """the reference's real-DEM Houston1km mesh (tests/golden/houston) cut into 3 RCB parts"""
    data = os.path.join(ROOT, "tests", "golden", "houston")
    gc = CS.houston_case(data)
    gm = gc.mesh
    xyz, conn, side_sets = M.read_exodus(os.path.join(data, "Houston1km_with_z.exo"))
    fg = oracle_from_case(gc).apply(gc.dt, gc.u_local)
    parts = P.rcb_partition(gm.cell_centroids, 3)
    cls = M.boundaries_from_side_sets(side_sets, conn, {1: "bottom_wall"})
    covered = 0
    for r in range(3):
        # side sets name global cells: classify on the global mesh, carry the classes over by edge vertices
        m = P.partition_mesh(xyz, conn, 3, r, parts=parts)
        gl = m.cell_global_ids
        lm_boundaries = _carry_boundaries(gm, m, xyz)
        m.boundaries = lm_boundaries
        ctypes = [M.CONDITION_DIRICHLET if b.name == "bottom_wall" else M.CONDITION_REFLECTING for b in m.boundaries]
        no = m.num_owned_cells
        c = CS.Case("houston_part", m, CS.RDyFlowConfig(), ctypes, gc.u_local[gl].copy(), np.full(no, 0.015), np.zeros((no, 3)),
                    {i: np.zeros((b.num_edges, 3)) for i, b in enumerate(m.boundaries) if b.name == "bottom_wall"}, gc.dt)
        f = oracle_from_case(c).apply(c.dt, c.u_local)
        own = gl[m.cell_owned_to_local]
        covered += own.size
        assert rel_linf(f, fg[own]) <= 1e-13
    assert covered == gm.num_cells


def _carry_boundaries(gm, lm, gxyz):
    """boundaries of a local mesh from the global mesh's: boundary edges are matched by their end-point coordinates"""
    key = lambda p: (round(float(p[0]), 6), round(float(p[1]), 6))
    table = {}
    for bi, b in enumerate(gm.boundaries):
        for e in b.edge_ids:
            a, c = gm.xyz[gm.edge_vertex_ids[e, 0]], gm.xyz[gm.edge_vertex_ids[e, 1]]
            table[frozenset((key(a), key(c)))] = bi
    lists = {bi: [] for bi in range(len(gm.boundaries))}
    for e in lm.edge_boundary_ids:
        a, c = lm.xyz[lm.edge_vertex_ids[e, 0]], lm.xyz[lm.edge_vertex_ids[e, 1]]
        bi = table.get(frozenset((key(a), key(c))))
        if bi is not None:
            lists[bi].append(int(e))
    return [M.RDyBoundary(gm.boundaries[bi].id, gm.boundaries[bi].name, np.array(lists[bi], dtype=np.int32)) for bi in range(len(gm.boundaries))]
