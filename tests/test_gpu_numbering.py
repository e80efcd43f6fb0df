"""The numbering the drop-in really hands over: cells permuted along a Hilbert curve and NOTHING else
(RDyHipPermuteLocalCells, adapter/rdyhip_petsc.c: "the other points keep their numbers"), edges in DMPlex's own order, which
owes nothing to the cells (src/rdymesh.c:693-710), left / right cells from the support order and the orientation flip of
src/rdymesh.c:607-673 -- not "lower-numbered cell on the left, edges in order of first appearance" as mesh.build_mesh makes
them.  Every mesh here goes through mesh.dmplex_like_numbering (or its two halves); the checks are test_gpu_parity.check_all:
whole RHS, primitive variables, boundary fluxes with the NaN pattern, the Courant value and -- exactly -- its ids.

Ties: the reference keeps the FIRST edge in loop order that reaches the maximal Courant number (src/swe/swe_petsc.c:289-296).
With edges numbered independently of the cells, "first" is decided by the loop position alone; uniform states (a lake at
rest, the two flat pools of a dam break) tie on every edge of a kind, and the ids must still be the oracle's, bit for bit.
RDYHIP_PGRID=8 makes eight workgroups walk all tiles, so that every thread sees many cells that tie.
"""
import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import (LIMITER_MINMOD, LIMITER_VANLEER, SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT, WELL_BALANCING_HR)

from test_gpu_parity import check_all, run_both

pytestmark = pytest.mark.gpu


TILED_ONLY = ("hr", "so_minmod", "so_vanleer")     # hydrostatic reconstruction and second order live in the tiled kernels


def _skip_cell(rdyhip_kernel, variant):
    if rdyhip_kernel == "cell" and variant in TILED_ONLY:
        pytest.skip("tiled kernels only")


def _variant(case, variant):
    if variant == "hr":
        case.config.well_balancing = WELL_BALANCING_HR
    elif variant == "so_minmod":
        case.config.second_order = True
        case.config.limiter = LIMITER_MINMOD
    elif variant == "so_vanleer":
        case.config.second_order = True
        case.config.limiter = LIMITER_VANLEER
    elif variant == "xq":
        case.config.source_method = SOURCE_IMPLICIT_XQ2018
    return case


def _tri(nx, ny, seed, variant, dry=True, **kw):
    K = 2 * np.pi / (0.8 * nx)
    m = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=K), project_2d=(variant == "hr"))
    m = M.dmplex_like_numbering(m, seed=seed, **kw)
    return _variant(CS.friction_slope_case(m, nx, ny, dt=1e-2, dry_disc=dry, K=K), variant)


VARIANTS = ["first", "xq", "hr", "so_minmod", "so_vanleer"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_triangles_dmplex_numbering(variant, rdyhip_kernel):
    _skip_cell(rdyhip_kernel, variant)
    case = _tri(61, 47, 5, variant)
    assert (case.mesh.edge_cell_ids[0::2] > case.mesh.edge_cell_ids[1::2]).any()      # left is not the lower-numbered cell
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc, near_tie_ok=True)
    op.destroy()


@pytest.mark.parametrize("half", ["edges_only", "cells_only"])
@pytest.mark.parametrize("variant", ["first", "so_minmod"])
def test_each_half_of_the_renumbering(half, variant, rdyhip_kernel):
    _skip_cell(rdyhip_kernel, variant)
    K = 2 * np.pi / 40
    m = M.structured_tri_mesh(53, 31, 1.0, zfunc=CS.mms_bathymetry(K=K))
    if half == "edges_only":
        rng = np.random.default_rng(17)
        m = M.renumber_edges(m, rng.permutation(m.num_edges))
    else:
        m = M.renumber_cells(m, M.hilbert_cell_order(m.cell_centroids))
    case = _variant(CS.friction_slope_case(m, 53.0, 31.0, dt=1e-2, K=K), variant)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc, near_tie_ok=True)
    op.destroy()


@pytest.mark.parametrize("variant", ["first", "hr", "so_minmod"])
def test_quads_dmplex_numbering(variant, rdyhip_kernel):
    _skip_cell(rdyhip_kernel, variant)
    K = 2 * np.pi / 30
    m = M.structured_quad_mesh(45, 37, 1.0, 1.5, zfunc=CS.mms_bathymetry(K=K), project_2d=(variant == "hr"))
    m = M.dmplex_like_numbering(m, seed=9)
    case = _variant(CS.friction_slope_case(m, 45.0, 55.5, dt=5e-3, K=K), variant)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc, near_tie_ok=True)
    assert op.layout_info()["slots_per_cell"] == 4
    op.destroy()


def _mixed_mesh(nx, ny):
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    xyz = np.zeros(((nx + 1) * (ny + 1), 3))
    xyz[:, 0] = ii.ravel()
    xyz[:, 1] = jj.ravel()
    xyz[:, 2] = 0.05 * np.sin(xyz[:, 0]) + 0.02 * xyz[:, 1]
    v = lambda i, j: j * (nx + 1) + i
    conn = []
    for j in range(ny):
        for i in range(nx):
            if (i // 3 + j // 2) % 2 == 0:
                conn.append([v(i, j), v(i + 1, j), v(i + 1, j + 1), v(i, j + 1)])
            else:
                conn.append([v(i, j), v(i + 1, j), v(i + 1, j + 1), -1])
                conn.append([v(i, j), v(i + 1, j + 1), v(i, j + 1), -1])
    return M.build_mesh(xyz, np.array(conn, dtype=np.int32), boundary_classifier=M.box_side_boundaries(0, nx, 0, ny))


@pytest.mark.parametrize("variant", ["first", "so_vanleer"])
def test_mixed_tri_quad_dmplex_numbering(variant, rdyhip_kernel):
    _skip_cell(rdyhip_kernel, variant)
    nx, ny = 36, 28
    m = M.dmplex_like_numbering(_mixed_mesh(nx, ny), seed=21)
    case = _variant(CS.friction_slope_case(m, nx, ny, dt=1e-2, K=2 * np.pi / 19, dry_disc=True), variant)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc, near_tie_ok=True)
    op.destroy()


# ---- exact ties -------------------------------------------------------------------------------------------------------------

def _uniform_case(mesh, kind, lx):
    if kind == "lake_at_rest":
        return CS.dam_break_case(mesh, 1e9, perturb=0.0)          # h = 10 everywhere, at rest
    return CS.dam_break_case(mesh, lx, perturb=0.0)               # two flat pools, h = 10 | 5


@pytest.mark.parametrize("kind", ["lake_at_rest", "two_flat_pools"])
@pytest.mark.parametrize("variant", ["first", "hr", "so_minmod"])
@pytest.mark.parametrize("shape", ["tri", "quad"])
def test_courant_ids_on_exact_ties(monkeypatch, kind, variant, shape, rdyhip_kernel):
    _skip_cell(rdyhip_kernel, variant)
    monkeypatch.setenv("RDYHIP_PGRID", "8")        # 8 workgroups walk all tiles: a thread meets the same value in many cells
    if shape == "tri":
        m = M.structured_tri_mesh(120, 90, 1.0, project_2d=(variant == "hr"))
        lx = 120.0
    else:
        m = M.structured_quad_mesh(150, 120, 1.0, 1.0, project_2d=(variant == "hr"))
        lx = 150.0
    m = M.dmplex_like_numbering(m, seed=33)
    case = _variant(_uniform_case(m, kind, lx), variant)
    f, fr, op, orc = run_both(case)
    assert op.layout_info()["num_tiles"] >= 64
    check_all(case, f, fr, op, orc)
    op.destroy()


def test_courant_ids_on_exact_ties_full_grid_1M(rdyhip_kernel):
    """the persistent grid as it is (no knob): 1 M cells = 3 907 tiles on 768 workgroups, a thread sees five or six tiles"""
    if rdyhip_kernel == "cell":
        pytest.skip("one variant is enough at this size")
    m = M.dmplex_like_numbering(M.structured_tri_mesh(1000, 500, 1.0), seed=41)
    case = CS.dam_break_case(m, 1000.0, perturb=0.0)
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc)
    op.destroy()


def test_one_million_cells_dmplex_numbering(rdyhip_kernel):
    """C2's size with all boundary types, friction, bed slope, a dry disc -- on the drop-in's numbering"""
    if rdyhip_kernel == "cell":
        pytest.skip("one variant is enough at this size")
    case = _tri(1000, 500, 7, "first")
    f, fr, op, orc = run_both(case)
    check_all(case, f, fr, op, orc, near_tie_ok=True)
    op.destroy()


# ---- several ranks: a rank's diagnostic BEFORE the reduction is the reference's -----------------------------------------------
# The reference's interior loop runs over ALL local internal edges -- edges between two ghost cells included -- and divides by
# min(area_l, area_r) whoever owns the cells (src/swe/swe_petsc.c:275-296); the MPI reduction then keeps the struct with the
# larger value (src/operator.c:705-715).  On ties the rank's own first edge in ITS loop order is what it reports, and which
# rank's struct survives a tie is left to MPI; timestep.reduce_courant takes the lowest rank.  So: per rank, the device's
# struct must equal the oracle's on that rank's local mesh -- ids exactly, in tied (uniform) states without any tilting --
# and the reduced struct follows.  One process is enough: the ghost values of a known state need no exchange.

@pytest.mark.parametrize("state", ["lake_at_rest", "two_flat_pools"])
@pytest.mark.parametrize("variant", ["first", "so_minmod"])
@pytest.mark.parametrize("kind", ["strips", "rcb_quads"])
@pytest.mark.parametrize("edge_order", ["random", "ghost_edges_first"])
def test_courant_ids_per_rank_without_tilting(monkeypatch, kind, variant, state, edge_order, rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("the cell-centric kernel (A/B reference point) evaluates a cut edge from its owned side only")
    import torch
    from helpers import oracle_from_case
    from rdycore_amd.operator import CourantNumberDiagnostics
    monkeypatch.setenv("RDYHIP_PGRID", "8")
    world = 3
    got, want = [], []
    for rank in range(world):
        if kind == "strips":
            m = M.strip_partition_tri_mesh(40, 48, rank, world, 1.0, order="tiled", tile=8)
            lx = 120.0
        else:
            # (the dam-break domain on a lattice of exactly representable coordinates: the ties of a uniform state are exact
            # only where equal edges have bitwise-equal lengths and areas; on the benchmark's own 10 m / 5120 lattice they agree to
            # rounding, which makes near-ties of them -- DESIGN.md section 2)
            from rdycore_amd import partition as P
            m = P.partitioned_structured_mesh("quad", 192, 96, (0.5, 0.5), rank, world, keep=CS.dam_break_keep(192, 96))
            lx = 96.0
        m = M.dmplex_like_numbering(m, seed=50 + rank)
        assert m.num_cells > m.num_owned_cells and (m.cell_is_owned[: m.num_owned_cells] != 0).all()
        el, er = m.edge_cell_ids[0::2], m.edge_cell_ids[1::2]
        ghost_ghost = (er >= 0) & (m.cell_is_owned[el] == 0) & (m.cell_is_owned[np.maximum(er, 0)] == 0)
        assert ghost_ghost.any()                                                       # edges between two ghost cells exist
        if edge_order == "ghost_edges_first":
            # ... and come first in the loop: the reference then reports one of THEM in a tied state, an edge no owned cell touches
            order = np.concatenate([np.nonzero(ghost_ghost)[0], np.nonzero(~ghost_ghost)[0]])
            new_of_old = np.empty(m.num_edges, dtype=np.int64)
            new_of_old[order] = np.arange(m.num_edges)
            m = M.renumber_edges(m, new_of_old)
        case = _variant(_uniform_case(m, state, lx), variant)
        orc = oracle_from_case(case)
        op = CS.create_operator(case)
        u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
        f = torch.empty((m.num_owned_cells, 3), dtype=torch.float64, device="cuda")
        if case.config.second_order:
            # uniform states: every gradient is exactly zero, which is what the (zero-initialised) gradient field holds for the
            # ghost cells and the ghost-adjacent owned cells
            op.apply_phase(0, True, case.dt, u, f, reset_diagnostics=True, gradients_ready=True)
            orc.compute_gradients(case.u_local)
            orc.set_gradients_ready(True)
        else:
            op.rhs_function(case.dt, u, f)
        torch.cuda.synchronize()
        orc.apply(case.dt, case.u_local)
        op.update_diagnostics()
        d = op.get_diagnostics()
        cmax, ce, cc = orc.diagnostics()
        assert abs(d.max_courant_num - cmax) <= 1e-12 * max(1.0, cmax)
        assert (d.global_edge_id, d.global_cell_id) == (ce, cc), (rank, (d.global_edge_id, d.global_cell_id), (ce, cc))
        if edge_order == "ghost_edges_first" and kind == "rcb_quads" and state == "lake_at_rest" and variant == "first":
            # (second order: only the rank that owns an edge reports it, src/swe/swe_petsc.c:172-190 -- never one between two ghosts)
            e = int(np.nonzero(m.edge_global_ids == ce)[0][0])          # all edges of the uniform quad mesh tie: the first one wins
            assert m.cell_is_owned[m.edge_cell_ids[2 * e]] == 0 and m.cell_is_owned[m.edge_cell_ids[2 * e + 1]] == 0
        got.append(d)
        want.append(CourantNumberDiagnostics(cmax, ce, cc))
        op.destroy()

    def reduce(rows):      # FindCourantNumberDiagnostics over the ranks in order: strictly larger replaces (timestep.reduce_courant)
        best = CourantNumberDiagnostics(0.0, -1, -1)
        for r in rows:
            if r.max_courant_num > best.max_courant_num:
                best = r
        return best
    a, b = reduce(got), reduce(want)
    assert (a.global_edge_id, a.global_cell_id) == (b.global_edge_id, b.global_cell_id)
