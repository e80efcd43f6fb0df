"""Host mesh layer: the RDyMesh conventions of src/rdymesh.c that cannot be
checked against DMPlex here are checked through analytic invariants."""
import os

import numpy as np
import pytest

from rdycore_amd import mesh as M

HERE = os.path.dirname(os.path.abspath(__file__))


def closure_defect(m):
    L = m.edge_cell_ids[0::2]
    R = m.edge_cell_ids[1::2]
    sx = np.zeros(m.num_cells)
    sy = np.zeros(m.num_cells)
    # 2-D edge length goes with the 2-D normal (cn, sn)
    v = m.xyz[m.edge_vertex_ids[:, 1], :2] - m.xyz[m.edge_vertex_ids[:, 0], :2]
    l2 = np.linalg.norm(v, axis=1)
    np.add.at(sx, L, m.edge_cn * l2)
    np.add.at(sy, L, m.edge_sn * l2)
    ii = R >= 0
    np.add.at(sx, R[ii], -m.edge_cn[ii] * l2[ii])
    np.add.at(sy, R[ii], -m.edge_sn[ii] * l2[ii])
    return max(np.abs(sx).max(), np.abs(sy).max())


def normals_point_left_to_right(m):
    L = m.edge_cell_ids[0::2]
    R = m.edge_cell_ids[1::2]
    ii = R >= 0
    v = m.cell_centroids[R[ii], :2] - m.cell_centroids[L[ii], :2]
    ok_int = (v[:, 0] * m.edge_cn[ii] + v[:, 1] * m.edge_sn[ii] > 0).all()
    b = ~ii
    w = m.edge_centroids[b, :2] - m.cell_centroids[L[b], :2]
    ok_bnd = (w[:, 0] * m.edge_cn[b] + w[:, 1] * m.edge_sn[b] > 0).all()
    return ok_int and ok_bnd


@pytest.mark.parametrize("order", ["rowmajor", "tiled"])
def test_structured_triangles(order):
    nx, ny = 13, 9
    m = M.structured_tri_mesh(nx, ny, 0.5, zfunc=lambda x, y: 0.1 * x - 0.3 * y, order=order, tile=4)
    assert m.num_cells == 2 * nx * ny
    assert m.num_internal_edges == 3 * nx * ny - nx - ny      # SURVEY 8.d counts
    assert m.num_boundary_edges == 2 * (nx + ny)
    assert closure_defect(m) < 1e-12
    assert normals_point_left_to_right(m)
    assert np.allclose(np.hypot(m.edge_cn, m.edge_sn), 1.0)
    # planar bathymetry reproduces exact slopes (src/rdymesh.c:747-784)
    assert np.allclose(m.cell_dz_dx, 0.1) and np.allclose(m.cell_dz_dy, -0.3)
    assert sorted(b.name for b in m.boundaries) == ["bottom", "left", "right", "top"]
    assert sum(b.num_edges for b in m.boundaries) == m.num_boundary_edges


def test_c2_counts_match_survey():
    # nx=1000, ny=500 -> 1,000,000 cells, 1,498,500 interior + 3,000 boundary edges
    nx, ny = 100, 50
    m = M.structured_tri_mesh(nx, ny)
    assert (m.num_cells, m.num_internal_edges, m.num_boundary_edges) == (10000, 3 * nx * ny - nx - ny, 300)


def test_quads_area_weighted_slopes():
    m = M.structured_quad_mesh(6, 4, 2.0, 1.0, zfunc=lambda x, y: 0.25 * x + 0.5 * y)
    assert np.allclose(m.cell_areas, 2.0 * np.sqrt(1 + 0.25 ** 2 + 0.5 ** 2))   # 3-D area of the tilted cell
    assert np.allclose(m.cell_dz_dx, 0.25) and np.allclose(m.cell_dz_dy, 0.5)
    assert closure_defect(m) < 1e-12 and normals_point_left_to_right(m)


def test_reference_meshes_load():
    xyz, conn, region, tags, names = M.read_gmsh41(os.path.join(HERE, "golden", "planar_dam_10x5.msh"))
    m = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_edge_tags(tags, names))
    assert m.num_cells == 44 and (m.cell_nverts == 4).all()          # 10x5 minus the 2x3 wall
    assert abs(m.cell_areas.sum() - 44.0) < 1e-9
    assert {b.name: b.num_edges for b in m.boundaries} == {"boundary": 26, "top_wall": 4, "bottom_wall": 6}
    assert sorted(np.bincount(region)[1:].tolist()) == [20, 24]
    xyz, conn = M.read_exodus_tri(os.path.join(HERE, "golden", "mms_triangles_dx1.exo"))
    assert conn.shape == (100, 3)
    x2, c2 = M.refine_triangles(xyz, conn)
    m2 = M.build_mesh(x2, c2, boundary_classifier=M.single_boundary())
    assert m2.num_cells == 400 and m2.boundaries[0].num_edges == 40


def test_owned_numbering_and_ghost_layer():
    for rank in range(3):
        m = M.strip_partition_tri_mesh(5, 6, rank, 3)
        assert m.num_owned_cells == 60
        ghosts = m.num_cells - m.num_owned_cells
        assert ghosts == (6 if rank in (0, 2) else 12)            # ny triangles per interior side
        # owned first in owned numbering, ghosts after (src/rdymesh.c:159-177)
        assert (m.cell_local_to_owned[m.cell_is_owned == 1] < m.num_owned_cells).all()
        assert (m.cell_local_to_owned[m.cell_is_owned == 0] >= m.num_owned_cells).all()
        assert (m.cell_owned_to_local == np.nonzero(m.cell_is_owned)[0]).all()
        # every ghost touches an owned cell through an interior edge
        L, R = m.edge_cell_ids[0::2], m.edge_cell_ids[1::2]
        ii = R >= 0
        touched = set(R[ii][m.cell_is_owned[L[ii]] == 1]) | set(L[ii][m.cell_is_owned[R[ii]] == 1])
        assert set(np.nonzero(m.cell_is_owned == 0)[0]) <= touched
        # global ids are those of the undivided mesh
        assert m.cell_global_ids.max() < 2 * 15 * 6 and len(set(m.cell_global_ids)) == m.num_cells


def test_interleaved_ghost_numbering():
    xyz, conn, cqi, _ = M.structured_tri_connectivity(12, 5)
    m = M.extract_local_mesh(xyz, conn, (cqi >= 4) & (cqi < 8), ghosts="interleaved")
    assert not (m.cell_owned_to_local == np.arange(m.num_owned_cells)).all()
    assert sorted(m.cell_local_to_owned.tolist()) == list(range(m.num_cells))


def test_hilbert_cell_order_is_a_local_permutation():
    xyz, conn, _, _ = M.structured_tri_connectivity(64, 48)
    cent = xyz[conn].mean(axis=1)
    p = M.hilbert_cell_order(cent)
    assert sorted(p.tolist()) == list(range(conn.shape[0]))
    # consecutive cells along the curve are neighbours or nearly so (row-major order jumps a whole row at each row end)
    step = np.linalg.norm(np.diff(cent[p][:, :2], axis=0), axis=1)
    assert step.mean() < 1.0 and np.percentile(step, 99) < 3.0
    # and the generator's "hilbert" order is that permutation applied to the squares
    m = M.structured_tri_mesh(20, 12, order="hilbert")
    r = M.structured_tri_mesh(20, 12, order="rowmajor")
    assert m.num_cells == r.num_cells and np.isclose(m.cell_areas.sum(), r.cell_areas.sum())
