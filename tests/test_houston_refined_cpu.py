"""The unstructured real-DEM workload at miniature size (CPU): the reference's Houston1km mesh refined as -dm_refine does
(src/rdydm.c:82-188), state and forcing of Houston1km.DirichletBC.yaml.  The 11.2 M-cell form of the same construction is
the `houston_refined` workload of bench.py and the full-size GPU parity case (tests/test_gpu_golden_and_scale.py)."""
import os

import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd import operator as OP
from helpers import oracle_from_case, rel_linf

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "houston")


def test_refinement_keeps_area_outline_and_side_set():
    base = CS.houston_case(DATA).mesh
    c = CS.houston_refined_case(DATA, 2, "hilbert")
    m = c.mesh
    assert m.num_cells == base.num_cells * 16 and m.num_boundary_edges == base.num_boundary_edges * 4
    assert abs(m.cell_areas.sum() - base.cell_areas.sum()) <= 1e-9 * base.cell_areas.sum()
    nb = {b.name: b.num_edges for b in base.boundaries}
    nr = {b.name: b.num_edges for b in m.boundaries}
    assert nr == {k: 4 * v for k, v in nb.items()}                      # each labelled edge split in two, twice
    # the original's vertices of valence 4 and 5 (its ragged outline and re-entrant corners) survive beside the valence-6
    # vertices that regular refinement adds
    val = np.bincount(m.cell_conn[:, :3].ravel())
    assert (val == 4).any() and (val == 5).any() and (val == 6).sum() > 0.9 * val.size
    # wet / dry fronts: a good part of the cells is exactly dry, the rest carries at least a millimetre
    h = c.u_local[:, 0]
    assert 0.2 < (h == 0).mean() < 0.6 and h[h > 0].min() >= 1e-3
    assert c.ext_src[:, 0].min() > 0 and c.boundary_values[m.boundary_by_name("bottom_wall")][0, 0] > 1.0


@pytest.mark.parametrize("order", ["hilbert", "natural"])
def test_tiles_of_the_refined_mesh_are_compact(order):
    """runs of 256 consecutive cells of a Hilbert-ordered (or refinement-ordered) unstructured mesh are compact patches:
    edge records per cell and halo cells per tile as on the structured benchmark mesh (1.61 / 55)"""
    c = CS.houston_refined_case(DATA, 4, order)
    info = OP.probe_layout(c.config, c.mesh, c.condition_types)
    assert info["num_edge_records"] / c.mesh.num_cells < 1.70
    assert info["num_halo_entries"] / info["num_tiles"] < 64
    assert info["max_tile_edges"] <= 512            # two register-resident rounds of edge records suffice
    # ... and FULL: nothing but the kernels' capacities ends a tile.  (Round 5's cutter first also ended one where 16 cells shared no
    # edge with it; in the nested refinement order the next 64-cell subtree often starts that way, every tile stopped at 64 cells
    # and the mesh ran 2.3 x slower -- with the two metrics above unharmed.)
    assert c.mesh.num_owned_cells / info["num_tiles"] > 250


@pytest.mark.parametrize("hr", [False, True])
def test_rcb_parts_reproduce_the_single_rank_rhs(hr):
    world = 3
    gc = CS.houston_refined_case(DATA, 2, "hilbert", hr=hr)
    gm = gc.mesh
    fg = oracle_from_case(gc).apply(gc.dt, gc.u_local)
    assert np.isfinite(fg).all() and np.abs(fg).max() > 0
    cent = {tuple(np.round(c[:2], 3)): i for i, c in enumerate(gm.cell_centroids)}
    seen = np.zeros(gm.num_cells, dtype=int)
    for r in range(world):
        c = CS.houston_refined_case(DATA, 2, "hilbert", hr=hr, rank=r, world=world)
        m = c.mesh
        rows = np.array([cent[tuple(np.round(x[:2], 3))] for x in m.cell_centroids])
        assert np.allclose(c.u_local, gc.u_local[rows], rtol=0, atol=1e-14)     # the state is a function of position only
        f = oracle_from_case(c).apply(c.dt, c.u_local)
        own = rows[m.cell_owned_to_local]
        seen[own] += 1
        assert rel_linf(f, fg[own]) <= 1e-13
        assert m.num_cells > m.num_owned_cells
        assert sum(b.num_edges for b in m.boundaries) <= gm.num_boundary_edges
    assert np.all(seen == 1)
