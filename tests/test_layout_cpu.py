"""CPU: the host-side layout pass of rdyhip_create (validation of the mesh arrays, slot tables, tiles, halo and
second-order ring lists) through rdyhip_probe_layout -- no device is touched.  The same pass feeds every GPU test;
here its numbers and its argument errors (PETSc-valued codes, as rdyhip_create returns them) are checked without one."""
import os

import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import RDyFlowConfig, RDyHipError, probe_layout


# the capacities every tile is cut to (csrc/swe_kernels.h: TILE_MAX_REC, TILE_MAX_HALO_*; muscl_kernels.h: MUSCL_MAX_RING_*)
CAP = {3: dict(rec=512, halo=104, ring=256), 4: dict(rec=512, halo=112, ring=168)}


def _fits(i, second_order=False):
    c = CAP[i["slots_per_cell"]]
    ok = i["max_tile_edges"] <= c["rec"] and i["max_tile_halo_cells"] <= c["halo"] and i["lds_fixed_layout"] == 1 and i["lds_bytes"] <= 64 * 1024
    return ok and (not second_order or i["max_tile_ring2_cells"] <= c["ring"])


def test_tiles_are_cut_to_the_kernels_capacities_whatever_the_numbering():
    info = {o: probe_layout(RDyFlowConfig(), M.structured_tri_mesh(160, 128, order=o)) for o in ("rowmajor", "tiled", "hilbert")}
    n = 2 * 160 * 128
    for i in info.values():
        assert i["num_owned_cells"] == i["num_cells"] == n and i["slots_per_cell"] == 3 and i["tiled_kernel"] == 1
        assert i["num_halo_tiles"] == 0 and i["num_halo_cells"] == 0 and i["owned_is_prefix"] == 1
        assert i["num_boundary_edges"] == 2 * (160 + 128)
        # every internal edge appears once if both cells share a tile, twice if not; boundary edges once
        assert 1.5 * n - (160 + 128) + i["num_boundary_edges"] <= i["num_edge_records"] <= 3 * n
        assert _fits(i)
    # a block numbering: every tile is one 16 x 8 block of squares = 256 triangles; along a Hilbert curve nearly so; row by
    # row a tile is a strip whose halo cells (the rows above and below) fill the LDS planes long before 256 cells
    assert info["tiled"]["num_tiles"] == n // 256 and info["hilbert"]["num_tiles"] <= 1.03 * n / 256
    assert 2 * n // 256 < info["rowmajor"]["num_tiles"]
    rec = {o: i["num_edge_records"] / n for o, i in info.items()}
    assert rec["rowmajor"] > 1.95 and rec["tiled"] < 1.65 and rec["hilbert"] < 1.68
    assert info["rowmajor"]["num_halo_entries"] > 4 * info["tiled"]["num_halo_entries"]
    # a random numbering: three records per cell (every edge cut), three halo cells per cell -- tiles of ~32 cells, and
    # still the same kernels (rounds 1-4 ran such meshes with 256-cell tiles in > 64 KB of run-time-sized LDS)
    rng = np.random.default_rng(0)
    xyz, conn, _, _ = M.structured_tri_connectivity(64, 48)
    mesh = M.build_mesh(xyz, conn[rng.permutation(conn.shape[0])], boundary_classifier=M.single_boundary())
    for so in (False, True):
        i = probe_layout(RDyFlowConfig(second_order=so), mesh)
        assert i["num_edge_records"] / mesh.num_cells > 2.9 and i["num_tiles"] > mesh.num_cells // 48 and _fits(i, so)


def test_quad_tiles_hold_two_rounds_of_edge_records():
    """a 16 x 16 block of quads has 544 edges: 32 more than the edge phase keeps in registers (two rounds of 256).  Tiles are
    cut where the 513th record would come: a 16 x 15 block (511) -- which is how partition.partitioned_structured_mesh numbers
    quads -- or 220-odd cells of a Hilbert curve."""
    for order, lo in (("tiled", 236.0), ("hilbert", 215.0)):
        q = CS.dam_break_quads_mesh(640, 320, 0, 1, order=order)
        for so in (False, True):
            i = probe_layout(RDyFlowConfig(second_order=so), q)
            assert i["slots_per_cell"] == 4 and _fits(i, so) and i["num_owned_cells"] / i["num_tiles"] >= lo - (3.0 if so else 0.0), (order, so, i)
    # (a row-major quad mesh: strips again)
    assert _fits(probe_layout(RDyFlowConfig(), M.structured_quad_mesh(200, 150)))
    # the measurement knob that makes all tiles smaller (small parts: more independent tiles per CU)
    import os
    os.environ["RDYHIP_TILE_CELLS"] = "128"
    try:
        i = probe_layout(RDyFlowConfig(), M.structured_tri_mesh(160, 128, order="tiled"))
    finally:
        os.environ.pop("RDYHIP_TILE_CELLS")
    assert i["num_tiles"] == 2 * 160 * 128 // 128 and _fits(i)


def test_quads_ghosts_and_second_order_tables():
    q = probe_layout(RDyFlowConfig(), M.structured_quad_mesh(40, 30))
    assert q["slots_per_cell"] == 4
    # one rank of a strip partition: ghost cells, halo cells and halo tiles
    m = M.strip_partition_tri_mesh(40, 48, 1, 3, order="tiled", tile=8)
    i = probe_layout(RDyFlowConfig(), m)
    assert i["num_cells"] > i["num_owned_cells"] == 2 * 40 * 48 and i["owned_is_prefix"] == 1
    assert i["num_halo_cells"] == 2 * 48 and 0 < i["num_halo_tiles"] < i["num_tiles"]
    interleaved = M.extract_local_mesh(*M.structured_tri_connectivity(24, 10)[:2], M.structured_tri_connectivity(24, 10)[2] < 12,
                                       ghosts="interleaved")
    assert probe_layout(RDyFlowConfig(), interleaved)["owned_is_prefix"] == 0
    # second order: the kernel also stages a second ring; its halo tiles include tiles whose FIRST RING touches a ghost
    s = probe_layout(RDyFlowConfig(second_order=True), m)
    assert s["second_order_fused"] == 1 and s["max_tile_ring2_cells"] > s["max_tile_halo_cells"] and s["lds_bytes"] > i["lds_bytes"]
    assert _fits(s, True) and s["lds_bytes"] == 8 * (6 * 360 + 5 * 520) + 4 * 520 <= 40 * 1024       # MusclSoATri: four workgroups per CU
    assert s["num_halo_tiles"] >= i["num_halo_tiles"]


def _code(fn):
    with pytest.raises(RDyHipError) as e:
        fn()
    return e.value.code


def test_argument_errors_of_create_without_a_device():
    import copy
    base = M.structured_tri_mesh(6, 4, project_2d=True)
    cfg = RDyFlowConfig()
    assert probe_layout(cfg, base)["num_tiles"] == 1

    def broken(**kw):
        m = copy.copy(base)
        for k, v in kw.items():
            setattr(m, k, v)
        return m

    # cells.local_to_owned must be a bijection onto the owned cells (PETSC_ERR_USER)
    l2o = base.cell_local_to_owned.copy()
    l2o[1] = l2o[0]
    assert _code(lambda: probe_layout(cfg, broken(cell_local_to_owned=l2o))) == 83
    # an edge pointing at a cell that does not exist (PETSC_ERR_ARG_OUTOFRANGE)
    ec = base.edge_cell_ids.copy()
    ec[2 * int(base.edge_internal_ids[0]) + 1] = base.num_cells + 5
    assert _code(lambda: probe_layout(cfg, broken(edge_cell_ids=ec))) == 63
    # inconsistent sizes (PETSC_ERR_ARG_SIZ)
    assert _code(lambda: probe_layout(cfg, broken(num_owned_cells=base.num_cells + 1))) == 60
    # a cell with five edges: only triangles and quads exist (src/rdymesh.c:809)
    ec = base.edge_cell_ids.copy()
    e3 = base.edge_internal_ids[:3]
    ec[2 * e3.astype(int)] = 0
    ec[2 * e3.astype(int) + 1] = np.arange(1, 4)
    ec2 = ec.copy()
    for e in base.edge_internal_ids[3:6]:
        ec2[2 * int(e)] = 0
    assert _code(lambda: probe_layout(cfg, broken(edge_cell_ids=ec2))) == 83
    # unknown boundary condition type (src/swe/swe_petsc.c:568)
    assert _code(lambda: probe_layout(cfg, base, [7] * len(base.boundaries))) == 83
    # combinations the reference rejects as well
    assert _code(lambda: probe_layout(RDyFlowConfig(second_order=True, well_balancing=2), base)) == 83     # src/operator.c:388-389
    assert _code(lambda: probe_layout(RDyFlowConfig(well_balancing=1), base)) == 83                        # BS2002: CEED only
    assert _code(lambda: probe_layout(RDyFlowConfig(source_method=5), base)) == 83
    assert _code(lambda: probe_layout(RDyFlowConfig(second_order=True, limiter=9), base)) == 83
