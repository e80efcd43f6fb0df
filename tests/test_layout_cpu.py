"""CPU: the host-side layout pass of rdyhip_create (validation of the mesh arrays, slot tables, tiles, halo and
second-order ring lists) through rdyhip_probe_layout -- no device is touched.  The same pass feeds every GPU test;
here its numbers and its argument errors (PETSc-valued codes, as rdyhip_create returns them) are checked without one."""
import os

import numpy as np
import pytest

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import RDyFlowConfig, RDyHipError, probe_layout


def test_tile_numbers_follow_the_cell_numbering():
    info = {o: probe_layout(RDyFlowConfig(), M.structured_tri_mesh(160, 128, order=o)) for o in ("rowmajor", "tiled", "hilbert")}
    n = 2 * 160 * 128
    for i in info.values():
        assert i["num_owned_cells"] == i["num_cells"] == n and i["slots_per_cell"] == 3 and i["tiled_kernel"] == 1
        assert i["num_tiles"] == n // 256 and i["num_halo_tiles"] == 0 and i["num_halo_cells"] == 0 and i["owned_is_prefix"] == 1
        assert i["num_boundary_edges"] == 2 * (160 + 128)
        # every internal edge appears once if both cells share a tile, twice if not; boundary edges once
        assert 1.5 * n - (160 + 128) + i["num_boundary_edges"] <= i["num_edge_records"] <= 3 * n
    rec = {o: i["num_edge_records"] / n for o, i in info.items()}
    assert rec["rowmajor"] > 1.95 and rec["tiled"] < 1.65 and rec["hilbert"] < 1.68
    assert info["rowmajor"]["num_halo_entries"] > 4 * info["tiled"]["num_halo_entries"]
    assert info["tiled"]["lds_bytes"] < info["rowmajor"]["lds_bytes"] < 64 * 1024
    # a numbering with locality fits the kernels' fixed LDS capacities (plane offsets become immediates), row-major does not
    assert info["tiled"]["lds_fixed_layout"] == 1 and info["hilbert"]["lds_fixed_layout"] == 1 and info["rowmajor"]["lds_fixed_layout"] == 0
    # a random numbering: three records per cell (every edge cut), halo lists of hundreds of cells, > 64 KB of LDS
    rng = np.random.default_rng(0)
    xyz, conn, _, _ = M.structured_tri_connectivity(64, 48)
    mesh = M.build_mesh(xyz, conn[rng.permutation(conn.shape[0])], boundary_classifier=M.single_boundary())
    i = probe_layout(RDyFlowConfig(), mesh)
    assert i["num_edge_records"] / mesh.num_cells > 2.9 and i["max_tile_halo_cells"] > 256 and i["lds_bytes"] > 64 * 1024


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
    # second order: the fused kernel also stages a second ring; its halo tiles include tiles whose FIRST RING touches a ghost
    s = probe_layout(RDyFlowConfig(second_order=True), m)
    assert s["second_order_fused"] == 1 and s["max_tile_ring2_cells"] > s["max_tile_halo_cells"] and s["lds_bytes"] > i["lds_bytes"]
    assert s["lds_fixed_layout"] == 1 and s["lds_bytes"] == 8 * (6 * 360 + 5 * 520 + 3 * 8) + 4 * 520       # MusclSoATri
    assert s["num_halo_tiles"] >= i["num_halo_tiles"]
    old = os.environ.get("RDYHIP_MUSCL")
    os.environ["RDYHIP_MUSCL"] = "split"
    try:
        assert probe_layout(RDyFlowConfig(second_order=True), m)["second_order_fused"] == 0
    finally:
        if old is None:
            os.environ.pop("RDYHIP_MUSCL")
        else:
            os.environ["RDYHIP_MUSCL"] = old


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


def test_second_order_flux_storage_is_chosen_from_the_tiles(monkeypatch):
    """fused second-order kernel: the edge fluxes take the gradients' LDS storage when every tile's edges fit the kernel's
    register rounds (2 x 256 for triangles, 3 x 256 for quads) -- a smaller workgroup footprint, four per CU on the
    benchmark meshes -- and the separate region otherwise (rdyhip_api.hip: layout_build; RDYHIP_MUSCL_EF_OVERLAY=0 forces it)"""
    def lds(mesh, case, overlay):
        monkeypatch.setenv("RDYHIP_MUSCL_EF_OVERLAY", overlay)
        case.config.second_order = True
        return probe_layout(case.config, mesh, case.condition_types)

    tri = M.structured_tri_mesh(96, 64, order="tiled")
    ctri = CS.friction_slope_case(tri, 96.0, 64.0, dt=1e-3)
    a, b = lds(tri, ctri, "1"), lds(tri, ctri, "0")
    assert a["max_tile_edges"] <= 512 and a["lds_bytes"] < b["lds_bytes"] and a["lds_bytes"] <= 40 * 1024
    quad = CS.dam_break_quads_mesh(640, 320, 0, 1, order="tiled")
    cq = CS.dam_break_quads_case(quad)
    a, b = lds(quad, cq, "1"), lds(quad, cq, "0")
    assert 512 < a["max_tile_edges"] <= 768 and a["lds_bytes"] < b["lds_bytes"] and a["lds_bytes"] <= 40 * 1024
    # a numbering without locality: more edges per tile than the register rounds hold -> the separate region either way
    rng = np.random.default_rng(0)
    xyz, conn, _, _ = M.structured_tri_connectivity(96, 64)
    rnd = M.build_mesh(xyz, conn[rng.permutation(conn.shape[0])], boundary_classifier=M.box_side_boundaries(0.0, 96.0, 0.0, 64.0))
    cr = CS.friction_slope_case(rnd, 96.0, 64.0, dt=1e-3)
    a, b = lds(rnd, cr, "1"), lds(rnd, cr, "0")
    assert a["max_tile_edges"] > 512 and a["lds_bytes"] == b["lds_bytes"]
