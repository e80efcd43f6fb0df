"""Synthetic workloads for the SWE right-hand side:
states, bathymetry, Manning fields and boundary data on the meshes of mesh.py.
Shared by bench.py and the tests; numpy only.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import numpy as np

from .mesh import (CONDITION_CRITICAL_OUTFLOW, CONDITION_DIRICHLET, CONDITION_REFLECTING, RDyMesh)
from .operator import RDyFlowConfig, SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT


@dataclasses.dataclass
class Case:
    name: str
    mesh: RDyMesh
    config: RDyFlowConfig
    condition_types: List[int]
    u_local: np.ndarray                 # [num_cells,3]
    mannings: np.ndarray                # [num_owned]
    ext_src: np.ndarray                 # [num_owned,3]
    boundary_values: Dict[int, np.ndarray]   # boundary index -> [num_edges,3]
    dt: float


def mms_fields(x, y, H=1.0, U=0.5, V=0.5, N=0.02, K=2 * np.pi / 200.0):
    """t=0 fields of driver/tests/swe_roe/mms_conv_study.yaml:11-46, rescaled."""
    s, c = np.sin, np.cos
    h = H * (1 + s(K * x) * s(K * y))
    u = U * c(K * x) * s(K * y)
    v = V * s(K * x) * c(K * y)
    n = N * (1 + s(K * x) * s(K * y))
    return h, u, v, n


def mms_bathymetry(Z=0.25, K=2 * np.pi / 200.0):
    return lambda x, y: Z * np.sin(K * x) * np.sin(K * y)


def dam_break_case(mesh: RDyMesh, lx: float, dt: float = 1e-3, perturb: float = 0.01, seed: int = 12345,
                   source_method: int = SOURCE_SEMI_IMPLICIT) -> Case:
    """C2: flat-bed dam break, h = 10 (x < lx/2) / 5, Manning 0.015 (ex2b.yaml:58,84-93),
    all-reflecting boundaries.  hu, hv get a small seeded perturbation so the
    fluxes are not trivially symmetric."""
    xc = mesh.cell_centroids[:, 0]
    yc = mesh.cell_centroids[:, 1]
    u = np.zeros((mesh.num_cells, 3))
    u[:, 0] = np.where(xc < 0.5 * lx, 10.0, 5.0)
    if perturb:
        # a smooth, partition-independent perturbation (depends on position only)
        u[:, 1] = perturb * u[:, 0] * np.sin(0.37 * xc + 0.11 * yc + seed % 7)
        u[:, 2] = perturb * u[:, 0] * np.cos(0.23 * xc - 0.19 * yc + seed % 5)
    no = mesh.num_owned_cells
    return Case("dam_break", mesh, RDyFlowConfig(source_method=source_method), [CONDITION_REFLECTING] * len(mesh.boundaries), u,
                np.full(no, 0.015), np.zeros((no, 3)), {}, dt)


def friction_slope_case(mesh: RDyMesh, lx: float, ly: float, dt: float = 1e-3,
                        source_method: int = SOURCE_SEMI_IMPLICIT, dry_disc: bool = True,
                        K: Optional[float] = None) -> Case:
    """C3: MMS-style analytic state over sinusoidal bathymetry (the mesh must
    have been built with zfunc=mms_bathymetry(K=K)), Manning field, rain-like
    water source 1e-5, a dry disc, and one boundary of each type:
    left = Dirichlet (analytic state), right = critical outflow, rest reflecting."""
    K = 2 * np.pi / 200.0 if K is None else K
    xc = mesh.cell_centroids[:, 0]
    yc = mesh.cell_centroids[:, 1]
    h, uu, vv, _ = mms_fields(xc, yc, K=K)
    if dry_disc:
        r2 = (xc - 0.5 * lx) ** 2 + (yc - 0.5 * ly) ** 2
        h = np.where(r2 < (0.1 * ly) ** 2, 0.0, h)
    u = np.stack([h, h * uu, h * vv], axis=1)
    oc = mesh.owned_centroids()
    _, _, _, n = mms_fields(oc[:, 0], oc[:, 1], K=K)
    no = mesh.num_owned_cells
    src = np.zeros((no, 3))
    src[:, 0] = 1e-5
    ctypes: List[int] = []
    bvals: Dict[int, np.ndarray] = {}
    for i, b in enumerate(mesh.boundaries):
        if b.name == "left":
            ctypes.append(CONDITION_DIRICHLET)
            ec = mesh.edge_centroids[b.edge_ids]
            hb, ub, vb, _ = mms_fields(ec[:, 0], ec[:, 1], K=K)
            bvals[i] = np.stack([hb, hb * ub, hb * vb], axis=1)
        elif b.name == "right":
            ctypes.append(CONDITION_CRITICAL_OUTFLOW)
        else:
            ctypes.append(CONDITION_REFLECTING)
    return Case("friction_slope", mesh, RDyFlowConfig(source_method=source_method), ctypes, u, n, src, bvals, dt)


def houston_case(data_dir: str) -> Case:
    """driver/tests/swe_roe/Houston1km.DirichletBC.yaml on share/meshes/Houston1km_with_z.exo: 2746 triangles
    over a real DEM, initial state from Houston1km.ic.*.bin (natural cell order, [cell][h, hu, hv]), Manning
    0.015, side set 1 = Dirichlet boundary (height 0 unless a bc series drives it), every other boundary
    edge reflecting (src/rdysetup.c:342-431), dt = 30 s, coupling interval 60 s, stop 4200 s."""
    import os
    from . import mesh as M
    xyz, conn, side_sets = M.read_exodus(os.path.join(data_dir, "Houston1km_with_z.exo"))
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_side_sets(side_sets, conn, {1: "bottom_wall"}))
    u = M.read_petsc_vec(os.path.join(data_dir, "Houston1km.ic.int32.bin")).reshape(mesh.num_cells, 3)
    ctypes = [CONDITION_DIRICHLET if b.name == "bottom_wall" else CONDITION_REFLECTING for b in mesh.boundaries]
    bvals = {i: np.zeros((b.num_edges, 3)) for i, b in enumerate(mesh.boundaries) if b.name == "bottom_wall"}
    no = mesh.num_owned_cells
    return Case("houston1km", mesh, RDyFlowConfig(), ctypes, u, np.full(no, 0.015), np.zeros((no, 3)), bvals, 30.0)


def levee_hr_case(data_dir: str) -> Case:
    """driver/tests/swe_roe/levee.hr.yaml (the reference's hydrostatic-reconstruction test) on share/meshes/levee.exo:
    506 triangles, per-cell bed elevation from levee_elevation.*.bin (grid.cell_elevation: it replaces the centroid z
    and thereby the HR operator's zc, src/rdysetup.c:1078-1100, src/swe/swe_petsc.c:1213-1215), initial state from
    levee_ic.*.bin (dry cells on the levee), Manning 0.033, no side sets: every boundary edge reflecting; x-y projected
    lengths and areas as HR requires (src/rdymesh.c:1478-1509); 600 steps of 0.1 s."""
    import os
    from . import mesh as M
    from .operator import WELL_BALANCING_HR
    xyz, conn, side_sets = M.read_exodus(os.path.join(data_dir, "levee.exo"))
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_side_sets(side_sets, conn), project_2d=True)
    mesh.cell_zc = M.read_petsc_vec(os.path.join(data_dir, "levee_elevation.int32.bin"))[: mesh.num_cells].copy()
    u = M.read_petsc_vec(os.path.join(data_dir, "levee_ic.int32.bin")).reshape(mesh.num_cells, 3)
    no = mesh.num_owned_cells
    return Case("levee_hr", mesh, RDyFlowConfig(well_balancing=WELL_BALANCING_HR), [CONDITION_REFLECTING] * len(mesh.boundaries), u,
                np.full(no, 0.033), np.zeros((no, 3)), {}, 60.0 / 600.0)


def mixed_elements_case(data_dir: str) -> Case:
    """driver/tests/swe_roe/mixed_elements_ic_file.yaml on share/meshes/DamBreak_grid5x10_mixed_elements.exo: 20 quads +
    96 triangles (two element blocks), initial state and Manning n from binary files in natural cell order, no side sets
    (every boundary edge reflecting), 1000 steps of 0.018 s."""
    import os
    from . import mesh as M
    xyz, conn, side_sets = M.read_exodus(os.path.join(data_dir, "DamBreak_grid5x10_mixed_elements.exo"))
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_side_sets(side_sets, conn))
    u = M.read_petsc_vec(os.path.join(data_dir, "DamBreak_grid5x10_mixed_elements_wetdownstream.ic.int32.bin")).reshape(mesh.num_cells, 3)
    n = M.read_petsc_vec(os.path.join(data_dir, "manning_grid5x10_mixed_elements.int32.bin"))[: mesh.num_cells]
    no = mesh.num_owned_cells
    return Case("mixed_elements", mesh, RDyFlowConfig(), [CONDITION_REFLECTING] * len(mesh.boundaries), u, n.copy(), np.zeros((no, 3)), {},
                0.005 * 3600.0 / 1000.0)


def quad_tri_case(data_dir: str):
    """driver/tests/swe_roe/quad_tri_mesh.yaml on share/meshes/quad_tri_mesh.exo: 4 quads + 8 triangles in three element
    blocks (= regions quad, tri_1, tri_2), h = 3 everywhere, Manning 0.015, boundaries right / left / bottom Dirichlet
    (h = 5), top critical outflow, a runoff source of 2e-4 m/s on tri_1 only, 10 steps of 5e-4 s.  Returns the Case and
    the cells' region ids (the multi-homogeneous forcing of the reference's test addresses regions 1 and 3 and
    boundaries 1, 2 and 4)."""
    import os
    from . import mesh as M
    xyz, conn, side_sets, region = M.read_exodus(os.path.join(data_dir, "quad_tri_mesh.exo"), return_regions=True)
    names = {1: "right", 2: "left", 3: "top", 4: "bottom"}
    side_sets = {k: v for k, v in side_sets.items() if k in names}     # side set 5 is not a boundary of the yaml
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_side_sets(side_sets, conn, names))
    u = np.zeros((mesh.num_cells, 3))
    u[:, 0] = 3.0
    ctypes, bvals = [], {}
    for i, b in enumerate(mesh.boundaries):
        if b.name == "top":
            ctypes.append(CONDITION_CRITICAL_OUTFLOW)
        elif b.name in ("right", "left", "bottom"):
            ctypes.append(CONDITION_DIRICHLET)
            bvals[i] = np.tile([5.0, 0.0, 0.0], (b.num_edges, 1))
        else:
            ctypes.append(CONDITION_REFLECTING)
    src = np.zeros((mesh.num_owned_cells, 3))
    src[region == 2, 0] = 0.0002
    case = Case("quad_tri", mesh, RDyFlowConfig(), ctypes, u, np.full(mesh.num_owned_cells, 0.015), src, bvals, 0.005 / 10)
    return case, region


def create_operator(case: Case):
    """CreateOperator + the data setters the reference's setup calls
    (InitMaterialProperties / InitSourceConditions / InitDirichletBoundaryConditions,
    src/rdysetup.c:1546-1555), for a Case."""
    from .operator import Operator
    op = Operator.create(case.config, case.mesh, case.condition_types)
    op.set_domain_mannings_n(case.mannings)
    for comp in range(3):
        op.set_domain_external_source(comp, case.ext_src[:, comp])
    for b, vals in case.boundary_values.items():
        op.set_boundary_values(b, vals)
    return op


def ex2b_case(msh_path: str) -> Case:
    """C1: driver/tests/swe_roe/ex2b.yaml on share/meshes/planar_dam_10x5.msh --
    44 quads, h = 10 upstream / 5 downstream (ex2b.yaml:84-93), Manning 0.015
    (:58), implicit_xq2018 friction (:7), "bottom_wall" critical outflow, the
    other two boundaries reflecting (:71-82), dt = 0.005 h / 1000 steps = 0.018 s
    (:17-20)."""
    from . import mesh as M
    xyz, conn, region, edge_tag, names = M.read_gmsh41(msh_path)
    mesh = M.build_mesh(xyz, conn, boundary_classifier=M.boundaries_from_edge_tags(edge_tag, names))
    u = np.zeros((mesh.num_cells, 3))
    u[:, 0] = np.where(region == 1, 10.0, 5.0)
    ctypes = [CONDITION_CRITICAL_OUTFLOW if b.name == "bottom_wall" else CONDITION_REFLECTING for b in mesh.boundaries]
    no = mesh.num_owned_cells
    return Case("ex2b", mesh, RDyFlowConfig(source_method=SOURCE_IMPLICIT_XQ2018), ctypes, u,
                np.full(no, 0.015), np.zeros((no, 3)), {}, 0.005 * 3600.0 / 1000.0)


# ---------------------------------------------------------------------------
# the reference's own dam-break benchmark (docs/user/example-cases/dam-break)
# ---------------------------------------------------------------------------

def dam_break_keep(nxg: int, nyg: int):
    """Squares of the 10 m x 5 m partial dam-break domain that are NOT dam (dam-break-initial-condition.png: the dam
    occupies 4 <= x < 6 m except for the breach 2 <= y < 4 m).  On the 5120 x 2560 grid that leaves the 11,534,336
    cells of docs/user/example-cases/dam-break/index.md:10-11."""
    def keep(qi, qj):
        in_dam_x = (qi * 10 >= 4 * nxg) & (qi * 10 < 6 * nxg)
        in_breach = (qj * 5 >= 2 * nyg) & (qj * 5 < 4 * nyg)
        return ~(in_dam_x & ~in_breach)
    return keep


def dam_break_quads_mesh(nxg: int = 5120, nyg: int = 2560, rank: int = 0, world: int = 1, order: str = "tiled", tile=None) -> RDyMesh:
    """The quad mesh of the reference's dam-break benchmark (DamBreak_grid5120x2560: dx = dy = 10 m / 5120), this
    rank's part of an RCB partition; every domain-boundary edge (outer walls and the dam's faces) is one reflecting
    boundary (index.md:11-12)."""
    from . import partition as P
    return P.partitioned_structured_mesh("quad", nxg, nyg, (10.0 / nxg, 5.0 / nyg), rank, world, order=order, tile=tile,
                                         keep=dam_break_keep(nxg, nyg))


def dam_break_quads_case(mesh: RDyMesh, dt: float = 1.5625e-5) -> Case:
    """inputdeck_5120x2560.yaml:13-16, 34-56: h = 10 m upstream of the dam (x < 4 m), 5 m elsewhere, momenta zero,
    Manning 0.015, semi-implicit friction (the default), dt = 1.5625e-5 s, 100 Euler steps."""
    u = np.zeros((mesh.num_cells, 3))
    u[:, 0] = np.where(mesh.cell_centroids[:, 0] < 4.0, 10.0, 5.0)
    no = mesh.num_owned_cells
    return Case("dam_break_quads", mesh, RDyFlowConfig(), [CONDITION_REFLECTING] * len(mesh.boundaries), u, np.full(no, 0.015),
                np.zeros((no, 3)), {}, dt)


# ---------------------------------------------------------------------------
# C5: Harvey-scale stand-in (BASELINE.json configs[4]; the real Turning_30m mesh is not in the tree)
# ---------------------------------------------------------------------------

C5_ETA0 = 3.0   # initial water surface elevation [m]: leaves ~1/3 of the cells dry on the DEM below


def c5_dem(lx: float, ly: float):
    """Rough analytic DEM: a ramp towards the outlet side plus three sinusoids of decreasing wavelength
    ("sum of 3 sinusoids + ramp")."""
    def z(x, y):
        tp = 2.0 * np.pi
        return (5.0 * (1.0 - x / lx) + 2.0 * np.sin(tp * x / (0.2 * lx)) * np.sin(tp * y / (0.16 * ly))
                + 1.0 * np.sin(tp * (x + y) / (0.07 * lx)) + 0.5 * np.sin(tp * x / (0.0194 * lx)) * np.cos(tp * y / (0.0226 * ly)))
    return z


def c5_boundaries(lx: float, ly: float, tol: float = 1e-9):
    """critical-outflow segment on the low (right) side, 0.4 ly <= y <= 0.6 ly; every other boundary edge reflecting
    (the Harvey case: 13 outflow edges, everything else closed; harvey-flooding.md:3-7)"""
    from . import partition as P

    def namer(a, b):
        on_right = (np.abs(a[:, 0] - lx) < tol) & (np.abs(b[:, 0] - lx) < tol)
        ym = 0.5 * (a[:, 1] + b[:, 1])
        return np.where(on_right & (ym >= 0.4 * ly) & (ym <= 0.6 * ly), 1, 0)
    return P.owned_cell_boundaries(namer, ("walls", "outlet"))


def c5_mesh(nxg: int = 5000, nyg: int = 5000, rank: int = 0, world: int = 1, d: float = 1.0, order: str = "tiled") -> RDyMesh:
    from . import partition as P
    lx, ly = nxg * d, nyg * d
    return P.partitioned_structured_mesh("tri", nxg, nyg, d, rank, world, zfunc=c5_dem(lx, ly), order=order,
                                         boundary_classifier=c5_boundaries(lx, ly), project_2d=True)


def c5_case(mesh: RDyMesh, lx: float, ly: float, dt: float = 0.05) -> Case:
    """Flooded rough terrain: water surface at C5_ETA0 over the DEM (cells above it dry, about a third of them), a gentle
    flow field where wet, uniform rain (1e-5 m/s = 36 mm/h), Manning 0.03, hydrostatic reconstruction (needed on real DEMs:
    docs/theory/second_order_hydrostatic_reconstruction.md), outlet = critical outflow."""
    from .operator import WELL_BALANCING_HR
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    h = np.maximum(0.0, C5_ETA0 - mesh.cell_zc)
    tp = 2.0 * np.pi
    uu = 0.2 * np.sin(tp * yc / (0.12 * ly))
    vv = 0.1 * np.cos(tp * xc / (0.14 * lx))
    u = np.stack([h, h * uu, h * vv], axis=1)
    no = mesh.num_owned_cells
    src = np.zeros((no, 3))
    src[:, 0] = 1e-5
    ctypes = [CONDITION_CRITICAL_OUTFLOW if b.name == "outlet" else CONDITION_REFLECTING for b in mesh.boundaries]
    return Case("c5_flood", mesh, RDyFlowConfig(well_balancing=WELL_BALANCING_HR), ctypes, u, np.full(no, 0.03), src, {}, dt)


# ---------------------------------------------------------------------------
# an UNSTRUCTURED real-DEM mesh at benchmark scale: the reference's Houston1km mesh refined as `-dm_refine` does
# (src/rdydm.c:82-188); harvey-flooding.md:3-7 describes the production case this stands in for (Turning_30m, 2 926 532
# triangles over a real DEM, rain forcing, an ocean stage boundary)
# ---------------------------------------------------------------------------

HOUSTON_DT = 30.0   # Houston1km.DirichletBC.yaml:13-18, halved with every refinement level


def houston_refined_mesh(data_dir: str, levels: int = 6, order: str = "hilbert", project_2d: bool = False,
                         rank: int = 0, world: int = 1):
    """share/meshes/Houston1km_with_z.exo (2 746 triangles over the real DEM, in its file order) refined `levels` times by
    edge midpoints, vertex z interpolated: 2 746 x 4^levels triangles (6 levels: 11 247 616) with the irregular valences,
    the ragged outline and the side set of the original.  `order`: "natural" = children 4c..4c+3 of cell c (what DMPlex's
    refinement leaves), "hilbert" = cells renumbered along a Hilbert curve through their centroids (what INTEGRATION.md asks
    of a host before CreateOperator), "random" = a seeded permutation (worst case).  world > 1: this rank's RCB part.
    Returns (mesh, parent): parent[c] = Houston1km cell that local cell c descends from."""
    import os
    from . import mesh as M
    xyz, conn, side_sets = M.read_exodus(os.path.join(data_dir, "Houston1km_with_z.exo"))
    te = M.side_sets_to_tagged_edges(side_sets, conn)
    conn = conn[:, :3]
    parent = np.arange(conn.shape[0], dtype=np.int64)
    for _ in range(levels):
        xyz, conn, te = M.refine_triangles(xyz, conn, te)
        parent = np.repeat(parent, 4)
    cent = (xyz[conn[:, 0], :2] + xyz[conn[:, 1], :2] + xyz[conn[:, 2], :2]) / 3.0
    if order == "hilbert":
        perm = M.hilbert_cell_order(cent)
    elif order == "random":
        perm = np.random.default_rng(20170826).permutation(conn.shape[0])
    elif order == "natural":
        perm = None
    else:
        raise ValueError(order)
    if perm is not None:
        conn, parent, cent = conn[perm], parent[perm], cent[perm]
    if order == "hilbert":
        # vertices along the same curve, so that cells and their vertices stay close in memory (host-side build speed only)
        vperm = M.hilbert_cell_order(xyz)
        vinv = np.empty(xyz.shape[0], dtype=np.int64)
        vinv[vperm] = np.arange(xyz.shape[0])
        xyz, conn = xyz[vperm], vinv[conn].astype(np.int32)
        te = np.concatenate([vinv[te[:, :2]], te[:, 2:]], axis=1)
    cls = M.boundaries_from_tagged_edges(te, {1: "bottom_wall"}, num_vertices=xyz.shape[0])
    if world == 1:
        return M.build_mesh(xyz, conn, boundary_classifier=cls, project_2d=project_2d), parent
    from . import partition as P
    part = P.rcb_partition(cent, world)
    nvg = xyz.shape[0]

    def cls_local(mesh):   # the tags are in global vertex ids
        vg = mesh._vertex_global_ids
        be = mesh.edge_boundary_ids
        # artificial outer edges of ghost cells are no boundary of the domain: keep edges whose (only) cell is owned
        own = mesh.cell_is_owned[mesh.edge_cell_ids[2 * be]] != 0
        sub = dataclasses.replace(mesh, edge_boundary_ids=be[own], edge_vertex_ids=vg[mesh.edge_vertex_ids].astype(np.int64))
        return cls(sub)
    owned = part == rank
    lm = _extract_with_vertex_ids(xyz, conn, owned, cls_local, project_2d, nvg, part)
    keep = lm._cell_sel
    return lm, parent[keep]


def _extract_with_vertex_ids(xyz, conn, owned, classifier, project_2d, nvg, parts=None):
    """mesh.extract_local_mesh, with the selection of cells and the global vertex ids left on the mesh object for the
    caller (parent lookup, boundary tags in global vertex ids)"""
    from . import mesh as M
    conn4 = np.concatenate([conn, -np.ones((conn.shape[0], 1), conn.dtype)], axis=1) if conn.shape[1] == 3 else conn
    # the vertices extract_local_mesh keeps: those of the owned cells and of their edge-adjacent ghosts
    nv = xyz.shape[0]
    a = np.concatenate([conn[:, 0], conn[:, 1], conn[:, 2]]).astype(np.int64)
    b = np.concatenate([conn[:, 1], conn[:, 2], conn[:, 0]]).astype(np.int64)
    cell = np.tile(np.arange(conn.shape[0]), 3)
    key = np.minimum(a, b) * nv + np.maximum(a, b)
    o = np.argsort(key, kind="stable")
    ks = key[o]
    same = ks[1:] == ks[:-1]
    c1, c2 = cell[o[:-1]][same], cell[o[1:]][same]
    keep = owned.copy()
    keep[c2[owned[c1]]] = True
    keep[c1[owned[c2]]] = True
    used = np.unique(conn[keep])

    def wrapped(mesh):
        mesh._vertex_global_ids = used
        return classifier(mesh)
    lm = M.extract_local_mesh(xyz, conn4, owned, boundary_classifier=wrapped, project_2d=project_2d,
                              vertex_global_ids=np.arange(nv, dtype=np.int64), num_vertices_global=nvg, cell_parts=parts)
    assert lm.num_cells == lm._cell_sel.size      # the selection (source cell of each local cell) extract_local_mesh made
    return lm


def houston_refined_case(data_dir: str, levels: int = 6, order: str = "hilbert", hr: bool = False, time: float = 7200.0,
                         rank: int = 0, world: int = 1) -> Case:
    """The state and forcing of the reference's Houston1km test (Houston1km.DirichletBC.yaml with the homogeneous rain and
    stage series of driver/tests/swe_roe/CMakeLists.txt:110-130, taken at `time`) on the refined mesh: every child cell
    starts from its parent's water SURFACE (parent bed + parent depth, clipped at the child's own bed: wet/dry fronts along
    every slope) and the parent's velocity; Manning 0.015; rain from Houston1km.rain.*.bin; side set 1 a Dirichlet stage
    boundary from Houston1km.bc.*.bin (temporally interpolated), every other boundary edge reflecting; dt = 30 s / 2^levels."""
    import os
    from . import mesh as M
    from .operator import WELL_BALANCING_HR
    mesh, parent = houston_refined_mesh(data_dir, levels, order, project_2d=hr, rank=rank, world=world)
    # the parents' bed elevation (vertex mean) and state, in the file's natural cell order
    xyz0, conn0, _ = M.read_exodus(os.path.join(data_dir, "Houston1km_with_z.exo"))
    zc0 = xyz0[conn0[:, :3], 2].mean(axis=1)
    u0 = M.read_petsc_vec(os.path.join(data_dir, "Houston1km.ic.int32.bin")).reshape(-1, 3)
    eta = zc0 + u0[:, 0]
    vel = u0[:, 1:] / u0[:, :1]
    h = np.maximum(0.0, eta[parent] - mesh.cell_zc)
    h = np.where(h < 1e-3, 0.0, h)            # no sub-millimetre films: cells are either wet or exactly dry
    u = np.stack([h, h * vel[parent, 0], h * vel[parent, 1]], axis=1)
    rain = M.read_petsc_vec(os.path.join(data_dir, "Houston1km.rain.int32.bin")).reshape(-1, 2)
    bc = M.read_petsc_vec(os.path.join(data_dir, "Houston1km.bc.int32.bin")).reshape(-1, 2)
    i = int(np.searchsorted(rain[:, 0], time, side="right") - 1)
    rate = float(rain[max(i, 0), 1])                                        # piecewise constant (no -temporally_interpolate)
    stage = float(np.interp(time, bc[:, 0], bc[:, 1]))                      # -temporally_interpolate_bc
    no = mesh.num_owned_cells
    src = np.zeros((no, 3))
    src[:, 0] = rate
    ctypes = [CONDITION_DIRICHLET if b.name == "bottom_wall" else CONDITION_REFLECTING for b in mesh.boundaries]
    bvals = {i: np.tile([stage, 0.0, 0.0], (b.num_edges, 1)) for i, b in enumerate(mesh.boundaries) if b.name == "bottom_wall"}
    cfg = RDyFlowConfig(well_balancing=WELL_BALANCING_HR) if hr else RDyFlowConfig()
    return Case("houston_refined", mesh, cfg, ctypes, u, np.full(no, 0.015), src, bvals, HOUSTON_DT / 2 ** levels)


# ---------------------------------------------------------------------------
# a genuinely unstructured triangulation (vertex valences 3 .. 11, cell areas graded 1 : 9) with the C5 physics
# ---------------------------------------------------------------------------

def delaunay_mesh(n: int = 1210, rank: int = 0, world: int = 1, order: str = "hilbert", d: float = 1.0, seed: int = 2017) -> RDyMesh:
    """Delaunay triangulation (scipy / Qhull) of an (n+1)^2 lattice whose interior points are jittered by +-0.42 d and then
    warped smoothly (x += 0.08 L sin(2 pi x / L), likewise y: spacings 0.5 .. 1.5 d, areas 1 : 9) over the C5 DEM: about 2 n^2
    triangles (n = 1210: 2.93 M, the size of the reference's Turning_30m Harvey mesh, harvey-flooding.md:3-7) with the valence
    spread of a real mesh generator's output instead of a lattice's constant 6.  Cell order: "natural" = Qhull's, "hilbert",
    "random".  world > 1: this rank's RCB part with its edge-adjacent ghost layer."""
    from scipy.spatial import Delaunay
    from . import mesh as M
    from . import partition as P
    L = n * d
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="xy")
    ii, jj = ii.ravel(), jj.ravel()
    pts = np.stack([ii * d, jj * d], axis=1).astype(np.float64)
    inner = (ii > 0) & (ii < n) & (jj > 0) & (jj < n)
    pts[inner] += rng.uniform(-0.42 * d, 0.42 * d, (int(inner.sum()), 2))
    pts += 0.08 * L * np.sin(2.0 * np.pi * pts / L)          # monotone (derivative 1 +- 0.5): the box maps onto itself
    pts[ii == 0, 0], pts[ii == n, 0], pts[jj == 0, 1], pts[jj == n, 1] = 0.0, L, 0.0, L
    conn = Delaunay(pts).simplices.astype(np.int32)
    xyz = np.zeros((pts.shape[0], 3))
    xyz[:, :2] = pts
    xyz[:, 2] = c5_dem(L, L)(pts[:, 0], pts[:, 1])
    cent = (pts[conn[:, 0]] + pts[conn[:, 1]] + pts[conn[:, 2]]) / 3.0
    if order == "hilbert":
        perm = M.hilbert_cell_order(cent)
    elif order == "random":
        perm = np.random.default_rng(seed + 1).permutation(conn.shape[0])
    elif order == "natural":
        perm = np.arange(conn.shape[0])
    else:
        raise ValueError(order)
    conn, cent = conn[perm], cent[perm]
    if order == "hilbert":
        vperm = M.hilbert_cell_order(xyz)
        vinv = np.empty(xyz.shape[0], dtype=np.int64)
        vinv[vperm] = np.arange(xyz.shape[0])
        xyz, conn = xyz[vperm], vinv[conn].astype(np.int32)
    cls = c5_boundaries(L, L)
    if world == 1:
        return M.build_mesh(xyz, conn, boundary_classifier=cls, project_2d=True)
    parts = P.rcb_partition(cent, world)
    return M.extract_local_mesh(xyz, conn, parts == rank, boundary_classifier=cls, project_2d=True,
                                vertex_global_ids=np.arange(xyz.shape[0], dtype=np.int64), num_vertices_global=xyz.shape[0],
                                cell_parts=parts)
