"""Host-side domain decomposition: which cells a rank owns, and its local mesh.

The reference hands the mesh to PETSc's partitioner (ParMETIS when built in,
docs/common/installation.md:19-29) and then adds a 1-cell overlap
(DMPlexDistributeOverlap(dm, 1, ...), src/rdydm.c:145-157).  DMPlex is out of
scope here; what the operator needs from it is only "an owned mask per rank +
every cell sharing an edge with an owned cell as a ghost", which
`rdycore_amd.mesh.extract_local_mesh` builds from any owned mask.  This module
supplies the masks:

  * `rcb_partition` / `rcb_owned_mask`: recursive coordinate bisection of the cell
    centroids -- at every level the point set is cut perpendicular to its longer
    extent at the weighted median, so the parts are balanced to one cell and
    compact whatever the shape of the domain (holes, ragged coastlines).  Works
    for any number of parts (uneven splits for non powers of two).
  * `partitioned_structured_mesh`: a rank's piece of a structured triangle or quad
    mesh (optionally with cells removed, e.g. the dam of the reference's dam-break
    benchmark) cut by RCB, generated without ever forming the global connectivity:
    only the bounding block of the rank's part (+1 layer) is built.

All numpy; runs identically on every rank (deterministic), so no communication is
needed to agree on the partition.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import mesh as M


def _bisect(points: np.ndarray, idx: np.ndarray, part0: int, nparts: int, out: Optional[np.ndarray], only: Optional[int]):
    """Assigns parts part0 .. part0+nparts-1 to the points `idx` (ascending).  With `only` set, only the branch that
    contains that part is followed and the indices of that part are returned."""
    if nparts == 1:
        if out is not None:
            out[idx] = part0
        return idx
    pts = points[idx]
    ext = pts.max(axis=0) - pts.min(axis=0) if idx.size else np.zeros(2)
    axis = 0 if ext[0] >= ext[1] else 1
    n_lo = nparts // 2
    k = int(round(idx.size * n_lo / nparts))           # points that go to the lower half
    key = pts[:, axis]
    if 0 < k < idx.size:
        # the k smallest keys go low.  Points whose key EQUALS the k-th smallest (a whole column of a structured grid) are
        # split once, contiguously along the other axis -- argpartition alone would deal them out arbitrarily and turn the
        # cut inside that column into salt and pepper (more cut edges, more ghost cells, more halo tiles).
        kth = np.partition(key, k - 1)[k - 1]
        lo_sel = key < kth
        ties = np.nonzero(key == kth)[0]
        need = k - int(np.count_nonzero(lo_sel))
        if need > 0:
            t_order = np.argsort(pts[ties, 1 - axis], kind="stable")
            lo_sel[ties[t_order[:need]]] = True
    else:
        lo_sel = np.zeros(idx.size, dtype=bool) if k <= 0 else np.ones(idx.size, dtype=bool)
    lo, hi = idx[lo_sel], idx[~lo_sel]                  # both ascending again
    if only is not None:
        if only < part0 + n_lo:
            return _bisect(points, lo, part0, n_lo, out, only)
        return _bisect(points, hi, part0 + n_lo, nparts - n_lo, out, only)
    _bisect(points, lo, part0, n_lo, out, None)
    _bisect(points, hi, part0 + n_lo, nparts - n_lo, out, None)
    return None


def rcb_partition(points: np.ndarray, nparts: int) -> np.ndarray:
    """Part id (0..nparts-1) of every point [N,2] by recursive coordinate bisection."""
    points = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :2])
    if nparts < 1:
        raise ValueError("nparts must be >= 1")
    out = np.zeros(points.shape[0], dtype=np.int32)
    _bisect(points, np.arange(points.shape[0], dtype=np.int64), 0, nparts, out, None)
    return out


def rcb_owned_mask(points: np.ndarray, nparts: int, rank: int) -> np.ndarray:
    """`rcb_partition(points, nparts) == rank` without forming the other parts (only the branch that holds `rank`
    is bisected): O(N) work per rank."""
    points = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :2])
    if not 0 <= rank < nparts:
        raise ValueError("rank out of range")
    mine = _bisect(points, np.arange(points.shape[0], dtype=np.int64), 0, nparts, None, rank)
    mask = np.zeros(points.shape[0], dtype=bool)
    mask[mine] = True
    return mask


def partition_mesh(xyz: np.ndarray, conn: np.ndarray, nparts: int, rank: int, boundary_classifier=None,
                   project_2d: bool = False, parts: Optional[np.ndarray] = None) -> M.RDyMesh:
    """Rank `rank`'s local mesh (owned cells first, then the edge-adjacent ghost layer) of an arbitrary global mesh
    cut into `nparts` by RCB of the cell centroids (or by a given part array)."""
    conn = np.asarray(conn)
    if parts is None:
        nv = (conn >= 0).sum(axis=1)
        cent = np.where((conn >= 0)[:, :, None], xyz[np.maximum(conn, 0), :2], 0.0).sum(axis=1) / nv[:, None]
        parts = rcb_partition(cent, nparts)
    owned = np.asarray(parts) == rank
    return M.extract_local_mesh(xyz, conn, owned, boundary_classifier=boundary_classifier, project_2d=project_2d, cell_parts=parts)


def owned_cell_boundaries(namer: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None, names=("domain_boundary",)):
    """Boundary classifier for a local (partitioned) mesh: every edge without a right cell whose left cell is OWNED is a
    true domain-boundary edge (all neighbours of an owned cell are present as ghosts), while the outer edges of ghost
    cells are artefacts of the cut and belong to no boundary.  `namer(a, b) -> index into names` sorts the edges (given
    their end points [n,3]) into several boundaries; default: one boundary."""
    def f(mesh: M.RDyMesh):
        be = mesh.edge_boundary_ids
        left = mesh.edge_cell_ids[2 * be.astype(np.int64)]
        be = be[mesh.cell_is_owned[left] != 0]
        if namer is None:
            return [M.RDyBoundary(1, names[0], be.astype(np.int32))]
        a = mesh.xyz[mesh.edge_vertex_ids[be, 0]]
        b = mesh.xyz[mesh.edge_vertex_ids[be, 1]]
        which = namer(a, b)
        return [M.RDyBoundary(i + 1, nm, be[which == i].astype(np.int32)) for i, nm in enumerate(names)]
    return f


def _block_connectivity(kind: str, i0: int, i1: int, j0: int, j1: int, d: Tuple[float, float], order: str, tile):
    """Squares [i0,i1) x [j0,j1) of a global structured grid: vertices, cells (2 triangles per square with the global
    diagonal parity, or 1 quad), and each cell's global square index (qi, qj)."""
    nx, ny = i1 - i0, j1 - j0
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    xyz = np.zeros(((nx + 1) * (ny + 1), 3))
    xyz[:, 0] = (ii.ravel() + i0) * d[0]
    xyz[:, 1] = (jj.ravel() + j0) * d[1]
    qi, qj = np.meshgrid(np.arange(nx, dtype=np.int64), np.arange(ny, dtype=np.int64), indexing="xy")
    qi, qj = qi.ravel(), qj.ravel()
    if order == "tiled":
        # tx x ty blocks of squares aligned to the GLOBAL grid, so that the numbering of a cell's block does not depend on the cut
        tx, ty = (tile, tile) if np.isscalar(tile) else tile
        gi, gj = qi + i0, qj + j0
        nbx = (i1 + tx - 1) // tx + 1
        key = ((gj // ty) * nbx + (gi // tx)) * (tx * ty) + (gj % ty) * tx + (gi % tx)
        perm = np.argsort(key, kind="stable")
        qi, qj = qi[perm], qj[perm]
    elif order == "hilbert":
        perm = M.hilbert_cell_order(np.stack([qi + 0.5, qj + 0.5], axis=1))
        qi, qj = qi[perm], qj[perm]
    elif order != "rowmajor":
        raise ValueError(order)

    def vid(i, j):
        return j * (nx + 1) + i

    v00, v10, v11, v01 = vid(qi, qj), vid(qi + 1, qj), vid(qi + 1, qj + 1), vid(qi, qj + 1)
    if kind == "quad":
        conn = np.stack([v00, v10, v11, v01], axis=1).astype(np.int32)
        return xyz, conn, qi + i0, qj + j0, np.zeros(qi.size, dtype=np.int64)
    if kind != "tri":
        raise ValueError(kind)
    par = ((qi + i0 + qj + j0) % 2) == 0
    t0 = np.where(par[:, None], np.stack([v00, v10, v11], 1), np.stack([v00, v10, v01], 1))
    t1 = np.where(par[:, None], np.stack([v00, v11, v01], 1), np.stack([v10, v11, v01], 1))
    conn = np.empty((2 * qi.size, 3), dtype=np.int32)
    conn[0::2] = t0
    conn[1::2] = t1
    return xyz, conn, np.repeat(qi + i0, 2), np.repeat(qj + j0, 2), np.arange(conn.shape[0], dtype=np.int64) % 2


def partitioned_structured_mesh(kind: str, nxg: int, nyg: int, d, rank: int, world: int,
                                zfunc: Optional[Callable] = None, order: str = "tiled", tile=None,
                                keep: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None,
                                boundary_classifier=None, project_2d: bool = False) -> M.RDyMesh:
    """Rank `rank`'s local mesh of the nxg x nyg structured mesh (`kind` = "tri": two triangles per square, alternating
    diagonals; "quad") cut into `world` parts by RCB of the square centres.  `keep(qi, qj)` (bool per square) removes
    squares from the domain (their edges become domain boundaries).  Global cell id = cells_per_square * (qj*nxg + qi) + t.
    With world == 1 this is the whole mesh (every kept square owned).
    `tile` (order "tiled"): the block of squares numbered together, default 16 x 16 squares for triangles (two 256-cell tiles
    of the operator) and 16 x 15 for quads -- 240 cells with 511 edges, what one tile of the operator holds (its edge phase
    keeps 512 edge records in registers; a 16 x 16 block of quads has 544)."""
    if tile is None:
        tile = 16 if kind == "tri" else (16, 15)
    d = (float(d), float(d)) if np.isscalar(d) else (float(d[0]), float(d[1]))
    per = 2 if kind == "tri" else 1
    if world == 1:
        i0, i1, j0, j1 = 0, nxg, 0, nyg
        owned_sq = None
    else:
        qi, qj = np.meshgrid(np.arange(nxg, dtype=np.int32), np.arange(nyg, dtype=np.int32), indexing="xy")
        qi, qj = qi.ravel(), qj.ravel()
        if keep is not None:
            k = keep(qi, qj)
            qi, qj = qi[k], qj[k]
        pts = np.stack([(qi + 0.5) * d[0], (qj + 0.5) * d[1]], axis=1)
        # every square's part, not only this rank's branch: the owners of the ghost cells order them (by owner, then by
        # global id) so that the exchange receives in place (mesh.extract_local_mesh, rdyhip_local_cell_order)
        sq_part = rcb_partition(pts, world)
        mine = sq_part == rank
        del pts
        oi, oj = qi[mine], qj[mine]
        if oi.size == 0:
            raise ValueError(f"rank {rank} of {world} owns no cell of the {nxg}x{nyg} mesh")
        i0, i1 = max(int(oi.min()) - 1, 0), min(int(oi.max()) + 2, nxg)
        j0, j1 = max(int(oj.min()) - 1, 0), min(int(oj.max()) + 2, nyg)
        owned_sq = np.zeros((j1 - j0, i1 - i0), dtype=bool)
        owned_sq[oj - j0, oi - i0] = True
        inb = (qi >= i0) & (qi < i1) & (qj >= j0) & (qj < j1)
        part_sq = np.full((j1 - j0, i1 - i0), -1, dtype=np.int32)      # squares removed from the domain keep -1
        part_sq[qj[inb] - j0, qi[inb] - i0] = sq_part[inb]
        del qi, qj, oi, oj, mine, sq_part, inb
    xyz, conn, cqi, cqj, t = _block_connectivity(kind, i0, i1, j0, j1, d, order, tile)
    bi, bj = np.meshgrid(np.arange(i1 - i0 + 1, dtype=np.int64), np.arange(j1 - j0 + 1, dtype=np.int64), indexing="xy")
    vgid = (bj.ravel() + j0) * (nxg + 1) + bi.ravel() + i0      # vertex (i, j) of the global (nxg+1) x (nyg+1) lattice
    nvg = (nxg + 1) * (nyg + 1)
    del bi, bj
    if zfunc is not None:
        xyz[:, 2] = zfunc(xyz[:, 0], xyz[:, 1])
    if keep is not None:
        k = keep(cqi, cqj)
        conn, cqi, cqj, t = conn[k], cqi[k], cqj[k], t[k]
    gids = per * (cqj * nxg + cqi) + t
    if boundary_classifier is None:
        boundary_classifier = owned_cell_boundaries()
    if keep is None:
        nglobal = per * nxg * nyg
    else:
        gi, gj = np.meshgrid(np.arange(nxg, dtype=np.int32), np.arange(nyg, dtype=np.int32), indexing="xy")
        nglobal = per * int(np.count_nonzero(keep(gi.ravel(), gj.ravel())))
    if owned_sq is None:
        # one rank: every cell is owned, no ghost layer to find
        used = np.unique(conn)
        remap = np.full(xyz.shape[0], -1, dtype=np.int64)
        remap[used] = np.arange(used.size)
        return M.build_mesh(xyz[used], remap[conn].astype(np.int32), cell_global_ids=gids, num_cells_global=nglobal,
                            boundary_classifier=boundary_classifier, project_2d=project_2d,
                            vertex_global_ids=vgid[used], num_vertices_global=nvg)
    owned = owned_sq[cqj - j0, cqi - i0]
    return M.extract_local_mesh(xyz, conn, owned, cell_global_ids=gids, num_cells_global=nglobal,
                                boundary_classifier=boundary_classifier, project_2d=project_2d,
                                vertex_global_ids=vgid, num_vertices_global=nvg, cell_parts=part_sq[cqj - j0, cqi - i0])
