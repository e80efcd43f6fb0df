"""TEST INFRASTRUCTURE -- ctypes wrapper of the CPU oracle (oracle/swe_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; nothing under rdycore_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)


class OracleMesh(C.Structure):
    _fields_ = [
        ("num_cells", C.c_int), ("num_owned_cells", C.c_int), ("num_edges", C.c_int), ("num_internal_edges", C.c_int),
        ("is_owned", c_int_p), ("local_to_owned", c_int_p), ("cell_global_ids", c_ll_p),
        ("areas", c_double_p), ("dz_dx", c_double_p), ("dz_dy", c_double_p), ("zc", c_double_p),
        ("cell_ids", c_int_p), ("internal_edge_ids", c_int_p), ("edge_global_ids", c_ll_p),
        ("lengths", c_double_p), ("cn", c_double_p), ("sn", c_double_p),
        ("num_vertices", C.c_int), ("centroids", c_double_p), ("vertex_ids", c_int_p), ("points", c_double_p),
        ("edge_is_owned", c_int_p),
    ]


class OracleBoundary(C.Structure):
    _fields_ = [("num_edges", C.c_int), ("edge_ids", c_int_p), ("bc_type", C.c_int)]


class OracleConfig(C.Structure):
    _fields_ = [("tiny_h", C.c_double), ("h_anuga_regular", C.c_double), ("xq2018_threshold", C.c_double),
                ("source_method", C.c_int), ("well_balancing", C.c_int), ("second_order", C.c_int), ("limiter", C.c_int)]


class OracleCourant(C.Structure):
    _fields_ = [("max_courant_num", C.c_double), ("global_edge_id", C.c_longlong), ("global_cell_id", C.c_longlong)]


def build(force: bool = False, openmp: bool = False) -> str:
    name = "libswe_oracle_omp.so" if openmp else "libswe_oracle.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("swe_oracle.c", "forcing_oracle.c", "swe_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


_LIB_OMP = None


def lib(openmp: bool = False):
    """the serial oracle, or (openmp=True) the same source built with -fopenmp: all host cores, bitwise the same results
    for the first-order path (the HR and second-order interior-flux loops stay serial)"""
    global _LIB, _LIB_OMP
    if openmp:
        if _LIB_OMP is None:
            _LIB_OMP = _bind(C.CDLL(build(openmp=True)))
        return _LIB_OMP
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


def _bind(L):
    if True:
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(OracleMesh), C.POINTER(OracleConfig), C.c_int, C.POINTER(OracleBoundary)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        for name in ("oracle_apply", "oracle_apply_interior", "oracle_apply_rest"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_double, c_double_p, c_double_p]
            getattr(L, name).restype = C.c_int
        for name in ("oracle_boundary_values", "oracle_boundary_fluxes", "oracle_boundary_fluxes_accum"):
            getattr(L, name).restype = c_double_p
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        for name in ("oracle_external_sources", "oracle_material_properties", "oracle_flux_divergence",
                     "oracle_primitive_variables"):
            getattr(L, name).restype = c_double_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_compute_gradients.argtypes = [C.c_void_p, c_double_p]
        L.oracle_set_gradients_ready.argtypes = [C.c_void_p, C.c_int]
        L.oracle_gradients.restype = c_double_p
        L.oracle_gradients.argtypes = [C.c_void_p, C.c_int]
        for name in ("oracle_rhs_local", "oracle_ls_grad_coeffs"):
            getattr(L, name).restype = c_double_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_reset_diagnostics.argtypes = [C.c_void_p]
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_set_num_threads.restype = C.c_int
        L.oracle_get_diagnostics.argtypes = [C.c_void_p, C.POINTER(OracleCourant)]
        L.oracle_roe_flux.argtypes = [C.c_double] * 8 + [c_double_p, c_double_p]
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _lp(a):
    return a.ctypes.data_as(c_ll_p)


def roe_flux(hl, ul, vl, hr, ur, vr, sn, cn):
    f = np.zeros(3)
    amax = C.c_double(0.0)
    lib().oracle_roe_flux(hl, ul, vl, hr, ur, vr, sn, cn, _dp(f), C.byref(amax))
    return f, amax.value


class OracleOperator:
    """CPU oracle for ApplyOperator on one rank's mesh (mesh: rdycore_amd.mesh.RDyMesh)."""

    def __init__(self, mesh, bc_types: Sequence[int], tiny_h=1e-7, h_anuga_regular=0.0, xq2018_threshold=1e-10,
                 source_method=0, well_balancing=0, second_order=False, limiter=0,
                 all_edges_local=False, openmp=False):
        """all_edges_local (second order only): treat every local internal edge as owned by this rank, i.e. solve the
        cut edges redundantly on both ranks instead of the reference's owner-computes + reverse-add -- the scheme
        of the HIP path; the owned rows of F are then complete without the reverse exchange."""
        L = self._L = lib(openmp)
        self.mesh = mesh
        self._keep = []

        def keep(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a

        m = OracleMesh()
        m.num_cells, m.num_owned_cells = mesh.num_cells, mesh.num_owned_cells
        m.num_edges, m.num_internal_edges = mesh.num_edges, mesh.num_internal_edges
        m.is_owned = _ip(keep(mesh.cell_is_owned, np.int32))
        m.local_to_owned = _ip(keep(mesh.cell_local_to_owned, np.int32))
        m.cell_global_ids = _lp(keep(mesh.cell_global_ids, np.int64))
        m.areas = _dp(keep(mesh.cell_areas, np.float64))
        m.dz_dx = _dp(keep(mesh.cell_dz_dx, np.float64))
        m.dz_dy = _dp(keep(mesh.cell_dz_dy, np.float64))
        m.zc = _dp(keep(mesh.cell_zc, np.float64))
        m.cell_ids = _ip(keep(mesh.edge_cell_ids, np.int32))
        m.internal_edge_ids = _ip(keep(mesh.edge_internal_ids, np.int32))
        m.edge_global_ids = _lp(keep(mesh.edge_global_ids, np.int64))
        m.lengths = _dp(keep(mesh.edge_lengths, np.float64))
        m.cn = _dp(keep(mesh.edge_cn, np.float64))
        m.sn = _dp(keep(mesh.edge_sn, np.float64))
        if second_order:
            m.num_vertices = mesh.num_vertices
            m.centroids = _dp(keep(mesh.cell_centroids, np.float64))
            m.vertex_ids = _ip(keep(mesh.edge_vertex_ids, np.int32))
            m.points = _dp(keep(mesh.xyz, np.float64))
            m.edge_is_owned = _ip(keep(np.ones(mesh.num_edges) if all_edges_local else mesh.edge_is_owned(), np.int32))
        nb = len(mesh.boundaries)
        assert len(bc_types) == nb
        barr = (OracleBoundary * max(nb, 1))()
        for i, b in enumerate(mesh.boundaries):
            barr[i].num_edges = b.num_edges
            barr[i].edge_ids = _ip(keep(b.edge_ids, np.int32))
            barr[i].bc_type = int(bc_types[i])
        cfg = OracleConfig(tiny_h, h_anuga_regular, xq2018_threshold, int(source_method), int(well_balancing), int(bool(second_order)),
                           int(limiter))
        self.second_order = bool(second_order)
        self._h = L.oracle_create(C.byref(m), C.byref(cfg), nb, barr)
        self._keep.append((m, barr, cfg))
        self.num_boundaries = nb
        no = mesh.num_owned_cells

        def view(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)) if n > 0 else np.zeros(0)

        self.external_sources = view(L.oracle_external_sources(self._h), 3 * no).reshape(no, 3)
        self.mannings = view(L.oracle_material_properties(self._h), no)
        self.flux_divergence = view(L.oracle_flux_divergence(self._h), 3 * no).reshape(no, 3)
        self.primitive_variables = view(L.oracle_primitive_variables(self._h), 3 * no).reshape(no, 3)
        if second_order:
            nc, ni = mesh.num_cells, mesh.num_internal_edges
            self.gradients = [view(L.oracle_gradients(self._h, k), 2 * nc).reshape(nc, 2) for k in range(3)]
            self.rhs_local = view(L.oracle_rhs_local(self._h), 3 * nc).reshape(nc, 3)
            self.ls_grad_coeffs = view(L.oracle_ls_grad_coeffs(self._h), 4 * ni).reshape(ni, 4)
        self.boundary_values = []
        self.boundary_fluxes = []
        self.boundary_fluxes_accum = []
        for i, b in enumerate(mesh.boundaries):
            n = b.num_edges
            self.boundary_values.append(view(L.oracle_boundary_values(self._h, i), 3 * n).reshape(n, 3))
            self.boundary_fluxes.append(view(L.oracle_boundary_fluxes(self._h, i), 3 * n).reshape(n, 3))
            self.boundary_fluxes_accum.append(view(L.oracle_boundary_fluxes_accum(self._h, i), 3 * n).reshape(n, 3))

    def apply(self, dt: float, u_local: np.ndarray, f_global: Optional[np.ndarray] = None, stage: str = "") -> np.ndarray:
        """f_global += RHS(u_local); a zeroed f_global is made when none is given
        (the caller's VecZeroEntries, src/rdysetup.c:1130)."""
        u = np.ascontiguousarray(u_local, dtype=np.float64)
        assert u.size == 3 * self.mesh.num_cells
        if f_global is None:
            f_global = np.zeros((self.mesh.num_owned_cells, 3))
        assert f_global.flags.c_contiguous and f_global.size == 3 * self.mesh.num_owned_cells
        rc = getattr(self._L, "oracle_apply" + stage)(self._h, float(dt), _dp(u), _dp(f_global))
        if rc != 0:
            raise RuntimeError("oracle_apply failed")
        return f_global

    def apply_interior(self, dt, u_local, f_global=None):
        return self.apply(dt, u_local, f_global, stage="_interior")

    def apply_rest(self, dt, u_local, f_global):
        return self.apply(dt, u_local, f_global, stage="_rest")

    def compute_gradients(self, u_local: np.ndarray):
        """ComputeLeastSquaresGradients into self.gradients (ghost rows incomplete, as in the reference)."""
        u = np.ascontiguousarray(u_local, dtype=np.float64)
        self._L.oracle_compute_gradients(self._h, _dp(u))

    def set_gradients_ready(self, ready: bool):
        self._L.oracle_set_gradients_ready(self._h, int(ready))

    def gradients6(self) -> np.ndarray:
        """[num_cells, 6] = (dh/dx, dh/dy, dhu/dx, dhu/dy, dhv/dx, dhv/dy): the layout of the HIP operator's gradient array"""
        return np.concatenate(self.gradients, axis=1)

    def reset_diagnostics(self):
        self._L.oracle_reset_diagnostics(self._h)

    def diagnostics(self):
        d = OracleCourant()
        self._L.oracle_get_diagnostics(self._h, C.byref(d))
        return d.max_courant_num, d.global_edge_id, d.global_cell_id

    def close(self):
        if self._h:
            self._L.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- forcing loops (oracle/forcing_oracle.c) ---------------------------------
def forcing_current_data(table, cur_time, temporally_interpolate):
    t = np.ascontiguousarray(table, dtype=np.float64).ravel()
    idx, val = C.c_int(-1), C.c_double(0.0)
    lib().oracle_forcing_current_data(_dp(t), C.c_int(t.size // 2), C.c_double(cur_time), C.c_int(int(temporally_interpolate)),
                                      C.byref(idx), C.byref(val))
    return idx.value, val.value


def forcing_set_raster(data_vec, offset, data2mesh_idx):
    d = np.ascontiguousarray(data_vec, dtype=np.float64)
    m = np.ascontiguousarray(data2mesh_idx, dtype=np.int32)
    out = np.zeros(m.size)
    lib().oracle_forcing_set_raster(_dp(d), C.c_int(offset), _ip(m), C.c_int(m.size), _dp(out))
    return out


def forcing_set_unstructured(data_vec, stride, data2mesh_idx):
    d = np.ascontiguousarray(data_vec, dtype=np.float64)
    m = np.ascontiguousarray(data2mesh_idx, dtype=np.int32)
    out = np.zeros(m.size * stride)
    lib().oracle_forcing_set_unstructured(_dp(d), C.c_int(stride), _ip(m), C.c_int(m.size), _dp(out))
    return out.reshape(m.size, stride)


def forcing_raster_map(mesh_xc, mesh_yc, ncols, nrows, cellsize, data_xc, data_yc):
    x, y = (np.ascontiguousarray(a, dtype=np.float64) for a in (mesh_xc, mesh_yc))
    px, py = (np.ascontiguousarray(a, dtype=np.float64) for a in (data_xc, data_yc))
    out = np.zeros(x.size, dtype=np.int32)
    lib().oracle_forcing_raster_map(C.c_int(x.size), _dp(x), _dp(y), C.c_int(ncols), C.c_int(nrows), C.c_double(cellsize), _dp(px), _dp(py),
                                    _ip(out))
    return out


def forcing_unstructured_map(mesh_xc, mesh_yc, data_xc, data_yc):
    x, y = (np.ascontiguousarray(a, dtype=np.float64) for a in (mesh_xc, mesh_yc))
    px, py = (np.ascontiguousarray(a, dtype=np.float64) for a in (data_xc, data_yc))
    out = np.zeros(x.size, dtype=np.int32)
    lib().oracle_forcing_unstructured_map(C.c_int(x.size), _dp(x), _dp(y), C.c_int(px.size), _dp(px), _dp(py), _ip(out))
    return out
