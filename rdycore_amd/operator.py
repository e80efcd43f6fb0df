"""Host-side mirror of RDycore's Operator interface for the SWE right-hand side.

Same names, argument meaning and error behaviour as the reference's
`CreateOperator / ApplyOperator / DestroyOperator` boundary and its data
setters (include/private/rdyoperatorimpl.h:208-271, src/operator.c), on top of
the C ABI in include/rdyhip.h.  torch is used only for device memory and
streams: `u_local` / `f_global` are float64 CUDA tensors laid out like the
PETSc Vecs they stand for ([num_cells,3] local, [num_owned_cells,3] global).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import RDyHipError  # noqa: F401  (re-export)
from .mesh import (CONDITION_CRITICAL_OUTFLOW, CONDITION_DIRICHLET, CONDITION_REFLECTING, RDyMesh)

SOURCE_SEMI_IMPLICIT = 0     # RDyFlowSourceMethod, include/private/rdyconfigimpl.h:52-56
SOURCE_IMPLICIT_XQ2018 = 1
RIEMANN_ROE = 0
WELL_BALANCING_NONE = 0      # RDyWellBalanceMethod, include/private/rdyconfigimpl.h:58-62
WELL_BALANCING_HR = 2        # hydrostatic reconstruction
LIMITER_MINMOD, LIMITER_NONE, LIMITER_VANLEER = 0, 1, 2   # RDyLimiterType, include/private/rdyconfigimpl.h:64-71

PHASE_ALL, PHASE_INTERIOR, PHASE_HALO = 0, 1, 2


@dataclasses.dataclass
class RDyFlowConfig:
    """The scalars of RDyConfig that reach the SWE operator, with the defaults
    of src/yaml_input.c:851-865."""
    tiny_h: float = 1e-7
    h_anuga_regular: float = 0.0
    xq2018_threshold: float = 1e-10
    source_method: int = SOURCE_SEMI_IMPLICIT
    riemann: int = RIEMANN_ROE
    well_balancing: int = WELL_BALANCING_NONE
    second_order: bool = False        # numerics.second_order: MUSCL reconstruction (src/swe/swe_petsc.c:98-213)
    limiter: int = LIMITER_MINMOD     # numerics.limiter
    cached_f_stores: bool = False     # RDYHIP_CONFIG_CACHED_F_STORES: F is read back by a separate update kernel (TSEULER's VecAXPY)


@dataclasses.dataclass
class CourantNumberDiagnostics:
    """include/private/rdyoperatorimpl.h:21-25"""
    max_courant_num: float
    global_edge_id: int
    global_cell_id: int


class _DeviceArray:
    """Wraps a raw device pointer for torch.as_tensor (zero copy)."""

    def __init__(self, ptr: int, shape: Tuple[int, ...], owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}


def _ptr(t) -> int:
    return int(t.data_ptr())


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _abi_arguments(config: "RDyFlowConfig", mesh: RDyMesh, condition_types: Optional[Sequence[int]]):
    """RDyHipConfig / RDyHipMesh / RDyHipBoundary[] for a mesh (the arrays are borrowed: keep `keep` alive during the call)"""
    nb = len(mesh.boundaries)
    if condition_types is None:
        condition_types = [CONDITION_REFLECTING] * nb
    if len(condition_types) != nb:
        raise RDyHipError(83, f"{len(condition_types)} boundary conditions for {nb} boundaries")
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a

    m = _lib.RDyHipMesh()
    m.num_cells, m.num_owned_cells = mesh.num_cells, mesh.num_owned_cells
    m.num_edges, m.num_internal_edges = mesh.num_edges, mesh.num_internal_edges
    m.cell_is_owned = arr(mesh.cell_is_owned, np.int32).ctypes.data_as(_lib.c_int32_p)
    m.cell_local_to_owned = arr(mesh.cell_local_to_owned, np.int32).ctypes.data_as(_lib.c_int32_p)
    m.cell_global_ids = arr(mesh.cell_global_ids, np.int64).ctypes.data_as(_lib.c_int64_p)
    m.cell_areas = arr(mesh.cell_areas, np.float64).ctypes.data_as(_lib.c_double_p)
    m.cell_dz_dx = arr(mesh.cell_dz_dx, np.float64).ctypes.data_as(_lib.c_double_p)
    m.cell_dz_dy = arr(mesh.cell_dz_dy, np.float64).ctypes.data_as(_lib.c_double_p)
    m.edge_cell_ids = arr(mesh.edge_cell_ids, np.int32).ctypes.data_as(_lib.c_int32_p)
    m.edge_internal_ids = arr(mesh.edge_internal_ids, np.int32).ctypes.data_as(_lib.c_int32_p)
    m.edge_global_ids = arr(mesh.edge_global_ids, np.int64).ctypes.data_as(_lib.c_int64_p)
    m.edge_lengths = arr(mesh.edge_lengths, np.float64).ctypes.data_as(_lib.c_double_p)
    m.edge_cn = arr(mesh.edge_cn, np.float64).ctypes.data_as(_lib.c_double_p)
    m.edge_sn = arr(mesh.edge_sn, np.float64).ctypes.data_as(_lib.c_double_p)
    m.cell_zc = arr(mesh.cell_zc, np.float64).ctypes.data_as(_lib.c_double_p)
    if config.second_order:
        m.num_vertices = mesh.num_vertices
        m.cell_centroids = arr(mesh.cell_centroids, np.float64).ctypes.data_as(_lib.c_double_p)
        m.edge_vertex_ids = arr(mesh.edge_vertex_ids, np.int32).ctypes.data_as(_lib.c_int32_p)
        m.vertex_points = arr(mesh.xyz, np.float64).ctypes.data_as(_lib.c_double_p)
        if mesh.num_cells > mesh.num_owned_cells:      # edges.is_owned: who reports the Courant number of a cut edge (swe_petsc.c:172-190)
            m.edge_is_owned = arr(mesh.edge_is_owned(), np.int32).ctypes.data_as(_lib.c_int32_p)
    barr = (_lib.RDyHipBoundary * max(nb, 1))()
    for i, b in enumerate(mesh.boundaries):
        barr[i].num_edges = b.num_edges
        barr[i].edge_ids = arr(b.edge_ids, np.int32).ctypes.data_as(_lib.c_int32_p)
        barr[i].condition_type = int(condition_types[i])
    cfg = _lib.RDyHipConfig(config.tiny_h, config.h_anuga_regular, config.xq2018_threshold,
                            int(config.source_method), int(config.riemann), int(config.well_balancing),
                            1 if config.second_order else 0, int(config.limiter), 1 if getattr(config, "cached_f_stores", False) else 0)
    return cfg, m, nb, barr, list(condition_types), keep


def probe_layout(config: "RDyFlowConfig", mesh: RDyMesh, condition_types: Optional[Sequence[int]] = None) -> dict:
    """rdyhip_probe_layout: the host-side layout pass of rdyhip_create alone (no GPU needed) -- validates the mesh
    exactly as create does and returns the layout numbers (tiles, edge records, halo lists, LDS per workgroup)."""
    cfg, m, nb, barr, _, _keep = _abi_arguments(config, mesh, condition_types)
    info = _lib.RDyHipLayoutInfo()
    _lib.check(_lib.load().rdyhip_probe_layout(C.byref(cfg), C.byref(m), nb, barr, C.byref(info)))
    return {k: getattr(info, k) for k, _ in info._fields_}


class Operator:
    """Operator (include/private/rdyoperatorimpl.h:103-201), native MI355X backend."""

    def __init__(self, handle, mesh: RDyMesh, config: RDyFlowConfig, condition_types: Sequence[int]):
        self._h = handle
        self.mesh = mesh
        self.config = config
        self.condition_types = list(condition_types)
        self.num_components = 3

    # -- CreateOperator (src/operator.c:348-417) ---------------------------
    @classmethod
    def create(cls, config: RDyFlowConfig, mesh: RDyMesh, condition_types: Optional[Sequence[int]] = None) -> "Operator":
        """`condition_types[b]` is the flow condition type of `mesh.boundaries[b]`
        (a boundary with no condition is reflecting, src/rdysetup.c:283-503)."""
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("rdycore_amd.Operator needs a HIP device (no CPU fallback)")
        torch.cuda.current_device()  # make sure the HIP context exists on the chosen device
        cfg, m, nb, barr, condition_types, _keep = _abi_arguments(config, mesh, condition_types)
        h = C.c_void_p()
        _lib.check(lib.rdyhip_create(C.byref(cfg), C.byref(m), nb, barr, C.byref(h)))
        return cls(h, mesh, config, condition_types)

    # -- DestroyOperator (src/operator.c:421-493) --------------------------
    def destroy(self):
        if self._h is not None and self._h.value:
            _lib.check(_lib.load().rdyhip_destroy(C.byref(self._h)))
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- argument checks shared by the apply calls -------------------------
    def _check_vecs(self, u_local: torch.Tensor, f_global: torch.Tensor):
        for name, t, n in (("u_local", u_local, self.mesh.num_cells), ("f_global", f_global, self.mesh.num_owned_cells)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                raise RDyHipError(83, f"{name} must be a contiguous float64 device tensor")
            if t.numel() != 3 * n:
                # src/swe/swe_petsc.c:234
                raise RDyHipError(83, f"Number of dof in {name} must be 3! ({t.numel()} values for {n} cells)")

    # -- ApplyOperator (src/operator.c:680-690): f_global += F(u_local) ----
    def apply(self, dt: float, u_local: torch.Tensor, f_global: torch.Tensor):
        self._check_vecs(u_local, f_global)
        _lib.check(_lib.load().rdyhip_apply(self._h, float(dt), _ptr(u_local), _ptr(f_global), _stream()))

    # -- OperatorRHSFunction's zero + reset + apply (src/rdysetup.c:1130-1139), fused
    def rhs_function(self, dt: float, u_local: torch.Tensor, f_global: torch.Tensor):
        self._check_vecs(u_local, f_global)
        _lib.check(_lib.load().rdyhip_rhs_function(self._h, float(dt), _ptr(u_local), _ptr(f_global), _stream()))

    def apply_phase(self, phase: int, overwrite: bool, dt: float, u_local: torch.Tensor, f_global: torch.Tensor,
                    reset_diagnostics: bool = False, gradients_ready: bool = False):
        self._check_vecs(u_local, f_global)
        flags = (1 if overwrite else 0) | (2 if reset_diagnostics else 0) | (4 if gradients_ready else 0)
        _lib.check(_lib.load().rdyhip_apply_phase(self._h, int(phase), flags, float(dt), _ptr(u_local),
                                                 _ptr(f_global), _stream()))

    # -- TSStep_Euler fused with the RHS: u_out[owned] = u_local[owned] + dt * RHS(u_local) ----------------
    def euler_step(self, dt: float, u_local: torch.Tensor, u_out: torch.Tensor, f_global: Optional[torch.Tensor] = None,
                   phase: int = PHASE_ALL, reset_diagnostics: bool = True, gradients_ready: bool = False):
        """One forward-Euler step without a separate axpy pass (rdyhip_euler_step).  Not in place; the ghost rows
        of u_out are the next halo update's.  f_global=None: F is not stored at all where the kernel allows it."""
        n = self.mesh.num_cells
        for name, t in (("u_local", u_local), ("u_out", u_out)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == 3 * n):
                raise RDyHipError(83, f"{name} must be a contiguous float64 device tensor of {n} cells x 3")
        if f_global is not None:
            self._check_vecs(u_local, f_global)
        flags = (2 if reset_diagnostics else 0) | (4 if gradients_ready else 0)
        _lib.check(_lib.load().rdyhip_euler_step(self._h, int(phase), flags, float(dt), _ptr(u_local), _ptr(u_out),
                                                 _ptr(f_global) if f_global is not None else None, _stream()))

    # -- second order: ComputeLeastSquaresGradients for the owned cells (src/operator_fluxes_ceed.c:998-1042)
    def compute_gradients(self, u_local: torch.Tensor, phase: int = PHASE_ALL):
        _lib.check(_lib.load().rdyhip_compute_gradients(self._h, int(phase), _ptr(u_local), _stream()))

    @property
    def gradients(self) -> torch.Tensor:
        """[num_cells, 6] (dh/dx, dh/dy, dhu/dx, dhu/dy, dhv/dx, dhv/dy) by LOCAL cell; ghost rows are the caller's to fill"""
        return self._field(4, 6)

    # -- SetOperatorBoundaryValues (src/operator.c:1045-1061) --------------
    def set_boundary_values(self, boundary: int, values, comp_offset: int = 0, ordered: bool = False):
        """values[e, c] for component comp_offset+c of boundary edge e (RDySetFlowDirichletBoundaryValues,
        src/rdydata.c:88-106).  `ordered`: the stream-ordered form (rdyhip_set_boundary_values_on, current stream) -- no
        device synchronisation, no blocking copy; the default synchronises as the legacy entry point does."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        if ordered:
            _lib.check(_lib.load().rdyhip_set_boundary_values_on(self._h, int(boundary), int(comp_offset), int(v.shape[1]),
                                                                int(v.shape[0]), v.ctypes.data_as(_lib.c_double_p), _stream()))
            return
        _lib.check(_lib.load().rdyhip_set_boundary_values(self._h, int(boundary), int(comp_offset), int(v.shape[1]),
                                                         int(v.shape[0]), v.ctypes.data_as(_lib.c_double_p)))

    # -- ExtractOperatorBoundaryFluxes ----------------------------------------
    def boundary_fluxes(self, boundary: int, accumulated: bool = False) -> np.ndarray:
        n = self.mesh.boundaries[boundary].num_edges
        out = np.zeros((n, 3))
        _lib.check(_lib.load().rdyhip_get_boundary_fluxes(self._h, int(boundary), 1 if accumulated else 0, n,
                                                         out.ctypes.data_as(_lib.c_double_p)))
        return out

    def reset_boundary_fluxes_accum(self):
        _lib.check(_lib.load().rdyhip_reset_boundary_fluxes_accum(self._h))

    # -- external sources (RDySet{Regional,Domain}{Water,XMomentum,YMomentum}Source, src/rdydata.c:225-366)
    def set_domain_external_source(self, comp: int, values, ordered: bool = False):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if v.size != self.mesh.num_owned_cells:
            raise RDyHipError(60, f"size ({v.size}) does not match the number of owned cells ({self.mesh.num_owned_cells})")
        if ordered:
            _lib.check(_lib.load().rdyhip_set_external_source_on(self._h, int(comp), int(v.size), None, v.ctypes.data_as(_lib.c_double_p), _stream()))
            return
        _lib.check(_lib.load().rdyhip_set_external_source(self._h, int(comp), int(v.size), None, v.ctypes.data_as(_lib.c_double_p)))

    def set_regional_external_source(self, owned_cell_ids, comp: int, values, ordered: bool = False):
        ids = np.ascontiguousarray(owned_cell_ids, dtype=np.int32).ravel()
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if v.size != ids.size:
            raise RDyHipError(60, f"size ({v.size}) does not match the region size ({ids.size})")
        if ordered:
            _lib.check(_lib.load().rdyhip_set_external_source_on(self._h, int(comp), int(v.size), ids.ctypes.data_as(_lib.c_int32_p),
                                                                v.ctypes.data_as(_lib.c_double_p), _stream()))
            return
        _lib.check(_lib.load().rdyhip_set_external_source(self._h, int(comp), int(v.size), ids.ctypes.data_as(_lib.c_int32_p),
                                                         v.ctypes.data_as(_lib.c_double_p)))

    # -- Manning's n (RDySet{Regional,Domain}ManningsN, src/rdydata.c:506-539)
    def set_domain_mannings_n(self, values, ordered: bool = False):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if v.size != self.mesh.num_owned_cells:
            raise RDyHipError(60, f"size ({v.size}) does not match the number of owned cells ({self.mesh.num_owned_cells})")
        if ordered:
            _lib.check(_lib.load().rdyhip_set_mannings_on(self._h, int(v.size), None, v.ctypes.data_as(_lib.c_double_p), _stream()))
            return
        _lib.check(_lib.load().rdyhip_set_mannings(self._h, int(v.size), None, v.ctypes.data_as(_lib.c_double_p)))

    def set_regional_mannings_n(self, owned_cell_ids, values, ordered: bool = False):
        ids = np.ascontiguousarray(owned_cell_ids, dtype=np.int32).ravel()
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if v.size != ids.size:
            raise RDyHipError(60, f"size ({v.size}) does not match the region size ({ids.size})")
        if ordered:
            _lib.check(_lib.load().rdyhip_set_mannings_on(self._h, int(v.size), ids.ctypes.data_as(_lib.c_int32_p),
                                                         v.ctypes.data_as(_lib.c_double_p), _stream()))
            return
        _lib.check(_lib.load().rdyhip_set_mannings(self._h, int(v.size), ids.ctypes.data_as(_lib.c_int32_p),
                                                  v.ctypes.data_as(_lib.c_double_p)))

    def refresh_field(self, field: int, values):
        """rdyhip_refresh_field: a whole input field (1: external sources [owned,3]; 2: Manning n [owned]) from a host array or a
        device tensor, ordered on the current stream"""
        if isinstance(values, torch.Tensor) and values.is_cuda:
            t = values.contiguous()
            _lib.check(_lib.load().rdyhip_refresh_field(self._h, int(field), _ptr(t), int(t.numel()), 1, _stream()))
            return
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        _lib.check(_lib.load().rdyhip_refresh_field(self._h, int(field), v.ctypes.data, int(v.size), 0, _stream()))

    # -- device-resident fields ------------------------------------------------
    def _field(self, field: int, ncomp: int) -> torch.Tensor:
        p = C.c_void_p()
        n = C.c_int64()
        _lib.check(_lib.load().rdyhip_field_ptr(self._h, int(field), C.byref(p), C.byref(n)))
        if n.value == 0:
            return torch.zeros((0, ncomp), dtype=torch.float64, device="cuda")
        t = torch.as_tensor(_DeviceArray(p.value, (n.value,), self), device="cuda")
        return t.view(-1, ncomp) if ncomp > 1 else t

    @property
    def primitive_variables(self) -> torch.Tensor:
        """Operator.primitive_variables: [owned,3] (h,u,v), written by every apply."""
        return self._field(0, 3)

    @property
    def external_sources(self) -> torch.Tensor:
        return self._field(1, 3)

    @property
    def mannings_n(self) -> torch.Tensor:
        return self._field(2, 1)

    def enable_flux_divergence(self, enable: bool = True):
        _lib.check(_lib.load().rdyhip_enable_flux_divergence(self._h, 1 if enable else 0))

    @property
    def flux_divergence(self) -> torch.Tensor:
        return self._field(3, 3)

    # -- diagnostics (src/operator.c:772-784, 867-893) ---------------------
    def reset_diagnostics(self):
        _lib.check(_lib.load().rdyhip_reset_diagnostics(self._h, _stream()))

    def update_diagnostics(self):
        _lib.check(_lib.load().rdyhip_update_diagnostics(self._h, _stream()))

    def get_diagnostics(self) -> CourantNumberDiagnostics:
        c = _lib.RDyHipCourant()
        _lib.check(_lib.load().rdyhip_get_diagnostics(self._h, C.byref(c)))
        return CourantNumberDiagnostics(c.max_courant_num, c.global_edge_id, c.global_cell_id)

    # -- explicit Euler update kept on the device ----------------------------
    def axpy_owned(self, dt: float, f_global: torch.Tensor, u_local: torch.Tensor):
        self._check_vecs(u_local, f_global)
        _lib.check(_lib.load().rdyhip_axpy_owned(self._h, float(dt), _ptr(f_global), _ptr(u_local), _stream()))

    def copy_owned_rows(self, u_global: torch.Tensor, u_local: torch.Tensor):
        """the local half of DMGlobalToLocal (src/rdysetup.c:1133-1134): u_local[owned cell o] = u_global[o]"""
        self._check_vecs(u_local, u_global)
        _lib.check(_lib.load().rdyhip_copy_owned_rows(self._h, _ptr(u_global), _ptr(u_local), _stream()))

    def layout_info(self) -> dict:
        info = _lib.RDyHipLayoutInfo()
        _lib.check(_lib.load().rdyhip_layout_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}


def pack_rows(src: torch.Tensor, row_ids: torch.Tensor, buf: torch.Tensor):
    """buf[i, :] = src[row_ids[i], :] for a [rows, ncomp] device array (the gradient field)"""
    _lib.check(_lib.load().rdyhip_pack_rows(_ptr(src), int(src.shape[1]), _ptr(row_ids), int(row_ids.numel()), _ptr(buf), _stream()))


def unpack_rows(dst: torch.Tensor, row_ids: torch.Tensor, buf: torch.Tensor):
    _lib.check(_lib.load().rdyhip_unpack_rows(_ptr(dst), int(dst.shape[1]), _ptr(row_ids), int(row_ids.numel()), _ptr(buf), _stream()))


def pack_cells(u_local: torch.Tensor, cell_ids: torch.Tensor, buf: torch.Tensor):
    """buf[i] = u_local[cell_ids[i]] on the current stream (halo send side)."""
    _lib.check(_lib.load().rdyhip_pack_cells(_ptr(u_local), _ptr(cell_ids), int(cell_ids.numel()), _ptr(buf), _stream()))


def unpack_cells(u_local: torch.Tensor, cell_ids: torch.Tensor, buf: torch.Tensor):
    """u_local[cell_ids[i]] = buf[i] on the current stream (halo receive side)."""
    _lib.check(_lib.load().rdyhip_unpack_cells(_ptr(u_local), _ptr(cell_ids), int(cell_ids.numel()), _ptr(buf), _stream()))
