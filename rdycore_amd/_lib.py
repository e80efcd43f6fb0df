"""ctypes binding of the C ABI in include/rdyhip.h.

There is deliberately no fallback: if librdyhip.so is missing or fails to
load, importing the operator fails loudly (the product path is the HIP
extension or nothing).
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class RDyHipConfig(C.Structure):
    _fields_ = [("tiny_h", C.c_double), ("h_anuga_regular", C.c_double), ("xq2018_threshold", C.c_double),
                ("source_method", C.c_int32), ("riemann", C.c_int32), ("well_balancing", C.c_int32), ("second_order", C.c_int32),
                ("limiter", C.c_int32), ("flags", C.c_int32)]


class RDyHipMesh(C.Structure):
    _fields_ = [
        ("num_cells", C.c_int32), ("num_owned_cells", C.c_int32), ("num_edges", C.c_int32), ("num_internal_edges", C.c_int32),
        ("cell_is_owned", c_int32_p), ("cell_local_to_owned", c_int32_p), ("cell_global_ids", c_int64_p),
        ("cell_areas", c_double_p), ("cell_dz_dx", c_double_p), ("cell_dz_dy", c_double_p),
        ("edge_cell_ids", c_int32_p), ("edge_internal_ids", c_int32_p), ("edge_global_ids", c_int64_p),
        ("edge_lengths", c_double_p), ("edge_cn", c_double_p), ("edge_sn", c_double_p), ("cell_zc", c_double_p),
        ("num_vertices", C.c_int32), ("cell_centroids", c_double_p), ("edge_vertex_ids", c_int32_p), ("vertex_points", c_double_p),
        ("edge_is_owned", c_int32_p),
    ]


class RDyHipBoundary(C.Structure):
    _fields_ = [("num_edges", C.c_int32), ("edge_ids", c_int32_p), ("condition_type", C.c_int32)]


class RDyHipCourant(C.Structure):
    _fields_ = [("max_courant_num", C.c_double), ("global_edge_id", C.c_int64), ("global_cell_id", C.c_int64)]


class RDyHipLayoutInfo(C.Structure):
    _fields_ = [("num_owned_cells", C.c_int32), ("num_cells", C.c_int32), ("slots_per_cell", C.c_int32),
                ("num_boundary_edges", C.c_int32), ("num_halo_cells", C.c_int32),
                ("tiled_kernel", C.c_int32), ("num_tiles", C.c_int32), ("num_halo_tiles", C.c_int32),
                ("max_tile_edges", C.c_int32), ("max_tile_halo_cells", C.c_int32), ("num_halo_entries", C.c_int64),
                ("num_edge_records", C.c_int64), ("owned_is_prefix", C.c_int32),
                ("device_bytes", C.c_int64), ("bytes_per_apply", C.c_int64),
                ("second_order_fused", C.c_int32), ("max_tile_ring2_cells", C.c_int32),
                ("persistent_grid", C.c_int32), ("lds_bytes", C.c_int32), ("lds_fixed_layout", C.c_int32)]


class RDyHipHaloFormInfo(C.Structure):
    _fields_ = [("form", C.c_int32), ("source", C.c_int32), ("trial_steps", C.c_int32), ("in_order_ms", C.c_double), ("two_stream_ms", C.c_double)]


# every symbol include/rdyhip.h declares: name -> (restype, argtypes)
_H = C.c_void_p  # RDyHipOperator
SYMBOLS = {
    "rdyhip_last_error": (C.c_char_p, []),
    "rdyhip_version": (C.c_int32, []),
    "rdyhip_create": (C.c_int, [C.POINTER(RDyHipConfig), C.POINTER(RDyHipMesh), C.c_int32, C.POINTER(RDyHipBoundary), C.POINTER(_H)]),
    "rdyhip_destroy": (C.c_int, [C.POINTER(_H)]),
    "rdyhip_apply": (C.c_int, [_H, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_rhs_function": (C.c_int, [_H, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_apply_phase": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_compute_gradients": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p]),
    "rdyhip_set_boundary_values": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "rdyhip_get_boundary_fluxes": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "rdyhip_reset_boundary_fluxes_accum": (C.c_int, [_H]),
    "rdyhip_set_external_source": (C.c_int, [_H, C.c_int32, C.c_int32, c_int32_p, c_double_p]),
    "rdyhip_set_mannings": (C.c_int, [_H, C.c_int32, c_int32_p, c_double_p]),
    "rdyhip_set_boundary_values_on": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p, C.c_void_p]),
    "rdyhip_set_external_source_on": (C.c_int, [_H, C.c_int32, C.c_int32, c_int32_p, c_double_p, C.c_void_p]),
    "rdyhip_set_mannings_on": (C.c_int, [_H, C.c_int32, c_int32_p, c_double_p, C.c_void_p]),
    "rdyhip_refresh_field": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "rdyhip_forcing_fill_source": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_void_p]),
    "rdyhip_forcing_gather_source": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_double, C.c_void_p]),
    "rdyhip_forcing_fill_boundary": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_void_p]),
    "rdyhip_forcing_gather_boundary": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "rdyhip_forcing_nearest_map": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                             C.c_void_p]),
    "rdyhip_field_ptr": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), c_int64_p]),
    "rdyhip_enable_flux_divergence": (C.c_int, [_H, C.c_int32]),
    "rdyhip_reset_diagnostics": (C.c_int, [_H, C.c_void_p]),
    "rdyhip_update_diagnostics": (C.c_int, [_H, C.c_void_p]),
    "rdyhip_get_diagnostics": (C.c_int, [_H, C.POINTER(RDyHipCourant)]),
    "rdyhip_pack_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rdyhip_unpack_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rdyhip_pack_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rdyhip_unpack_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rdyhip_axpy_owned": (C.c_int, [_H, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_euler_step": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_halo_create": (C.c_int, [_H, C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, C.POINTER(C.c_void_p)]),
    "rdyhip_halo_destroy": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rdyhip_halo_overlaps": (C.c_int32, [C.c_void_p]),
    "rdyhip_halo_direct_receive": (C.c_int32, [C.c_void_p]),
    "rdyhip_halo_fuse_pack": (C.c_int, [C.c_void_p, C.c_int32]),
    "rdyhip_halo_pack_fused": (C.c_int32, [C.c_void_p]),
    "rdyhip_halo_form_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(RDyHipHaloFormInfo)]),
    "rdyhip_halo_set_form": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rdyhip_halo_invalidate": (C.c_int, [C.c_void_p]),
    "rdyhip_halo_set_transport": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_halo_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rdyhip_rhs_overlapped": (C.c_int, [_H, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_euler_step_overlapped": (C.c_int, [_H, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rdyhip_comm_init_rank": (C.c_int, [C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]),
    "rdyhip_comm_destroy": (C.c_int, [C.c_void_p]),
    "rdyhip_rccl_version": (C.c_int32, []),
    "rdyhip_comm_count": (C.c_int, [C.c_void_p, c_int32_p]),
    "rdyhip_halo_plan_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_int32_p, c_int64_p, C.POINTER(C.c_void_p)]),
    "rdyhip_halo_plan_requests": (C.c_int, [C.c_void_p, C.POINTER(c_int32_p), C.POINTER(c_int64_p)]),
    "rdyhip_halo_plan_finish": (C.c_int, [C.c_void_p, c_int32_p, c_int64_p, C.c_int32, c_int32_p, c_int64_p]),
    "rdyhip_halo_plan_get": (C.c_int, [C.c_void_p, c_int32_p, C.POINTER(c_int32_p), C.POINTER(c_int32_p), C.POINTER(c_int32_p),
                                       C.POINTER(c_int32_p), C.POINTER(c_int32_p)]),
    "rdyhip_halo_plan_destroy": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rdyhip_hilbert_cell_order": (C.c_int, [C.c_int32, c_double_p, C.c_int32, c_int32_p, c_int32_p]),
    "rdyhip_local_cell_order": (C.c_int, [C.c_int32, c_double_p, C.c_int32, c_int32_p, c_int32_p, c_int64_p, c_int32_p]),
    "rdyhip_copy_owned_rows": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rdyhip_probe_layout": (C.c_int, [C.POINTER(RDyHipConfig), C.POINTER(RDyHipMesh), C.c_int32, C.POINTER(RDyHipBoundary),
                                      C.POINTER(RDyHipLayoutInfo)]),
    "rdyhip_layout_info": (C.c_int, [_H, C.POINTER(RDyHipLayoutInfo)]),
}

# RDyHipTransportFn: int (*)(void *ctx, const double *d_send, double *d_recv, int32_t ncomp, void *stream)
TRANSPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)

_LIB = None
ABI_VERSION = 110     # RDYHIP_VERSION of include/rdyhip.h this binding was written against


class RDyHipError(RuntimeError):
    """A non-zero return code from the C ABI (PETSc-valued error codes)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"rdyhip error {code}: {message}")
        self.code = code
        self.message = message


def load(build_if_missing: bool = False):
    """dlopen rdycore_amd/csrc/librdyhip.so and bind every declared symbol."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname
    # libamdhip64.so.7).  It must be in the process before librdyhip.so is
    # opened, so that our DT_NEEDED libamdhip64.so.7 binds to that same copy;
    # two HIP runtimes in one process cannot both own the device.
    import torch  # noqa: F401
    path = _build.lib_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build_native()
        else:
            raise ImportError(
                f"{path} not found: the HIP extension is required (run `python -m rdycore_amd.build` or "
                "__graft_entry__.build()); there is no CPU fallback for the operator")
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
        except AttributeError:
            if "RDYHIP_LIB" in os.environ:   # A/B timing of an older build of the same ABI: later additions are absent
                missing.append(name)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if "RDYHIP_LIB" in os.environ:
        # another build of the library was asked for: say what it is, so that an ABI mismatch shows up here and not later as an
        # AttributeError or a call through default-int ctypes signatures
        import sys
        have = int(lib.rdyhip_version()) if hasattr(lib, "rdyhip_version") else -1
        if missing or have != ABI_VERSION:
            print(f"rdycore_amd: RDYHIP_LIB={path}: library version {have}, this binding expects {ABI_VERSION}; "
                  f"symbols it lacks: {', '.join(missing) if missing else 'none'}", file=sys.stderr)
    _LIB = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().rdyhip_last_error()
        raise RDyHipError(rc, msg.decode() if msg else "")
