"""Forcing ingestion kept on the device: mirror of RDyApplyForcing
(src/forcing/rdyforcing.c:688-770) for the operator's external water source and
its homogeneous / unstructured Dirichlet values.

The datasets, their data->mesh maps and the regions' cell lists are uploaded to
HBM once; every `apply(time)` then enqueues the fill/gather kernels of
include/rdyhip.h's forcing section on the current stream.  Only the scalar time
lookup of RDyForcingGetCurrentData runs on the host.  File I/O (PETSc binary
Vec files, hourly file names) is the caller's: datasets are handed over as
arrays laid out like the reference's data_vec.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .operator import Operator, _ptr, _stream

MM_PER_HR_2_M_PER_SEC = 1.0 / (1000.0 * 3600.0)   # src/forcing/rdyforcing_dataset.c:303


def current_data(table: np.ndarray, cur_time: float, temporally_interpolate: bool) -> Tuple[int, float]:
    """RDyForcingGetCurrentData (src/forcing/rdyforcing_dataset.c:32-67) on a
    [ndata, 2] (time, value) table: the interval [t_i, t_{i+1}) holding cur_time,
    else the last entry."""
    t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 2)
    ndata = t.shape[0]
    for i in range(ndata - 1):
        time_dn, data_dn = t[i]
        time_up, data_up = t[i + 1]
        if time_dn <= cur_time < time_up:
            if temporally_interpolate:
                return i, float((cur_time - time_dn) / (time_up - time_dn) * (data_up - data_dn) + data_dn)
            return i, float(data_dn)
    return ndata - 1, float(t[ndata - 1, 1])


@dataclasses.dataclass
class HomogeneousDataset:
    """RDyHomogeneousDataset (src/forcing/rdyforcing_dataset.c:69-80, 320-344): a
    (time, value) series, spatially uniform; refilled only when the interval
    changes unless it is interpolated in time."""
    table: np.ndarray
    temporally_interpolate: bool = False
    cur_idx: int = -1
    prev_idx: int = -1

    def advance(self, cur_time: float) -> Optional[float]:
        """the value to write, or None when the reference would leave the array untouched"""
        self.cur_idx, value = current_data(self.table, cur_time, self.temporally_interpolate)
        if self.temporally_interpolate or self.cur_idx != self.prev_idx:
            self.prev_idx = self.cur_idx
            return value
        return None


def _device_ids(ids, device) -> Optional[torch.Tensor]:
    if ids is None:
        return None
    return torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int32), device=device)


def nearest_map(xc, yc, data_xc, data_yc, min_dist0: float, device) -> torch.Tensor:
    """data2mesh_idx by brute force on the GPU (rdyhip_forcing_nearest_map).
    min_dist0 >= 0: RDyForcingCreateRasterDatasetMapping; < 0: RDyForcingCreateUnstructuredDatasetMap."""
    dx = torch.as_tensor(np.ascontiguousarray(xc, dtype=np.float64), device=device)
    dy = torch.as_tensor(np.ascontiguousarray(yc, dtype=np.float64), device=device)
    px = torch.as_tensor(np.ascontiguousarray(data_xc, dtype=np.float64), device=device)
    py = torch.as_tensor(np.ascontiguousarray(data_yc, dtype=np.float64), device=device)
    out = torch.zeros(dx.numel(), dtype=torch.int32, device=device)   # PetscCalloc1, rdyforcing.c:262
    _lib.check(_lib.load().rdyhip_forcing_nearest_map(int(dx.numel()), _ptr(dx), _ptr(dy), int(px.numel()), _ptr(px), _ptr(py),
                                                      float(min_dist0), _ptr(out), _stream()))
    return out


class RasterDataset:
    """RDyRasterDataset (src/forcing/rdyforcing_dataset.c:113-146, rdyforcing.c:242-285):
    data_vec = [ncols, nrows, xlc, ylc, cellsize, values row-major from the top row], mm/h."""

    HEADER_OFFSET = 5

    def __init__(self, data_vec: np.ndarray, mesh_xc, mesh_yc, device, dtime_in_hour: float = 1.0):
        v = np.ascontiguousarray(data_vec, dtype=np.float64).ravel()
        self.ncols, self.nrows = int(v[0]), int(v[1])
        self.xlc, self.ylc, self.cellsize = float(v[2]), float(v[3]), float(v[4])
        if v.size != self.HEADER_OFFSET + self.ncols * self.nrows:
            raise _lib.RDyHipError(83, "raster data_vec length does not match its header")
        self.dtime_in_hour = dtime_in_hour
        self.ndata_file = 1
        self.device = device
        icol = np.arange(self.ncols, dtype=np.float64)
        irow = np.arange(self.nrows, dtype=np.float64)
        # rdyforcing.c:254-260
        xs = self.xlc + icol * self.cellsize + self.cellsize / 2.0
        ys = self.ylc + (self.nrows - 1 - irow) * self.cellsize + self.cellsize / 2.0
        self.data_xc = np.tile(xs, self.nrows)
        self.data_yc = np.repeat(ys, self.ncols)
        self.d_data = torch.as_tensor(v, device=device)
        min_dist0 = (max(self.ncols, self.nrows) + 1) * self.cellsize   # rdyforcing_map.c:115
        self.d_map = nearest_map(mesh_xc, mesh_yc, self.data_xc, self.data_yc, min_dist0, device)

    def needs_next_file(self, cur_time: float) -> bool:
        """rdyforcing_dataset.c:298"""
        return cur_time / 3600.0 >= self.ndata_file * self.dtime_in_hour

    def load_next(self, data_vec: np.ndarray):
        """RDyForcingOpenNextRasterDataset (rdyforcing_dataset.c:166-196): same header, new values"""
        v = np.ascontiguousarray(data_vec, dtype=np.float64).ravel()
        if (int(v[0]), int(v[1]), float(v[2]), float(v[3]), float(v[4])) != (self.ncols, self.nrows, self.xlc, self.ylc, self.cellsize):
            raise _lib.RDyHipError(83, "The header of the previous and new rainfall do not match")
        self.d_data.copy_(torch.as_tensor(v), non_blocking=False)
        self.ndata_file += 1


class UnstructuredDataset:
    """RDyUnstructuredDataset (src/forcing/rdyforcing_dataset.c:201-236): data_vec = [ndata, stride, values...]"""

    OFFSET = 2

    def __init__(self, data_vec: np.ndarray, expected_stride: int, data_xc, data_yc, mesh_xc, mesh_yc, device):
        v = np.ascontiguousarray(data_vec, dtype=np.float64).ravel()
        self.ndata, self.stride = int(v[0]), int(v[1])
        if (v.size - 2) // self.stride != self.ndata or self.stride != expected_stride:
            raise _lib.RDyHipError(83, "unstructured data_vec is inconsistent with its header")
        self.device = device
        self.d_data = torch.as_tensor(v, device=device)
        self.d_map = nearest_map(mesh_xc, mesh_yc, data_xc, data_yc, -1.0, device)


class Forcing:
    """One source dataset and one boundary dataset bound to an Operator, like the
    RDyForcing object's `source` and `boundary` members."""

    def __init__(self, op: Operator):
        self.op = op
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._sources: List[tuple] = []
        self._boundaries: List[tuple] = []

    # -- source/sink (water, component 0: RDySetRegionalWaterSource) ------------
    def add_constant_source(self, owned_cell_ids, rate: float):
        """FORCING_DATASET_CONSTANT"""
        ids = _device_ids(owned_cell_ids, self.device)
        self._sources.append(("constant", ids, self._n(ids), float(rate)))

    def add_homogeneous_source(self, owned_cell_ids, dataset: HomogeneousDataset):
        """FORCING_DATASET_HOMOGENEOUS / one entry of FORCING_DATASET_MULTI_HOMOGENEOUS"""
        ids = _device_ids(owned_cell_ids, self.device)
        self._sources.append(("homogeneous", ids, self._n(ids), dataset))

    def add_raster_source(self, owned_cell_ids, dataset: RasterDataset):
        """FORCING_DATASET_RASTER; the dataset's map has one entry per region cell"""
        ids = _device_ids(owned_cell_ids, self.device)
        assert dataset.d_map.numel() == self._n(ids)
        self._sources.append(("raster", ids, self._n(ids), dataset))

    def add_unstructured_source(self, owned_cell_ids, dataset: UnstructuredDataset):
        """FORCING_DATASET_UNSTRUCTURED (stride 1)"""
        ids = _device_ids(owned_cell_ids, self.device)
        assert dataset.d_map.numel() == self._n(ids) and dataset.stride == 1
        self._sources.append(("unstructured", ids, self._n(ids), dataset))

    # -- Dirichlet boundary values ---------------------------------------------
    def add_homogeneous_boundary(self, boundary: int, dataset: HomogeneousDataset):
        self._boundaries.append(("homogeneous", int(boundary), dataset))

    def add_unstructured_boundary(self, boundary: int, dataset: UnstructuredDataset):
        assert dataset.stride == 3 and dataset.d_map.numel() == self.op.mesh.boundaries[boundary].num_edges
        self._boundaries.append(("unstructured", int(boundary), dataset))

    def _n(self, ids) -> int:
        return self.op.mesh.num_owned_cells if ids is None else int(ids.numel())

    def apply(self, time: float):
        """RDyApplyForcing(rdy, forcing, time): enqueue this step's fills on the current stream."""
        L, h, st = _lib.load(), self.op._h, _stream()
        for kind, ids, n, d in self._sources:
            pid = _ptr(ids) if ids is not None else None
            if kind == "constant":
                _lib.check(L.rdyhip_forcing_fill_source(h, 0, n, pid, d, st))
            elif kind == "homogeneous":
                value = d.advance(time)
                if value is not None:
                    _lib.check(L.rdyhip_forcing_fill_source(h, 0, n, pid, value, st))
            elif kind == "raster":
                _lib.check(L.rdyhip_forcing_gather_source(h, 0, n, pid, _ptr(d.d_data), _ptr(d.d_map), 1, d.HEADER_OFFSET,
                                                          MM_PER_HR_2_M_PER_SEC, st))
            else:
                _lib.check(L.rdyhip_forcing_gather_source(h, 0, n, pid, _ptr(d.d_data), _ptr(d.d_map), d.stride, d.OFFSET, 1.0, st))
        for kind, b, d in self._boundaries:
            ne = self.op.mesh.boundaries[b].num_edges
            if kind == "homogeneous":
                value = d.advance(time)
                if value is not None:
                    _lib.check(L.rdyhip_forcing_fill_boundary(h, b, ne, value, st))
            else:
                _lib.check(L.rdyhip_forcing_gather_boundary(h, b, ne, _ptr(d.d_data), _ptr(d.d_map), d.stride, d.OFFSET, st))
