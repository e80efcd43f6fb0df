"""Where the multi-rank step's time goes, on ONE GPU: a strip rank with ghost cells on both sides, its exchange looped
back to itself through a one-rank RCCL communicator (bench.py --self-exchange).  Times, per step: the whole
rdyhip_rhs_overlapped; the two compute phases alone; the exchange alone (pack + RCCL + unpack); pack + unpack without
RCCL; the host's enqueue time.   usage (GPU box): python tools/step_breakdown.py [nx ny]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rdycore_amd import _lib
from rdycore_amd import cases as CS
from rdycore_amd import mesh as M

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2500, 2000)
torch.cuda.set_device(0)
lib = _lib.load()
K = 2 * np.pi / 200.0
mesh = M.strip_partition_tri_mesh(nx, ny, 1, 3, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled")
case = CS.friction_slope_case(mesh, 3.0 * nx, 1.0 * ny, dt=1e-3, K=K)
op = CS.create_operator(case)
ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
gset = np.zeros(mesh.num_cells, dtype=bool)
gset[ghost] = True
cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
sendc = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut])).astype(np.int32)
n = min(sendc.size, ghost.size)
sendc, ghost = np.ascontiguousarray(sendc[:n]), np.ascontiguousarray(ghost[:n])
uid = C.create_string_buffer(128)
_lib.check(lib.rdyhip_comm_unique_id(uid))
comm = C.c_void_p()
_lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
pp = lambda a: a.ctypes.data_as(_lib.c_int32_p)
hh = C.c_void_p()
_lib.check(lib.rdyhip_halo_create(op._h, comm, 1, pp(i32([0])), pp(i32([n])), pp(sendc), pp(i32([n])), pp(ghost), C.byref(hh)))
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
st = int(torch.cuda.current_stream().cuda_stream)
d_send = torch.as_tensor(sendc, device="cuda")
d_recv = torch.as_tensor(ghost, device="cuda")
buf = torch.empty((n, 3), dtype=torch.float64, device="cuda")


def timed(fn, k=200, lead=60):
    for _ in range(lead):
        fn()
    torch.cuda.synchronize()
    for _ in range(lead):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / k * 1e3
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k, 4), round(host, 4)


up, fp = int(u.data_ptr()), int(f.data_ptr())
res = {"cells": mesh.num_owned_cells, "ghost_cells": int(n), "info": {k: op.layout_info()[k] for k in ("num_tiles", "num_halo_tiles", "persistent_grid")}}
res["single_launch_rhs"] = timed(lambda: op.rhs_function(case.dt, u, f))
res["overlapped_step"] = timed(lambda: _lib.check(lib.rdyhip_rhs_overlapped(op._h, hh, float(case.dt), up, fp, st)))
res["halo_overlaps"] = int(lib.rdyhip_halo_overlaps(hh))
res["in_order_exchange_then_single_launch"] = timed(lambda: (_lib.check(lib.rdyhip_halo_exchange(hh, up, 3, st)), op.rhs_function(case.dt, u, f)))
res["phases_only_interior_then_halo"] = timed(lambda: (op.apply_phase(1, True, case.dt, u, f, reset_diagnostics=True), op.apply_phase(2, True, case.dt, u, f)))
res["interior_phase_only"] = timed(lambda: op.apply_phase(1, True, case.dt, u, f, reset_diagnostics=True))
res["halo_phase_only"] = timed(lambda: op.apply_phase(2, True, case.dt, u, f))
res["exchange_only_pack_rccl_unpack"] = timed(lambda: _lib.check(lib.rdyhip_halo_exchange(hh, up, 3, st)))
res["pack_unpack_only"] = timed(lambda: (_lib.check(lib.rdyhip_pack_cells(up, int(d_send.data_ptr()), int(n), int(buf.data_ptr()), st)),
                                         _lib.check(lib.rdyhip_unpack_cells(up, int(d_recv.data_ptr()), int(n), int(buf.data_ptr()), st))))
# the same step driven from Python with torch streams (round 1's halo.py): fork / join through torch events
s2 = torch.cuda.Stream(priority=-1)
main = torch.cuda.current_stream()


def torch_step():
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        _lib.check(lib.rdyhip_halo_exchange(hh, up, 3, int(s2.cuda_stream)))
    op.apply_phase(1, True, case.dt, u, f, reset_diagnostics=True)
    main.wait_stream(s2)
    op.apply_phase(2, True, case.dt, u, f)


res["overlapped_step_torch_streams"] = timed(torch_step)
res["columns"] = "[ms per step on the GPU (HIP events around 200 steps), ms per step of host enqueue time]"
print(json.dumps(res))
_lib.check(lib.rdyhip_halo_destroy(C.byref(hh)))
_lib.check(lib.rdyhip_comm_destroy(comm))
op.destroy()
