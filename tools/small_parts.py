"""What one rank of a strong-scaled run of the reference's own meshes costs per step, on ONE GPU: the part an
8-way RCB partition gives a rank, with the real multi-rank step of the C ABI and its exchange looped back through a one-rank RCCL
communicator (the rank's own boundary cells travel to its ghost rows: same launches, same bytes through ncclSend / ncclRecv, no
xGMI hop).  Per part, microseconds per step (HIP events around 300 back-to-back steps, after 100 untimed ones):

  kernel_rhs      rdyhip_rhs_function alone (ONE launch: the figure step / kernel is quoted against)
  kernel_euler    rdyhip_euler_step alone (the fused Euler step, two state arrays ping-pong, dt = 0 keeps the state put)
  r03_rhs         rdyhip_rhs_overlapped as round 3 ran it: pack launch, RCCL, unpack launch, kernel   (RDYHIP_DIRECT_RECV=0)
  rhs_direct      rdyhip_rhs_overlapped with the direct receive: pack launch, RCCL into the ghost rows, kernel
  euler_direct    rdyhip_euler_step_overlapped, direct receive, pack launch:        pack, RCCL, kernel
  euler_fused     the same with rdyhip_halo_fuse_pack: the pack rides on the previous step's kernel -- RCCL, kernel, in the form the
                  halo's own trial chooses (round 5: rdyhip_halo_form_info; `forms` holds what it chose and the two timings)
  euler_fused_in_order / _overlapped   the two forms forced (RDYHIP_OVERLAP=0 / 1): RCCL, kernel in order; or the two-stream form
                  (round 4's third form, signalled, is tools/probes/round4_forms.patch)
  exchange        rdyhip_halo_exchange alone (pack, RCCL, direct receive)

usage (GPU box): python tools/small_parts.py > gpurun_out/small_parts.txt"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from rdycore_amd import _lib
from rdycore_amd import cases as CS

torch.cuda.set_device(0)
lib = _lib.load()
i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
pp = lambda a: a.ctypes.data_as(_lib.c_int32_p)


def timed(fn, k=300, lead=100):
    for _ in range(lead):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / k * 1e6
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 2), round(host, 2)


def self_halo(op, mesh, comm, direct, overlap=None):
    """the halo bench.py --self-exchange builds: this rank's ghost-adjacent owned cells are sent to its own ghost rows"""
    ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
    gset = np.zeros(mesh.num_cells, dtype=bool)
    gset[ghost] = True
    cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
    cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
    sendc = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut])).astype(np.int32)
    n = min(sendc.size, ghost.size)
    sendc, ghost = np.ascontiguousarray(sendc[:n]), np.ascontiguousarray(ghost[:n])
    os.environ["RDYHIP_DIRECT_RECV"] = "1" if direct else "0"
    if overlap is not None:
        os.environ["RDYHIP_OVERLAP"] = str(int(overlap))
    hh = C.c_void_p()
    _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, pp(i32([0])), pp(i32([n])), pp(sendc), pp(i32([n])), pp(ghost), C.byref(hh)))
    os.environ.pop("RDYHIP_DIRECT_RECV")
    os.environ.pop("RDYHIP_OVERLAP", None)
    assert lib.rdyhip_halo_direct_receive(hh) == (1 if direct else 0), "the ghosts of this part are not one run of rows"
    return hh, int(n)


def one(tag, argv, scaling="strong"):
    args = bench.parse(["--no-cpu-baseline"] + argv)
    sav = args.scaling
    args.scaling = scaling
    case = bench.build_case(args, args.emulate_rank, args.emulate_world)
    args.scaling = sav
    mesh = case.mesh
    op = CS.create_operator(case)
    uid = C.create_string_buffer(128)
    _lib.check(lib.rdyhip_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    u2 = u.clone()
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)
    up, u2p, fp, dt = int(u.data_ptr()), int(u2.data_ptr()), int(f.data_ptr()), float(case.dt)
    pair = [up, u2p]

    def pingpong(call):
        def fn():
            call(pair[0], pair[1])
            pair.reverse()
        return fn

    res = {"part": tag, "cells": mesh.num_owned_cells, "ghosts": mesh.num_cells - mesh.num_owned_cells, "tiles": op.layout_info()["num_tiles"]}
    # (second order: the kernel alone takes the ghost cells' gradients as they are -- flag 4 -- since no exchange fills them here)
    so = 4 if args.second_order else 0
    res["kernel_rhs"] = timed(lambda: op.apply_phase(0, True, dt, u, f, reset_diagnostics=True, gradients_ready=bool(so)))
    res["kernel_euler"] = timed(pingpong(lambda a, b: _lib.check(lib.rdyhip_euler_step(op._h, 0, 2 | so, 0.0, a, b, None, st))))
    h0, n = self_halo(op, mesh, comm, direct=False)
    res["halo_cells"] = n
    res["overlapped_form"] = int(lib.rdyhip_halo_overlaps(h0))
    res["r03_rhs"] = timed(lambda: _lib.check(lib.rdyhip_rhs_overlapped(op._h, h0, dt, up, fp, st)))
    res["r03_euler"] = timed(pingpong(lambda a, b: _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h0, 0.0, a, b, None, st))))
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h0)))
    h1, _ = self_halo(op, mesh, comm, direct=True)
    res["rhs_direct"] = timed(lambda: _lib.check(lib.rdyhip_rhs_overlapped(op._h, h1, dt, up, fp, st)))
    res["euler_direct"] = timed(pingpong(lambda a, b: _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h1, 0.0, a, b, None, st))))
    res["exchange"] = timed(lambda: _lib.check(lib.rdyhip_halo_exchange(h1, up, 3, st)))
    _lib.check(lib.rdyhip_halo_fuse_pack(h1, 1))
    res["euler_fused"] = timed(pingpong(lambda a, b: _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h1, 0.0, a, b, None, st))))
    res["forms"] = {}
    for kind, name in ((0, "rhs"), (1, "euler")):
        fi = _lib.RDyHipHaloFormInfo()
        _lib.check(lib.rdyhip_halo_form_info(h1, kind, C.byref(fi)))
        res["forms"][name] = {"form": "two_streams" if fi.form else "in_order", "source": int(fi.source), "in_order_us": round(1e3 * fi.in_order_ms, 2),
                              "two_stream_us": round(1e3 * fi.two_stream_ms, 2)}
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h1)))
    # the fused-pack Euler step in its two forms, forced: in order (transfer, one launch) and two streams (transfer beside the
    # interior launch, the ghost-adjacent tiles in a second launch on the exchange stream)
    for name, ov in (("euler_fused_in_order", 0), ("euler_fused_overlapped", 1)):
        hx, _ = self_halo(op, mesh, comm, direct=True, overlap=ov)
        _lib.check(lib.rdyhip_halo_fuse_pack(hx, 1))
        res[name] = timed(pingpong(lambda a, b: _lib.check(lib.rdyhip_euler_step_overlapped(op._h, hx, 0.0, a, b, None, st))))
        _lib.check(lib.rdyhip_halo_destroy(C.byref(hx)))
    _lib.check(lib.rdyhip_comm_destroy(comm))
    op.destroy()
    k = res["kernel_rhs"][0]
    res["step_over_kernel"] = {key: round(res[key][0] / (res["kernel_euler"][0] if key.startswith("euler") or key == "r03_euler" else k), 3)
                               for key in ("r03_rhs", "r03_euler", "rhs_direct", "euler_direct", "euler_fused", "euler_fused_in_order", "euler_fused_overlapped")
                               if key in res}
    res["columns"] = "[us per step on the GPU, us per step of host enqueue time]"
    print(json.dumps(res), flush=True)
    del u, u2, f


if __name__ == "__main__":
    parts = [
        # the reference's 2.88 M-quad dam break (docs/user/example-cases/dam-break/index.md:24-26) and the Houston1km mesh refined five
        # times (2.81 M triangles, the size of its Turning_30m Harvey mesh), each cut 8 ways: 0.36 M / 0.35 M cells per rank
        ("dambreak_2560x1280 rank0/8", ["--workload", "dambreak_quads", "--nx", "2560", "--ny", "1280", "--emulate-world", "8", "--emulate-rank", "0"]),
        ("dambreak_2560x1280 rank3/8", ["--workload", "dambreak_quads", "--nx", "2560", "--ny", "1280", "--emulate-world", "8", "--emulate-rank", "3"]),
        ("houston_L5 rank0/8", ["--workload", "houston_refined", "--levels", "5", "--emulate-world", "8", "--emulate-rank", "0"]),
        ("houston_L5 rank3/8", ["--workload", "houston_refined", "--levels", "5", "--emulate-world", "8", "--emulate-rank", "3"]),
        # 1 M-cell and 2.9 M-cell parts: the 8-way cut of an 8 M-quad / the 23 M-quad... kept to what the reference publishes:
        # the 5120 x 2560 dam break (11.5 M quads) cut 8 ways = 1.44 M cells per rank, cut 4 ways = 2.88 M
        ("dambreak_5120x2560 rank3/8", ["--workload", "dambreak_quads", "--emulate-world", "8", "--emulate-rank", "3"]),
        ("dambreak_5120x2560 rank1/4", ["--workload", "dambreak_quads", "--emulate-world", "4", "--emulate-rank", "1"]),
        ("houston_L6 rank3/8", ["--workload", "houston_refined", "--levels", "6", "--emulate-world", "8", "--emulate-rank", "3"]),
        ("houston_L6 rank1/4", ["--workload", "houston_refined", "--levels", "6", "--emulate-world", "4", "--emulate-rank", "1"]),
        # second order on two of the small parts (the fused pack of the state exchange rides on the MUSCL Euler-step kernel too)
        ("second_order dambreak_2560x1280 rank0/8", ["--workload", "dambreak_quads", "--nx", "2560", "--ny", "1280", "--emulate-world", "8", "--emulate-rank", "0", "--second-order"]),
        ("second_order houston_L5 rank3/8", ["--workload", "houston_refined", "--levels", "5", "--emulate-world", "8", "--emulate-rank", "3", "--second-order"]),
        # the weak-scaling benchmark's rank: an inner strip of 10 M cells with ghost columns on both sides (bench.py --gpus N)
        ("c3_strip_10M rank1/3", ["--emulate-world", "3", "--emulate-rank", "1"], "weak"),
    ]
    sel = sys.argv[1:]
    for part in parts:
        if not sel or any(s in part[0] for s in sel):
            one(*part)
