#!/usr/bin/env python3
"""Benchmark of the SWE right-hand-side evaluation (BASELINE.json metric:
M cell-updates/s + achieved HBM GB/s).

  python bench.py --gpus N --steps K --warmup W

A "step" is one OperatorRHSFunction (src/rdysetup.c:1120-1172): for N > 1 the
ghost update of u_local, then the fused Roe-flux + source kernel over every
owned cell, including the Courant-number reduction.  Workload (N = 1):
BASELINE.json configs[2] -- the 10 M-cell friction + bed-slope mesh the
north-star's >= 40 % HBM target is quoted on (SURVEY.md section 8.d "C3");
for N > 1 each rank owns one such block (strips along x, configs[3], weak
scaling).  State is resident in HBM before the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

ALG_BYTES_PER_CELL = 176.0     # algorithmic bytes per cell-update
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--nx", type=int, default=2500, help="squares per rank along x")
    p.add_argument("--ny", type=int, default=2000, help="squares along y")
    p.add_argument("--order", default="tiled", choices=["rowmajor", "tiled", "hilbert"],
                   help="cell numbering of the synthetic mesh: generator order, 16x16-square blocks, or squares along a Hilbert curve")
    p.add_argument("--source", default="semi_implicit", choices=["semi_implicit", "implicit_xq2018"])
    p.add_argument("--workload", default="c3", choices=["c3", "c2"],
                   help="c3: friction + bed slope + all BC types (default; use --nx 2500 --ny 2000); "
                        "c2: flat-bed dam break, all reflecting (BASELINE configs[1]: --nx 1000 --ny 500)")
    p.add_argument("--hr", action="store_true", help="hydrostatic-reconstruction variant of the operator (SURVEY 8.f row 2)")
    p.add_argument("--second-order", action="store_true",
                   help="MUSCL second-order variant (SURVEY 8.f row 4): gradient kernel + reconstructing flux kernel per RHS")
    p.add_argument("--limiter", default="minmod", choices=["minmod", "none", "van_leer"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-host-cores CPU figure (the 1-core cpu_baseline stays)")
    p.add_argument("--kernel", default=None, choices=["tiled", "cell"], help="kernel variant (default: library default = tiled)")
    p.add_argument("--cpu-sample", default="1000x500", help="nx x ny of the CPU-baseline sample mesh")
    return p.parse_args()


LIMITERS = {"minmod": 0, "none": 1, "van_leer": 2}
# second order: 176 B + least-squares coefficients (3 slots x 16) + centroid->midpoint displacements (1.5 edges x 32);
# the split form (RDYHIP_MUSCL=split) also writes and reads the gradient array (2 x 48) and reads the state twice (24)
ALG_BYTES_PER_CELL_SECOND_ORDER = 176.0 + 48.0 + 48.0
ALG_BYTES_PER_CELL_SECOND_ORDER_SPLIT = ALG_BYTES_PER_CELL_SECOND_ORDER + 96.0 + 24.0


def build_case(nx, ny, rank, world, order, source, workload="c3", hr=False, second_order=False, limiter="minmod"):
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    from rdycore_amd.operator import SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT
    K = 2 * np.pi / 200.0
    src = SOURCE_SEMI_IMPLICIT if source == "semi_implicit" else SOURCE_IMPLICIT_XQ2018
    zf = CS.mms_bathymetry(K=K) if workload == "c3" else None
    if world == 1:
        mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=zf, order=order, project_2d=hr)
    else:
        mesh = M.strip_partition_tri_mesh(nx, ny, rank, world, 1.0, zfunc=zf, order=order)
    if workload == "c2":
        case = CS.dam_break_case(mesh, nx * world * 1.0, dt=1e-3, source_method=src)
    else:
        case = CS.friction_slope_case(mesh, nx * world * 1.0, ny * 1.0, dt=1e-3, source_method=src, K=K)
    if hr:
        from rdycore_amd.operator import WELL_BALANCING_HR
        case.config.well_balancing = WELL_BALANCING_HR
    case.config.second_order = bool(second_order)
    case.config.limiter = LIMITERS[limiter]
    return case


def cpu_baseline(sample: str, source: str, workload: str = "c3", hr: bool = False, second_order: bool = False, limiter: str = "minmod"):
    """The CPU oracle (a plain-C restatement of the reference's PETSc path, one
    core) timed on a bounded sample of the same workload."""
    from oracle import oracle as O  # test infrastructure; used here only as the timed CPU baseline
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from_case
    nx, ny = map(int, sample.split("x"))
    case = build_case(nx, ny, 0, 1, "rowmajor", source, workload, hr, second_order, limiter)
    orc = oracle_from_case(case)
    f = np.zeros((case.mesh.num_owned_cells, 3))
    orc.apply(case.dt, case.u_local, f)  # warm
    times = []
    t_end = time.time() + 12.0
    while len(times) < 3 or (time.time() < t_end and len(times) < 10):
        f[:] = 0.0
        t0 = time.perf_counter()
        orc.apply(case.dt, case.u_local, f)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    nc = case.mesh.num_owned_cells
    return {"value": round(nc / med / 1e6, 3), "unit": "M cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} RHS evaluations of the same workload on a {nx}x{ny}x2 = {nc}-cell mesh, "
                      f"oracle/swe_oracle.c (gcc -O2, 1 thread), median {med * 1e3:.1f} ms/RHS"}


def _cpu_strip_worker(args):
    """One host core: the oracle on one strip (with its ghost cells) of the sample mesh."""
    nx, ny, rank, world, source, workload, hr, reps, second_order, limiter = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from_case
    case = build_case(nx, ny, rank, world, "rowmajor", source, workload, hr, second_order, limiter)
    orc = oracle_from_case(case)
    f = np.zeros((case.mesh.num_owned_cells, 3))
    orc.apply(case.dt, case.u_local, f)
    ts = []
    for _ in range(reps):
        f[:] = 0.0
        t0 = time.perf_counter()
        orc.apply(case.dt, case.u_local, f)
        ts.append(time.perf_counter() - t0)
    return case.mesh.num_owned_cells, float(np.median(ts))


def cpu_baseline_all_cores(sample: str, source: str, workload: str, hr: bool, second_order: bool = False, limiter: str = "minmod"):
    """no MPI launcher exists here, so P independent oracle processes run on P strip
    partitions of the sample mesh (ghost cells present, not exchanged) -- an upper bound on what the
    MPI-parallel reference could do on these host cores."""
    import multiprocessing as mp
    nx, ny = map(int, sample.split("x"))
    cores = len(os.sched_getaffinity(0))
    p = max(1, min(cores, 32))
    while nx % p:
        p -= 1
    with mp.get_context("spawn").Pool(p) as pool:
        res = pool.map(_cpu_strip_worker, [(nx // p, ny, r, p, source, workload, hr, 5, second_order, limiter) for r in range(p)])
    cells = sum(r[0] for r in res)
    tmax = max(r[1] for r in res)
    return {"value": round(cells / tmax / 1e6, 2), "unit": "M cell-updates/s", "cores": p, "kind": "port",
            "sample": f"{p} independent oracle processes, one x-strip each of the {nx}x{ny}x2-cell sample mesh, no halo exchange "
                      f"(upper bound on an MPI run), slowest strip {tmax * 1e3:.1f} ms/RHS"}


def load_traffic(workload_key: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes, if present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        return t.get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    if args.kernel:
        os.environ["RDYHIP_KERNEL"] = args.kernel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "gloo": rehearsal of several ranks on one GPU (host-staged halo)
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from rdycore_amd import cases as CS
    from rdycore_amd.halo import HaloExchange

    t0 = time.time()
    case = build_case(args.nx, args.ny, rank, world, args.order, args.source, args.workload, args.hr, args.second_order, args.limiter)
    mesh = case.mesh
    op = CS.create_operator(case)
    halo = HaloExchange(mesh, dev) if world > 1 else None
    u = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device=dev)
    setup_s = time.time() - t0
    n_owned = mesh.num_owned_cells

    def step():
        if halo is not None:
            halo.rhs_overlapped(op, case.dt, u, f)
        else:
            op.rhs_function(case.dt, u, f)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        rdev = dev if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_owned], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_cells = int(tot.item())
    else:
        total_cells = n_owned

    # ---- dominant kernel: average duration with HIP events on the launch stream
    # (one rhs call = reset (1 thread) + swe_rhs_kernel + Courant finalize (1 block))
    k_iters = max(10, min(args.steps, 50))
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(k_iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(k_iters)]
    def kernel_only():
        if args.second_order and halo is not None:
            # the ghost gradients of the last timed step are still in place: the flux launch alone
            op.apply_phase(0, True, case.dt, u, f, reset_diagnostics=True, gradients_ready=True)
        else:
            op.rhs_function(case.dt, u, f)

    for i in range(k_iters):
        starts[i].record()
        kernel_only()
        ends[i].record()
    torch.cuda.synchronize()
    kern_all = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    kern_isolated_ms = float(np.mean(kern_all))
    # the figure the roofline uses: HIP events around the whole timed region / steps (one RHS = one launch, so this is the
    # launch-to-launch period including the dispatch gap); with several ranks the region also holds the exchange, so the
    # individually bracketed launches are used there
    kern_ms = ev0.elapsed_time(ev1) / args.steps if world == 1 else kern_isolated_ms

    # multi-GPU: the ghost update on its own (pack, P2P over RCCL, unpack), not overlapped
    halo_ms = None
    if halo is not None:
        torch.cuda.synchronize()
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0.record()
        for _ in range(k_iters):
            halo.exchange(u)
        h1.record()
        torch.cuda.synchronize()
        halo_ms = h0.elapsed_time(h1) / k_iters

    # ---- extra (not the metric): one whole forward-Euler step, the update fused into the RHS kernel's stores
    # (rdyhip_euler_step, F never written) against the RHS + axpy pair -- SURVEY.md 8.f row 1
    u2 = torch.empty_like(u)
    u3 = u.clone()

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def fused_step():
        if halo is not None:
            halo.step_overlapped(op, case.dt, u, u2)
        else:
            op.euler_step(case.dt, u, u2)

    def pair_step():
        if halo is not None:
            halo.rhs_overlapped(op, case.dt, u, f)
        else:
            op.rhs_function(case.dt, u, f)
        op.axpy_owned(0.0, f, u3)      # dt = 0: same traffic, the scratch state stays put

    ef, ep = timed(fused_step, k_iters), timed(pair_step, k_iters)
    euler = {"fused_ms_per_step": round(ef, 5), "rhs_plus_axpy_ms_per_step": round(ep, 5),
             "fused_steps_per_s": round(1e3 / ef, 1)}
    del u2, u3
    step()   # leave F and the diagnostics of a plain RHS evaluation behind for the sanity checks below

    # sanity: the result is finite and the Courant diagnostic is alive
    op.update_diagnostics()
    courant = op.get_diagnostics().max_courant_num
    finite = bool(torch.isfinite(f).all().item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells / (elapsed / args.steps) / 1e6
        achieved = n_owned * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9
        info = op.layout_info()
        if args.workload == "c3":
            workload = (f"C3: synthetic {args.nx * world}x{args.ny}x2 = {total_cells}-cell triangle mesh "
                        f"({n_owned} cells/GPU), MMS-style state over sinusoidal bathymetry, Manning field, rain source, "
                        f"dry disc, Dirichlet + critical-outflow + reflecting boundaries, {args.source} friction, dt=1e-3")
        else:
            workload = (f"C2: synthetic {args.nx * world}x{args.ny}x2 = {total_cells}-cell triangle mesh ({n_owned} cells/GPU), "
                        f"flat-bed dam break h = 10 / 5 with perturbed momenta, Manning 0.015, reflecting walls, {args.source} friction, dt=1e-3")
        out = {
            "metric": "M cell-updates/s (SWE RHS eval)",
            "value": round(value, 1),
            "unit": "M cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "cells_per_gpu": n_owned, "cell_order": args.order,
                       "partition": "single" if world == 1 else f"strips_x{world}",
                       "well_balancing": "hydrostatic_reconstruction" if args.hr else "none",
                       "spatial_order": ("second (MUSCL, %s limiter)" % args.limiter) if args.second_order else "first",
                       "halo_bytes_per_rank": halo.bytes_sent_per_exchange if halo else 0,
                       "halo_exchange_alone_ms": round(halo_ms, 5) if halo_ms is not None else None,
                       "setup_seconds": round(setup_s, 1), "max_courant": courant, "finite": finite},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": load_traffic(f"{args.nx}x{args.ny}_{args.order}_{args.source}") if args.workload == "c3" else None,
                         "kernel": (("swe_rhs_muscl_fused_kernel<3,%d>" if info["second_order_fused"] else
                                     "muscl_gradient_kernel<3> + swe_rhs_muscl_kernel<3,%d>") % (0 if args.source == "semi_implicit" else 1))
                         if args.second_order else
                         "%s<3,%d>" % ("swe_rhs_tiled_kernel" if info["tiled_kernel"] else "swe_rhs_kernel",
                                       0 if args.source == "semi_implicit" else 1),
                         "tile_edge_records_per_cell": round(info["num_edge_records"] / max(n_owned, 1), 4),
                         "kernel_avg_ms": round(kern_ms, 5),
                         "kernel_isolated_avg_ms": round(kern_isolated_ms, 5), "kernel_isolated_median_ms": round(float(np.median(kern_all)), 5),
                         "kernel_isolated_min_ms": round(float(np.min(kern_all)), 5),
                         "algorithmic_bytes_per_launch": int(n_owned * ALG_BYTES_PER_CELL),
                         "layout_bytes_per_launch": int(info["bytes_per_apply"])},
        }
        out["euler_step"] = euler
        if args.second_order:
            # the 176-B figure above keeps variants comparable; the second-order path's own model:
            b2 = ALG_BYTES_PER_CELL_SECOND_ORDER if info["second_order_fused"] else ALG_BYTES_PER_CELL_SECOND_ORDER_SPLIT
            a2 = n_owned * b2 / (kern_ms * 1e-3) / 1e9
            out["roofline"]["second_order_model"] = {"bytes_per_cell_update": b2, "achieved": round(a2, 1),
                                                     "frac": round(a2 / HBM_PEAK_GBPS, 4)}
            out["roofline"]["traffic"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.source, args.workload, args.hr, args.second_order, args.limiter)
            if not args.no_cpu_all_cores:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.cpu_sample, args.source, args.workload, args.hr, args.second_order,
                                                                           args.limiter)
                except Exception as exc:  # a reported extra, never a reason to lose the bench line
                    out["cpu_baseline_all_cores"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    op.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
