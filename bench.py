#!/usr/bin/env python3
"""Benchmark of the SWE right-hand-side evaluation (BASELINE.json metric:
M cell-updates/s + achieved HBM GB/s).

  python bench.py --gpus N --steps K --warmup W

A "step" is one OperatorRHSFunction (src/rdysetup.c:1120-1172): for N > 1 the
ghost update of u_local, then the fused Roe-flux + source kernel over every
owned cell, including the Courant-number reduction.  Workloads:

  c3 (default)     BASELINE.json configs[2]: the 10 M-cell friction + bed-slope triangle mesh the
                   north-star's >= 40 % HBM target is quoted on; N > 1: one such block per rank
                   (strips along x, configs[3], weak scaling) or --scaling strong (RCB parts of one mesh)
  c2               configs[1]: flat-bed dam break on triangles (--nx 1000 --ny 500 = 1 M cells)
  dambreak_quads   the reference's own published benchmark problem: 5120 x 2560 quads minus the dam =
                   11,534,336 cells, h = 10 / 5 m, n = 0.015, dt = 1.5625e-5 s, reflecting walls
                   (docs/user/example-cases/dam-break/index.md:10-13, inputdeck_5120x2560.yaml); strong scaling
  c5               configs[4] stand-in: nx x ny x 2 triangles over a rough DEM, ~40 % dry, rain, critical-outflow
                   segment, hydrostatic reconstruction; strong scaling over RCB parts (default 5000 x 5000 for
                   N = 8; --emulate-world / --emulate-rank time one rank's part on one GPU)

`--gpus N` with N > 1 started as a plain command launches its own N ranks (one process per GPU, fresh
children, RANK / WORLD_SIZE / MASTER_* set; rdycore_amd/launch.py) and relays rank 0's line; under
torch.distributed.run (WORLD_SIZE already set) it is simply one of the ranks.  State is resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL = 176.0     # algorithmic bytes per cell-update (triangles: 1.5 edges per cell)
ALG_BYTES_PER_CELL_QUADS = 192.0   # the same count with 2 edges per cell: 80 + 32 x 2 + 48
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
LIMITERS = {"minmod": 0, "none": 1, "van_leer": 2}
# second order: 176 B + cell centroids (16) + one edge midpoint per edge (1.5 x 16) -- the least-squares coefficients
# (3 x 16 B per cell in the reference) and the centroid->midpoint displacements (1.5 x 32 B) are formed on the chip from
# them
ALG_BYTES_PER_CELL_SECOND_ORDER = 176.0 + 16.0 + 24.0
CPU_FULL_MESH_MAX_CELLS = 12_000_000   # cpu_baseline runs on the benchmark mesh itself up to this size (C3: 10 M cells, ~1 s per RHS on one core)
KERNEL_SOURCES = ["swe_kernels.h", "swe_device.h", "muscl_kernels.h"]
HOUSTON_DATA = os.path.join(ROOT, "tests", "golden", "houston")   # the reference's Houston1km fixtures (data files, in the repo)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="c3", choices=["c3", "c2", "dambreak_quads", "c5", "houston_refined", "delaunay"])
    p.add_argument("--levels", type=int, default=6, help="houston_refined: refinement levels of the 2 746-triangle Houston1km mesh (6: 11.2 M cells)")
    p.add_argument("--nx", type=int, default=None, help="squares along x (weak scaling: per rank; strong: of the whole mesh)")
    p.add_argument("--ny", type=int, default=None, help="squares along y")
    p.add_argument("--scaling", default=None, choices=["weak", "strong"],
                   help="weak: per-rank work fixed (default for c3 / c2); strong: one mesh cut into N parts by RCB (default for dambreak_quads / c5)")
    p.add_argument("--order", default=None, choices=["rowmajor", "tiled", "hilbert", "natural", "random"],
                   help="cell numbering of the mesh: generator order, 16x16-square blocks (default of the structured workloads), cells along "
                        "a Hilbert curve (default of the unstructured ones); unstructured only: natural = the order refinement / the "
                        "triangulator leaves, random = a seeded permutation")
    p.add_argument("--quad-block", default=None, help="dambreak_quads, --order tiled: the block of squares numbered together, e.g. 16x16 (default 16x15 = one tile of the operator)")
    p.add_argument("--moving-state", action="store_true",
                   help="dambreak_quads: a smooth velocity field and a tilted surface instead of the benchmark's two flat pools at rest (in "
                        "which every edge of a kind reaches the same Courant number: the diagnostic's tie path runs in every tile)")
    p.add_argument("--source", default="semi_implicit", choices=["semi_implicit", "implicit_xq2018"])
    p.add_argument("--hr", action="store_true", help="hydrostatic-reconstruction variant of the operator (SURVEY 8.f row 2)")
    p.add_argument("--second-order", action="store_true", help="MUSCL second-order variant (SURVEY 8.f row 4)")
    p.add_argument("--limiter", default="minmod", choices=sorted(LIMITERS))
    p.add_argument("--emulate-world", type=int, default=0, help="with --gpus 1: time the part one rank of an N-rank strong-scaling run would own")
    p.add_argument("--emulate-rank", type=int, default=0)
    p.add_argument("--self-exchange", action="store_true",
                   help="with --gpus 1 --emulate-world W: run the multi-rank step (rdyhip_rhs_overlapped: pack, RCCL send/recv, unpack on the "
                        "exchange stream, interior tiles meanwhile, halo tiles after) with a one-rank RCCL communicator whose only peer is "
                        "this rank -- the ghost rows receive this rank's own boundary cells.  Same launches, same bytes through RCCL, no "
                        "xGMI hop: what the N > 1 step costs on the GPU apart from the link")
    p.add_argument("--condition-seconds", type=float, default=1.0,
                   help="untimed RHS launches for this long before --warmup (brings the device out of the idle clock state the host-only "
                        "setup leaves it in; disclosed in config.conditioning)")
    p.add_argument("--halo", default=None, choices=["torch", "c"],
                   help="N > 1: who drives the exchange -- torch.distributed P2P from Python, or the C ABI's RCCL path (default: c with nccl)")
    p.add_argument("--inject-fault", default="none", choices=["none", "exchange"],
                   help="test hook of the N > 1 self-check: 'exchange' makes the last rank expect a wrong ghost value, so that the run must end "
                        "with the one-line {\"error\": ...} and a non-zero exit code (tests/test_gpu_multirank.py)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-host-cores CPU figure (the 1-core cpu_baseline stays)")
    p.add_argument("--no-order-study", action="store_true", help="skip the row-major / Hilbert numbering figures (c3, N = 1 only)")
    p.add_argument("--kernel", default=None, choices=["tiled", "cell"], help="kernel variant (default: library default = tiled)")
    p.add_argument("--cpu-sample", default=None, help="nx x ny of the CPU-baseline sample mesh")
    p.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 self-launch: give up after this many seconds")
    p.add_argument("--watchdog-seconds", type=float, default=1200.0,
                   help="every rank dumps its Python stacks and exits non-zero if it is still running after this long (a hung "
                        "collective then becomes a reported failure instead of a silent hang); 0: off")
    a = p.parse_args(argv)
    defaults = {"c3": (2500, 2000), "c2": (1000, 500), "dambreak_quads": (5120, 2560), "c5": (5000, 5000), "houston_refined": (0, 0),
                "delaunay": (1210, 1210)}
    unstructured = a.workload in ("houston_refined", "delaunay")
    if a.order is None:
        a.order = "hilbert" if unstructured else "tiled"
    if a.order not in (("hilbert", "natural", "random") if unstructured else ("rowmajor", "tiled", "hilbert")):
        p.error(f"--order {a.order} does not apply to --workload {a.workload}")
    if a.nx is None:
        a.nx = defaults[a.workload][0]
    if a.ny is None:
        a.ny = defaults[a.workload][1]
    if a.workload == "houston_refined":
        a.nx, a.ny = a.levels, 0          # build_case's nx is the number of refinement levels here
    if a.workload == "delaunay":
        a.ny = a.nx
    if a.scaling is None:
        a.scaling = "strong" if a.workload in ("dambreak_quads", "c5", "houston_refined", "delaunay") else "weak"
    if a.workload in ("c5", "delaunay"):
        a.hr = True
    if a.cpu_sample is None:
        a.cpu_sample = {"c3": "1000x500", "c2": "1000x500", "dambreak_quads": "1280x640", "c5": "700x700", "houston_refined": "4x0",
                        "delaunay": "500x500"}[a.workload]
    return a


def build_case(args, rank, world, nx=None, ny=None, order=None):
    """The Case (mesh + state + operator data) of `rank` in a `world`-rank run of the chosen workload."""
    import numpy as np
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    from rdycore_amd import partition as P
    from rdycore_amd.operator import SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT, WELL_BALANCING_HR
    nx = args.nx if nx is None else nx
    ny = args.ny if ny is None else ny
    order = args.order if order is None else order
    src = SOURCE_SEMI_IMPLICIT if args.source == "semi_implicit" else SOURCE_IMPLICIT_XQ2018
    wl = args.workload
    strong = args.scaling == "strong"
    if wl in ("c3", "c2"):
        K = 2 * np.pi / 200.0
        zf = CS.mms_bathymetry(K=K) if wl == "c3" else None
        nxg = nx if (strong or world == 1) else nx * world
        if world == 1:
            mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=zf, order=order, project_2d=args.hr)
        elif strong:
            mesh = P.partitioned_structured_mesh("tri", nx, ny, 1.0, rank, world, zfunc=zf, order=order,
                                                 boundary_classifier=M.box_side_boundaries(0.0, nx * 1.0, 0.0, ny * 1.0), project_2d=args.hr)
        else:
            mesh = M.strip_partition_tri_mesh(nx, ny, rank, world, 1.0, zfunc=zf, order=order)
        if wl == "c2":
            case = CS.dam_break_case(mesh, nxg * 1.0, dt=1e-3, source_method=src)
        else:
            case = CS.friction_slope_case(mesh, nxg * 1.0, ny * 1.0, dt=1e-3, source_method=src, K=K)
    elif wl == "dambreak_quads":
        nxg = nx if (strong or world == 1) else nx * world
        mesh = CS.dam_break_quads_mesh(nxg, ny, rank, world, order=order, tile=tuple(map(int, args.quad_block.split("x"))) if args.quad_block else None)
        case = CS.dam_break_quads_case(mesh)
        case.config.source_method = src
        if args.moving_state:
            xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
            case.u_local[:, 0] *= 1.0 + 1e-3 * xc / 10.0 + 2e-3 * yc / 5.0
            case.u_local[:, 1] = 0.3 * case.u_local[:, 0] * np.sin(1.7 * xc + 0.9 * yc)
            case.u_local[:, 2] = 0.2 * case.u_local[:, 0] * np.cos(1.1 * xc - 2.3 * yc)
    elif wl == "houston_refined":
        # nx = refinement levels here
        case = CS.houston_refined_case(HOUSTON_DATA, nx, order if order in ("hilbert", "natural", "random") else "hilbert", hr=args.hr,
                                       rank=rank, world=world)
        case.config.source_method = src
    elif wl == "delaunay":
        mesh = CS.delaunay_mesh(nx, rank, world, order=order if order in ("hilbert", "natural", "random") else "hilbert")
        case = CS.c5_case(mesh, nx * 1.0, nx * 1.0)
        case.config.source_method = src
    else:
        nxg = nx if (strong or world == 1) else nx * world
        mesh = CS.c5_mesh(nxg, ny, rank, world, order=order)
        case = CS.c5_case(mesh, nxg * 1.0, ny * 1.0)
        case.config.source_method = src
    if args.hr:
        case.config.well_balancing = WELL_BALANCING_HR
    case.config.second_order = bool(args.second_order)
    case.config.limiter = LIMITERS[args.limiter]
    return case


def _time_oracle(orc, case, min_reps, max_reps, budget_s):
    import numpy as np
    f = np.zeros((case.mesh.num_owned_cells, 3))
    orc.apply(case.dt, case.u_local, f)  # warm
    times = []
    t_end = time.time() + budget_s
    while len(times) < min_reps or (time.time() < t_end and len(times) < max_reps):
        f[:] = 0.0
        t0 = time.perf_counter()
        orc.apply(case.dt, case.u_local, f)
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times)


def cpu_baseline(args, case=None):
    """The CPU oracle (a plain-C restatement of the reference's PETSc path, ApplyOperator of src/operator.c:656-672, one core)
    timed on `case` -- the benchmark mesh itself -- or, without one, on the bounded sample of the same workload."""
    from oracle import oracle as O  # noqa: F401  test infrastructure; used here only as the timed CPU baseline
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from_case
    if case is None:
        nx, ny = map(int, args.cpu_sample.split("x"))
        case = build_case(args, 0, 1, nx, ny, "rowmajor")
        what = f"a {nx}x{ny}-square = {case.mesh.num_owned_cells}-cell sample mesh of the same workload"
        med, reps = _time_oracle(oracle_from_case(case), case, 3, 10, 12.0)
    else:
        what = f"the benchmark mesh itself ({case.mesh.num_owned_cells} cells, the numbering the GPU ran)"
        med, reps = _time_oracle(oracle_from_case(case), case, 3, 5, 8.0)
    nc = case.mesh.num_owned_cells
    return {"value": round(nc / med / 1e6, 3), "unit": "M cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{reps} RHS evaluations on {what}, oracle/swe_oracle.c (gcc -O2, 1 thread), median {med * 1e3:.1f} ms/RHS"}


def _cpu_part_worker(a):
    """One host core: the oracle on one RCB part (with its ghost cells) of the sample mesh."""
    argv, nx, ny, rank, world, reps = a
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from_case
    args = parse(argv)
    args.scaling = "strong"
    case = build_case(args, rank, world, nx, ny, "rowmajor")
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


def cpu_baseline_all_cores(args, argv):
    """no MPI launcher exists here, so P independent oracle processes run on the P parts of an RCB
    partition of the sample mesh (ghost cells present, not exchanged) -- an upper bound on what the MPI-parallel
    reference could do on these host cores."""
    import multiprocessing as mp
    nx, ny = map(int, args.cpu_sample.split("x"))
    cores = len(os.sched_getaffinity(0))
    p = max(1, min(cores, 32))
    with mp.get_context("spawn").Pool(p) as pool:
        res = pool.map(_cpu_part_worker, [(argv, nx, ny, r, p, 5) for r in range(p)])
    cells = sum(r[0] for r in res)
    tmax = max(r[1] for r in res)
    return {"value": round(cells / tmax / 1e6, 2), "unit": "M cell-updates/s", "cores": p, "kind": "port",
            "sample": f"{p} independent oracle processes, one RCB part each of the {nx}x{ny}-square sample mesh, no halo exchange "
                      f"(upper bound on an MPI run), slowest part {tmax * 1e3:.1f} ms/RHS"}


def cpu_baseline_openmp(args, case=None):
    """SURVEY.md 8.d (ii), second line: ONE process, all host cores through OpenMP -- oracle/libswe_oracle_omp.so, the
    oracle's own source built with -fopenmp (Riemann batch and source terms over all cores, each cell's flux sum by one
    thread in the serial order: bitwise the serial result, tests/test_oracle_openmp.py).  On `case` (the benchmark mesh
    itself) or, without one, on the sample mesh."""
    import numpy as np
    cores = max(1, min(len(os.sched_getaffinity(0)), 32))   # as cpu_baseline_all_cores: the affinity mask of a GPU box lists more
                                                            # cores than its CPU share holds
    from oracle import oracle as O  # noqa: F401  test infrastructure; used here only as the timed CPU baseline
    cores = int(O.lib(openmp=True).oracle_set_num_threads(cores))   # not OMP_NUM_THREADS: torch has initialised libgomp long ago
    if case is None:
        nx, ny = map(int, args.cpu_sample.split("x"))
        case = build_case(args, 0, 1, nx, ny, "rowmajor")
        what = f"the {nx}x{ny}-square = {case.mesh.num_owned_cells}-cell sample mesh"
    else:
        what = f"the benchmark mesh itself ({case.mesh.num_owned_cells} cells)"
    cfg = case.config
    if cfg.second_order or cfg.well_balancing:
        return {"skipped": "the OpenMP build parallelises the first-order path only"}
    orc = O.OracleOperator(case.mesh, case.condition_types, cfg.tiny_h, cfg.h_anuga_regular, cfg.xq2018_threshold, cfg.source_method,
                           cfg.well_balancing, openmp=True)
    orc.mannings[:] = case.mannings
    orc.external_sources[:] = case.ext_src
    for b, vals in case.boundary_values.items():
        orc.boundary_values[b][:] = vals
    f = np.zeros((case.mesh.num_owned_cells, 3))
    orc.apply(case.dt, case.u_local, f)
    ts = []
    for _ in range(10):
        f[:] = 0.0
        t0 = time.perf_counter()
        orc.apply(case.dt, case.u_local, f)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    nc = case.mesh.num_owned_cells
    return {"value": round(nc / med / 1e6, 2), "unit": "M cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"10 RHS evaluations on {what}, oracle/swe_oracle.c built with -fopenmp, "
                      f"{cores} threads in one process, median {med * 1e3:.1f} ms/RHS"}


def kernel_sha(second_order: bool = False) -> str:
    """hash of the kernel sources (comments and white space stripped): ties a stored PMC traffic figure to the code it was
    measured on (the first-order / HR kernels do not depend on muscl_kernels.h)"""
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        if name == "muscl_kernels.h" and not second_order:
            continue
        with open(os.path.join(ROOT, "rdycore_amd", "csrc", name), "r") as fh:
            text = fh.read()
        # the code, not its commentary: comments and white space do not change what the compiler sees
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def traffic_key(args) -> str:
    """the key of a run's PMC entry in profiles/traffic.json (tools/make_traffic.py writes what this function names)"""
    key = f"{args.workload}_{args.nx}x{args.ny}_{args.order}_{args.source}"
    if args.hr and args.workload not in ("c5", "delaunay"):
        key += "_hr"
    if args.second_order:
        key += "_second_order_" + args.limiter
    if args.emulate_world > 1:
        key += f"_rank{args.emulate_rank}of{args.emulate_world}"
        if args.self_exchange:
            key += "_self_exchange"       # bytes per STEP: the interior and the halo launch together
    return key


def load_traffic(workload_key: str, layout_bytes: int, second_order: bool = False, quiet: bool = False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), only if they were collected on
    exactly this MACHINE CODE of the measured kernel(s) -- the entry names the kernel and holds the hash of its bytes in the
    library the profiled run loaded; the library this run loads must have the same bytes (rdycore_amd/codeobj.py) -- AND this
    device layout (made by the host code in rdyhip_api.hip: its byte count per launch is compared); a stale or unverifiable
    entry is reported as such, never silently."""
    from rdycore_amd import build as _build, codeobj
    path = os.path.join(ROOT, "profiles", "traffic.json")
    src = {"source": "profiles/traffic.json", "key": workload_key, "kernel_source_sha": kernel_sha(second_order)}
    try:
        with open(path) as fh:
            t = json.load(fh)
    except Exception as exc:
        src["status"] = f"unreadable: {exc!r}"
        return None, src
    ent = t.get(workload_key)
    if not ent:
        src["status"] = "no PMC passes for this workload"
        return None, src
    measured = [(ent.get("kernel"), ent.get("code_sha"))]
    if ent.get("also_in_the_step"):
        measured.append((ent["also_in_the_step"].get("kernel"), ent["also_in_the_step"].get("code_sha")))
    try:
        now = [codeobj.kernel_sha(_build.lib_path(), k) for k, _ in measured]
    except Exception as exc:
        src["status"] = f"cannot verify the code of the measured kernel: {exc!r}"
        if not quiet:
            print(f"bench.py: profiles/traffic.json[{workload_key}]: {src['status']}; roofline.traffic = null", file=sys.stderr)
        return None, src
    src["kernel"], src["code_sha"] = measured[0][0], now[0]
    if [m[1] for m in measured] != now or int(ent.get("layout_bytes_per_launch", -1)) != int(layout_bytes):
        src["status"] = (f"STALE: measured on code {[m[1] for m in measured]} / layout {ent.get('layout_bytes_per_launch')} B, "
                         f"current: {now} / {layout_bytes} B -- rerun tools/profile_gpu.sh")
        if not quiet:
            print(f"bench.py: profiles/traffic.json[{workload_key}] is stale; roofline.traffic = null", file=sys.stderr)
        return None, src
    src["status"] = "measured on this machine code of the kernel"
    return ent.get("hbm_bytes_per_launch"), src


def run_rank(args, argv):
    if args.watchdog_seconds > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)
    # ONE line on stdout, whatever the libraries print: RCCL writes a version banner to stdout when a communicator is
    # created.  File descriptor 1 is pointed at stderr for the whole run; the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    if args.kernel:
        os.environ["RDYHIP_KERNEL"] = args.kernel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "gloo": rehearsal of several ranks on one GPU (host-staged halo)
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"--gpus {world} needs {world} devices, {ndev} visible (BENCH_BACKEND=gloo rehearses several ranks on one GPU)")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # RCCL's warnings land in stderr (fd 1 points there): kept with the run's log
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd.halo import HaloExchange

    def all_ranks_ok(ok: bool) -> bool:
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def abort_run(what: str, mine: str = ""):
        """first contact with N > 1 devices must not end in a silent hang or a plausible-looking number: every rank reports,
        rank 0 prints ONE line {"error": ...} instead of the bench line, everybody exits non-zero"""
        print(f"bench.py rank {rank}: {what} {mine}", file=sys.stderr)
        notes = [None] * world
        if world > 1:
            dist.all_gather_object(notes, mine)
        else:
            notes = [mine]
        if rank == 0:
            os.write(real_stdout, (json.dumps({"error": what, "per_rank": notes, "n_gpus": world, "metric": "M cell-updates/s (SWE RHS eval)",
                                               "value": None}) + "\n").encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)

    stages = {}
    t0 = time.time()
    if world == 1 and args.emulate_world > 1:
        sav = args.scaling
        if not args.self_exchange:
            args.scaling = "strong"
        case = build_case(args, args.emulate_rank, args.emulate_world)
        args.scaling = sav
    else:
        case = build_case(args, rank, world)
    mesh = case.mesh
    stages["mesh_and_state_s"] = round(time.time() - t0, 2)
    t1 = time.time()
    op = CS.create_operator(case)
    stages["operator_create_s"] = round(time.time() - t1, 2)
    halo_mode = args.halo or ("c" if backend == "nccl" else "torch")
    halo, halo_note = None, None
    if world > 1:
        try:
            halo = HaloExchange(mesh, dev, transport=halo_mode, op=op)
            ok = 1
        except Exception as exc:      # e.g. RCCL refuses to build the library's own communicator on this node
            ok, halo_note = 0, repr(exc)
            print(f"bench.py rank {rank}: halo transport '{halo_mode}' failed: {exc!r}", file=sys.stderr)
        agree = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)       # every rank must drive the exchange the same way
        if int(agree.item()) == 0:
            if halo_mode == "torch":
                raise SystemExit(f"halo exchange setup failed: {halo_note}")
            if halo is not None:
                halo.destroy()
            halo = HaloExchange(mesh, dev, transport="torch", op=op)
            halo_note = f"fell back from the C-side RCCL exchange to torch.distributed P2P: {halo_note}"
        stages.update({k: round(v, 3) for k, v in halo.timings.items()})
        # the library's communicator must span every rank of this run (a silent single-rank communicator would make each
        # rank exchange with itself and still print a plausible line)
        spans = halo.rccl_ranks()
        if not all_ranks_ok(not (halo.transport == "c" and backend == "nccl" and halo_note is None and spans != world)):
            abort_run("the library's RCCL communicator does not span the run", f"ncclCommCount = {spans}, world = {world}")
    self_halo, self_rccl_ranks = None, None
    if args.self_exchange:
        if world != 1 or args.emulate_world < 2:
            raise SystemExit("--self-exchange needs --gpus 1 and --emulate-world W")
        import ctypes as C
        from rdycore_amd import _lib
        lib = _lib.load()
        ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
        # the owned cell across each ghost cell's cut edge: what a neighbour would ask this rank for
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
        hh = C.c_void_p()
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        pp = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        cnt = i32([n])
        _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, pp(i32([0])), pp(cnt), pp(sendc), pp(cnt), pp(ghost), C.byref(hh)))
        self_halo = (lib, hh, comm, n)
        nr = C.c_int32(0)
        _lib.check(lib.rdyhip_comm_count(comm, C.byref(nr)))
        self_rccl_ranks = int(nr.value)
    u = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device=dev)
    setup_s = time.time() - t0
    n_owned = mesh.num_owned_cells

    # ---- exchange self-check before anything is timed (N > 1): the analytic state of every LOCAL cell, ghosts included, is known
    # on this rank (build_case fills all local cells), so what the neighbours send can be checked bit for bit -- first the plain
    # ghost update, then the whole multi-rank step
    if halo is not None:
        truth = u.clone()
        ghost_rows = torch.as_tensor(np.nonzero(mesh.cell_is_owned == 0)[0], device=dev)
        if args.inject_fault == "exchange" and rank == world - 1 and ghost_rows.numel():
            truth[ghost_rows[:1]] += 1.0
        mine = ""
        try:
            for name, fn in (("exchange", lambda: halo.exchange(u)), ("overlapped step", lambda: halo.rhs_overlapped(op, case.dt, u, f))):
                u[ghost_rows] = float("nan")
                torch.cuda.synchronize()
                t1 = time.time()
                fn()
                torch.cuda.synchronize()
                stages[f"first_{name.replace(' ', '_')}_s"] = round(time.time() - t1, 3)
                bad = int((~(u == truth).all(dim=1)).sum().item())
                if bad and not mine:        # no early exit: every rank takes part in both exchanges, whatever it found
                    mine = f"{bad} of {int(ghost_rows.numel())} ghost cells differ from their owners' values after the first {name}"
            if not mine and not bool(torch.isfinite(f).all().item()):
                mine = "non-finite RHS after the first overlapped step"
        except Exception as exc:
            mine = f"exception in the exchange self-check: {exc!r}"
        if not all_ranks_ok(not mine):
            abort_run("exchange self-check failed", mine)
        u.copy_(truth)
        del truth

    def step():
        if halo is not None:
            halo.rhs_overlapped(op, case.dt, u, f)
        elif self_halo is not None:
            _lib.check(self_halo[0].rdyhip_rhs_overlapped(op._h, self_halo[1], float(case.dt), int(u.data_ptr()), int(f.data_ptr()),
                                                          int(torch.cuda.current_stream().cuda_stream)))
        else:
            op.rhs_function(case.dt, u, f)

    # ---- device conditioning (untimed, disclosed): the mesh setup above is seconds of host-only work, after which
    # the first launches run at the idle clock state (profiles/r02_launch_series.json); a driver that times 20 steps
    # after 5 warm-up steps would otherwise measure the ramp, not the kernel
    n_cond = 0
    if args.condition_seconds > 0:
        # the number of launches is agreed between the ranks (a step holds an exchange: a rank that left a time-based loop
        # one round earlier than its neighbour would leave that neighbour waiting for ever): one batch is timed, the slowest
        # rank's time decides the count for everybody
        for _ in range(10):          # priming batch: one-time costs (RCCL connects its peers at the first send / recv) stay out
            step()                   # of the batch time that sizes the phase
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t_batch = time.perf_counter() - t_c
        if world > 1:
            tb = torch.tensor([t_batch], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            t_batch = float(tb.item())
        n_batches = int(min(2000, max(0, np.ceil(args.condition_seconds / max(t_batch, 1e-6)) - 1)))
        for _ in range(n_batches):
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        n_cond = 10 * (n_batches + 2)
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
        tmax = torch.tensor([n_owned], dtype=torch.int64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        max_cells = int(tmax.item())
    else:
        total_cells = max_cells = n_owned

    # ---- dominant kernel: average duration with HIP events on the launch stream (one rhs call = ONE launch)
    k_iters = max(10, min(args.steps, 50))
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(k_iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(k_iters)]

    def kernel_only():
        if self_halo is not None:
            return step()
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

    def timed(fn, n, sync_ranks=True, lead=40):
        """ms per call over n back-to-back calls.  `lead` untimed calls go first with no gap before the timed ones: after
        ANY idle moment (a synchronize is enough) launches 5..40 run 20-30 % slow (profiles/r02_launch_series.json)."""
        if world > 1 and sync_ranks:
            torch.cuda.synchronize()
            dist.barrier()
        for _ in range(lead):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # steady state: the launch-to-launch period of back-to-back launches, median over batches (what a long run sees)
    periods = []
    if world == 1:
        for _ in range(40):
            kernel_only()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        evs[0].record()
        for b in range(12):
            for _ in range(10):
                kernel_only()
            evs[b + 1].record()
        torch.cuda.synchronize()
        periods = [evs[b].elapsed_time(evs[b + 1]) / 10 for b in range(12)]
    period_median_ms = float(np.median(periods)) if periods else None

    # multi-GPU: the ghost update on its own (pack, P2P over RCCL, unpack), not overlapped
    halo_ms = timed(lambda: halo.exchange(u), k_iters, lead=5) if halo is not None else None

    # ---- extra (not the metric): one whole forward-Euler step, the update fused into the RHS kernel's stores
    # (rdyhip_euler_step, F never written) against the RHS + axpy pair -- SURVEY.md 8.f row 1
    u2 = torch.empty_like(u)
    u3 = u.clone()

    # with a halo the two state arrays ping-pong as a time loop's do (dt = 0: the same work, the state stays put), so that the
    # pack of each exchange can ride on the previous step's kernel (rdyhip_halo_fuse_pack) -- the chain a run of steps has
    pack_fused = False
    if halo is not None and halo._halo is not None:
        pack_fused = halo.fuse_pack(True)
    elif self_halo is not None:
        pack_fused = self_halo[0].rdyhip_halo_fuse_pack(self_halo[1], 1) == 0
    pp = [u, u2]
    if halo is not None or self_halo is not None:
        u2.copy_(u)

    def fused_step():
        if halo is not None:
            halo.step_overlapped(op, 0.0, pp[0], pp[1])
            pp.reverse()
        elif self_halo is not None:
            _lib.check(self_halo[0].rdyhip_euler_step_overlapped(op._h, self_halo[1], 0.0, int(pp[0].data_ptr()), int(pp[1].data_ptr()), None,
                                                                 int(torch.cuda.current_stream().cuda_stream)))
            pp.reverse()
        else:
            op.euler_step(case.dt, u, u2)

    def pair_step():
        step()
        op.axpy_owned(0.0, f, u3)      # dt = 0: same traffic, the scratch state stays put

    ef, ep = timed(fused_step, 60), timed(pair_step, 60)
    euler = {"fused_ms_per_step": round(ef, 5), "rhs_plus_axpy_ms_per_step": round(ep, 5), "fused_steps_per_s": round(1e3 / ef, 1)}
    if halo is not None or self_halo is not None:
        euler["pack_fused_into_kernel"] = bool(pack_fused)
        if halo is not None:
            halo.invalidate()
    # ---- extra (not the metric): the drop-in's real loop, not back-to-back launches.  RDyAdvance (src/rdyadvance.c:261-383) as
    # the driver's time loop calls it: refresh of the rain source from a host array (RDySetDomainWaterSource -> the
    # stream-ordered setter), 20 explicit steps, the Courant struct read back (a synchronisation, as with adaptive dt) -- the
    # device restarts from idle every interval and runs its next few dozen launches 20-35 % slow
    advance = None
    if world == 1 and halo is None and self_halo is None:
        rain = np.full(n_owned, 1e-5)
        src_before = op.external_sources.clone()

        def one_advance(nsteps=20, sync=True):
            op.set_domain_external_source(0, rain, ordered=True)
            cur, nxt = u, u2
            for _ in range(nsteps):
                op.euler_step(0.0, cur, nxt)       # dt = 0: every interval does the same work
                cur, nxt = nxt, cur
            if sync:
                op.update_diagnostics()

        u2.copy_(u)
        rows = {}
        for name, sync in (("adaptive_dt", True), ("fixed_dt", False)):
            one_advance(sync=sync)
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            for _ in range(12):
                one_advance(sync=sync)
            torch.cuda.synchronize()
            rows[name] = (time.perf_counter() - t_a) / (12 * 20) * 1e3
        op.refresh_field(1, src_before)
        torch.cuda.synchronize()
        advance = {"steps_per_advance": 20, "advances": 12,
                   "refresh": "domain-wide rain source from a host array (8 B per cell) through the stream-ordered setter, every advance",
                   "ms_per_step_adaptive_dt": round(rows["adaptive_dt"], 5), "ms_per_step_fixed_dt": round(rows["fixed_dt"], 5),
                   "back_to_back_ms_per_step": round(ef, 5),
                   "vs_back_to_back_adaptive_dt": round(rows["adaptive_dt"] / ef, 4), "vs_back_to_back_fixed_dt": round(rows["fixed_dt"] / ef, 4),
                   "note": "adaptive_dt: the 16-byte Courant struct is read back after every advance (src/rdyadvance.c:366-372), so the device "
                           "idles between advances; fixed_dt: nothing synchronises"}
        del src_before
    del u2, u3, pp
    step()   # leave F and the diagnostics of a plain RHS evaluation behind for the sanity checks below

    # sanity: the result is finite and the Courant diagnostic is alive (cross-rank struct-max, src/operator.c:705-751)
    from rdycore_amd.timestep import reduce_courant
    op.update_diagnostics()
    cd = reduce_courant(op.get_diagnostics(), dev)
    finite = bool(torch.isfinite(f).all().item())
    if world > 1:
        fin = torch.tensor([1 if finite else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(fin, op=dist.ReduceOp.MIN)
        finite = bool(fin.item())

    # what every rank did (rank 0's line carries all of them): the form of the step, direct receive, cells, ghosts
    def form_of(kind):
        """rdyhip_halo_form_info: the form this rank's steps of a kind take (in order / two streams), chosen by the halo's own
        16-step trial on the communicator of this run, with the two timings"""
        if halo is not None and halo._halo is not None:
            return halo.form_info(kind)
        if self_halo is not None:
            import ctypes as C
            fi = _lib.RDyHipHaloFormInfo()
            _lib.check(self_halo[0].rdyhip_halo_form_info(self_halo[1], 1 if kind == "euler" else 0, C.byref(fi)))
            return {"form": "two_streams" if fi.form else "in_order", "source": HaloExchange.FORM_SOURCES[fi.source], "trial_steps": int(fi.trial_steps),
                    "in_order_ms": round(float(fi.in_order_ms), 5), "two_stream_ms": round(float(fi.two_stream_ms), 5)}
        return None

    mine_info = {"rank": rank, "cells": n_owned, "ghost_cells": int(mesh.num_cells - n_owned),
                 "rhs_step_form": form_of("rhs"), "euler_step_form": form_of("euler"),
                 "halo_overlapped": (int(_lib.load().rdyhip_halo_overlaps(halo._halo)) if halo is not None and halo._halo is not None else None),
                 "direct_receive": (bool(halo.direct_receive) if halo is not None and halo._halo is not None else None),
                 "peers": (len(halo._plan_peers) if halo is not None else 0), "device": dev_index}
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, mine_info)
    else:
        per_rank = [mine_info]

    order_study = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_order_study and args.emulate_world <= 1 \
            and not args.second_order and not args.hr:
        # SURVEY.md 8.d "report both": the same mesh in the generator's row-major numbering and along a Hilbert curve
        order_study = {args.order: round(n_owned / period_median_ms / 1e3, 1)}
        info_main = op.layout_info()
        order_study["edge_records_per_cell"] = {args.order: round(info_main["num_edge_records"] / n_owned, 4)}
        for o2 in ("rowmajor", "hilbert", "tiled"):
            if o2 == args.order:
                continue
            c2 = build_case(args, 0, 1, order=o2)
            op2 = CS.create_operator(c2)
            uu = torch.tensor(c2.u_local, dtype=torch.float64, device=dev)
            ms = timed(lambda: op2.rhs_function(c2.dt, uu, f), 100, sync_ranks=False)
            order_study[o2] = round(n_owned / ms / 1e3, 1)
            order_study["edge_records_per_cell"][o2] = round(op2.layout_info()["num_edge_records"] / n_owned, 4)
            op2.destroy()
            del uu, c2, op2
        order_study["unit"] = "M cell-updates/s (steady-state launch period, same box, same run)"

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells / (elapsed / args.steps) / 1e6
        info = op.layout_info()
        quads = info["slots_per_cell"] == 4
        alg = ALG_BYTES_PER_CELL_QUADS if quads else ALG_BYTES_PER_CELL
        achieved = n_owned * alg / (kern_ms * 1e-3) / 1e9
        nxg = args.nx if (args.scaling == "strong" or world == 1) else args.nx * world
        part = "single" if world == 1 else (f"rcb_{world}" if args.scaling == "strong" else f"strips_x{world}")
        if world == 1 and args.emulate_world > 1:
            part = f"rank {args.emulate_rank} of rcb_{args.emulate_world} (ghost cells present, not exchanged)"
            if self_halo is not None:
                part = (f"rank {args.emulate_rank} of {'rcb_' if args.scaling == 'strong' else 'strips_x'}{args.emulate_world}; every step is rdyhip_rhs_overlapped with a one-rank RCCL "
                        f"communicator whose only peer is this rank ({self_halo[3]} cells = {self_halo[3] * 24} B sent to and received from itself)")
        friction = f"{args.source} friction"
        if args.workload == "c3":
            workload = (f"C3: synthetic {nxg}x{args.ny}x2 = {total_cells}-cell triangle mesh "
                        f"({n_owned} cells/GPU), MMS-style state over sinusoidal bathymetry, Manning field, rain source, "
                        f"dry disc, Dirichlet + critical-outflow + reflecting boundaries, {friction}, dt=1e-3")
        elif args.workload == "c2":
            workload = (f"C2: synthetic {nxg}x{args.ny}x2 = {total_cells}-cell triangle mesh ({n_owned} cells/GPU), "
                        f"flat-bed dam break h = 10 / 5 with perturbed momenta, Manning 0.015, reflecting walls, {friction}, dt=1e-3")
        elif args.workload == "dambreak_quads":
            workload = (f"the reference's dam-break benchmark (docs/user/example-cases/dam-break): {nxg}x{args.ny} quads minus the dam = "
                        f"{total_cells} cells ({n_owned} on rank 0), dx = dy = 10 m / {nxg}, h = 10 / 5 m at rest, Manning 0.015, "
                        f"all walls reflecting, {friction}, dt = 1.5625e-5 s")
        elif args.workload == "houston_refined":
            dry = float((case.u_local[mesh.cell_owned_to_local, 0] == 0.0).mean())
            workload = (f"the reference's Houston1km real-DEM mesh (share/meshes/Houston1km_with_z.exo, 2746 triangles, ragged outline) refined "
                        f"{args.levels} times as -dm_refine does = {total_cells} triangles ({n_owned} on rank 0), vertex z interpolated; state = the "
                        f"parents' water surface of Houston1km.ic clipped at each child's bed ({dry:.0%} of the cells dry), rain and Dirichlet stage "
                        f"from the reference's Houston1km.rain / .bc series at t = 7200 s, Manning 0.015, other boundary edges reflecting, "
                        f"{friction}, dt = {case.dt:g} s")
        elif args.workload == "delaunay":
            dry = float((case.u_local[mesh.cell_owned_to_local, 0] == 0.0).mean())
            workload = (f"genuinely unstructured: Delaunay triangulation of a jittered, smoothly graded {args.nx + 1}^2 point set (vertex valences "
                        f"3..11, cell areas 1:9+) = {total_cells} triangles ({n_owned} on rank 0) over the C5 DEM, {dry:.0%} of the cells dry, rain "
                        f"1e-5 m/s, Manning 0.03, critical-outflow segment + reflecting walls, hydrostatic reconstruction, {friction}, dt = 0.05 s")
        else:
            dry = float((case.u_local[mesh.cell_owned_to_local, 0] == 0.0).mean())
            workload = (f"C5 stand-in for the Harvey mesh: synthetic {nxg}x{args.ny}x2 triangles over a rough analytic DEM (ramp + 3 sinusoids), "
                        f"{total_cells} cells in this run ({n_owned} on rank 0), {dry:.0%} of them dry, rain 1e-5 m/s, Manning 0.03, "
                        f"critical-outflow segment + reflecting walls, hydrostatic reconstruction, {friction}, dt = 0.05 s")
        traffic, traffic_src = (None, None)
        if world == 1:
            traffic, traffic_src = load_traffic(traffic_key(args), int(info["bytes_per_apply"]), args.second_order)
        if args.second_order:
            kname = "swe_rhs_muscl_fused_kernel<%d,%d>" % (info["slots_per_cell"], 0 if args.source == "semi_implicit" else 1)
        else:
            kname = "%s<%d,%d%s>" % ("swe_rhs_tiled_kernel" if info["tiled_kernel"] else "swe_rhs_kernel", info["slots_per_cell"],
                                     0 if args.source == "semi_implicit" else 1, ",HR" if args.hr else "")
        out = {
            "metric": "M cell-updates/s (SWE RHS eval)",
            "value": round(value, 1),
            "unit": "M cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "cells_per_gpu": n_owned, "max_cells_per_gpu": max_cells, "total_cells": total_cells,
                       "cell_order": args.order, "partition": part,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": (backend if world > 1 else None),
                       "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if world > 1 and backend == "nccl" else None,
                       "halo_driver": (halo.transport if halo is not None else None), "halo_note": halo_note,
                       # ncclCommCount of the communicator the library's exchange runs on: proof that RCCL spanned N ranks
                       "rccl_ranks": (halo.rccl_ranks() if halo is not None else (self_rccl_ranks if self_halo is not None else None)),
                       "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       # 1: the exchange runs on the library's stream behind the interior tiles; 0: the in-order form of small parts
                       "halo_overlapped": ((int(_lib.load().rdyhip_halo_overlaps(halo._halo)) if halo is not None and halo._halo is not None else None)
                                           if self_halo is None else int(self_halo[0].rdyhip_halo_overlaps(self_halo[1]))),
                       # 1: ncclRecv lands in u_local's ghost rows (ghosts numbered peer by peer in arrival order): no unpack launch
                       "halo_direct_receive": (halo.direct_receive if halo is not None and halo._halo is not None else
                                               (bool(self_halo[0].rdyhip_halo_direct_receive(self_halo[1])) if self_halo is not None else None)),
                       "well_balancing": "hydrostatic_reconstruction" if args.hr else "none",
                       "spatial_order": ("second (MUSCL, %s limiter)" % args.limiter) if args.second_order else "first",
                       "halo_bytes_per_rank": halo.bytes_sent_per_exchange if halo else 0,
                       "halo_exchange_alone_ms": round(halo_ms, 5) if halo_ms is not None else None,
                       "conditioning": (f"{n_cond} untimed RHS launches ({args.condition_seconds:g} s) before --warmup"
                                        if n_cond else "none"),
                       "setup_seconds": round(setup_s, 1), "setup_stages": stages, "per_rank": per_rank, "max_courant": cd.max_courant_num,
                       "max_courant_edge": cd.global_edge_id, "max_courant_cell": cd.global_cell_id, "finite": finite},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname,
                         "algorithmic_bytes_per_cell": alg,
                         "tile_edge_records_per_cell": round(info["num_edge_records"] / max(n_owned, 1), 4),
                         "halo_cells_per_tile": round(info["num_halo_entries"] / max(info["num_tiles"], 1), 2),
                         "kernel_avg_ms": round(kern_ms, 5),
                         "kernel_isolated_avg_ms": round(kern_isolated_ms, 5), "kernel_isolated_median_ms": round(float(np.median(kern_all)), 5),
                         "kernel_isolated_min_ms": round(float(np.min(kern_all)), 5),
                         "steady_state_period_median_ms": round(period_median_ms, 5) if period_median_ms else None,
                         "steady_state_frac": round(n_owned * alg / (period_median_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if period_median_ms else None,
                         "algorithmic_bytes_per_launch": int(n_owned * alg),
                         "layout_bytes_per_launch": int(info["bytes_per_apply"]),
                         "persistent_workgroups": int(info["persistent_grid"]), "lds_bytes_per_workgroup": int(info["lds_bytes"]),
                         "cells_per_tile": round(n_owned / max(info["num_tiles"], 1), 1)},
        }
        if quads:
            out["roofline"]["frac_176B_model"] = round(n_owned * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        # the whole forward-Euler step by the same byte model (it stores u_out where the RHS stores F; F itself is not written)
        euler["frac_of_hbm_roofline"] = round(n_owned * alg / (euler["fused_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if world == 1 and halo is None and self_halo is None:
            # PMC bytes of the Euler-step kernel itself, from the same profiled runs (their Euler launches), same guard
            et, esrc = load_traffic(traffic_key(args) + "_euler_step", int(info["bytes_per_apply"]), args.second_order, quiet=True)
            euler["traffic"], euler["traffic_source"] = et, esrc
        out["euler_step"] = euler
        if advance is not None:
            out["advance_pattern"] = advance
        if order_study:
            out["cell_order_study"] = order_study
        if args.second_order:
            # the 176-B figure above keeps variants comparable; the second-order path's own model:
            b2 = ALG_BYTES_PER_CELL_SECOND_ORDER
            if quads:      # two edges per cell instead of 1.5: 192 + centroid 16 + midpoints 2 x 16
                b2 += ALG_BYTES_PER_CELL_QUADS - ALG_BYTES_PER_CELL + 8.0
            a2 = n_owned * b2 / (kern_ms * 1e-3) / 1e9
            out["roofline"]["second_order_model"] = {"bytes_per_cell_update": b2, "achieved": round(a2, 1),
                                                     "frac": round(a2 / HBM_PEAK_GBPS, 4)}
        if not args.no_cpu_baseline and world == 1:
            # the reference's CPU path beside the GPU figure, ON THE BENCHMARK MESH ITSELF where it fits (C3's
            # 10 M cells, ~1 s per RHS on one core); the 1 M-cell sample of earlier rounds stays as a second key
            full = case if (n_owned <= CPU_FULL_MESH_MAX_CELLS and args.emulate_world <= 1) else None
            out["cpu_baseline"] = cpu_baseline(args, full)
            if full is not None:
                out["cpu_baseline_sample"] = cpu_baseline(args)
            if not args.no_cpu_all_cores:
                try:
                    out["cpu_baseline_openmp"] = cpu_baseline_openmp(args, full)     # all host cores, one process, the benchmark mesh
                except Exception as exc:  # a reported extra, never a reason to lose the bench line
                    out["cpu_baseline_openmp"] = {"error": repr(exc)}
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, argv)   # one process per core on the parts of the sample
                except Exception as exc:
                    out["cpu_baseline_all_cores"] = {"error": repr(exc)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if halo is not None:
        halo.destroy()
    if self_halo is not None:
        import ctypes as C
        _lib.check(self_halo[0].rdyhip_halo_destroy(C.byref(self_halo[1])))
        _lib.check(self_halo[0].rdyhip_comm_destroy(self_halo[2]))
    op.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain command: become the launcher.  Nothing here may touch the GPU (no HIP call, no
        # torch.cuda.*): the ranks are fresh children, this process only relays rank 0's line.
        from rdycore_amd.launch import launch_ranks
        rc = launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + argv, timeout=args.launch_timeout,
                          keep=lambda line: line.lstrip().startswith("{"))     # the JSON line; library chatter goes to stderr
        sys.exit(rc)
    run_rank(args, argv)


if __name__ == "__main__":
    main()
