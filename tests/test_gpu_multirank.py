"""Several ranks on ONE GPU (gloo process group, bytes staged through the host): the rank-local pieces of the
N > 1 path -- partition (strips or RCB), pack / unpack kernels, side-stream exchange overlapped with the interior
tiles, halo tiles afterwards, the cross-rank Courant struct-max -- give the single-rank RHS.  Two drivers:
transport="torch" (Python calls the ABI's pack / phase / unpack entry points one by one) and transport="c"
(ONE call: rdyhip_rhs_overlapped / rdyhip_euler_step_overlapped, csrc/halo_exchange.h, with the bytes going through the
ABI's transport callback because RCCL cannot connect several ranks that share a device).  RCCL itself is exercised by
the one-rank self-exchange test at the bottom."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cases(kind, rank, world, second_order):
    """(this rank's Case, the undivided Case, the undivided mesh's partition-independent edge id function)"""
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    from rdycore_amd import partition as P
    if kind == "strips":
        nxp, ny = 40, 48                      # per-rank strip: 3840 owned cells = 15 tiles
        nxg = nxp * world
        K = 2 * np.pi / 37
        z = CS.mms_bathymetry(K=K)
        mesh = M.strip_partition_tri_mesh(nxp, ny, rank, world, 1.0, zfunc=z, order="tiled", tile=8)
        case = CS.friction_slope_case(mesh, nxg, ny, dt=1e-2, K=K)
        g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
        gc = CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K)
        for c in (case, gc):
            # the analytic state repeats every 37 squares and is symmetric in x <-> y, so several edges on different ranks
            # reach the SAME maximal Courant number to the last bit; a tilt (a function of position only) makes the
            # maximum unique, so that the ids of the cross-rank struct-max are well defined
            xc, yc = c.mesh.cell_centroids[:, 0], c.mesh.cell_centroids[:, 1]
            c.u_local[:, 1] *= 1.0 + 1e-3 * xc / nxg + 2e-3 * yc / ny
        ekey = lambda e: M.edge_vertex_key(g, e)
    elif kind == "rcb_c5":
        nx, ny = 200, 200                     # the C5 miniature: 80 000 triangles over the rough DEM, HR, ~40 % dry
        case = CS.c5_case(CS.c5_mesh(nx, ny, rank, world), float(nx), float(ny))
        g = CS.c5_mesh(nx, ny)
        gc = CS.c5_case(g, float(nx), float(ny))
        ekey = lambda e: e                     # the oracle reports edges.global_ids, which this mesh already carries partition-independent
    elif kind == "rcb_quads":
        case = CS.dam_break_quads_case(CS.dam_break_quads_mesh(320, 160, rank, world))
        g = CS.dam_break_quads_mesh(320, 160)
        gc = CS.dam_break_quads_case(g)
        for c in (case, gc):                  # a moving state (the benchmark's initial state is at rest)
            xc, yc = c.mesh.cell_centroids[:, 0], c.mesh.cell_centroids[:, 1]
            # (the tilt of the depth makes the maximal Courant number unique: on the two flat pools several edges reach the
            # same value to the last bit, and which of them a run reports depends on its loop order, i.e. on the partition)
            c.u_local[:, 0] *= 1.0 + 1e-3 * xc / 10.0 + 2e-3 * yc / 5.0
            c.u_local[:, 1] = 0.3 * c.u_local[:, 0] * np.sin(1.7 * xc + 0.9 * yc)
            c.u_local[:, 2] = 0.2 * c.u_local[:, 0] * np.cos(1.1 * xc - 2.3 * yc)
        ekey = lambda e: e
    elif kind == "rcb_houston":
        # unstructured: the reference's Houston1km mesh refined three times (175 744 triangles), ragged outline, wet / dry
        # fronts, rain + stage forcing; parts of very different shapes, the owners of the ghosts known from the part array
        data = os.path.join(ROOT, "tests", "golden", "houston")
        case = CS.houston_refined_case(data, 3, "hilbert", rank=rank, world=world)
        gc = CS.houston_refined_case(data, 3, "hilbert")
        g = gc.mesh
        ekey = lambda e: M.edge_vertex_key(g, e)
    else:
        raise ValueError(kind)
    case.config.second_order = second_order
    gc.config.second_order = second_order
    return case, gc, ekey


def _worker(rank, world, port, kernel, q, second_order=False, transport="torch", kind="strips", overlap="1"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # these meshes are small: left to itself the library would run them in its in-order form (rdyhip_halo_overlaps);
    # "1" forces the overlapped two-stream form the big meshes use, "0" the in-order one
    if overlap == "trial":
        os.environ.pop("RDYHIP_OVERLAP", None)       # the halo chooses by itself: the first 16 steps of a kind alternate
    else:
        os.environ["RDYHIP_OVERLAP"] = overlap
    if kernel:
        os.environ["RDYHIP_KERNEL"] = kernel
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rdycore_amd import cases as CS
        from rdycore_amd.halo import HaloExchange
        from rdycore_amd.timestep import reduce_courant
        from helpers import oracle_from_case, rel_linf
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        case, gc, ekey = _cases(kind, rank, world, second_order)
        mesh = case.mesh
        op = CS.create_operator(case)
        halo = HaloExchange(mesh, dev, transport=transport, op=op)
        if transport == "c":
            from rdycore_amd import _lib
            if overlap != "trial":
                assert _lib.load().rdyhip_halo_overlaps(halo._halo) == int(overlap)
        u_np = case.u_local.copy()
        u_np[mesh.cell_is_owned == 0] = np.nan          # ghosts unknown until exchanged
        u = torch.tensor(u_np, dtype=torch.float64, device=dev)
        f = torch.full((mesh.num_owned_cells, 3), 5.0, dtype=torch.float64, device=dev)
        for _ in range(3):                               # repeated steps reuse buffers and streams
            halo.rhs_overlapped(op, case.dt, u, f)
        torch.cuda.synchronize()
        assert torch.equal(u, torch.tensor(case.u_local, device=dev)), "ghost update wrong"
        # the whole forward-Euler step with the same overlap (update fused into the kernels' stores)
        u2 = torch.full_like(u, float("nan"))
        halo.step_overlapped(op, case.dt, u, u2)
        torch.cuda.synchronize()
        own = torch.as_tensor(mesh.cell_owned_to_local, device=dev).long()
        assert torch.allclose(u2[own], u[own] + case.dt * f, rtol=0, atol=1e-13), "fused Euler step differs from RHS + axpy"
        if transport == "c":
            # the two shortcuts of the exchange chain (include/rdyhip.h).  Direct receive: these meshes number their ghosts peer
            # by peer in arrival order, so every exchange above already landed in the ghost rows themselves (no unpack launch).
            # Fused pack: inside one advance() the pack of each exchange rides on the previous step's kernel -- the same five
            # steps with the pack launches back give the same bits.
            from rdycore_amd import _lib
            from rdycore_amd.timestep import EulerStepper
            lib = _lib.load()
            assert halo.direct_receive
            if kernel != "cell":
                ua = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
                uc = ua.clone()
                dts = 0.1 * case.dt
                EulerStepper(op, halo=halo).advance(ua, dts, 5 * dts)
                assert lib.rdyhip_halo_pack_fused(halo._halo) == 1
                assert halo.form_info("euler")["source"] == ("forced" if overlap != "trial" else "trial_running")
                # (second order, fused form: the state pack rides on the kernel, the gradient exchange still packs with a launch)
                stepper = EulerStepper(op, halo=halo)
                assert halo.fuse_pack(False) is False
                stepper.advance(uc, dts, 5 * dts)
                torch.cuda.synchronize()
                assert bool(torch.isfinite(ua[own]).all()) and torch.equal(ua, uc), "fused pack changes the trajectory"
            else:
                assert halo.fuse_pack(True) is False         # second order / the cell-centric kernel keep their pack launch
        # the plain (not overlapped) ghost update of a fresh array
        u3 = torch.tensor(u_np, dtype=torch.float64, device=dev)
        halo.exchange(u3)
        torch.cuda.synchronize()
        assert torch.equal(u3, u)
        halo.rhs_overlapped(op, case.dt, u, f)           # leave the diagnostics of a plain RHS behind
        # single-rank truth from the oracle on the undivided mesh
        og = oracle_from_case(gc)
        fg = og.apply(gc.dt, gc.u_local)
        g2row = {int(g): i for i, g in enumerate(gc.mesh.cell_global_ids)}
        rows = np.array([g2row[int(g)] for g in mesh.cell_global_ids[mesh.cell_owned_to_local]])
        err = rel_linf(f.cpu().numpy(), fg[rows])
        if kind == "strips":
            # two classical Runge-Kutta steps (temporal: rk4): four overlapped RHS evaluations each, against the
            # single-rank oracle loop on the undivided mesh
            from rdycore_amd.timestep import EulerStepper
            from helpers import oracle_rk4
            urk = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
            EulerStepper(op, halo=halo, temporal="rk4").advance(urk, case.dt, 2 * case.dt)
            torch.cuda.synchronize()
            ug = oracle_rk4(og, gc.u_local, gc.dt, 2)
            own = mesh.cell_owned_to_local
            err = max(err, rel_linf(urk.cpu().numpy()[own], ug[rows]))
            # the water-volume budget ACROSS ranks over six overlapped Euler steps: a cut edge is evaluated by both ranks
            # from identical operands, so what one rank's cell loses the other rank's cell gains, to the last bit
            ub = torch.tensor(case.u_local, dtype=torch.float64, device=dev)
            area = mesh.cell_areas[own]
            v0 = float((area * ub.cpu().numpy()[own, 0]).sum())
            st = EulerStepper(op, halo=halo)
            out = 0.0
            dtb = 0.1 * case.dt                          # small enough for the second-order scheme at this case's dry disc
            for _ in range(6):
                st.advance(ub, dtb, dtb)
                for b, bnd in enumerate(mesh.boundaries):
                    mine = mesh.cell_is_owned[mesh.edge_cell_ids[2 * bnd.edge_ids]] != 0
                    fl = op.boundary_fluxes(b)[:, 0]
                    out += dtb * float(np.nansum((fl * mesh.edge_lengths[bnd.edge_ids])[mine]))
            torch.cuda.synchronize()
            v1 = float((area * ub.cpu().numpy()[own, 0]).sum())
            rain = 6 * dtb * float((area * case.ext_src[:, 0]).sum())
            tot = torch.tensor([v1 - v0, rain - out, v0], dtype=torch.float64)
            dist.all_reduce(tot)
            assert abs(float(tot[0] - tot[1])) <= 1e-12 * float(tot[2]), ("volume budget across ranks", tot.tolist())
            halo.rhs_overlapped(op, case.dt, u, f)       # diagnostics of a plain RHS again
            og.reset_diagnostics()
            og.apply(gc.dt, gc.u_local)
        # UpdateOperatorDiagnostics: local 16 bytes + the struct-max across ranks, ids included (src/operator.c:705-715, 879)
        op.update_diagnostics()
        red = reduce_courant(op.get_diagnostics(), dev)
        cg, eg, cellg = og.diagnostics()
        ids_ok = red.global_cell_id == cellg and (red.global_edge_id == ekey(eg) if eg >= 0 else red.global_edge_id == -1)
        info = op.layout_info()
        if not ids_ok:
            print(f"rank {rank}: Courant struct-max {red} vs the oracle's ({cg}, edge {eg} = key {ekey(eg) if eg >= 0 else -1}, cell {cellg})", file=sys.stderr)
        if overlap == "trial" and transport == "c":
            # through the rest of the trial and past it: same bits in whatever form a step runs, and every rank has chosen
            fa = f.clone()
            for _ in range(20):
                halo.rhs_overlapped(op, case.dt, u, f)
            torch.cuda.synchronize()
            assert torch.equal(f, fa)
            fi = halo.form_info("rhs")
            assert fi["source"] == "measured" and fi["trial_steps"] == 16 and fi["in_order_ms"] > 0 and fi["two_stream_ms"] > 0, fi
        q.put((rank, err, abs(red.max_courant_num - cg), bool(ids_ok), info["num_halo_tiles"], info["num_tiles"]))
        halo.destroy()
        op.destroy()
    finally:
        dist.destroy_process_group()


def _run(world, args, join=240):
    assert world <= 5, "at most six processes may use the GPU of a box together, and the test runner itself is one of them"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + args[:1] + (q,) + args[1:]) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(join)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(q.get(timeout=5) for _ in range(world))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["torch", "c"])
def test_three_ranks_one_gpu_overlapped_rhs(rdyhip_kernel, transport):
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(3, (rdyhip_kernel, False, transport, "strips")):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12 and ids_ok
        if rdyhip_kernel != "cell":
            assert 0 < nhalo_tiles < ntiles


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["torch", "c"])
def test_three_ranks_one_gpu_second_order(rdyhip_kernel, transport):
    """the second-order path across ranks: state exchange, gradients, gradient exchange (6 values per cell),
    fluxes of interior tiles overlapped, halo tiles afterwards; no reverse exchange"""
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(3, (rdyhip_kernel, True, transport, "strips")):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12 and ids_ok


@pytest.mark.timeout(300)
@pytest.mark.parametrize("second_order", [False, True])
def test_three_ranks_one_gpu_in_order_form(rdyhip_kernel, second_order):
    """the form small parts get (no overlap: exchange, gradients + their exchange, one launch over all tiles, in order on the
    caller's stream) gives the same RHS, Euler step, RK4 steps, volume budget and Courant struct"""
    if rdyhip_kernel == "cell" and second_order:
        pytest.skip("second order is implemented by the tiled kernels")
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(3, (rdyhip_kernel, second_order, "c", "strips", "0")):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12 and ids_ok


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["rcb_c5", "rcb_quads"])
def test_three_rcb_ranks_one_gpu(rdyhip_kernel, kind):
    """RCB parts (rdycore_amd/partition.py) instead of strips: the C5 miniature (rough DEM, hydrostatic reconstruction, dry
    cells, outlet) and the reference's dam-break quad mesh with its hole, each on 3 ranks through the C-side overlapped
    RHS = the oracle on the undivided mesh"""
    if rdyhip_kernel == "cell" and kind == "rcb_c5":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(3, (rdyhip_kernel, False, "c", kind)):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12 and ids_ok


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind,second_order", [("strips", False), ("rcb_c5", False), ("rcb_quads", True), ("rcb_houston", False), ("rcb_houston", True)])
def test_five_ranks_one_gpu(rdyhip_kernel, kind, second_order):
    """the rehearsal of a full node as far as one device allows (six processes may share a card, the test runner being one
    of them): strips with three inner ranks, RCB-5 parts with three and more peers, uneven sizes and shapes, the second-order
    double exchange, RK4 (strips) -- every rank's RHS = the single-rank oracle's rows, the cross-rank Courant struct-max
    with its ids.  (Eight ranks: the same pattern discovery and exchange on the CPU, tests/test_halo_gloo.py,
    tests/test_halo_plan_cpu.py.)"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough for the five-rank rehearsal")
    res = _run(5, (rdyhip_kernel, second_order, "c", kind), join=420)
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in res:
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-10 and ids_ok
        assert 0 < nhalo_tiles <= ntiles


@pytest.mark.timeout(600)
@pytest.mark.parametrize("second_order", [False, True])
def test_five_ranks_choose_the_form_of_their_step(rdyhip_kernel, second_order):
    """nothing forced: every rank's halo alternates between the two forms of the step over its first 16 calls, times them and
    keeps the faster one (rdyhip_halo_form_info) -- five gloo ranks on one device, every step of the trial and after it = the
    single-rank oracle's rows"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(5, (rdyhip_kernel, second_order, "c", "strips", "trial"), join=420):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-10 and ids_ok


@pytest.mark.timeout(300)
def test_a_slow_transfer_flips_the_form(rdyhip_kernel):
    """The trial decides by what it measures: the same part, the same exchange pattern, a transport callback that copies the send
    rows to the ghost rows on the device -- once as fast as it can, once behind a device-side delay about as long as the interior
    tiles' launch.  Fast: the step in order wins (the two cross-stream hops of the other form cost more than the copy they hide).
    Slow: two streams win (the delay runs beside the interior launch).  Same bits either way."""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    import ctypes as C
    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    from rdycore_amd.operator import _DeviceArray
    lib = _lib.load()
    torch.cuda.set_device(0)
    nxp, ny, world = 1200, 1250, 3                          # 3.0 M owned cells: a launch of ~100 us, long enough to hide something behind
    K = 2 * np.pi / 97
    mesh = M.strip_partition_tri_mesh(nxp, ny, 1, world, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled")
    case = CS.friction_slope_case(mesh, nxp * world, ny, dt=1e-3, K=K)
    op = CS.create_operator(case)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
    gset = np.zeros(mesh.num_cells, dtype=bool)
    gset[ghost] = True
    cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
    cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
    sendc = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut])).astype(np.int32)
    n = min(sendc.size, ghost.size)
    sendc, ghost = i32(sendc[:n]), i32(ghost[:n])
    st = int(torch.cuda.current_stream().cuda_stream)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda:0")
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda:0")
    owner = object()
    delay = {"cycles": 0}

    def transport(ctx, d_send, d_recv, ncomp, stream):
        try:
            s = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
            with torch.cuda.stream(s):
                if delay["cycles"]:
                    torch.cuda._sleep(delay["cycles"])                     # the "link": device time on the stream the bytes travel on
                src = torch.as_tensor(_DeviceArray(d_send, (n * ncomp,), owner), device="cuda:0")
                dst = torch.as_tensor(_DeviceArray(d_recv, (n * ncomp,), owner), device="cuda:0")
                dst.copy_(src)
            return 0
        except Exception:                                                  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1
    cb = _lib.TRANSPORT_FN(transport)

    def run(cycles):
        delay["cycles"] = cycles
        h = C.c_void_p()
        _lib.check(lib.rdyhip_halo_create(op._h, None, 1, p(i32([0])), p(i32([n])), p(sendc), p(i32([n])), p(ghost), C.byref(h)))
        _lib.check(lib.rdyhip_halo_set_transport(h, C.cast(cb, C.c_void_p), None))
        for _ in range(24):
            _lib.check(lib.rdyhip_rhs_overlapped(op._h, h, case.dt, int(u.data_ptr()), int(f.data_ptr()), st))
        torch.cuda.synchronize()
        fi = _lib.RDyHipHaloFormInfo()
        _lib.check(lib.rdyhip_halo_form_info(h, 0, C.byref(fi)))
        _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
        return f.clone(), fi

    # a delay of ~100 us, whatever the counter torch.cuda._sleep spins on ticks at
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    e0.record()
    torch.cuda._sleep(100_000)
    e1.record()
    torch.cuda.synchronize()
    cycles = max(1000, int(100_000 * 0.100 / max(e0.elapsed_time(e1), 1e-4)))
    f_fast, fast = run(0)
    f_slow, slow = run(cycles)
    assert torch.equal(f_fast, f_slow)
    assert fast.source == 1 and slow.source == 1
    assert fast.form == 0, (fast.in_order_ms, fast.two_stream_ms)
    assert slow.form == 1, (slow.in_order_ms, slow.two_stream_ms)
    assert slow.in_order_ms > fast.in_order_ms + 0.05                      # the delay is on the critical path of the in-order step ...
    assert slow.two_stream_ms < slow.in_order_ms                           # ... and beside the interior launch in the other form
    op.destroy()


@pytest.mark.timeout(300)
def test_rccl_self_exchange_one_rank(rdyhip_kernel):
    """RCCL itself, as far as one device allows: a one-rank communicator (rdyhip_comm_init_rank) and a halo whose only
    peer is this rank -- ncclSend / ncclRecv to self inside one group on the library's stream -- moves the packed cells
    into the "ghost" rows; then the overlapped RHS through that halo equals the plain RHS"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    import ctypes as C
    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    lib = _lib.load()
    torch.cuda.set_device(0)
    assert lib.rdyhip_rccl_version() > 20000
    K = 2 * np.pi / 37
    mesh = M.structured_tri_mesh(48, 40, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=8)
    case = CS.friction_slope_case(mesh, 48.0, 40.0, dt=1e-2, K=K)
    op = CS.create_operator(case)
    uid = C.create_string_buffer(128)
    _lib.check(lib.rdyhip_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
    n = 500
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    send, recv = i32(np.arange(0, n)), i32(np.arange(2000, 2000 + n))      # "owned" cells 0..499 -> "ghost" rows 2000..2499
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    h = C.c_void_p()
    _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, p(i32([0])), p(i32([n])), p(send), p(i32([n])), p(recv), C.byref(h)))
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda:0")
    v = u.clone()
    st = int(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rdyhip_halo_exchange(h, int(v.data_ptr()), 3, st))
    torch.cuda.synchronize()
    expect = u.clone()
    expect[2000:2000 + n] = u[0:n]
    assert torch.equal(v, expect)
    # overlapped RHS through the same halo: the exchange rewrites rows 2000.. of u before the halo phase; with no ghost
    # cells in the mesh every tile is an interior tile, which reads u while the exchange writes it -- so feed a state whose
    # rows 2000.. already equal rows 0.. (the exchange then rewrites identical values)
    w = expect.clone()
    f1 = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda:0")
    f2 = torch.empty_like(f1)
    _lib.check(lib.rdyhip_rhs_overlapped(op._h, h, case.dt, int(w.data_ptr()), int(f1.data_ptr()), st))
    op.rhs_function(case.dt, expect, f2)
    torch.cuda.synchronize()
    assert torch.equal(w, expect) and torch.equal(f1, f2)
    # rows 2000.. are consecutive: the transfer above landed in the array itself (direct receive).  The fused pack over RCCL:
    # four ping-pong Euler steps whose packs ride on the kernels = the same steps with a pack launch each
    assert lib.rdyhip_halo_direct_receive(h) == 1 and lib.rdyhip_halo_pack_fused(h) == 0
    res = []
    for fuse in (1, 0):
        _lib.check(lib.rdyhip_halo_fuse_pack(h, fuse))
        assert lib.rdyhip_halo_pack_fused(h) == fuse
        a, b = expect.clone(), torch.empty_like(expect)
        for _ in range(4):
            _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h, 0.1 * case.dt, int(a.data_ptr()), int(b.data_ptr()), None, st))
            a, b = b, a
        torch.cuda.synchronize()
        res.append(a.clone())
    assert bool(torch.isfinite(res[0]).all()) and torch.equal(res[0], res[1])
    # a state written behind the library's back must be announced: without rdyhip_halo_invalidate the stale send rows travel
    _lib.check(lib.rdyhip_halo_fuse_pack(h, 1))
    a, b = expect.clone(), torch.empty_like(expect)
    _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h, 0.0, int(a.data_ptr()), int(b.data_ptr()), None, st))   # dt = 0: b = a on the owned rows
    torch.cuda.synchronize()
    b[0:n] += 1.0                                             # the host edits the cells that are sent
    _lib.check(lib.rdyhip_halo_invalidate(h))
    _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h, 0.0, int(b.data_ptr()), int(a.data_ptr()), None, st))
    torch.cuda.synchronize()
    assert torch.equal(b[2000:2000 + n], b[0:n])              # the exchange of the second step carried the edited rows
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
    _lib.check(lib.rdyhip_comm_destroy(comm))
    op.destroy()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,second_order", [("strips", False), ("rcb_houston", False), ("strips", True), ("rcb_houston", True)])
def test_step_forms_self_exchange(rdyhip_kernel, kind, second_order):
    """The forms of the multi-rank step (include/rdyhip.h) give the same bits.  One rank's part of a partitioned mesh (real ghost
    rows), the exchange looped back through a one-rank RCCL communicator (its boundary cells travel to its own ghost rows):
    twelve Euler steps -- with an RHS evaluation and a host edit of the state in between -- (a) with a pack launch per step, in
    order; (b) fused pack, in order and on two streams; (c) no pack fused, two streams; (d) the halo left to itself: its first
    sixteen steps of a kind alternate between the forms (the trial), then the faster one stays -- and rdyhip_halo_form_info
    says which, with both timings."""
    if rdyhip_kernel == "cell":
        pytest.skip("the fused pack rides on the tiled kernels")
    import ctypes as C
    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    lib = _lib.load()
    torch.cuda.set_device(0)
    if kind == "strips":
        nxp, ny, world = 160, 96, 3                         # 30 720 owned cells = 120 tiles: eight XCD chunks
        K = 2 * np.pi / 37
        mesh = M.strip_partition_tri_mesh(nxp, ny, 1, world, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=8)
        case = CS.friction_slope_case(mesh, nxp * world, ny, dt=1e-2, K=K)
    else:
        case = CS.houston_refined_case(os.path.join(ROOT, "tests", "golden", "houston"), 3, "hilbert", rank=2, world=5)
        mesh = case.mesh
    case.config.second_order = second_order
    op = CS.create_operator(case)
    uid = C.create_string_buffer(128)
    _lib.check(lib.rdyhip_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
    gset = np.zeros(mesh.num_cells, dtype=bool)
    gset[ghost] = True
    cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
    cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
    sendc = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut])).astype(np.int32)
    n = min(sendc.size, ghost.size)
    assert n > 50
    sendc, ghost = i32(sendc[:n]), i32(ghost[:n])
    st = int(torch.cuda.current_stream().cuda_stream)
    dts = 0.1 * case.dt
    f = torch.empty((mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda:0")

    def form_info(h, kind_):
        info = _lib.RDyHipHaloFormInfo()
        _lib.check(lib.rdyhip_halo_form_info(h, kind_, C.byref(info)))
        return info

    def run(fuse, overlap, steps_after=4):
        if overlap is not None:
            os.environ["RDYHIP_OVERLAP"] = overlap
        try:
            h = C.c_void_p()
            _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, p(i32([0])), p(i32([n])), p(sendc), p(i32([n])), p(ghost), C.byref(h)))
            _lib.check(lib.rdyhip_halo_fuse_pack(h, fuse))
        finally:
            os.environ.pop("RDYHIP_OVERLAP", None)
        assert lib.rdyhip_halo_pack_fused(h) == fuse
        if overlap is not None:
            assert form_info(h, 1).source == 2 and form_info(h, 1).form == int(overlap) == lib.rdyhip_halo_overlaps(h)
        a = torch.tensor(case.u_local, dtype=torch.float64, device="cuda:0")
        b = a.clone()
        step = lambda x, y: _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h, dts, int(x.data_ptr()), int(y.data_ptr()), None, st))
        for _ in range(5):
            step(a, b)
            a, b = b, a
        _lib.check(lib.rdyhip_rhs_overlapped(op._h, h, dts, int(a.data_ptr()), int(f.data_ptr()), st))     # an RHS in between: packs again
        fa = f.clone()
        for _ in range(3):
            step(a, b)
            a, b = b, a
        torch.cuda.synchronize()
        a[torch.as_tensor(sendc[:7].astype(np.int64), device="cuda:0")] *= 1.01    # the host edits cells that are sent ...
        _lib.check(lib.rdyhip_halo_invalidate(h))                                  # ... and says so
        for _ in range(steps_after):
            step(a, b)
            a, b = b, a
        torch.cuda.synchronize()
        info = form_info(h, 1)
        _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
        return a.clone(), fa, info

    ref, fref, _ = run(0, "0")
    own = torch.as_tensor(mesh.cell_owned_to_local.astype(np.int64), device="cuda:0")
    assert bool(torch.isfinite(ref[own]).all()) and bool(torch.isfinite(fref).all())
    for fuse, overlap in ((1, "0"), (1, "1"), (0, "1"), (1, None), (0, None)):
        got, fgot, info = run(fuse, overlap)
        assert torch.equal(got, ref) and torch.equal(fgot, fref), (fuse, overlap)
        if overlap is None:
            assert info.source == 0 and info.trial_steps == 12           # twelve Euler steps so far: the trial is still running
    # ... and through to its end: 16 steps alternate, the 17th reads the timings and settles
    ref2, _, _ = run(0, "0", steps_after=12)
    got2, _, info = run(1, None, steps_after=12)
    assert torch.equal(got2, ref2)
    assert info.source == 1 and info.trial_steps == 16 and info.in_order_ms > 0.0 and info.two_stream_ms > 0.0, (info.source, info.trial_steps)
    assert info.form == (1 if info.two_stream_ms < info.in_order_ms else 0)
    _lib.check(lib.rdyhip_comm_destroy(comm))
    op.destroy()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("second_order", [False, True])
def test_send_cells_outside_the_ghost_adjacent_tiles(rdyhip_kernel, second_order):
    """the DM's overlap is vertex-adjacent (DMPlexDistributeOverlap, src/rdydm.c:150), so a cell that touches
    another rank through a vertex only is a send cell -- and can sit in a tile no ghost touches, which the INTERIOR launch of a
    two-stream step runs beside the transfer that reads the send buffer.  A halo whose send list holds such cells keeps its
    fused-pack Euler steps in order (rdyhip_halo_form_info: locked), whatever RDYHIP_OVERLAP says, and gives the bits of the
    steps with a pack launch each."""
    if rdyhip_kernel == "cell":
        pytest.skip("the fused pack rides on the tiled kernels")
    import ctypes as C
    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    lib = _lib.load()
    torch.cuda.set_device(0)
    nxp, ny, world = 160, 96, 3
    K = 2 * np.pi / 37
    mesh = M.strip_partition_tri_mesh(nxp, ny, 1, world, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=8)
    case = CS.friction_slope_case(mesh, nxp * world, ny, dt=1e-2, K=K)
    case.config.second_order = second_order
    op = CS.create_operator(case)
    info = op.layout_info()
    assert 0 < info["num_halo_tiles"] < info["num_tiles"] // 4
    uid = C.create_string_buffer(128)
    _lib.check(lib.rdyhip_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.rdyhip_comm_init_rank(1, 0, uid.raw, C.byref(comm)))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    ghost = np.nonzero(mesh.cell_is_owned == 0)[0].astype(np.int32)
    # send cells: the edge-adjacent ones AND owned cells from the middle of the strip (what a vertex-adjacent overlap may add)
    gset = np.zeros(mesh.num_cells, dtype=bool)
    gset[ghost] = True
    cl, cr = mesh.edge_cell_ids[0::2], mesh.edge_cell_ids[1::2]
    cut = (cr >= 0) & (gset[cl] != gset[np.maximum(cr, 0)])
    edge_adj = np.unique(np.where(gset[cl[cut]], cr[cut], cl[cut]))
    xc = mesh.cell_centroids[:, 0]
    mid = np.nonzero((mesh.cell_is_owned != 0) & (np.abs(xc - xc[mesh.cell_is_owned != 0].mean()) < 2.0))[0][:40]
    sendc = np.concatenate([edge_adj, mid]).astype(np.int32)
    n = min(sendc.size, ghost.size)
    sendc, ghostr = i32(np.concatenate([mid, edge_adj])[:n]), i32(ghost[:n])
    st = int(torch.cuda.current_stream().cuda_stream)
    dts = 0.1 * case.dt

    def run(fuse, overlap):
        os.environ["RDYHIP_OVERLAP"] = overlap
        try:
            h = C.c_void_p()
            _lib.check(lib.rdyhip_halo_create(op._h, comm, 1, p(i32([0])), p(i32([n])), p(sendc), p(i32([n])), p(ghostr), C.byref(h)))
            _lib.check(lib.rdyhip_halo_fuse_pack(h, fuse))
        finally:
            os.environ.pop("RDYHIP_OVERLAP")
        fi = _lib.RDyHipHaloFormInfo()
        _lib.check(lib.rdyhip_halo_form_info(h, 1, C.byref(fi)))
        a = torch.tensor(case.u_local, dtype=torch.float64, device="cuda:0")
        b = a.clone()
        for _ in range(6):
            _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h, dts, int(a.data_ptr()), int(b.data_ptr()), None, st))
            a, b = b, a
        torch.cuda.synchronize()
        _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
        return a.clone(), (fi.form, fi.source)

    ref, _ = run(0, "0")
    got, form = run(1, "1")                       # two streams asked for: refused for this pattern
    assert form == (0, 4), form                   # in order, RDYHIP_HALO_FORM_LOCKED_IN_ORDER
    assert torch.equal(got, ref)
    got, form = run(0, "1")                       # without the fused pack the two-stream form is fine (the pack launch runs first)
    assert form == (1, 2) and torch.equal(got, ref)
    _lib.check(lib.rdyhip_comm_destroy(comm))
    op.destroy()


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks(rdyhip_kernel):
    """`python bench.py --gpus 2` started as a plain command launches its own ranks (rdycore_amd/launch.py) and prints
    one line; BENCH_BACKEND=gloo lets both ranks share this box's one GPU"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    import json
    import subprocess
    env = dict(os.environ, BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    for extra, driver in (([], "torch"), (["--halo", "c"], "c"), (["--workload", "c5", "--nx", "160", "--ny", "160", "--halo", "c"], "c"),
                          (["--gpus", "5", "--workload", "houston_refined", "--levels", "3", "--halo", "c"], "c")):
        n = 5 if "houston_refined" in extra else 2
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--nx", "200", "--ny", "200",
               "--condition-seconds", "0.2", "--watchdog-seconds", "150", "--launch-timeout", "200"] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["config"]["world_size"] == n and d["config"]["backend"] == "gloo" and d["config"]["finite"] is True
        assert d["config"]["rccl_ranks"] is None                 # gloo rehearsal: the bytes do not travel over RCCL, and the line says so
        assert d["scaling"] == ("strong" if ("c5" in extra or "houston_refined" in extra) else "weak") and d["value"] > 0
        assert d["config"]["halo_driver"] == driver and d["config"]["halo_bytes_per_rank"] > 0
        assert d["config"]["max_courant"] > 0 and d["config"]["max_courant_cell"] >= 0
        # first-contact evidence: what every rank did, how long each setup stage took, the self-check ran (its two stages are timed)
        pr = d["config"]["per_rank"]
        assert [r["rank"] for r in pr] == list(range(n)) and all(r["cells"] > 0 and r["ghost_cells"] > 0 and r["peers"] >= 1 for r in pr)
        if driver == "c":
            assert all(r["direct_receive"] is True and r["halo_overlapped"] in (0, 1) for r in pr)
        st = d["config"]["setup_stages"]
        assert {"mesh_and_state_s", "operator_create_s", "plan_s", "first_exchange_s", "first_overlapped_step_s"} <= set(st)
    # the self-check's failure path: one rank expects a wrong ghost value -> ONE line {"error": ...}, exit code 3, no hang
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "5", "--warmup", "2", "--nx", "120", "--ny", "100",
           "--condition-seconds", "0", "--watchdog-seconds", "150", "--launch-timeout", "200", "--halo", "c", "--inject-fault", "exchange"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "exchange self-check failed" in json.loads(lines[0])["error"], (out.stdout, out.stderr[-2000:])
    assert "ghost cells differ" in json.loads(lines[0])["per_rank"][2]


@pytest.mark.timeout(600)
def test_bench_under_torch_distributed_run(rdyhip_kernel):
    """the driver's other way of starting N ranks: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (RANK / WORLD_SIZE come from the launcher, bench.py must not start ranks of its own): one line, from rank 0"""
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    import json
    import subprocess
    from rdycore_amd.launch import free_port
    env = dict(os.environ, BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--nx", "200", "--ny", "200", "--condition-seconds", "0.2", "--watchdog-seconds", "150"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["finite"] is True and d["value"] > 0


@pytest.mark.timeout(300)
def test_three_rcb_ranks_one_gpu_second_order_quads(rdyhip_kernel):
    """second order on RCB parts of the dam-break quad mesh: irregular halos (a first ring that touches ghosts on two
    sides of a part), the quads' three-round flux layout, the gradient exchange -- against the oracle on the undivided mesh"""
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    for rank, err, cerr, ids_ok, nhalo_tiles, ntiles in _run(3, (rdyhip_kernel, True, "c", "rcb_quads")):
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-10 and ids_ok
        assert 0 < nhalo_tiles < ntiles


def test_fused_pack_lifetime_and_argument_errors(rdyhip_kernel):
    """rdyhip_halo_fuse_pack: one halo per operator holds it, second order and the cell-centric kernel refuse it, a halo without
    peers may hold it (empty lists), and a halo that outlives its operator is destroyed without touching freed state"""
    import ctypes as C
    from rdycore_amd import _lib
    from rdycore_amd import cases as CS
    from rdycore_amd import mesh as M
    lib = _lib.load()
    torch.cuda.set_device(0)
    K = 2 * np.pi / 37
    mesh = M.structured_tri_mesh(48, 40, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=8)
    case = CS.friction_slope_case(mesh, 48.0, 40.0, dt=1e-2, K=K)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_lib.c_int32_p)

    def make_halo(op, n):
        h = C.c_void_p()
        send, recv = i32(np.arange(0, n)), i32(np.arange(2000, 2000 + n))
        _lib.check(lib.rdyhip_halo_create(op._h, None, 1 if n else 0, p(i32([0])), p(i32([n])), p(send), p(i32([n])), p(recv), C.byref(h)))
        return h

    op = CS.create_operator(case)
    h1, h2, h0 = make_halo(op, 300), make_halo(op, 200), make_halo(op, 0)
    if rdyhip_kernel == "cell":
        assert lib.rdyhip_halo_fuse_pack(h1, 1) == 83 and b"tiled" in lib.rdyhip_last_error()
    else:
        assert lib.rdyhip_halo_fuse_pack(h1, 1) == 0 and lib.rdyhip_halo_pack_fused(h1) == 1
        assert lib.rdyhip_halo_fuse_pack(h1, 1) == 0                                   # idempotent
        assert lib.rdyhip_halo_fuse_pack(h2, 1) == 83 and b"another halo" in lib.rdyhip_last_error()
        assert lib.rdyhip_halo_fuse_pack(h1, 0) == 0 and lib.rdyhip_halo_pack_fused(h1) == 0
        assert lib.rdyhip_halo_fuse_pack(h0, 1) == 0                                   # no peers: empty send lists
        # an Euler step with the (empty) lists attached is the plain step
        u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda:0")
        a, b = torch.empty_like(u), torch.empty_like(u)
        st = int(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.rdyhip_euler_step_overlapped(op._h, h0, case.dt, int(u.data_ptr()), int(a.data_ptr()), None, st))
        op.euler_step(case.dt, u, b)
        torch.cuda.synchronize()
        assert torch.equal(a[:mesh.num_owned_cells], b[:mesh.num_owned_cells])
        assert lib.rdyhip_halo_fuse_pack(h0, 0) == 0 and lib.rdyhip_halo_fuse_pack(h2, 1) == 0
    assert lib.rdyhip_halo_invalidate(None) == 83 and lib.rdyhip_halo_fuse_pack(None, 1) == 83
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h1)))
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h0)))
    op.destroy()                                       # h2 (holding the fused pack on the tiled kernel) outlives its operator
    _lib.check(lib.rdyhip_halo_destroy(C.byref(h2)))
    # second order packs in its Euler-step kernel too
    case.config.second_order = True
    if rdyhip_kernel != "cell":
        op2 = CS.create_operator(case)
        h = make_halo(op2, 100)
        assert lib.rdyhip_halo_fuse_pack(h, 1) == 0 and lib.rdyhip_halo_pack_fused(h) == 1
        _lib.check(lib.rdyhip_halo_destroy(C.byref(h)))
        op2.destroy()
