"""The N > 1 path on CPU: two gloo ranks, strip partition, ghost update through
rdycore_amd.halo.HaloExchange, RHS (oracle) equal to the single-rank RHS."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ghosts_mode, q, second_order=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rdycore_amd import cases as CS
        from rdycore_amd import mesh as M
        from rdycore_amd.halo import HaloExchange
        from helpers import oracle_from_case, rel_linf
        nxg, ny = 8 * world, 9
        K = 2 * np.pi / 13
        z = CS.mms_bathymetry(K=K)
        if ghosts_mode == "tail":
            mesh = M.strip_partition_tri_mesh(nxg // world, ny, rank, world, 1.0, zfunc=z)
        else:
            xyz, conn, cqi, cqj = M.structured_tri_connectivity(nxg, ny)
            xyz[:, 2] = z(xyz[:, 0], xyz[:, 1])
            own = (cqi >= rank * (nxg // world)) & (cqi < (rank + 1) * (nxg // world))
            mesh = M.extract_local_mesh(xyz, conn, own, boundary_classifier=M.box_side_boundaries(0, nxg, 0, ny),
                                        ghosts="interleaved")
        case = CS.friction_slope_case(mesh, nxg, ny, dt=1e-2, K=K)
        case.config.second_order = second_order
        truth = case.u_local.copy()
        u = torch.tensor(case.u_local)
        ghost = torch.tensor(mesh.cell_is_owned == 0)
        u[ghost] = float("nan")                       # ghosts unknown before the exchange
        halo = HaloExchange(mesh, torch.device("cpu"))
        halo.exchange(u)
        assert torch.equal(u, torch.tensor(truth)), "ghost cells differ from their owners' values"
        if second_order:
            # the HIP path's scheme on the oracle: gradients of the owned cells, ghost gradients through the same
            # HaloExchange (6 values per cell), then every local edge solved here -- no reverse exchange
            from oracle import oracle as O
            cfg = case.config
            orc = O.OracleOperator(mesh, case.condition_types, cfg.tiny_h, cfg.h_anuga_regular, cfg.xq2018_threshold, cfg.source_method,
                                   second_order=True, limiter=cfg.limiter, all_edges_local=True)
            orc.mannings[:] = case.mannings
            orc.external_sources[:] = case.ext_src
            for b, vals in case.boundary_values.items():
                orc.boundary_values[b][:] = vals
            orc.compute_gradients(u.numpy())
            g6 = torch.tensor(orc.gradients6())
            g6[ghost] = float("nan")
            halo.exchange(g6)
            for k in range(3):
                orc.gradients[k][:] = g6[:, 2 * k:2 * k + 2].numpy()
            orc.set_gradients_ready(True)
            f = orc.apply(case.dt, u.numpy())
        else:
            orc = oracle_from_case(case)
            f = orc.apply(case.dt, u.numpy())
        # single-rank answer
        g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
        gc = CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K)
        gc.config.second_order = second_order
        og = oracle_from_case(gc)
        fg = og.apply(gc.dt, gc.u_local)
        gid = mesh.cell_global_ids[mesh.cell_owned_to_local]
        err = rel_linf(f, fg[gid])
        # UpdateOperatorDiagnostics across ranks: the struct-max keeps the ids of the global maximum (src/operator.c:705-715, 879)
        from rdycore_amd.operator import CourantNumberDiagnostics
        from rdycore_amd.timestep import reduce_courant
        red = reduce_courant(CourantNumberDiagnostics(*orc.diagnostics()))
        cg, eg, cellg = og.diagnostics()
        assert red.max_courant_num == cg and cg > 0.0
        assert red.global_cell_id == cellg                           # global cell ids of the undivided mesh = its local ids
        assert red.global_edge_id == M.edge_vertex_key(g, eg)        # partition-independent edge id (mesh.build_mesh)
        q.put((rank, err, halo.bytes_sent_per_exchange, int(ghost.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ghosts_mode,second_order", [("tail", False), ("interleaved", False), ("tail", True)])
@pytest.mark.timeout(180)
def test_two_rank_halo_exchange_and_partitioned_rhs(ghosts_mode, second_order):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ghosts_mode, q, second_order)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    for rank, err, nbytes, nghost in res:
        assert err < 1e-13
        assert nghost == 9 and nbytes == 9 * 24         # ny ghost triangles per side, 3 doubles each


def _worker_rcb(rank, world, port, kind, q):
    """RCB parts of an unstructured mesh: uneven part sizes, ranks with three and more neighbours; the pattern comes from
    rdyhip_halo_plan_* (owner ranks carried by the mesh, or found by asking when it carries none)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rdycore_amd import cases as CS
        from rdycore_amd.halo import HaloExchange
        from helpers import oracle_from_case, rel_linf
        data = os.path.join(ROOT, "tests", "golden", "houston")
        if kind == "houston":
            case = CS.houston_refined_case(data, 2, "hilbert", rank=rank, world=world)
            gc = CS.houston_refined_case(data, 2, "hilbert")
            assert case.mesh.cell_owner_rank is not None
        else:                                     # the C5 miniature: its generator knows only its own part (no owner ranks)
            case = CS.c5_case(CS.c5_mesh(60, 50, rank, world), 60.0, 50.0)
            gc = CS.c5_case(CS.c5_mesh(60, 50), 60.0, 50.0)
            case.mesh.cell_owner_rank = None      # as a mesh cut without a part array at hand: the owners are found by asking
        mesh = case.mesh
        truth = case.u_local.copy()
        u = torch.tensor(case.u_local)
        ghost = torch.tensor(mesh.cell_is_owned == 0)
        u[ghost] = float("nan")
        halo = HaloExchange(mesh, torch.device("cpu"))
        halo.exchange(u)
        assert torch.equal(u, torch.tensor(truth)), "ghost cells differ from their owners' values"
        f = oracle_from_case(case).apply(case.dt, u.numpy())
        fg = oracle_from_case(gc).apply(gc.dt, gc.u_local)
        if kind == "houston":
            key = {tuple(np.round(c[:2], 3)): i for i, c in enumerate(gc.mesh.cell_centroids)}
            rows = np.array([key[tuple(np.round(c[:2], 3))] for c in mesh.cell_centroids[mesh.cell_owned_to_local]])
        else:
            g2row = {int(g): i for i, g in enumerate(gc.mesh.cell_global_ids)}
            rows = np.array([g2row[int(g)] for g in mesh.cell_global_ids[mesh.cell_owned_to_local]])
        # the contiguous-ghost plan: ghosts numbered peer by peer in arrival order (mesh.extract_local_mesh) make the receive
        # list one run of consecutive rows, which is what rdyhip_halo_create needs to receive in place
        in_place = bool(np.array_equal(halo._plan_recv_cells, np.arange(mesh.num_owned_cells, mesh.num_cells)))
        q.put((rank, rel_linf(f, fg[rows]), len(halo.send_ids), len(halo.recv_ids), mesh.num_owned_cells, int(ghost.sum()), in_place))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("houston", 8), ("c5", 3)])
@pytest.mark.timeout(300)
def test_rcb_ranks_through_the_halo_plan(kind, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rcb, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert all(err < 1e-13 for _, err, *_ in res)
    assert all(ns == nr and ns >= 1 for _, _, ns, nr, _, _, _ in res)    # the pattern is symmetric in its peers
    assert all(r[-1] for r in res)                                       # every rank can receive in place
    if world == 8:
        assert max(ns for _, _, ns, *_ in res) >= 3                      # ranks with three and more neighbours
        sizes = [r[4] for r in res]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == 2746 * 16
