"""Several ranks on ONE GPU (gloo transport, device buffers staged through the
host): the rank-local pieces of the N > 1 path -- strip partition, pack/unpack
kernels, side-stream exchange overlapped with the interior tiles, halo tiles
afterwards -- give the single-rank RHS.  RCCL itself cannot be exercised with
several ranks on one device; everything around it is."""
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


def _worker(rank, world, port, kernel, q, second_order=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if kernel:
        os.environ["RDYHIP_KERNEL"] = kernel
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rdycore_amd import cases as CS
        from rdycore_amd import mesh as M
        from rdycore_amd.halo import HaloExchange
        from helpers import oracle_from_case, rel_linf
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        nxp, ny = 40, 48                      # per-rank strip: 3840 owned cells = 15 tiles
        nxg = nxp * world
        K = 2 * np.pi / 37
        z = CS.mms_bathymetry(K=K)
        mesh = M.strip_partition_tri_mesh(nxp, ny, rank, world, 1.0, zfunc=z, order="tiled", tile=8)
        case = CS.friction_slope_case(mesh, nxg, ny, dt=1e-2, K=K)
        case.config.second_order = second_order
        op = CS.create_operator(case)
        halo = HaloExchange(mesh, dev)
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
        halo.rhs_overlapped(op, case.dt, u, f)           # leave the diagnostics of a plain RHS behind
        # single-rank truth from the oracle on the undivided mesh
        g = M.structured_tri_mesh(nxg, ny, 1.0, zfunc=z)
        gc = CS.friction_slope_case(g, nxg, ny, dt=1e-2, K=K)
        gc.config.second_order = second_order
        og = oracle_from_case(gc)
        fg = og.apply(gc.dt, gc.u_local)
        gid = mesh.cell_global_ids[mesh.cell_owned_to_local]
        err = rel_linf(f.cpu().numpy(), fg[gid])
        op.update_diagnostics()
        cmax = torch.tensor([op.get_diagnostics().max_courant_num], dtype=torch.float64)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)      # the MPI_Allreduce of src/operator.c:879
        info = op.layout_info()
        q.put((rank, err, abs(float(cmax) - og.diagnostics()[0]), info["num_halo_tiles"], info["num_tiles"]))
        op.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_ranks_one_gpu_overlapped_rhs(rdyhip_kernel):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rdyhip_kernel, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    for rank, err, cerr, nhalo_tiles, ntiles in res:
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12
        assert 0 < nhalo_tiles < ntiles


@pytest.mark.timeout(300)
def test_three_ranks_one_gpu_second_order(rdyhip_kernel):
    """the second-order path across ranks: state exchange, gradients, gradient exchange (6 values per cell),
    fluxes of interior tiles overlapped, halo tiles afterwards; no reverse exchange"""
    if rdyhip_kernel == "cell":
        pytest.skip("second order is implemented by the tiled kernels")
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rdyhip_kernel, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    for rank, err, cerr, nhalo_tiles, ntiles in res:
        assert err <= 1e-10, (rank, err)
        assert cerr <= 1e-12
