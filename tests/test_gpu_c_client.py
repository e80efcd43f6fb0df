"""The ABI from a plain C program: tests/c_client/rdyhip_client.c is compiled
with gcc as C11 against include/rdyhip.h, links librdyhip.so and the HIP
runtime, and is checked against oracle results written to a case file."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from rdycore_amd import build
from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from helpers import oracle_from_case


def write_case(path, case, overwrite, f_in, f_exp, pv_exp, courant, courant_cell=-1, owner=None, zc=None):
    m = case.mesh
    with open(path, "wb") as fh:
        fh.write(struct.pack("8i", m.num_cells, m.num_owned_cells, m.num_edges, m.num_internal_edges, len(m.boundaries),
                             case.config.source_method, 1 if overwrite else 0, int(case.config.well_balancing)))
        fh.write(struct.pack("4d", case.config.tiny_h, case.config.h_anuga_regular, case.config.xq2018_threshold, case.dt))
        for a, dt in ((m.cell_is_owned, np.int32), (m.cell_local_to_owned, np.int32), (m.cell_global_ids, np.int64),
                      (m.cell_areas, np.float64), (m.cell_dz_dx, np.float64), (m.cell_dz_dy, np.float64),
                      (m.edge_cell_ids, np.int32), (m.edge_internal_ids, np.int32), (m.edge_global_ids, np.int64),
                      (m.edge_lengths, np.float64), (m.edge_cn, np.float64), (m.edge_sn, np.float64)):
            fh.write(np.ascontiguousarray(a, dtype=dt).tobytes())
        for i, b in enumerate(m.boundaries):
            fh.write(struct.pack("2i", b.num_edges, case.condition_types[i]))
            fh.write(np.ascontiguousarray(b.edge_ids, dtype=np.int32).tobytes())
            fh.write(np.ascontiguousarray(case.boundary_values.get(i, np.zeros((b.num_edges, 3))), dtype=np.float64).tobytes())
        fh.write(np.ascontiguousarray(case.mannings, dtype=np.float64).tobytes())
        fh.write(np.ascontiguousarray(case.ext_src.T, dtype=np.float64).tobytes())     # [comp][owned]
        for a in (case.u_local, f_in, f_exp, pv_exp):
            fh.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        fh.write(struct.pack("d", courant))
        # trailer (multi-rank cases, tests/c_client/case_io.h): expected global cell of the Courant maximum, owner ranks, bed elevations
        fh.write(struct.pack("q2i", int(courant_cell), 1 if owner is not None else 0, 1 if zc is not None else 0))
        if owner is not None:
            fh.write(np.ascontiguousarray(owner, dtype=np.int32).tobytes())
        if zc is not None:
            fh.write(np.ascontiguousarray(zc, dtype=np.float64).tobytes())


def compile_client(tmp_path, name="rdyhip_client"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(build.lib_path())
    cdir = os.path.join(ROOT, "tests", "c_client")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           f"-I{os.path.join(ROOT, 'include')}", f"-I{cdir}", os.path.join(cdir, name + ".c"),
           f"-L{libdir}", "-lrdyhip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_clients_compile_as_c11(tmp_path):
    """every C client of the ABI builds with gcc -std=c11 -Wall -Werror and links librdyhip.so (no GPU needed for that)"""
    for name in ("rdyhip_client", "rdyhip_advance_client", "rdyhip_mr_client"):
        assert os.path.exists(compile_client(tmp_path, name))


def test_header_is_valid_c11(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "rdyhip.h"\nint main(void) { RDyHipConfig c = {1e-7, 0.0, 1e-10, RDYHIP_SOURCE_SEMI_IMPLICIT, RDYHIP_RIEMANN_ROE}; return (int)c.riemann; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", f"-I{os.path.join(ROOT, 'include')}", str(src)])


@pytest.mark.gpu
@pytest.mark.parametrize("overwrite", [True, False])
def test_c_client_matches_oracle(tmp_path, overwrite):
    build.build_native()
    K = 2 * np.pi / 23
    mesh = M.structured_tri_mesh(31, 19, 1.0, zfunc=CS.mms_bathymetry(K=K), order="tiled", tile=4)
    case = CS.friction_slope_case(mesh, 31, 19, dt=1e-2, source_method=1, K=K)
    orc = oracle_from_case(case)
    f_in = np.zeros((mesh.num_owned_cells, 3)) if overwrite else np.random.default_rng(5).normal(size=(mesh.num_owned_cells, 3)) * 0.05
    f_exp = orc.apply(case.dt, case.u_local, f_in.copy())
    path = str(tmp_path / "case.bin")
    write_case(path, case, overwrite, f_in, f_exp, orc.primitive_variables, orc.diagnostics()[0])
    exe = compile_client(tmp_path)
    env = dict(os.environ)
    env.pop("RDYHIP_LIB", None)
    out = subprocess.run([exe, path], capture_output=True, text=True, env=env, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_c_host_runs_rdyadvance_on_ex2b(tmp_path, adaptive):
    """RDyAdvance from a C host (tests/c_client/rdyhip_advance_client.c): ex2b's 1000 steps of dt = 0.018 s (ex2b.yaml:17-20)
    as fused Euler steps on two ping-pong state arrays, and twelve coupling intervals with the adaptive Courant -> dt rule of
    src/rdyadvance.c:303-343 -- final state, step count and final dt = the same loop driven by the oracle"""
    build.build_native()
    case = CS.ex2b_case(os.path.join(ROOT, "tests", "golden", "planar_dam_10x5.msh"))
    no = case.mesh.num_owned_cells
    path, out_path = str(tmp_path / "ex2b.bin"), str(tmp_path / "out.bin")
    write_case(path, case, True, np.zeros((no, 3)), np.zeros((no, 3)), np.zeros((no, 3)), 0.0)
    if adaptive:
        nint, interval, dt0, target, max_inc = 12, 0.2, 0.002, 0.4, 1.5
    else:
        nint, interval, dt0, target, max_inc = 1, 1000 * case.dt, case.dt, 0.5, 2.0
    exe = compile_client(tmp_path, "rdyhip_advance_client")
    env = dict(os.environ)
    env.pop("RDYHIP_LIB", None)
    run = subprocess.run([exe, path, out_path, str(nint), repr(interval), repr(dt0), "1" if adaptive else "0", repr(target), repr(max_inc)],
                         capture_output=True, text=True, env=env, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = open(out_path, "rb").read()
    steps, = struct.unpack("q", raw[:8])
    dt_c, t_c = struct.unpack("2d", raw[8:24])
    u_c = np.frombuffer(raw[24:], dtype=np.float64).reshape(-1, 3)
    # the same loop on the oracle
    orc = oracle_from_case(case)
    u, dt, t, n, cmax = case.u_local.copy(), dt0, 0.0, 0, None
    for _ in range(nint):
        if adaptive and cmax is not None:
            if cmax < target:
                dt = min(dt * min(target / cmax if cmax > 0 else float("inf"), max_inc), interval)
            else:
                dt *= target / cmax
        t_end = t + interval
        while t < t_end * (1.0 - 1e-14):
            h = min(dt, t_end - t)
            orc.reset_diagnostics()
            u = u + h * orc.apply(h, u)
            t += h
            n += 1
        cmax = orc.diagnostics()[0]
    assert steps == n and (n >= 1000 if not adaptive else n > 100)
    assert abs(dt_c - dt) <= 1e-12 * dt and abs(t_c - t) <= 1e-12 * t
    assert np.abs(u[:, 0] - case.u_local[:, 0]).max() > 0.5          # the dam has broken: the pools have levelled
    from helpers import rel_linf
    assert rel_linf(u_c, u) <= 1e-9


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,overlap", [("strips", "0"), ("strips", "1"), ("rcb_c5", "0")])
def test_c_multi_rank_host(tmp_path, kind, overlap, rdyhip_kernel):
    """the multi-rank path from C alone (tests/c_client/rdyhip_mr_client.c: three forked ranks, pipes for the plan's all-to-all
    and for the exchange through rdyhip_halo_set_transport): rdyhip_copy_owned_rows + rdyhip_rhs_overlapped as
    OperatorRHSFunctionHip calls them = the single-rank oracle's rows, ghost rows bit for bit, the Courant struct-max,
    fused-pack Euler steps = RHS + axpy.  Strips (first order, both forms of the step) and RCB parts of the C5 miniature (HR)."""
    if rdyhip_kernel == "cell" and kind == "rcb_c5":
        pytest.skip("hydrostatic reconstruction is implemented by the tiled kernel")
    build.build_native()
    world = 3
    if kind == "strips":
        nxp, ny, K = 40, 48, 2 * np.pi / 37
        z = CS.mms_bathymetry(K=K)
        g = M.structured_tri_mesh(nxp * world, ny, 1.0, zfunc=z)
        gc = CS.friction_slope_case(g, nxp * world, ny, dt=1e-2, K=K)
        parts = [CS.friction_slope_case(M.strip_partition_tri_mesh(nxp, ny, r, world, 1.0, zfunc=z, order="tiled", tile=8), nxp * world, ny, dt=1e-2, K=K)
                 for r in range(world)]
        for c in parts + [gc]:      # a tilt makes the Courant maximum unique (tests/test_gpu_multirank.py)
            xc, yc = c.mesh.cell_centroids[:, 0], c.mesh.cell_centroids[:, 1]
            c.u_local[:, 1] *= 1.0 + 1e-3 * xc / (nxp * world) + 2e-3 * yc / ny
    else:
        gc = CS.c5_case(CS.c5_mesh(120, 100), 120.0, 100.0)
        parts = [CS.c5_case(CS.c5_mesh(120, 100, r, world), 120.0, 100.0) for r in range(world)]
    og = oracle_from_case(gc)
    fg = og.apply(gc.dt, gc.u_local)
    cmax, _, ccell = og.diagnostics()
    g2row = {int(gid): i for i, gid in enumerate(gc.mesh.cell_global_ids)}
    prefix = str(tmp_path / "part")
    for r, c in enumerate(parts):
        m = c.mesh
        assert m.cell_owner_rank is not None
        rows = np.array([g2row[int(gid)] for gid in m.cell_global_ids[m.cell_owned_to_local]])
        no = m.num_owned_cells
        write_case(f"{prefix}.{r}.bin", c, True, np.zeros((no, 3)), fg[rows], np.zeros((no, 3)), cmax, courant_cell=ccell,
                   owner=m.cell_owner_rank, zc=m.cell_zc)
    exe = compile_client(tmp_path, "rdyhip_mr_client")
    env = dict(os.environ, RDYHIP_OVERLAP=overlap)
    env.pop("RDYHIP_LIB", None)
    run = subprocess.run([exe, str(world), prefix], capture_output=True, text=True, env=env, timeout=240)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert run.stdout.count("direct receive 1") == world and run.stdout.count(f"overlapped form {overlap}") == world
    # the fused pack rides on the tiled Euler-step kernels; the cell-centric kernel keeps its pack launch
    assert "courant ok" in run.stdout and run.stdout.count(f"fused pack {0 if rdyhip_kernel == 'cell' else 1}") == world
