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


def write_case(path, case, overwrite, f_in, f_exp, pv_exp, courant):
    m = case.mesh
    with open(path, "wb") as fh:
        fh.write(struct.pack("8i", m.num_cells, m.num_owned_cells, m.num_edges, m.num_internal_edges, len(m.boundaries),
                             case.config.source_method, 1 if overwrite else 0, 0))
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


def compile_client(tmp_path):
    exe = str(tmp_path / "rdyhip_client")
    libdir = os.path.dirname(build.lib_path())
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "tests", "c_client", "rdyhip_client.c"),
           f"-L{libdir}", "-lrdyhip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


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
