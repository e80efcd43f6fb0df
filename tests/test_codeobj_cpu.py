"""The guard that ties profiles/traffic.json to the machine code it was measured on (rdycore_amd/codeobj.py), checked on the
built library without a GPU: the parser finds every kernel, the hash of a kernel is stable across calls and differs between
instantiations, and every committed traffic entry names a kernel of THIS build with THESE bytes -- so a change to a measured
kernel (sources, compiler or flags) fails here, on the CPU, until the workload has been profiled again."""
import json
import os
import shutil

import pytest

from rdycore_amd import build, codeobj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("c++filt") is None and not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt"),
                                reason="no C++ demangler")


def test_code_object_is_found_and_every_rhs_kernel_has_a_body():
    lib = build.lib_path()
    co = codeobj.device_code_object(lib)
    assert co[:4] == b"\x7fELF" and len(co) > 1 << 20
    code = codeobj.kernel_bytes(lib)
    assert all(len(b) > 0 for b in code.values())
    h = codeobj.kernel_hashes(lib)
    rhs = [k for k in h if "swe_rhs_tiled_kernel<" in k or "swe_rhs_muscl_fused_kernel<" in k]
    assert len(rhs) == 84 and len({h[k] for k in rhs}) == len(rhs)           # no two instantiations share their code
    assert h == codeobj.kernel_hashes(lib)
    name = "void rdyhip::swe_rhs_tiled_kernel<3, 0, true, false, false, true>(rdyhip::KernelArgs, double, double const*, double*)"
    assert codeobj.kernel_sha(lib, name) == h[name] == codeobj.kernel_sha(lib, name[:70])   # an unambiguous prefix is accepted
    with pytest.raises(KeyError):
        codeobj.kernel_sha(lib, "void rdyhip::swe_rhs_tiled_kernel<3")                      # an ambiguous one is not


def test_traffic_entries_were_measured_on_the_kernels_of_this_build():
    lib = build.lib_path()
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert len(t) >= 18
    stale = []
    for key, ent in t.items():
        pairs = [(ent["kernel"], ent.get("code_sha"))]
        if ent.get("also_in_the_step"):
            pairs.append((ent["also_in_the_step"]["kernel"], ent["also_in_the_step"].get("code_sha")))
        for kernel, sha in pairs:
            try:
                now = codeobj.kernel_sha(lib, kernel)
            except KeyError as exc:
                stale.append((key, str(exc)))
                continue
            if now != sha:
                stale.append((key, kernel[:80], sha, now))
        # the figure itself: corrected counters, both directions, per launch (or per step)
        assert ent["hbm_bytes_per_launch"] == ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"] > 0
    assert not stale, f"profile these workloads again (tools/profile_all.sh): {stale}"


def test_rhs_and_euler_step_kernels_are_told_apart_by_their_template_arguments():
    """the profile tools (tools/make_traffic.py, parse_rocprof.py) pick the RHS kernel or its Euler-step instantiation by name:
    codeobj.is_euler_step_kernel must agree with the fifth template argument of EVERY instantiation in the built library (a
    new template parameter would silently send the traffic of one kernel to the other's entry), and every traffic.json entry
    must name the kind of kernel its key says"""
    h = codeobj.kernel_hashes(build.lib_path())
    rhs = [k for k in h if "swe_rhs_tiled_kernel<" in k or "swe_rhs_muscl_fused_kernel<" in k]
    n_euler = 0
    for k in rhs:
        args = [a.strip() for a in k[k.index("<") + 1:k.index(">")].split(",")]
        assert len(args) == (6 if "tiled" in k else 5), k
        assert codeobj.is_euler_step_kernel(k) == (args[4] == "true"), k
        n_euler += args[4] == "true"
    assert 0 < n_euler < len(rhs)
    assert not codeobj.is_euler_step_kernel("rdyhip::axpy_owned_kernel(int, int const*, double, double const*, double*)")
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, ent in t.items():
        assert codeobj.is_euler_step_kernel(ent["kernel"]) == key.endswith("_euler_step"), key
