"""The C-ABI library loads and exports every symbol include/rdyhip.h declares
(no compute calls: there is no GPU in the CPU test run)."""
import os
import re

from rdycore_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "rdyhip.h")) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rdyhip_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_for_gfx950_and_exports_every_declared_symbol():
    build.build_native()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"librdyhip.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.rdyhip_version() == 110


def test_code_object_targets_gfx950_only():
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={build.lib_path()}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_argument_errors_without_a_device():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    # null arguments are user errors (PETSC_ERR_USER = 83), reported before any HIP call
    assert lib.rdyhip_create(None, None, 0, None, C.byref(h)) == 83
    assert b"null" in lib.rdyhip_last_error()
    assert lib.rdyhip_pack_cells(None, None, -1, None, None) == 60
    assert lib.rdyhip_apply(None, 0.1, None, None, None) == 83
