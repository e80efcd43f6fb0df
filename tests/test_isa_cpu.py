"""What the compiler made of the kernels, checked on the built library itself (no GPU): the device code object is taken out of
librdyhip.so (.hip_fatbin -> clang-offload-bundler -> llvm-objdump) and every RHS kernel must still carry its non-temporal
stores.  Round 4 shipped, for a few hours, a store helper whose two cache policies sat in the arms of one if / else: the optimiser
sank them into ONE store and dropped the hint from both -- every F / pv store of every kernel, first order -6 % -- and nothing
but a same-box A/B against the previous round's build showed it."""
import os
import re
import shutil
import subprocess

import pytest

from rdycore_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]


@pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS) or shutil.which("c++filt") is None, reason="ROCm's llvm binutils are not installed")
def test_rhs_kernels_keep_their_nontemporal_stores(tmp_path):
    lib = build.lib_path()
    assert os.path.exists(lib)
    fat, co = str(tmp_path / "fatbin.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([TOOLS[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([TOOLS[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    asm = subprocess.run([TOOLS[2], "-d", co], capture_output=True, text=True, check=True).stdout
    kernels = {}
    name = None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = [0, 0]
        elif name and "global_store" in line:
            kernels[name][1 if re.search(r"\bnt\b", line) else 0] += 1
    rhs = {k: v for k, v in kernels.items() if "swe_rhs_tiled_kernel" in k or "swe_rhs_muscl_fused_kernel" in k}
    assert len(rhs) == 84                                    # 48 instantiations of the first-order / HR kernel, 36 of the second-order one
    bad = {k: v for k, v in rhs.items() if v[1] < 6}         # at least F (3 whole-line stores) and the primitive variables (3), hinted
    assert not bad, f"{len(bad)} RHS kernels lost their non-temporal stores, e.g. {list(bad.items())[:3]}"
    # the Euler-step variants also store the new state with the hint (u_out: 3 more)
    demangled = subprocess.run(["c++filt"], input="\n".join(rhs), capture_output=True, text=True, check=True).stdout.splitlines()
    # (the instantiations whose last template argument is false store F -- and, in the Euler-step kernels, u_out -- with the
    # default policy on purpose: a host that reads F straight back, a state that fits the Infinity Cache; they keep pv and fdiv hinted)
    euler = [(k, d) for k, d in zip(rhs, demangled) if re.search(r"swe_rhs_tiled_kernel<\d, \d, true, (true|false), true, (true|false)>", d)]
    hinted = [k for k, d in euler if not re.search(r", false>\(", d)]
    plain = [k for k, d in euler if re.search(r", false>\(", d)]
    assert hinted and all(rhs[k][1] >= 9 for k in hinted), [(k, rhs[k]) for k in hinted if rhs[k][1] < 9][:3]
    assert len(plain) == 8 and all(6 <= rhs[k][1] < 9 for k in plain), [(k, rhs[k]) for k in plain][:3]
