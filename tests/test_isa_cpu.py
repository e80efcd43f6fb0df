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


# What the compiler allocated for the headline instantiations (code object metadata + disassembly of the built library, no GPU):
# registers decide how many workgroups share a CU -- the second-order kernels must stay at <= 128 VGPRs (four workgroups per CU:
# one register more cost an Euler-step instantiation 15 % for an afternoon in round 5), the first-order ones at <= 168 (three) --,
# scratch must be zero, and the number of waits on vector memory says whether a load crept into a place where every wave waits
# for the whole prefetch batch (the first-order kernel has ONE wait per tile on its hot path; the rest sit in prologue, tail and
# cold branches).  Upper bounds with a little slack: a change that moves them is looked at, then the table is updated.
PINS = {
    # kernel (demangled prefix):                                  (max VGPRs, max vmcnt waits, min hinted stores, min hinted loads)
    # measured on the round-5 build: (143, 35, 9, 26), (141, 36, 9, 22), (148, 35, 9, 26), (145, 37, 14, 26), (150, 37, 14, 26),
    #                                (121, 41, 9, 16), (121, 43, 14, 16), (121, 46, 14, 16), (121, 41, 9, 17), (123, 43, 14, 17)
    "swe_rhs_tiled_kernel<3, 0, true, false, false, true>":      (150, 38, 9, 26),
    "swe_rhs_tiled_kernel<3, 0, true, true, false, true>":       (150, 39, 9, 22),
    "swe_rhs_tiled_kernel<4, 0, true, false, false, true>":      (156, 38, 9, 26),
    "swe_rhs_tiled_kernel<3, 0, true, false, true, true>":       (152, 40, 14, 26),
    "swe_rhs_tiled_kernel<4, 0, true, false, true, true>":       (158, 40, 14, 26),
    "swe_rhs_muscl_fused_kernel<3, 0, true, 0, false>":          (128, 44, 9, 16),
    "swe_rhs_muscl_fused_kernel<3, 0, true, 0, true>":           (128, 46, 14, 16),
    "swe_rhs_muscl_fused_kernel<3, 0, true, 2, true>":           (128, 49, 14, 16),
    "swe_rhs_muscl_fused_kernel<4, 0, true, 0, false>":          (128, 44, 9, 17),
    "swe_rhs_muscl_fused_kernel<4, 0, true, 0, true>":           (128, 46, 14, 17),
}


@pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS) or shutil.which("c++filt") is None, reason="ROCm's llvm binutils are not installed")
def test_headline_kernels_keep_their_registers_and_waits(tmp_path):
    from rdycore_amd import codeobj
    lib = build.lib_path()
    res = codeobj.kernel_resources(lib)
    fat, co = str(tmp_path / "fatbin.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([TOOLS[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([TOOLS[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    asm = subprocess.run([TOOLS[2], "-d", co], capture_output=True, text=True, check=True).stdout
    bodies, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            bodies[name] = []
        elif name:
            bodies[name].append(line.split("//")[0].strip())
    names = subprocess.run(["c++filt"], input="\n".join(bodies), capture_output=True, text=True, check=True).stdout.splitlines()
    by_name = dict(zip(names, bodies.values()))
    # every RHS kernel: no scratch, no VGPR spills
    rhs = {k: v for k, v in res.items() if "swe_rhs_tiled_kernel<" in k or "swe_rhs_muscl_fused_kernel<" in k}
    assert len(rhs) == 84
    assert all(v["scratch"] == 0 and v["vgpr_spills"] == 0 for v in rhs.values()), {k: v for k, v in rhs.items() if v["scratch"] or v["vgpr_spills"]}
    # the second-order kernels: four workgroups per CU, all 36 instantiations
    so = {k: v["vgpr"] for k, v in rhs.items() if "swe_rhs_muscl_fused_kernel<" in k}
    assert len(so) == 36 and max(so.values()) <= 128, so
    assert max(v["vgpr"] for v in rhs.values()) <= 168
    bad = {}
    for prefix, (max_vgpr, max_waits, min_nt_st, min_nt_ld) in PINS.items():
        hits = [k for k in by_name if k.startswith("void rdyhip::" + prefix)]
        assert len(hits) == 1, (prefix, hits)
        body = by_name[hits[0]]
        r = res[hits[0]]
        waits = sum(1 for ln in body if ln.startswith("s_waitcnt") and "vmcnt" in ln)
        nt_st = sum(1 for ln in body if ln.startswith("global_store") and re.search(r"\bnt\b", ln))
        nt_ld = sum(1 for ln in body if ln.startswith("global_load") and re.search(r"\bnt\b", ln))
        if r["vgpr"] > max_vgpr or waits > max_waits or nt_st < min_nt_st or nt_ld < min_nt_ld:
            bad[prefix] = {"vgpr": r["vgpr"], "vmcnt_waits": waits, "nt_stores": nt_st, "nt_loads": nt_ld}
    assert not bad, bad
