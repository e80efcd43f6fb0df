"""The machine code of the kernels in the built library, without a GPU and without binutils: librdyhip.so (ELF) -> its
.hip_fatbin section (clang offload bundle) -> the gfx950 code object (ELF) -> the bytes of every kernel function.

Used to tie a measurement to the code it was taken on: profiles/traffic.json stores, per workload, the hash of the measured
kernel's bytes (`code_sha`), and bench.py reports the PMC traffic figure only while the library it runs has the same bytes for
that kernel.  A change to one instantiation (say the Euler-step variants) leaves the entries of the others valid, a change of
compiler or flags invalidates all of them -- which the hash of the SOURCES (round 3's guard) got wrong both ways."""
from __future__ import annotations

import hashlib
import os
import shutil
import struct
import re
import subprocess

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "gfx950"


def _elf_sections(blob: bytes):
    """{name: (offset, size, addr)} and the raw section header tuples of a little-endian ELF64 image"""
    if blob[:4] != b"\x7fELF" or blob[4] != 2 or blob[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    heads = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
        heads.append((name, typ, addr, off, size, link, entsize))
    stroff, strsize = heads[shstrndx][3], heads[shstrndx][4]
    strtab = blob[stroff:stroff + strsize]
    cstr = lambda tab, o: tab[o:tab.index(b"\0", o)].decode()
    return {cstr(strtab, h[0]): h for h in heads}, cstr


def device_code_object(lib: str, target: str = TARGET) -> bytes:
    with open(lib, "rb") as fh:
        blob = fh.read()
    secs, _ = _elf_sections(blob)
    if ".hip_fatbin" not in secs:
        raise ValueError(f"{lib} has no .hip_fatbin section")
    _, _, _, off, size, _, _ = secs[".hip_fatbin"]
    fat = blob[off:off + size]
    pos = fat.find(BUNDLE_MAGIC)
    if pos < 0:
        raise ValueError("no clang offload bundle in .hip_fatbin (compressed bundles are not handled)")
    fat = fat[pos:]
    n, = struct.unpack_from("<Q", fat, len(BUNDLE_MAGIC))
    p = len(BUNDLE_MAGIC) + 8
    for _ in range(n):
        eoff, esize, idlen = struct.unpack_from("<QQQ", fat, p)
        ident = fat[p + 24:p + 24 + idlen].decode()
        p += 24 + idlen
        if ident.startswith("hip") and ident.rstrip("-").endswith(target) and esize:
            return fat[eoff:eoff + esize]
    raise ValueError(f"no {target} code object in {lib}")


def kernel_bytes(lib: str) -> dict:
    """{mangled name: code bytes} of every function the code object's symbol table gives a size for"""
    co = device_code_object(lib)
    secs, cstr = _elf_sections(co)
    _, _, _, symoff, symsize, link, entsize = secs[".symtab"]
    by_index = {}
    shoff, = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", co, 0x3A)
    for i in range(shnum):
        by_index[i] = struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize)
    strh = by_index[link]
    strtab = co[strh[4]:strh[4] + strh[5]]
    out = {}
    for i in range(symsize // entsize):
        name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", co, symoff + i * entsize)
        if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= shnum:      # STT_FUNC with a body
            continue
        sec = by_index[shndx]
        start = sec[4] + (value - sec[3])
        out[cstr(strtab, name)] = co[start:start + size]
    return out


def _demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not (shutil.which(tool) or os.path.exists(tool)):
        raise RuntimeError("no C++ demangler (c++filt / llvm-cxxfilt) on this machine")
    res = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, res))


_cache = {}


def kernel_hashes(lib: str) -> dict:
    """{demangled kernel name: first 16 hex digits of the SHA-256 of its code bytes}"""
    key = (lib, os.path.getmtime(lib))
    if key not in _cache:
        code = kernel_bytes(lib)
        names = _demangle(list(code))
        _cache[key] = {names[m]: hashlib.sha256(b).hexdigest()[:16] for m, b in code.items()}
    return _cache[key]


def kernel_sha(lib: str, kernel: str) -> str:
    """hash of the kernel whose demangled name is `kernel` (as rocprofv3 prints it); a prefix is accepted while it is unambiguous"""
    h = kernel_hashes(lib)
    if kernel in h:
        return h[kernel]
    hits = {v for k, v in h.items() if k.startswith(kernel)}
    if len(hits) != 1:
        raise KeyError(f"{len(hits)} different kernels match {kernel!r}")
    return hits.pop()


def kernel_resources(lib: str) -> dict:
    """{demangled kernel name: {vgpr, agpr, sgpr, scratch, lds, vgpr_spills, sgpr_spills}} from the code object's
    NT_AMDGPU_METADATA note (msgpack): what the compiler allocated per kernel -- registers decide how many workgroups share a
    CU, scratch must stay zero in every hot kernel (tests/test_isa_cpu.py pins the headline instantiations)."""
    import msgpack
    co = device_code_object(lib)
    secs, _ = _elf_sections(co)
    out = {}
    for sname, h in secs.items():
        if h[1] != 7:      # SHT_NOTE
            continue
        off, size = h[3], h[4]
        p = off
        while p + 12 <= off + size:
            namesz, descsz, typ = struct.unpack_from("<III", co, p)
            p += 12
            name = co[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = co[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and typ == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = {"vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                                       "scratch": k.get(".private_segment_fixed_size", 0), "lds": k.get(".group_segment_fixed_size", 0),
                                       "vgpr_spills": k.get(".vgpr_spill_count", 0), "sgpr_spills": k.get(".sgpr_spill_count", 0)}
    names = _demangle(list(out))
    return {names[m]: v for m, v in out.items()}


# swe_rhs_tiled_kernel<S, SRC, OVW, HR, EULER, FNT> / swe_rhs_muscl_fused_kernel<S, SRC, OVW, LIM, EULER>: the fifth template
# argument says whether a kernel is the RHS or its Euler-step instantiation (rdyhip_euler_step).  A profiled bench run launches
# both about as often; the profile tools tell them apart by this, never by launch counts.
_EULER_RX = re.compile(r"swe_rhs_(tiled_kernel<\d, \d, (?:true|false), (?:true|false), true, (?:true|false)>|muscl_fused_kernel<\d, \d, (?:true|false), \d, true>)")


def is_euler_step_kernel(name: str) -> bool:
    return bool(_EULER_RX.search(name))
