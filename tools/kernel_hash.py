"""sha256 of ONE kernel's machine code inside libshine_hip.so (no external tools: the .so's .hip_fatbin section holds
uncompressed __CLANG_OFFLOAD_BUNDLE__ containers whose gfx950 entries are plain ELF64 code objects).

    python tools/kernel_hash.py [lib.so] [symbol substring ...]

Used to tie counter files to code: tools/pmc_to_json.py stamps `kernel_code_sha256` into profiles/r03_pmc_*.json and
bench.py reports the PMC-derived fields only while the library it loaded still contains that exact kernel.
"""
import hashlib
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shine_mapping_amd", "lib",
                           "libshine_hip.so")
# mangled-name fragments of the kernels bench.py quotes counters for: k_step_v3<L, WAVES, EIK, PROF, EXT, MARK, FAR>
# (kitti-large: the far build — tables beyond the Infinity Cache, picked by table size)
STEP_KERNELS = {
    ("maicity", 4): "k_step_v3ILi4ELi8ELb0ELb0ELb0ELb0ELb0EE",
    ("maicity", 3): "k_step_v3ILi3ELi8ELb0ELb0ELb0ELb0ELb0EE",
    ("kitti", 3): "k_step_v3ILi3ELi8ELb1ELb0ELb0ELb0ELb0EE",
    ("kitti-large", 3): "k_step_v3ILi3ELi8ELb1ELb0ELb0ELb0ELb1EE",
}


def _code_objects(blob):
    """yield (triple, bytes) of every bundle entry"""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        base = pos
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        cur = base + len(MAGIC) + 8
        if n > 64:  # not a header (the magic string can also occur as data)
            pos = base + 1
            continue
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            cur += 24
            triple = blob[cur:cur + tlen].decode("ascii", "replace")
            cur += tlen
            yield triple, blob[base + off: base + off + size]
        pos = base + len(MAGIC)


def _elf_functions(elf):
    """{symbol name: code bytes} of the STT_FUNC symbols of an ELF64 little-endian object"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from(
            "<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    out = {}
    for s in secs:
        if s["type"] != 2:  # SHT_SYMTAB
            continue
        strtab = secs[s["link"]]
        for j in range(s["size"] // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, s["off"] + 24 * j)
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx >= len(secs):  # STT_FUNC
                continue
            end = elf.index(b"\0", strtab["off"] + st_name)
            name = elf[strtab["off"] + st_name:end].decode("ascii", "replace")
            sec = secs[st_shndx]
            start = sec["off"] + (st_value - sec["addr"])
            out[name] = elf[start:start + st_size]
    return out


def kernel_code_sha256(fragment, lib_path=None):
    """sha256 (hex) of the machine code of the one gfx950 kernel whose mangled name contains `fragment`; None if the
    library is missing or holds no such kernel; raises if the fragment is ambiguous."""
    lib_path = lib_path or DEFAULT_LIB
    if not os.path.isfile(lib_path):
        return None
    blob = open(lib_path, "rb").read()
    hits = {}
    for triple, obj in _code_objects(blob):
        if "gfx950" not in triple:
            continue
        for name, code in _elf_functions(obj).items():
            if fragment in name:
                hits[name] = hashlib.sha256(code).hexdigest()
    if not hits:
        return None
    if len(hits) > 1:
        raise ValueError("kernel name fragment %r is ambiguous: %s" % (fragment, sorted(hits)))
    return next(iter(hits.values()))


def step_kernel_sha256(workload, levels, lib_path=None):
    frag = STEP_KERNELS.get((workload, int(levels)))
    return kernel_code_sha256(frag, lib_path) if frag else None


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else DEFAULT_LIB
    frags = [a for a in sys.argv[1:] if not a.endswith(".so")] or sorted(set(STEP_KERNELS.values()))
    for f in frags:
        print(f, kernel_code_sha256(f, lib))
