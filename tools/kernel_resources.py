#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code object inside librt_amd.so: VGPRs, SGPRs, spills, private (scratch) bytes
per lane, LDS -- read from the code object's own metadata (the AMDGPU msgpack note), not from a compiler log, so it
describes the library that is actually loaded.

    python tools/kernel_resources.py [path/to/librt_amd.so] [substring ...]

Also the source of the stale-profile guard: `fingerprint(so, kernel)` = kernel symbol + its register / spill / scratch
figures + a hash of that kernel's machine code.  profiles/pmc_traffic.json records it when the PMC passes are summarised;
bench.py quotes those counters only while the loaded library still has that fingerprint.
"""
import hashlib
import os
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SO = os.path.join(ROOT, "ray-tracing_amd", "librt_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_elfs(so_path=DEFAULT_SO, arch="gfx950"):
    """Every device ELF for `arch` in the library's .hip_fatbin: one (uncompressed) clang offload bundle per .hip translation
    unit (csrc/rt_kernels.hip, csrc/rt_tools.hip)."""
    blob = open(so_path, "rb").read()
    out = []
    base = blob.find(MAGIC)
    while base >= 0:
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tlen].decode()
            off += tlen
            if arch in triple and size:
                out.append(blob[base + o: base + o + size])
        base = blob.find(MAGIC, base + len(MAGIC))
    if not out:
        raise RuntimeError(f"{so_path}: no {arch} code object in an uncompressed clang offload bundle")
    return out


def device_elf(so_path=DEFAULT_SO, arch="gfx950"):
    """The device ELF that holds the render kernels (the one with k_stream in its symbol table)."""
    elfs = device_elfs(so_path, arch)
    for e in elfs:
        if b"k_stream" in e:
            return e
    return elfs[0]


def _notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not an ELF64"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:        # SHT_NOTE
            continue
        o, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = o, o + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernel_table(so_path=DEFAULT_SO):
    """{demangled-ish kernel symbol: dict(vgpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_bytes, max_flat_wg)}"""
    elf = device_elf(so_path)
    for name, ntype, desc in _notes(elf):
        if name == b"AMDGPU" and ntype == 32:
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            out = {}
            for k in md.get("amdhsa.kernels", []):
                out[k[".name"]] = dict(vgpr=k.get(".vgpr_count"), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count"),
                                       vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                       scratch_bytes=k.get(".private_segment_fixed_size", 0),
                                       lds_bytes=k.get(".group_segment_fixed_size", 0), max_flat_wg=k.get(".max_flat_workgroup_size"))
            return out
    raise RuntimeError("no AMDGPU metadata note in the device ELF")


def kernel_code_hashes(so_path=DEFAULT_SO):
    """{kernel symbol: sha256[:16] of its machine code} from the device ELF's symbol table (STT_FUNC symbols)."""
    elf = device_elf(so_path)
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, sh)
        secs.append(dict(type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    out = {}
    for sec in secs:
        if sec["type"] != 2:        # SHT_SYMTAB
            continue
        strtab = secs[sec["link"]]
        for k in range(sec["size"] // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, sec["off"] + 24 * k)
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= len(secs):     # STT_FUNC, defined
                continue
            end = elf.index(b"\0", strtab["off"] + st_name)
            name = elf[strtab["off"] + st_name:end].decode()
            text = secs[st_shndx]
            o = text["off"] + (st_value - text["addr"])
            out[name] = hashlib.sha256(elf[o:o + st_size]).hexdigest()[:16]
    return out


def fingerprint(so_path=DEFAULT_SO, kernel_substr="k_stream"):
    """What a committed PMC summary is valid for: every kernel whose symbol contains kernel_substr, with its resources and
    the hash of its machine code (other kernels of the library may change without making the summary stale)."""
    tab = kernel_table(so_path)
    code = kernel_code_hashes(so_path)
    return {"kernels": {n: dict(v, code_sha16=code.get(n)) for n, v in sorted(tab.items()) if kernel_substr in n}}


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else DEFAULT_SO
    pats = [a for a in sys.argv[1:] if a != so]
    tab = kernel_table(so)
    print(f"{so}: device ELF sha16 {hashlib.sha256(device_elf(so)).hexdigest()[:16]}")
    print(f"{'kernel':70s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'lds':>7s}")
    for name, r in sorted(tab.items()):
        if pats and not any(p in name for p in pats):
            continue
        print(f"{name[:70]:70s} {r['vgpr']:5d} {r['sgpr']:5d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch_bytes']:8d} {r['lds_bytes']:7d}")
