#!/usr/bin/env python3
"""Code size (bytes of machine code), registers and LDS of every gfx950 kernel in the built library.

    python tools/kernel_code_sizes.py [path/to/libs2svc_hip.so] [--filter substr] [--min-bytes N]

Why this matters here: a kernel of the training chain runs once per launch with ONE workgroup per CU on an instruction cache
that is invalidated at every dispatch, so straight-line code is fetched cold -- measured ~0.45 us per KB executed
(profiles/r02_xcd_barrier_bench.txt, profiles/r03_fused_layers_phase_breakdown.txt).  A 30 KB kernel cannot run in less than
~14 us whatever its loads do; this table is the first thing to look at for a launch-bound kernel.

The library holds one clang offload bundle per translation unit in its `.hip_fatbin` section; the script cuts out the gfx950
code objects and reads their symbol tables (pure Python ELF parsing, no ROCm tool needed).
"""
import argparse
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def bundles(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def elf_funcs(elf):
    """(name, size) of every STT_FUNC symbol of an ELF64 little-endian image"""
    if elf[:4] != b"\x7fELF":
        return []
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    out = []
    for s in secs:
        if s[1] != 2:                       # SHT_SYMTAB
            continue
        strtab = secs[s[6]]
        for i in range(s[5] // 24):
            name, info, _other, _shndx, _value, size = struct.unpack_from("<IBBHQQ", elf, s[4] + i * 24)
            if info & 0xF == 2 and size:    # STT_FUNC
                e = elf.index(b"\0", strtab[4] + name)
                out.append((elf[strtab[4] + name:e].decode(), size))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")
    except Exception:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "seq2seq_vc_amd", "csrc", "libs2svc_hip.so"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--min-bytes", type=int, default=0)
    a = ap.parse_args()
    blob = open(a.lib, "rb").read()
    rows = {}
    for elf in bundles(blob):
        for name, size in elf_funcs(elf):
            rows[name] = size
    names = sorted(rows, key=lambda n: -rows[n])
    pretty = demangle(names)
    print(f"# {a.lib}: {len(rows)} kernels, {sum(rows.values()) / 1024:.0f} KB of gfx950 code")
    print(f"{'bytes':>8s}  {'~cold us':>8s}  kernel")
    for n, p in zip(names, pretty):
        if rows[n] < a.min_bytes or (a.filter and a.filter not in p):
            continue
        p = p.replace("(anonymous namespace)::", "")
        cut = p.find("(")
        print(f"{rows[n]:8d}  {rows[n] / 1024 * 0.45:8.1f}  {p[:cut] if cut > 0 else p}")


if __name__ == "__main__":
    sys.exit(main())
