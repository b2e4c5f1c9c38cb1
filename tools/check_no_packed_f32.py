#!/usr/bin/env python3
"""Disassemble every gfx950 code object of the built library and count the packed-fp32 VALU instructions (v_pk_add_f32,
v_pk_mul_f32, v_pk_fma_f32).  The library is built with -fno-slp-vectorize -fno-vectorize because SLP-formed packed fp32 code
gave, rarely and only inside a full training step, a wrong HIGH-half result in the last 16-lane quarter of a partially active
wave (DESIGN.md section 5 "Hazard", tools/repro_spline_slp.py, profiles/r04_repro_spline_slp.txt): the shipped code must hold none.

    python tools/check_no_packed_f32.py [path/to/libs2svc_hip.so]      # prints {"v_pk_*_f32": count, ...}; exit 1 if any
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.kernel_code_sizes import bundles  # noqa: E402

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
PAT = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32)\b")


def count(lib):
    blob = open(lib, "rb").read()
    total, per_kernel = {}, {}
    for elf in bundles(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        cur = "?"
        for line in out.splitlines():
            if line.endswith(">:"):
                cur = line.split("<")[-1][:-2]
            m = PAT.search(line)
            if m:
                total[m.group(1)] = total.get(m.group(1), 0) + 1
                per_kernel[cur] = per_kernel.get(cur, 0) + 1
    return total, per_kernel


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "seq2seq_vc_amd", "csrc", "libs2svc_hip.so")
    total, per_kernel = count(lib)
    print(json.dumps({"library": os.path.relpath(lib, ROOT), "packed_fp32_instructions": total,
                      "kernels": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])[:20])}))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
