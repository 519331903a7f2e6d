#!/usr/bin/env python3
"""Extracts the gfx950 code object from a hipcc-built shared library (the clang offload bundle in .hip_fatbin) and, optionally,
disassembles one kernel:  python tools/extract_codeobj.py lib.so out.co [kernel-name-substring]"""
import struct, subprocess, sys
data = open(sys.argv[1], "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(magic)
assert pos >= 0, "no offload bundle"
n, = struct.unpack_from("<Q", data, pos + 24)
p = pos + 32
for _ in range(n):
    off, size, idlen = struct.unpack_from("<QQQ", data, p)
    p += 24
    ident = data[p:p + idlen].decode()
    p += idlen
    if "gfx950" in ident:
        open(sys.argv[2], "wb").write(data[pos + off:pos + off + size])
        print("wrote", sys.argv[2], ident, size)
if len(sys.argv) > 3:
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={sys.argv[3]}", sys.argv[2]],
                         capture_output=True, text=True).stdout
    sys.stdout.write(out)
