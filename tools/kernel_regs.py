#!/usr/bin/env python3
"""usage: tools/kernel_regs.py [lib.so] -- register / spill / scratch table of every path kernel in a built library.
Carves the gfx950 code objects out of the library's offload bundles and reads their metadata notes with llvm-readelf."""
import os
import re
import struct
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pine_amd", "lib", "libpine_gpu.so")
data = open(lib, "rb").read()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
rows = []
pos = 0
with tempfile.TemporaryDirectory() as tmp:
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        if n == 0 or n > 64:  # (the magic as a string literal of the host code, not a bundle)
            pos += 24
            continue
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            if tl > 256 or off > len(data) or size > len(data):
                break
            triple = data[q + 24:q + 24 + tl].decode(errors="replace")
            q += 24 + tl
            if "amdgcn" in triple and size:
                f = os.path.join(tmp, "co.elf")
                open(f, "wb").write(data[pos + off:pos + off + size])
                txt = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
                for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", txt, flags=re.S):
                    name, body = m.group(1), m.group(2)
                    g = lambda k: int(re.search(r"\.%s:\s*(\d+)" % k, body).group(1))
                    rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
        pos += 24
def short(name):
    m = re.search(r"(path_queue_kernel|path_trace_kernel)ILj(\d+)ELi(\d+)", name)
    if m:
        ns = "fast " if "pine_gpu_fast" in name else ""
        return f"{ns}{m.group(1)}<{m.group(2)},{m.group(3)}>"
    return None
print(f"{'kernel':48s} vgpr vgpr_spill sgpr sgpr_spill scratch_bytes_per_lane")
for r in sorted(rows):
    s = short(r[0])
    if s:
        print(f"{s:48s} {r[1]:4d} {r[2]:10d} {r[3]:4d} {r[4]:10d} {r[5]:6d}")
