#!/usr/bin/env python3
"""usage: tools/spill_map.py file.s [kernel-substring] -- where a kernel's scratch (spill) instructions sit: per source
line (from -gline-tables-only .loc directives) the number of scratch_load / scratch_store instructions, plus the
kernel's register metadata.  Build the .s with: hipcc ... --cuda-device-only -S -gline-tables-only"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "path_queue_kernel"
files = {}
cur_fn = None
loc = None
by_line = collections.Counter()
kinds = collections.Counter()
total = collections.Counter()
for line in open(path, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur_fn = m.group(1)
        continue
    if cur_fn is None or want not in cur_fn:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s*(scratch_(?:load|store)\w*|v_writelane_b32|v_readlane_b32|v_accvgpr_(?:read|write)\w*)', line)
    if m:
        op = m.group(1)
        k = "scratch_load" if op.startswith("scratch_load") else "scratch_store" if op.startswith("scratch_store") else op
        total[k] += 1
        if k.startswith("scratch"):
            by_line[(loc, k)] += 1
print("totals:", dict(total))
agg = collections.defaultdict(lambda: [0, 0])
for (l, k), n in by_line.items():
    agg[l][0 if k == "scratch_load" else 1] += n
for l, (ld, st) in sorted(agg.items(), key=lambda x: (x[0][0], x[0][1]) if x[0] else ("", 0)):
    print(f"{l[0] if l else '?'}:{l[1] if l else 0:5d}  loads {ld:3d}  stores {st:3d}")
