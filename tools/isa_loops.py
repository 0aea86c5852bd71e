#!/usr/bin/env python3
"""usage: tools/isa_loops.py file.s [kernel-substring] [min-instructions] -- the loops of a kernel (a backward branch to an earlier
label) with their static VALU / SALU / LDS / VMEM instruction counts and the source files (.loc) their instructions come from.
A quick way to see what one trip of the traversal loop costs."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "path_queue"
min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files, cur = {}, None
insts = []   # (op, file, line, text)
labels = {}  # label -> index into insts
loc = ("?", 0)
for line in open(path, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r'^(_Z\w+|pine_\w+):', line)
    if m:
        cur = m.group(1)
        continue
    if cur is None or want not in cur:
        continue
    m = re.match(r'^(\.LBB\w+):', line)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z_0-9]+)(\s.*)?$', line)
    if m and not m.group(1).startswith("."):
        insts.append((m.group(1), loc[0], loc[1], line.strip()))
loops = []
for i, (op, f, l, text) in enumerate(insts):
    if op.startswith(("s_cbranch", "s_branch")):
        t = text.split()[-1]
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i))
for a, b in sorted(loops, key=lambda x: (x[1] - x[0])):
    body = insts[a:b + 1]
    if len(body) < min_insts:
        continue
    c = collections.Counter()
    byfile = collections.Counter()
    for op, f, l, _ in body:
        k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        c[k] += 1
        if k == "valu":
            byfile[f] += 1
    lines = sorted({(f, l) for _, f, l, _ in body if f == "pine_trav.h" or f == "pine_queue_kernel.h"})
    span = f"{lines[0][0]}:{lines[0][1]}..{lines[-1][1]}" if lines else ""
    print(f"loop [{a}:{b}] insts {len(body):5d} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}  valu by file {dict(byfile)}  {span}")
