#!/usr/bin/env python3
"""usage: tools/isa_lines.py file.s [kernel-substring] [--top N] -- static instruction counts of a kernel per source line
(-gline-tables-only .loc directives): VALU split by issue cost (full rate 2 cycles, half rate 4, quarter 8: tools/valu_rate.hip),
SALU, LDS, VMEM.  The weighted column is VALU pipe cycles."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "path_queue_kernel"
HALF = re.compile(r"v_(cmp|cmpx|cndmask|max3|min3|med3|lshl|lshr|ashr|mul_lo|mul_hi|div_scale|div_fmas|div_fixup|bfe|bfi|alignbit|cvt_f64|cvt_f32_f64|cvt_i32_f64|cvt_u32_f64|"
                  r"\w+_f64|mad_u64|mad_i64|lshl_add_u64|add_co|addc_co|sub_co|subb_co|readlane|writelane|readfirstlane|mbcnt|perm|and_or|or3|xad|add3|lshl_or|lshl_add)")
QUARTER = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
files = {}
cur = None
loc = ("?", 0)
rows = collections.defaultdict(lambda: collections.Counter())
for line in open(path, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
        continue
    if cur is None or want not in cur:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z_0-9]+)\s', line)
    if not m:
        continue
    op = m.group(1)
    c = rows[loc]
    if op.startswith("v_"):
        if QUARTER.match(op): c["v8"] += 1
        elif HALF.match(op): c["v4"] += 1
        else: c["v2"] += 1
    elif op.startswith("s_"):
        c["s"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["vmem"] += 1
tot = collections.Counter()
byfile = collections.defaultdict(lambda: collections.Counter())
for (f, l), c in rows.items():
    tot.update(c)
    byfile[f].update(c)
def fmt(c):
    w = 2 * c["v2"] + 4 * c["v4"] + 8 * c["v8"]
    return f"valu {c['v2'] + c['v4'] + c['v8']:6d} (full {c['v2']:5d} half {c['v4']:5d} quarter {c['v8']:4d}) pipe-cycles {w:6d}  salu {c['s']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}"
print("TOTAL", fmt(tot))
for f, c in sorted(byfile.items(), key=lambda x: -(2 * x[1]["v2"] + 4 * x[1]["v4"] + 8 * x[1]["v8"])):
    print(f"{f:28s}", fmt(c))
if "--lines" in sys.argv:
    fsel = sys.argv[sys.argv.index("--lines") + 1]
    for (f, l), c in sorted(rows.items()):
        if f == fsel:
            print(f"{f}:{l:5d}", fmt(c))
