"""Summarise rocprofv3 --pmc CSVs for the path kernel: mean counter value per dispatch."""
import csv, collections, glob, json, sys
agg = collections.defaultdict(list)
kname = vg = lds = grid = None
args = [a for a in sys.argv[1:] if not a.startswith("--")]
def opt(name): return next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--" + name + "=")), None)
traffic_out = opt("traffic-json")
# --counters-json=FILE --config=c2 --mode=specialised|precompiled: merge this kernel's mean counters into FILE under "config:mode",
# stamped with the hash of the kernel sources they were measured on (bench.py reads the file and flags a stale entry)
counters_out, cfg_name, cfg_mode = opt("counters-json"), opt("config"), opt("mode")
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "path_trace" in r["Kernel_Name"] or "path_queue" in r["Kernel_Name"] or "pine_scene_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pine_gpu::", "")
                vg = r.get("VGPR_Count"); lds = r.get("LDS_Block_Size"); grid = r.get("Grid_Size")
print(f"kernel {kname} VGPR={vg} static_LDS={lds} grid={grid}   (rocprofv3 reports the kernel's STATIC LDS only; these kernels take all of theirs -- "
      f"130 - 160 KB per workgroup -- as dynamic shared memory: see `lds_bytes` of plan stats / tools/sections.py, and VGPR is the descriptor's granule count)")
m = {k: sum(v) / len(v) for k, v in agg.items()}
for k in sorted(m):
    print(f"{k:28s} {m[k]:.5e}  (n={len(agg[k])})")
def g(k): return m.get(k, float('nan'))
print("--- derived")
print(f"wave lifetime cycles (x4)     {g('SQ_WAVE_CYCLES')*4/g('SQ_WAVES'):.4e}")
print(f"frac ACTIVE_INST_ANY          {g('SQ_ACTIVE_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f}")
print(f"frac WAIT_ANY                 {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.3f}")
print(f"frac WAIT_INST_ANY            {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f}")
print(f"frac ACTIVE_INST_VALU         {g('SQ_ACTIVE_INST_VALU')/g('SQ_WAVE_CYCLES'):.3f}")
print(f"VALU lane utilisation         {g('SQ_THREAD_CYCLES_VALU')/(g('SQ_ACTIVE_INST_VALU')*64):.3f}")
print(f"VALU insts / SALU / VMEM_RD / LDS  {g('SQ_INSTS_VALU'):.3e} {g('SQ_INSTS_SALU'):.3e} {g('SQ_INSTS_VMEM_RD'):.3e} {g('SQ_INSTS_LDS'):.3e}")
if 'FETCH_SIZE' in m: print(f"FETCH_SIZE KB {g('FETCH_SIZE'):.4e} (x2 per guide for wide reads)  WRITE_SIZE KB {g('WRITE_SIZE'):.4e}")

if traffic_out and 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are KB at the L2's fabric side (Infinity-Cache
    # hits included); on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes -> double it.
    json.dump({"kernel": kname, "fetch_size_kb": m['FETCH_SIZE'], "write_size_kb": m['WRITE_SIZE'],
               "traffic_bytes_per_launch": (2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0,
               "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 read-request correction of the guide); fabric-side bytes, Infinity-Cache hits included",
               "dispatches_averaged": len(agg['FETCH_SIZE'])}, open(traffic_out, "w"), indent=1)

if counters_out and cfg_name and m:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    try:
        allc = json.load(open(counters_out))
    except Exception:
        allc = {}
    entry = {"kernel": kname, "mode": cfg_mode or "precompiled", "counters": m, "dispatches_averaged": {k: len(v) for k, v in agg.items()},
             "kernel_source_hash": bench.kernel_source_hash(),
             "what": "mean per dispatch of the path kernel over `python bench.py --headline-only` under rocprofv3 --pmc, one counter group per pass"}
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        entry["traffic_bytes_per_launch"] = (2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0
        entry["traffic_correction"] = "2 x FETCH_SIZE + WRITE_SIZE KB (gfx950 read-request correction of the guide); fabric side, Infinity-Cache hits included"
    allc[f"{cfg_name}:{entry['mode']}"] = entry
    json.dump(allc, open(counters_out, "w"), indent=1, sort_keys=True)
