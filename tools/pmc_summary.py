"""Summarise rocprofv3 --pmc CSVs for the path kernel: mean counter value per dispatch."""
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "path_trace" in r["Kernel_Name"] or "path_queue" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                vg = r.get("VGPR_Count"); lds = r.get("LDS_Block_Size"); grid = r.get("Grid_Size")
print(f"kernel path kernel VGPR={vg} LDS={lds} grid={grid}")
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
