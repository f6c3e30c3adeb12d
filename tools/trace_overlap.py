"""Diagnostics: from a rocprofv3 kernel trace (csv), the last few steps' kernels with start / end relative to the step's first kernel
(k_describe_scan / k_walk_slab) and their queue: do the side-stream launches run beside k_tile?  tools/trace_overlap.py <kernel_trace.csv> [steps]"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
firsts = [i for i, r in enumerate(rows) if "k_describe_scan" in r["Kernel_Name"] or "k_walk_slab" in r["Kernel_Name"]]
for i0 in firsts[-steps - 1:-1]:
    t0 = int(rows[i0]["Start_Timestamp"])
    i1 = next((j for j in firsts if j > i0), len(rows))
    print("step:")
    for r in rows[i0:i1]:
        name = r["Kernel_Name"].replace("void l2r::", "").replace("l2r::", "")
        name = name[:name.index("(")] if "(" in name else name
        print("  q%-2s %-44s %8.1f .. %8.1f us  (%6.1f)  grid %s" % (r["Queue_Id"], name[:44], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
