#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs (one directory per pass)."""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2:] or ["spmv_blk_kernel<2, 2>", "spmv_stream_kernel<1, 0>", "tri_blk_kernel", "tri_stream_sf_kernel",
                        "tri_blk_sf_kernel", "tri_stream_kernel", "vec_axpy"]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
meta = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(float)
    info = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = next((w for w in want if w in k), None)
        if m is None:
            continue
        key = (m + (" L" if re.search(r"kernel<1, [01]", k) and "tri" in m else " U" if "tri" in m else ""), r["Dispatch_Id"])
        per_dispatch[(key, r["Counter_Name"])] += float(r["Counter_Value"])
        info[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"])
    for ((name, _), cn), v in per_dispatch.items():
        a = acc[name][cn]; a[0] += v; a[1] += 1
    for (name, _), (dur, vg, lds, grid) in info.items():
        a = acc[name]["_dur_ns(pmc run)"]; a[0] += dur; a[1] += 1
        meta[name] = (vg, lds)
for name in sorted(acc):
    print(f"== {name}  vgpr={meta[name][0]} lds={meta[name][1]}")
    for cn in sorted(acc[name]):
        s, n = acc[name][cn]
        print(f"   {cn:42s} mean/dispatch {s / n:16.1f}   (n={n})")
