#!/usr/bin/env python3
"""Grid / workgroup / LDS / register footprint of every kernel in a rocprofv3 kernel trace (one line per distinct kernel).

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py ...;  python tools/trace_shapes.py out
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    d = seen.setdefault(k, {"n": 0, "ns": 0, "r": r})
    d["n"] += 1
    d["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, d in sorted(seen.items(), key=lambda kv: -kv[1]["ns"]):
    r = d["r"]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    print("%8.1f us x%4d  wgs %5d x %4d thr  lds %6s  vgpr %4s agpr %4s scratch %4s  %s" % (
        d["ns"] / d["n"] / 1e3, d["n"], grid // max(wg, 1), wg, r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"),
        r.get("Accum_VGPR_Count", "?"), r.get("Scratch_Size", r.get("Private_Segment_Size", "?")), k[:90]))
