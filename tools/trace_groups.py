#!/usr/bin/env python3
"""Median GPU-side duration of consecutive launches of one kernel family in a rocprofv3 kernel trace, in groups of N
(N = 1 warm-up + N - 1 timed): python tools/trace_groups.py DIR NAME_SUBSTRING N"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3])
for i in range(0, len(rows), n):
    grp = rows[i:i + n]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp[1:])
    wg = int(grp[0]["Workgroup_Size_X"])
    print("%-44s wgs %4d x %4d thr  median %.1f us" % (grp[0]["Kernel_Name"][5:49], int(grp[0]["Grid_Size_X"]) // wg, wg, d[len(d) // 2]))
