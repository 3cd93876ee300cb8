#!/usr/bin/env python3
"""Median per-dispatch counter values per kernel from rocprofv3 --pmc CSV outputs: python tools/pmc_table.py dir [substr]"""
import collections, csv, glob, statistics as st, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if sub in k:
        print(k[:100])
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} {st.median(v):16.0f}  (n={len(v)})")
