#!/usr/bin/env python3
"""Table of ms/step and per-kernel average durations for the variants of an A/B run (tools/ab_run.sh)."""
import csv
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:58]


def main():
    out, names = sys.argv[1], sys.argv[2:]
    stats, order = {}, []
    for v in names:
        try:
            ms = json.loads(open(os.path.join(out, v + ".json")).read().strip().splitlines()[-1])["ms_per_step"]
        except Exception as e:  # noqa: BLE001
            ms = float("nan")
        rows = {}
        p = os.path.join(out, v, "p_kernel_stats.csv")
        if os.path.isfile(p):
            for r in csv.DictReader(open(p)):
                if int(r["Calls"]) >= 50:
                    rows[short(r["Name"])] = float(r["AverageNs"]) / 1e3
        stats[v] = (ms, rows)
        for k in rows:
            if k not in order:
                order.append(k)
    print(f"{'kernel':60s}" + "".join(f"{v:>10s}" for v in names))
    print(f"{'ms_per_step (graph replay, 200 steps)':60s}" + "".join(f"{stats[v][0]:10.4f}" for v in names))
    for k in order:
        print(f"{k:60s}" + "".join(f"{stats[v][1].get(k, float('nan')):10.2f}" for v in names))
    print(f"{'sum of kernels (us)':60s}" + "".join(f"{sum(stats[v][1].values()):10.1f}" for v in names))


if __name__ == "__main__":
    main()
