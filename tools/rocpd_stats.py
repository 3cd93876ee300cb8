#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (*.db): name, grid, calls, mean / min duration in us.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [substring]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
    names = dict(c.execute(f"select id, {name_col} from {ks}"))
    stats = defaultdict(list)
    for kid, st, en, gx, wx in c.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd} order by start"):
        stats[(names.get(kid, str(kid)), gx // max(wx, 1))].append((en - st) / 1000.0)
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for (n, g), v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        if sub in n:
            print(f"{n[:90]:90s} wgs {g:6d} calls {len(v):5d} mean {sum(v) / len(v):8.1f} us  min {min(v):8.1f}")


if __name__ == "__main__":
    main()
