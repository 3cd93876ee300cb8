#!/usr/bin/env python3
"""Per-kernel duration and HBM traffic of the 4096-block latent step / eval forward (tools/profile_4096.sh)."""
import collections
import csv
import json
import os
import re
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return k.split("(")[0][:70]


def counters(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    D, tag = sys.argv[1], sys.argv[2]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(D, "stats", "p_kernel_trace.csv"))):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fetch = counters(os.path.join(D, "fetch", "p_counter_collection.csv"), "FETCH_SIZE")
    write = counters(os.path.join(D, "write", "p_counter_collection.csv"), "WRITE_SIZE")
    run = json.loads(open(os.path.join(D, "run.json")).read().strip().splitlines()[-1])
    # the big dispatches of a kernel: those within a factor of two of its longest one (the warm-up / init launches
    # of the same kernels at other sizes drop out)
    rows = []
    for k, v in dur.items():
        big = [x for x in v if x >= 0.5 * max(v)]
        if st.median(big) < 50.0:
            continue
        f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
        fb = st.median([x for x in f if x >= 0.5 * max(f)]) if max(f) > 0 else 0.0
        wb = st.median([x for x in w if x >= 0.5 * max(w)]) if max(w) > 0 else 0.0
        hbm = (2 * fb + wb) * 1024          # KB -> B; gfx950: FETCH_SIZE counts 128-B requests at 64 B (guide, HBM section)
        rows.append((st.median(big), len(big), k, hbm))
    rows.sort(reverse=True)
    tot_us = sum(r[0] * r[1] for r in rows)
    out = [f"# 4096 resident 32^3 blocks, one MI355X (BASELINE.json configs[2]): `python3 tools/latent4096.py`", "",
           f"Un-profiled run: latent step (NVFPCC.py:225-251, forward + losses + backward-data + Adam on the latents) "
           f"{run['latent_step_ms']} ms = **{run['latent_step_blocks_per_s']} blocks/s** = {run['latent_step_tflops']} TFLOP/s "
           f"(0.8048 GFLOP/block, {run['latent_step_tflops'] / 157.3 * 100:.1f} % of the fp32 peak); eval forward "
           f"{run['eval_forward_ms']} ms = **{run['eval_forward_blocks_per_s']} blocks/s** = {run['eval_forward_tflops']} TFLOP/s "
           f"({run['eval_forward_tflops'] / 157.3 * 100:.1f} %); peak device memory {run['peak_mem_GB']} GB of 288.", "",
           "Kernels of at least 50 us (rocprofv3 --kernel-trace for the durations; --pmc FETCH_SIZE and --pmc WRITE_SIZE in",
           "passes of their own; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of",
           "MI355X_MICROARCH.md -- an upper bound for our 4-byte-per-lane loads):", "",
           "| kernel | dispatches | us (median) | HBM MB / dispatch | GB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
    for us, n, k, hbm in rows:
        gbs = hbm / us / 1e3
        out.append(f"| `{short(k)}` | {n} | {us:.0f} | {hbm / 1e6:.0f} | {gbs:.0f} | {gbs / 8000 * 100:.0f} % |")
    out.append("")
    out.append(f"Sum of these kernels: {tot_us / 1e3:.1f} ms over the profiled run.")
    open(os.path.join(ROOT, "profiles", f"{tag}_latent4096.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
