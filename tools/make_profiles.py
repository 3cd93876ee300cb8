#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of four `bench.py` runs into the committed summaries under profiles/.

    rocprofv3 --kernel-trace --stats -d D/stats -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --steps 50
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d D/fetch ... -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --steps 20 --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d D/write ... (same)
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d D/sq ... (same)
    python tools/make_profiles.py D r01

FETCH_SIZE is doubled (gfx950 counts 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section); counters are collected
in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass).
"""
import collections
import csv
import json
import os
import re
import shutil
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def med_counter(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), st.median(v)) for k, v in acc.items()}


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    return k.split("(")[0][:110] if not k.startswith("void") else k[5:].split("(float")[0].split("(Wg")[0].split("(Heads")[0][:110]


def main():
    base, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(base, "stats", "b16_kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_b16_kernel_stats.csv"))
    fetch = med_counter(os.path.join(base, "fetch", "b16_counter_collection.csv"), "FETCH_SIZE")
    write = med_counter(os.path.join(base, "write", "b16_counter_collection.csv"), "WRITE_SIZE")
    out = ["# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of: python3 bench.py --steps 20 "
           "--no-cpu-baseline --no-pmc --no-epoch --no-graph (batch 16)",
           "# per-dispatch medians, KB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts "
           "128-B requests at 64 B, MI355X_MICROARCH.md section HBM)",
           "kernel,dispatches,FETCH_SIZE_KB_median,WRITE_SIZE_KB_median,hbm_bytes_corrected"]
    tab = {}
    for k, (n, f) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        w = write.get(k, (0, 0.0))[1]
        tab[k] = (2 * f + w) * 1024
        out.append(f'"{short(k)}",{n},{f:.1f},{w:.1f},{tab[k]:.0f}')
    open(os.path.join(prof, f"{tag}_pmc_traffic_b16.csv"), "w").write("\n".join(out) + "\n")

    def find(*subs):
        for k in tab:
            if all(s in k for s in subs):
                return tab[k]
        return None
    tj = {"batch": 16, "chanstr": "8,16,8,8", "source": f"profiles/{tag}_pmc_traffic_b16.csv",
          "hbm_bytes_per_launch": {
              "conv2_bwd_data": find("conv_k4_wino1<W1Cfg<32, 3") or find("conv_k4_wino<WCfg<32, 3>") or find("conv_k4_mfma<MCvFlat<8, 18") or find("conv_gather_glds<GCfg<8, 8, 4, 1, 4, 9"),
              "conv2_fwd": find("conv_k4_wino1<W1Cfg<35, 0") or find("conv_k4_wino<WCfg<35, 0>") or find("conv_k4_mfma<MCv<8, 0, 1, 16"),
              "conv2_bwd_weight": find("wgrad_mfma3_kernel") or find("wgrad_k4_mfma<MCfg<32")},
          "note": "conv2_bwd_weight is the one-launch kernel that also holds the up2, conv1, up1 and conv0 weight gradients"}
    json.dump(tj, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
             "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]
    sqp = os.path.join(base, "sq", "b16_counter_collection.csv")
    vals = {n: med_counter(sqp, n) for n in names}
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(base, "sq", "b16_kernel_trace.csv"))):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    step = [k for k in dur if len(dur[k]) >= 10 and not k.startswith("void at::")]     # the step's kernels, not the set-up's
    ker = sorted(step, key=lambda k: -st.median(dur[k]))[:18]
    lines = ["SQ counters (rocprofv3 --pmc, one pass) of `python3 bench.py --steps 20 --no-cpu-baseline --no-pmc --no-epoch --no-graph`, batch 16; "
             "medians per dispatch.",
             "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 "
             "shader engines.", "",
             "| kernel | us (profiled) | SQ_INSTS_MFMA | MFMA busy | ACTIVE_INST_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE_CYCLES "
             "| WAIT_ANY / WAVE_CYCLES | non-MFMA VALU insts |", "|---|---|---|---|---|---|---|---|"]
    for k in ker:
        g = lambda n: vals[n].get(k, (0, 0.0))[1]
        wc = g("SQ_WAVE_CYCLES") or 1
        kc = g("SQ_BUSY_CYCLES") / 32 or 1
        lines.append(f"| `{short(k)}` | {st.median(dur[k]):.1f} | {g('SQ_INSTS_MFMA'):.3g} | "
                     f"{g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / kc * 100:.0f} % | {g('SQ_ACTIVE_INST_ANY') / wc * 100:.0f} % | "
                     f"{g('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | {g('SQ_WAIT_ANY') / wc * 100:.0f} % | "
                     f"{g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA'):.3g} |")
    open(os.path.join(prof, f"{tag}_pmc_sq_b16.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(tj))


if __name__ == "__main__":
    main()
