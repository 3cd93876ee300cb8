#!/bin/bash
# Per-kernel times of the captured train step (rocprofv3 --kernel-trace --stats), narrow and wide decoder:
#   tools/kprobe.sh            (on the GPU box; tables under gpurun_out/kp/)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/kp; mkdir -p $O
cd $R
for cfg in "s:" "w:--chanstr 16,32,16,16 --ch 8"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d $O/$tag -o k --output-format csv -- python3 bench.py $extra --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --steps 30 --warmup 5 --repeats 1 > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
  cp "$(find $O/$tag -name '*kernel_stats.csv')" $O/$tag.csv
  find $O/$tag -name "*kernel_trace.csv" -delete
done
python3 - <<'P'
import csv, re, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/kp")
for tag in ("s", "w"):
    rows = list(csv.DictReader(open(f"{O}/{tag}.csv")))
    print(f"== {tag}")
    tot = 0.0
    nstep = max(int(r["Calls"]) for r in rows if "step_head" in r["Name"])
    for r in rows[:34]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:86]
        print(f"{n:88s} {r['Calls']:>4s} {float(r['AverageNs'])/1000:8.1f}")
        # the kernels of the captured step: launched once per step (the roofline probe adds repeats of three of them
        # under the same name -- same kernel, the average stands -- and the five-gradient launch without its tail
        # workgroup as a row of its own, which is left out)
        c = int(r["Calls"])
        probe_row = "wgrad_mfma3_kernel" in n and c != nstep
        if c >= nstep - 2 and "rocclr" not in n and not probe_row:
            tot += float(r["AverageNs"]) / 1000
    print(f"-- sum of the step's kernels: {tot:.1f} us")
P
