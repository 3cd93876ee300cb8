#!/bin/bash
# A/B runs on the GPU box: tools/ab_run.sh OUTDIR spec1 spec2 ...
#   spec = LIB[+ENV=VAL[+ENV=VAL...]]   LIB "base" = the in-tree library, otherwise nvfpcc_amd/ab/libnvf_hip_LIB.so
# per variant: bench.py ms/step (graph replay) and a rocprofv3 kernel-stats pass; summary printed by tools/ab_summary.py
set -o pipefail
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
names=()
for spec in "$@"; do
  IFS='+' read -r -a parts <<< "$spec"
  lib=${parts[0]}
  v=$(echo "$spec" | tr '+=,' '___')
  names+=("$v")
  (
    if [ "$lib" != base ]; then export NVF_LIB=$PWD/nvfpcc_amd/ab/libnvf_hip_$lib.so; fi
    for kv in "${parts[@]:1}"; do export "$kv"; done
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --steps 200 --warmup 20 > "$out/$v.json" 2> "$out/$v.err" || { echo "bench $v failed"; tail -5 "$out/$v.err"; exit 1; }
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/$v" -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --steps 50 > "$out/$v.prof.log" 2>&1 || { echo "prof $v failed"; tail -5 "$out/$v.prof.log"; exit 1; }
    rm -f "$out/$v/p_kernel_trace.csv"
  ) || exit 1
done
python3 tools/ab_summary.py "$out" "${names[@]}"
