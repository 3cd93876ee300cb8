#!/bin/bash
# rocprofv3 passes of tools/latent4096.py (BASELINE configs[2]) -> profiles/<tag>_latent4096.md
#   tools/profile_4096.sh gpurun_out/p4096 r01
set -o pipefail
D=$1; TAG=$2
mkdir -p "$D"
export TMPDIR=/tmp
python3 tools/latent4096.py > "$D/run.json" 2> "$D/run.err" || { tail -5 "$D/run.err"; exit 1; }
rocprofv3 --kernel-trace --stats -d "$D/stats" -o p --output-format csv -- python3 tools/latent4096.py > "$D/stats.log" 2>&1 || { tail -5 "$D/stats.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$D/fetch" -o p --output-format csv -- python3 tools/latent4096.py --reps 1 > "$D/fetch.log" 2>&1 || { tail -5 "$D/fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$D/write" -o p --output-format csv -- python3 tools/latent4096.py --reps 1 > "$D/write.log" 2>&1 || { tail -5 "$D/write.log"; exit 1; }
python3 tools/summarize_4096.py "$D" "$TAG" > "$D/summary.log" 2>&1 || { tail -8 "$D/summary.log"; exit 1; }
cp profiles/${TAG}_latent4096.md "$D/"
head -2 "$D/fetch/p_counter_collection.csv" > "$D/fetch_header.txt"; head -2 "$D/stats/p_kernel_trace.csv" > "$D/trace_header.txt"
rm -rf "$D/fetch" "$D/write" "$D/stats/p_kernel_trace.csv"
cat "$D/summary.log"
