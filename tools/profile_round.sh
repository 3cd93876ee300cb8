#!/bin/bash
# The four rocprofv3 passes behind profiles/<tag>_* (run on the GPU box), then tools/make_profiles.py:
#   tools/profile_round.sh gpurun_out/prof r01
# kernel stats of the default bench (graph replay); FETCH_SIZE, WRITE_SIZE and the SQ counters in passes of their own
# (host launches: --no-graph, so every dispatch carries its kernel name).
set -o pipefail
D=$1; TAG=$2
mkdir -p "$D"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$D/stats" -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --steps 50 > "$D/stats.log" 2>&1 || { tail -5 "$D/stats.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$D/fetch" -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --steps 20 --no-graph > "$D/fetch.log" 2>&1 || { tail -5 "$D/fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$D/write" -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --steps 20 --no-graph > "$D/write.log" 2>&1 || { tail -5 "$D/write.log"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d "$D/sq" -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --steps 20 --no-graph > "$D/sq.log" 2>&1 || { tail -5 "$D/sq.log"; exit 1; }
python3 tools/make_profiles.py "$D" "$TAG" > "$D/make_profiles.log" 2>&1 || { tail -5 "$D/make_profiles.log"; exit 1; }
mkdir -p "$D/profiles_out" && cp profiles/${TAG}_bench_b16_kernel_stats.csv profiles/${TAG}_pmc_traffic_b16.csv profiles/${TAG}_traffic.json profiles/${TAG}_pmc_sq_b16.md "$D/profiles_out/"
rm -rf "$D/fetch" "$D/write" "$D/stats/b16_kernel_trace.csv"
find "$D/sq" -name "*counter_collection.csv" -delete
find "$D/sq" -name "*kernel_trace.csv" -delete
tail -20 "$D/make_profiles.log"
