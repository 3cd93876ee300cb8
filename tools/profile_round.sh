#!/bin/bash
# The four rocprofv3 passes behind profiles/<tag>_* (tools/make_profiles.py): run on the GPU box.
#   tools/profile_round.sh r04
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; D=$R/gpurun_out/prof_round; rm -rf $D; mkdir -p $D
cd $R
B="python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --repeats 1"
rocprofv3 --kernel-trace --stats -d $D/stats -o b16 --output-format csv -- $B --steps 50 > $D/stats.log 2>&1 || { tail -5 $D/stats.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o b16 --output-format csv -- $B --steps 20 --no-graph > $D/fetch.log 2>&1 || { tail -5 $D/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o b16 --output-format csv -- $B --steps 20 --no-graph > $D/write.log 2>&1 || { tail -5 $D/write.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $D/sq -o b16 --output-format csv -- $B --steps 20 --no-graph > $D/sq.log 2>&1 || { tail -5 $D/sq.log; exit 1; }
for d in stats fetch write sq; do f=$(find $D/$d -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/$d/b16_kernel_stats.csv; f=$(find $D/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $D/$d/b16_counter_collection.csv; f=$(find $D/$d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $D/$d/b16_kernel_trace.csv; done
python3 tools/make_profiles.py $D ${1:-r05} > $D/make.log 2>&1; tail -20 $D/make.log
find $D -name "*kernel_trace.csv" -size +20M -delete
mkdir -p $R/gpurun_out/profiles_out && cp $R/profiles/${1:-r05}_* $R/gpurun_out/profiles_out/
