#!/bin/bash
# One rocprofv3 --pmc pass with the SQ counters over a host-launched bench run; prints the per-kernel medians
# (tools/pmc_table.py).  Run on the GPU box: bash tools/sq_pass.sh [extra bench args]
set -o pipefail
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D=gpurun_out/prof_sq; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
  -d $D/sq -o b16 --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --repeats 1 --steps 20 --no-graph "$@" > $D/sq.log 2>&1 || { tail -5 $D/sq.log; exit 1; }
python3 tools/pmc_table.py $D/sq > gpurun_out/sq_table.txt
find $D -name "*.csv" -size +5M -delete
