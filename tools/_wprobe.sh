set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wp; mkdir -p $O
cd $R
W="--chanstr 16,32,16,16 --ch 8 --no-cpu-baseline --no-pmc --no-epoch --steps 30 --warmup 5 --repeats 1"
rocprofv3 --kernel-trace --stats -d $O/tail -o w --output-format csv -- python3 bench.py $W > $O/tail.log 2>&1 &&
for d in tail; do f=$(find $O/$d -name "*kernel_stats.csv"); cp $f $O/$d.csv; done; find $O -name "*kernel_trace.csv" -delete
