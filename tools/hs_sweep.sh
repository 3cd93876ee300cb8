# A/B on one box: per-kernel sums of the captured step for a few environment settings
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$name -o k --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --steps 30 --warmup 5 --repeats 1 > gpurun_out/ab_$name.log 2>&1 || { tail -5 gpurun_out/ab_$name.log; return 1; }
  python3 - "$name" <<'P'
import csv,glob,sys
name=sys.argv[1]
f=glob.glob(f"gpurun_out/ab_{name}/**/*kernel_stats.csv",recursive=True)[0]
tot=0; show=[]
for r in csv.DictReader(open(f)):
    c=int(r["Calls"]); a=float(r["AverageNs"])/1000
    if c in (45,46,47,127) and "rocclr" not in r["Name"]:
        tot+=a
        if any(k in r["Name"] for k in ("heads3_wgrad","wgrad_reduce_and","finals_tail","wgrad_mfma3")): show.append(f"{r['Name'].split('(')[0][-28:]}={a:.1f}")
print(f"{name:24s} sum {tot:7.1f}  " + "  ".join(show))
P
}
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_net.py -m gpu -q -x > gpurun_out/t10.log 2>&1; tail -3 gpurun_out/t10.log
run sep256 NVF_HEADS_IN_TRUNK5=0 NVF_HEADS_SLABS=256
run sep1024 NVF_HEADS_IN_TRUNK5=0 NVF_HEADS_SLABS=1024
run in256 NVF_HEADS_IN_TRUNK5=1 NVF_HEADS_SLABS=256
run in512 NVF_HEADS_IN_TRUNK5=1 NVF_HEADS_SLABS=512
run in1024 NVF_HEADS_IN_TRUNK5=1 NVF_HEADS_SLABS=1024
run in256b NVF_HEADS_IN_TRUNK5=1 NVF_HEADS_SLABS=256
