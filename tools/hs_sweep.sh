# A/B on one box: per-kernel sums of the captured step for a few environment settings
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$name -o k --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc --no-epoch --no-sweep --no-direct --sustained-s 0 --steps 30 --warmup 5 --repeats 1 > gpurun_out/ab_$name.log 2>&1 || { tail -5 gpurun_out/ab_$name.log; return 1; }
  python3 - "$name" <<'P'
import csv,glob,sys
name=sys.argv[1]
f=glob.glob(f"gpurun_out/ab_{name}/**/*kernel_stats.csv",recursive=True)[0]
tot=0; show=[]
for r in csv.DictReader(open(f)):
    c=int(r["Calls"]); a=float(r["AverageNs"])/1000
    if c >= 70 and "rocclr" not in r["Name"] and not ("wgrad_mfma3_kernel" in r["Name"] and c >= 80):
        tot+=a
        if any(k in r["Name"] for k in ("heads3_wgrad","wgrad_reduce","finals_tail","wgrad_mfma3")): show.append(f"{r['Name'].split('(')[0][-28:]}={a:.1f}")
print(f"{name:24s} sum {tot:7.1f}  " + "  ".join(show))
P
}
show_conv() { python3 - "$1" <<'P'
import csv,glob,sys
f=glob.glob(f"gpurun_out/ab_{sys.argv[1]}/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "conv_k4_mfma" in r["Name"] and int(r["Calls"])>=45: print("   ", r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["AverageNs"])/1000,1))
P
}
show_ct() { python3 - "$1" <<'P'
import csv,glob,sys
f=glob.glob(f"gpurun_out/ab_{sys.argv[1]}/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "convT_k5s2_mfma" in r["Name"] and int(r["Calls"])>=45: print("   ", r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"])/1000,1))
P
}
showk() { python3 - "$1" "$2" <<'P'
import csv,glob,sys
f=glob.glob(f"gpurun_out/ab_{sys.argv[1]}/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"] and int(r["Calls"])>=45: print("   ", r["Name"].split("(")[0][-46:], r["Calls"], round(float(r["AverageNs"])/1000,1))
P
}
# edit below: one line per setting, e.g.
#   run base X=1
run base2 X=1
