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
python3 - <<'P'
import torch, sys
sys.path.insert(0, ".")
from nvfpcc_amd import ops
g = torch.Generator().manual_seed(1)
for cin, dout, vs in ((8, 16, (5, 6)), (16, 8, (5,))):
    din = 2 * dout + 3
    gy = torch.randn(3, 8, din, din, din, generator=g).cuda()
    w = (torch.randn(cin, 8, 5, 5, 5, generator=g) * 0.1).cuda()
    mask = torch.randn(3, cin, dout, dout, dout, generator=g).cuda()
    wf, wb = ops.pack_convT_weight(w)
    wp = ops.pack_s2k5_mfma(wb, 8, cin)
    ref = ops.conv3d_s2k5_mfma(gy, wp, cin, mask=mask, variant=0)
    for v in vs:
        y = ops.conv3d_s2k5_mfma(gy, wp, cin, mask=mask, variant=v)
        print("cin", cin, "variant", v, "bit-identical to the default:", torch.equal(y, ref))
P
run base A=1; showk base conv_s2k5
run up2b5 NVF_VAR_UP2B=5; showk up2b5 conv_s2k5
run up2b6 NVF_VAR_UP2B=6; showk up2b6 conv_s2k5
run up1b5 NVF_VAR_UP1B=5; showk up1b5 conv_s2k5
