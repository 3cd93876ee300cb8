#!/usr/bin/env python3
"""Build A/B variants of libnvf_hip.so for tuning runs: the named sources are recompiled with extra -D flags, every
other object is taken from the regular in-tree build.  The variant is used through the NVF_LIB hook (nvfpcc_amd/_lib.py).

    python tools/ab_build.py sumT wgrad.hip -DNVF_SUM_T=1024        # -> nvfpcc_amd/ab/libnvf_hip_sumT.so
    NVF_LIB=nvfpcc_amd/ab/libnvf_hip_sumT.so python bench.py --no-cpu-baseline
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvfpcc_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    srcs = [a for a in sys.argv[2:] if a.endswith(".hip")]
    flags = [a for a in sys.argv[2:] if not a.endswith(".hip")]
    B.build()
    out_dir = os.path.join(B.HERE, "ab")
    os.makedirs(out_dir, exist_ok=True)
    objs, procs = [], []
    for src in B.SOURCES:
        if src in srcs:
            obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", *flags, "-c",
                   os.path.join(B.CSRC, src), "-o", obj]
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        else:
            obj = os.path.join(B.CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"hipcc failed on {src}:\n{out}")
    lib = os.path.join(out_dir, f"libnvf_hip_{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
