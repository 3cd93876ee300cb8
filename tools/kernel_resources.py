#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch/occupancy of every kernel in nvfpcc_amd/csrc (hipcc -Rpass-analysis)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvfpcc_amd", "csrc")
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "SGPRs Spill": "sspill", "VGPRs Spill": "vspill",
        "LDS Size [bytes/block]": "lds"}


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    files = sys.argv[1:] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    for f in files:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                            os.path.join(CSRC, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
        rows, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            for k, short in KEYS.items():
                m = re.search(re.escape(k) + r": (\d+)", line)
                if m and cur is not None:
                    cur[short] = int(m.group(1))
        print(f"== {f}")
        for row in rows:
            name = re.sub(r"\(.*", "", demangle(row["name"]))
            name = name.replace("void ", "")[:78]
            print(f"{name:78s} vgpr={row.get('vgpr'):4d} sgpr={row.get('sgpr'):4d} occ={row.get('occ')} "
                  f"lds={row.get('lds'):6d} scratch={row.get('scratch')} spill(s/v)={row.get('sspill')}/{row.get('vspill')}")


if __name__ == "__main__":
    main()
