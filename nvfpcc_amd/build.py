"""Builds libnvf_hip.so (gfx950) in-tree with hipcc.  No torch headers involved: the
library is a plain C-ABI shared object (include/nvf_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnvf_hip.so")
CODEC_LIB = os.path.join(HERE, "libnvf_codec.so")
SOURCES = ["conv_direct.hip", "conv_mfma.hip", "conv_wino.hip", "conv_wino1.hip", "conv16_wino.hip", "conv16_wino1.hip", "conv16_mfma.hip", "wgrad16_mfma.hip", "wgrad16_wino.hip", "convt16_mfma.hip", "convt_mfma.hip", "convs2_mfma.hip", "pack_mfma.hip", "heads.hip", "heads_wgrad_mfma.hip", "wgrad.hip", "pointwise.hip", "stem.hip", "finals.hip", "preprocess.hip"]


def _stale(lib=LIB):
    if not os.path.isfile(lib):
        return True
    t = os.path.getmtime(lib)
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".o")]
    deps += [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_codec(force=False):
    """Host-side range coder (g++, no GPU code): libnvf_codec.so, C ABI in include/nvf_codec.h."""
    if not force and not _stale(CODEC_LIB):
        return CODEC_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-fPIC", "-shared", "-std=c++17",
           os.path.join(CSRC, "range_coder.cpp"), "-o", CODEC_LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on range_coder.cpp:\n{r.stdout}")
    return CODEC_LIB


def build(force=False, verbose=False):
    build_codec(force)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    inc = os.path.join(HERE, "..", "include")
    hdr_t = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")] +
                [os.path.getmtime(os.path.join(inc, f)) for f in os.listdir(inc)])
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.isfile(obj) and
                os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src)))):
            continue          # object is newer than its source and every header
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
