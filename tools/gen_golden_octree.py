#!/usr/bin/env python3
"""Golden octree partition from the REAL reference executable (oracle/_ref/get_octree, built by `make -C oracle`
from /root/reference/get_octree.cpp) on the seeded synthetic cloud of tests/golden_inputs.py."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_inputs import synthetic_cloud, write_cloud_ply  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "get_octree")


def main():
    pts = synthetic_cloud()
    with tempfile.TemporaryDirectory() as d:
        ply = os.path.join(d, "cloud.ply")
        write_cloud_ply(ply, pts)
        subprocess.run([EXE, ply, os.path.join(d, "o.txt"), os.path.join(d, "s.txt")], check=True)
        origins = np.loadtxt(os.path.join(d, "o.txt"), delimiter=",")
        subtree = open(os.path.join(d, "s.txt")).read()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "octree.npz"), origins=origins.astype(np.int64),
                        n_points=np.int64(len(pts)),
                        points_sha=np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8),
                        subtree_len=np.int64(len(subtree)),
                        subtree_sha=np.frombuffer(hashlib.sha256(subtree.encode()).digest(), np.uint8))
    print("points", len(pts), "origins", origins.shape, "subtree bits", len(subtree))


if __name__ == "__main__":
    main()
