"""Pre-processing: octree partition pinned to the reference's own get_octree
executable (tests/golden/octree.npz; live cross-check where oracle/_ref is built); distance / occupancy grids on
the GPU against the CPU oracle (oracle/preprocess_oracle.py, scipy cKDTree restatement of util_get_grids.py)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest
import torch

from nvfpcc_amd import preprocess as pp
from tests.golden_inputs import synthetic_cloud, write_cloud_ply

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "get_octree")


def test_octree_partition_matches_reference_executable(golden_dir):
    G = np.load(os.path.join(golden_dir, "octree.npz"))
    pts = synthetic_cloud()
    assert hashlib.sha256(pts.tobytes()).digest() == G["points_sha"].tobytes()
    origins, subtree = pp.octree_level5(pts)
    assert np.array_equal(origins, G["origins"])                      # same cubes, same traversal order
    assert len(subtree) == int(G["subtree_len"])
    assert hashlib.sha256(subtree.encode()).digest() == G["subtree_sha"].tobytes()


@pytest.mark.skipif(not os.path.isfile(REF_EXE), reason="oracle/_ref not built (make -C oracle)")
def test_octree_live_against_reference_executable(tmp_path):
    rng = np.random.default_rng(5)
    pts = np.unique(rng.integers(0, 1024, size=(3000, 3)), axis=0)    # scattered points: many sparse cubes
    ply = str(tmp_path / "c.ply")
    write_cloud_ply(ply, pts)
    subprocess.run([REF_EXE, ply, str(tmp_path / "o.txt"), str(tmp_path / "s.txt")], check=True)
    origins, subtree = pp.octree_level5(pp.read_ply_xyz(ply))
    assert np.array_equal(origins, np.loadtxt(str(tmp_path / "o.txt"), delimiter=",").astype(np.int64))
    assert subtree == open(str(tmp_path / "s.txt")).read()
    pp.write_origins_txt(str(tmp_path / "mine.txt"), origins)
    assert open(str(tmp_path / "mine.txt")).read() == open(str(tmp_path / "o.txt")).read()


def test_ply_reader_roundtrip(tmp_path):
    pts = synthetic_cloud()[:500]
    ply = str(tmp_path / "c.ply")
    write_cloud_ply(ply, pts)
    assert np.array_equal(pp.read_ply_xyz(ply), pts)


@pytest.mark.gpu
def test_distance_grids_equal_the_kdtree_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle import preprocess_oracle as PO
    pts = synthetic_cloud()
    origins, _ = pp.octree_level5(pts)
    gt, dist = pp.build_grids(pts, origins)
    gt_ref, dist_ref = PO.grids(pts, origins)
    assert gt.shape == (len(origins), 1, 32, 32, 32) and gt.dtype == np.uint8 and dist.dtype == np.float64
    assert np.array_equal(dist, dist_ref)          # integer squared distances: exact
    assert np.array_equal(gt, gt_ref)
    assert int(gt.sum()) == len(pts)               # every input point is an occupied voxel of exactly one cube
    # scattered cloud: nearest point usually lies in another cube (exercises the +-2 block neighbourhood)
    rng = np.random.default_rng(6)
    pts2 = np.unique(rng.integers(300, 460, size=(400, 3)), axis=0)
    o2, _ = pp.octree_level5(pts2)
    gt2, d2 = pp.build_grids(pts2, o2)
    gt2r, d2r = PO.grids(pts2, o2)
    assert np.array_equal(d2, d2r) and np.array_equal(gt2, gt2r)
