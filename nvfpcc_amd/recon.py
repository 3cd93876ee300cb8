"""Occupancy grids -> point cloud (reconstruction half of encode()/decode(), NVFPCC.py:501-554, 624-650).

The reference thresholds with MinkowskiEngine (to_sparse + pruning) one block at a time and writes the
PLY through open3d; here the decoder runs batched (bit-identical to batch 1 by construction of the
kernels), a ballot-compaction kernel emits origin + (z, y, x) in raster order, and the PLY writer is ours.
"""
import numpy as np
import torch

from . import ops


@torch.no_grad()
def reconstruct_points(net, latents, origins, thh, batch=64, q=2):
    """latents [N,ch,2,2,2] (already rounded) on the device -> (int64 [n,3] points, per-block counts).

    Voxel coordinate order inside a block is (d0, d1, d2) of the 32^3 grid, the order
    torch.nonzero(out[b, 0] > thh) yields; points = coords + origins[b] (NVFPCC.py:535-538, 635-637)."""
    dev = latents.device
    origins = torch.as_tensor(np.asarray(origins)).to(torch.int32)
    pts, counts = [], []
    for lo in range(0, latents.shape[0], batch):
        hi = min(lo + batch, latents.shape[0])
        out = net.reconstruct(latents[lo:hi].contiguous(), q)
        p, c = ops.threshold_points(out, thh, origins[lo:hi].to(dev))
        pts.append(p.cpu())
        counts.append(c.cpu())
    return torch.cat(pts, 0).long().numpy(), torch.cat(counts, 0).numpy()


def write_ply_ascii(path, points):
    """ASCII PLY with double x/y/z, the layout open3d writes for write_ascii=True (NVFPCC.py:554, 650)."""
    pts = np.round(np.asarray(points, np.float64))
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment Created by nvfpcc_amd\n")
        f.write(f"element vertex {pts.shape[0]}\nproperty double x\nproperty double y\nproperty double z\nend_header\n")
        for x, y, z in pts:
            f.write(f"{x:.0f} {y:.0f} {z:.0f}\n")


def read_ply_ascii(path):
    with open(path) as f:
        n = 0
        for line in f:
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line.strip() == "end_header":
                break
        if n == 0:
            return np.zeros((0, 3), np.float64)
        return np.loadtxt(f, dtype=np.float64).reshape(n, -1)[:, :3]
