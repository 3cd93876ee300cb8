"""HIP decoder on trained, 4-bit-quantised weights against the real reference's decode path (README.md:63:
"rc_enc.ply and rc_dec.ply are identical"; goldens: tools/gen_golden_trained.py, NVFPCC.py:557-650).

Stated tolerance: probabilities <= 1e-5 abs; occupancy identical on every voxel with |p - thh| > 2e-6 at
thh in {0.5, 0.6, 0.64, 0.65}; the decoded point set of `NVFPCC.py decode` equals the reference's, point for point."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.test_trained_golden import CFG, load_pack, state_from_pack, check_probabilities

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden_points(G, thh):
    occ = np.unpackbits(G[f"occ/{thh}"], axis=1).reshape(-1, 32, 32, 32).astype(bool)
    return np.concatenate([np.argwhere(occ[b]) + G["origins"][b].astype(np.int64) for b in range(occ.shape[0])])


@pytest.mark.parametrize("batch", [1, 5])
@pytest.mark.parametrize("tag", ["S", "W"])
def test_reconstruct_equals_the_reference_on_trained_weights(tag, batch, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd import network
    from nvfpcc_amd.model import Net
    from nvfpcc_amd.seeds import synthetic_seed
    pack, G = load_pack(golden_dir, tag)
    ch, channels = CFG[tag]
    network.reset_seed(synthetic_seed())
    net = Net(None, "Gaussian", ch, ",".join(str(c) for c in channels), verbose=False)
    net.load_state_dict(state_from_pack(pack), strict=False)
    net = net.to("cuda")
    lat = torch.from_numpy(G["latents"].astype(np.float32)).to("cuda")
    with torch.no_grad():
        probs = torch.cat([net.reconstruct(lat[i:i + batch].contiguous(), 2) for i in range(0, lat.shape[0], batch)])
    flips = check_probabilities(G, probs.reshape(lat.shape[0], -1).cpu().numpy())
    assert flips == 0, "no voxel of these fixtures sits within 2e-6 of a threshold"


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tag,thh", [("S", 0.6), ("S", 0.64), ("W", 0.6)])
def test_cli_decode_writes_the_reference_point_set(tag, thh, tmp_path, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd.recon import read_ply_ascii
    pack, G = load_pack(golden_dir, tag)
    cwd = str(tmp_path)
    shutil.copy(os.path.join(golden_dir, f"trained_{tag}_pack.pk"), os.path.join(cwd, "pack.pk"))
    ch, channels = CFG[tag]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "NVFPCC.py"), "decode", "pack.pk", "--batchsize", "1", "--thh",
                        str(thh), "--N", str(G["latents"].shape[0]), "--chanstr", ",".join(map(str, channels)), "--ch",
                        str(ch)], cwd=cwd, env=dict(os.environ, PYTHONPATH=ROOT), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    dec = read_ply_ascii(os.path.join(cwd, "rc_dec.ply")).astype(np.int64)
    want = golden_points(G, thh)
    # same points in the same order (raster order inside a block, blocks in origin order); MinkowskiEngine's own
    # order inside a block is the one third-party detail not pinned, so the contract is the sorted set
    assert dec.shape == want.shape
    assert np.array_equal(np.unique(dec, axis=0), np.unique(want, axis=0))
    assert np.array_equal(dec, want)
