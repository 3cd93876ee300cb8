"""Trained, 4-bit-quantised decoders against the REAL reference's decode path (README.md:63 "rc_dec.ply identical").

tests/golden/trained_{S,W}_pack.pk were written by the HIP command line on an MI355X (tools/make_trained_fixture.py:
201 epochs on 12 synthetic blocks, manipulate_weights, encode); tests/golden/trained_{S,W}.npz hold what the reference's
own Net gives for them on the CPU, block by block at batch 1 (tools/gen_golden_trained.py, /root/reference/NVFPCC.py:
557-638, latents through the reference's arithmetic-coding executable).  This file is the CPU half: the in-process range
coder and the oracle on trained weights; tests/test_gpu_trained.py is the HIP half."""
import os
import pickle

import numpy as np
import pytest
import torch

from nvfpcc_amd import latent_codec, weight_codec
from nvfpcc_amd.seeds import synthetic_seed
from oracle import nvf_oracle as O

CFG = {"S": (3, (8, 16, 8, 8)), "W": (8, (16, 32, 16, 16))}


def load_pack(golden_dir, tag):
    with open(os.path.join(golden_dir, f"trained_{tag}_pack.pk"), "rb") as f:
        pack = pickle.load(f)
    return pack, np.load(os.path.join(golden_dir, f"trained_{tag}.npz"))


def state_from_pack(pack, qp=16.0):
    """The tensors decode() loads (NVFPCC.py:566-581): de-quantised kernels + as-is parameters."""
    wp = pack["net_weight_pack"]
    pool = weight_codec.entropy_decode(wp["bit_stream"], wp["inv_codebook"], wp["element_length"], wp["shape_list"])
    nd = {k: torch.from_numpy(v).float() / qp for k, v in zip(wp["keys_quantize"], pool)}
    nd.update({k: torch.from_numpy(np.asarray(v)).float() for k, v in zip(wp["keys_code_as_is"], wp["as_is_pool"])})
    return nd


def check_probabilities(G, probs, tol=1e-5, margin=2e-6):
    """probs [n, 32768] float32 against the golden: sampled probabilities, every near-threshold voxel, per-block sums,
    and the occupancy at the four thresholds on every voxel farther than `margin` from the threshold."""
    n = probs.shape[0]
    assert np.abs(probs[:, G["sample_index"]] - G["sample_p"][:n]).max() <= tol
    assert np.allclose(probs.astype(np.float64).sum(1), G["sum_p"][:n], rtol=2e-6)
    flips = 0
    for t in G["thh"]:
        want = np.unpackbits(G[f"occ/{t}"], axis=1).astype(bool)[:n]
        got = probs > np.float32(t)
        near = G[f"near/{t}/index"]
        near = near[near[:, 0] < n]
        if near.size:
            assert np.abs(probs[near[:, 0], near[:, 1]] - G[f"near/{t}/p"][:near.shape[0]]).max() <= tol
        diff = np.argwhere(got != want)
        flips += diff.shape[0]
        for b, i in diff:      # a flip is allowed only within `margin` of the threshold
            assert abs(float(probs[b, i]) - t) <= margin, (t, b, i, probs[b, i])
    return flips


@pytest.mark.parametrize("tag", ["S", "W"])
def test_latent_stream_decodes_like_the_reference_executable(tag, golden_dir):
    pack, G = load_pack(golden_dir, tag)
    lat = latent_codec.arithmetic_dec(pack["latent_pack"]).numpy()
    assert lat.shape == G["latents"].shape and np.array_equal(lat, G["latents"].astype(np.float32))
    assert pack["origins"].dtype == np.int16 and np.array_equal(pack["origins"], G["origins"])


@pytest.mark.parametrize("tag,n", [("S", 12), ("W", 4)])
def test_oracle_reconstructs_trained_weights_like_the_reference(tag, n, golden_dir):
    """Pins the oracle's decoder on a trained, quantised network with sharp probabilities (the other goldens use
    seed-init + perturbed weights)."""
    pack, G = load_pack(golden_dir, tag)
    ch, channels = CFG[tag]
    P, _ = O.build_state(ch, channels, synthetic_seed())
    P.update(state_from_pack(pack))
    lat = torch.from_numpy(G["latents"][:n].astype(np.float32))
    torch.set_num_threads(8)
    with torch.no_grad():
        probs = np.concatenate([O.decoder(P, lat[i:i + 1], 2)[0].reshape(1, -1).numpy() for i in range(n)])
    flips = check_probabilities(G, probs, tol=2e-6, margin=1e-6)
    assert flips == 0
