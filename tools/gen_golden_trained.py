#!/usr/bin/env python3
"""Step 2 of the trained-weights golden (build container only: needs /root/reference and oracle/_ref).

Takes the pack.pk that the HIP command line produced on the GPU box (tools/make_trained_fixture.py: 201 training
epochs on 12 synthetic blocks, 4-bit weights, encode) and runs the REAL reference's decode path on the CPU
(/root/reference/NVFPCC.py:557-638), one block at a time as it does:

  * the latent stream is decoded by the reference's own `module_arithmeticcoding d 1 1` executable (:601-607);
  * the de-quantised kernels (symbols / qp) and the as-is tensors are loaded into the reference's Net with
    `load_state_dict(strict=False)` (:574-581) -- its frozen `*_init` buffers come from the same SEED3 stand-in;
  * `net.reconstruct(latent[i:i+1], q=2)` gives the occupancy probabilities of every block (:626-628).

The Huffman container itself is unpacked with the build's weight_codec (the reference's needs the PyPI `bitstream`
package; its codebook construction is pinned separately, tests/golden/huffman.npz).

Stored (tests/golden/trained_<tag>.npz): the decoded latents, the bit-packed occupancy `out > thh` at the four
thresholds the README uses, every probability within 1e-4 of a threshold, 1024 sampled probabilities per block
and per-block sums.  tests/golden/trained_<tag>_pack.pk is the input, committed beside it.

    python tools/gen_golden_trained.py gpurun_out/trained_S S gpurun_out/trained_W W
"""
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden import import_reference, RefNet      # noqa: E402
from nvfpcc_amd import weight_codec                  # noqa: E402
from tests.golden_inputs import sample_index         # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "module_arithmeticcoding")
OUT = os.path.join(ROOT, "tests", "golden")
CFG = {"S": (3, (8, 16, 8, 8)), "W": (8, (16, 32, 16, 16))}
THH = (0.5, 0.6, 0.64, 0.65)
QP = 16.0


def ref_decode_latents(lp):
    s = tuple(lp["shape"])
    tile = lambda t: torch.tile(t, (s[0], 1, s[2], s[3], s[4])).detach().cpu().numpy().astype(np.float32).reshape(-1)
    blob = lp["length"].tobytes() + (tile(lp["mu"]) + 512).tobytes() + tile(lp["sigma"]).tobytes() + \
        lp["latent_byte_stream"]
    out = subprocess.run([EXE, "d", "1", "1"], input=blob, stdout=subprocess.PIPE, check=True).stdout
    return (np.frombuffer(out, np.int16) - 512).astype(np.float32).reshape(s)


def main():
    torch.set_num_threads(8)
    net_mod, _, _ = import_reference()
    args = sys.argv[1:]
    for src, tag in zip(args[0::2], args[1::2]):
        ch, channels = CFG[tag]
        with open(os.path.join(src, "pack.pk"), "rb") as f:
            pack = pickle.load(f)
        wp = pack["net_weight_pack"]
        dec_pool = weight_codec.entropy_decode(wp["bit_stream"], wp["inv_codebook"], wp["element_length"], wp["shape_list"])
        nd = {k: torch.from_numpy(v).float() / QP for k, v in zip(wp["keys_quantize"], dec_pool)}
        nd.update({k: torch.from_numpy(np.asarray(v)).float() for k, v in zip(wp["keys_code_as_is"], wp["as_is_pool"])})
        net = RefNet(net_mod, ch, channels)
        rt = net.load_state_dict(nd, strict=False)
        assert not rt.unexpected_keys, rt.unexpected_keys
        latents = ref_decode_latents(pack["latent_pack"])
        n = latents.shape[0]
        g = {"latents": latents.astype(np.int16), "thh": np.array(THH, np.float64),
             "origins": np.asarray(pack["origins"])}
        probs = np.empty((n, 32, 32, 32), np.float32)
        with torch.no_grad():
            for i in range(n):          # batch 1, as NVFPCC.py:624-628
                out, _, _ = net.reconstructor(torch.from_numpy(latents[i:i + 1]), 2)
                probs[i] = out[0, 0].numpy()
        flat = probs.reshape(n, -1)
        for t in THH:
            g[f"occ/{t}"] = np.packbits(flat > np.float32(t), axis=1)
            near = np.argwhere(np.abs(flat.astype(np.float64) - t) < 1e-4)
            g[f"near/{t}/index"] = near.astype(np.int32)
            g[f"near/{t}/p"] = flat[near[:, 0], near[:, 1]]
        idx = sample_index(32768, 1024)
        g["sample_index"] = np.asarray(idx, np.int32)
        g["sample_p"] = flat[:, idx]
        g["sum_p"] = flat.astype(np.float64).sum(1)
        sharp = float(np.mean((flat < 0.05) | (flat > 0.95)))
        np.savez_compressed(os.path.join(OUT, f"trained_{tag}.npz"), **g)
        shutil.copy(os.path.join(src, "pack.pk"), os.path.join(OUT, f"trained_{tag}_pack.pk"))
        print(tag, "blocks", n, "points@0.6", int((flat > 0.6).sum()), "fraction of p outside [.05,.95]: %.4f" % sharp,
              "near-threshold voxels", {t: int(g[f'near/{t}/p'].size) for t in THH},
              "npz bytes", os.path.getsize(os.path.join(OUT, f"trained_{tag}.npz")))


if __name__ == "__main__":
    main()
