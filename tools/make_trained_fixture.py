#!/usr/bin/env python3
"""Step 1 of the trained-weights golden (runs on the GPU box): train a small decoder with the HIP command line,
quantise it to the 1/16 grid, encode and decode, and leave the artefacts under gpurun_out/trained_<tag>/:

    pack.pk       the codec's output: Huffman-coded 4-bit kernels + as-is floats + origins + AC-coded latents
    emb.npy       the trained latent table (encode's input)
    rc_enc.ply / rc_dec.ply   what the HIP encoder / decoder reconstructed

Step 2 (tools/gen_golden_trained.py, build container only) loads pack.pk into the REAL reference network on the CPU
and stores its occupancy; tests/test_gpu_trained.py compares the HIP decoder with that.

    python tools/make_trained_fixture.py [S] [W]
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvfpcc_amd.synth import write_dataset  # noqa: E402

CASES = {"S": ["--chanstr", "8,16,8,8", "--ch", "3", "--wemb", "5"],
         "W": ["--chanstr", "16,32,16,16", "--ch", "8", "--wemb", "8"]}
N_BLOCKS, EPOCHS, PHASE, BATCH = 12, 201, 80, 4


def run(cmd, cwd):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=dict(os.environ, PYTHONPATH=ROOT),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-4000:])
        raise SystemExit(r.returncode)
    return r.stdout


def main():
    tags = [t for t in sys.argv[1:] if t in CASES] or list(CASES)
    cli = os.path.join(ROOT, "NVFPCC.py")
    for tag in tags:
        cwd = tempfile.mkdtemp(prefix=f"nvf_trained_{tag}_")
        write_dataset(os.path.join(cwd, "toy"), N_BLOCKS)
        common = CASES[tag][:4]
        log = run([cli, "train", "toy.ply", "--checkpoint_dir", "ckpts", "--batchsize", str(BATCH), "--lambda", "200",
                   "--lr", "1e-3", "--w1", "10", "--w2", "57", "--shuffle", "True", "--epochs", str(EPOCHS),
                   "--phase_change", str(PHASE)] + CASES[tag], cwd)
        last = "%04d" % (EPOCHS - 1)
        print(tag, [ln for ln in log.splitlines() if f"Epoch {last}" in ln])
        run([os.path.join(ROOT, "manipulate_weights.py"), f"ckpts/{last}.ckpt", "q4.ckpt", "16"], cwd)
        print(run([cli, "encode", "toy.ply", "--batchsize", "5", "--load_weights", "q4.ckpt", "--load_emb",
                   f"ckpts/{last}_emb.ckpt", "--thh", "0.6", "--pack_fn", "pack.pk"] + common, cwd)[-600:])
        run([cli, "decode", "pack.pk", "--batchsize", "1", "--thh", "0.6", "--N", str(N_BLOCKS)] + common, cwd)
        out = os.path.join(ROOT, "gpurun_out", f"trained_{tag}")
        os.makedirs(out, exist_ok=True)
        for f in ("pack.pk", "rc_enc.ply", "rc_dec.ply"):
            shutil.copy(os.path.join(cwd, f), os.path.join(out, f))
        # pack.pk as the reference writes it holds sigma / mu as tensors on the encoder's device (NVFPCC.py:471-477);
        # the committed fixture must unpickle on a CPU-only host too
        import pickle
        with open(os.path.join(out, "pack.pk"), "rb") as f:
            pack = pickle.load(f)
        lp = pack["latent_pack"]
        lp["sigma"], lp["mu"] = lp["sigma"].detach().cpu(), lp["mu"].detach().cpu()
        with open(os.path.join(out, "pack.pk"), "wb") as f:
            pickle.dump(pack, f)
        emb = torch.load(os.path.join(cwd, "ckpts", f"{last}_emb.ckpt"), map_location="cpu")
        np.save(os.path.join(out, "emb.npy"), emb.float().numpy())
        print(tag, "->", out, {f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)})


if __name__ == "__main__":
    main()
