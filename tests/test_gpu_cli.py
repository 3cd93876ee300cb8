"""End-to-end on the GPU through the command line: train a few epochs on a tiny synthetic cloud, quantise,
encode to pack.pk (+ rc_enc.ply), decode (+ rc_dec.ply) -- README.md:63's claim "rc_enc.ply and rc_dec.ply
are identical" made a test, at different encode / decode batch sizes."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_train_encode_decode_roundtrip(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd.recon import read_ply_ascii
    from nvfpcc_amd.synth import write_dataset, make_origins
    cwd = str(tmp_path)
    n = 24
    write_dataset(os.path.join(cwd, "toy"), n)
    cli = os.path.join(ROOT, "NVFPCC.py")
    common = ["--chanstr", "8,16,8,8", "--ch", "3"]
    out = run([cli, "train", "toy.ply", "--checkpoint_dir", "ckpts", "--batchsize", "8", "--lambda", "200", "--lr",
               "1e-3", "--w1", "10", "--w2", "57", "--wemb", "5", "--shuffle", "True", "--epochs", "11",
               "--phase_change", "5"] + common, cwd)
    assert "[Epoch 0010 TRAIN" in out and "[Epoch 0010 TEST" in out
    sd = torch.load(os.path.join(cwd, "ckpts", "0010.ckpt"), map_location="cpu")
    assert len(sd) == 50 and sd["reconstructor.conv2.kernel"].shape == (8, 8, 4, 4, 4)
    emb = torch.load(os.path.join(cwd, "ckpts", "0010_emb.ckpt"), map_location="cpu")
    assert emb.shape == (n, 3, 2, 2, 2) and not torch.equal(emb, torch.ones_like(emb))
    first = float(out.split("[Epoch 0000 TRAIN")[1].split("Loss: ")[1].split()[0])
    last = float(out.split("[Epoch 0010 TRAIN")[1].split("Loss: ")[1].split()[0])
    assert last < first, (first, last)
    run([os.path.join(ROOT, "manipulate_weights.py"), "ckpts/0010.ckpt", "q4.ckpt", "16"], cwd)
    run([cli, "encode", "toy.ply", "--batchsize", "5", "--load_weights", "q4.ckpt", "--load_emb",
         "ckpts/0010_emb.ckpt", "--thh", "0.5", "--pack_fn", "pack.pk"] + common, cwd)
    run([cli, "decode", "pack.pk", "--batchsize", "1", "--thh", "0.5", "--N", str(n)] + common, cwd)
    with open(os.path.join(cwd, "pack.pk"), "rb") as f:
        pack = pickle.load(f)
    assert list(pack) == ['net_weight_pack', 'origins', 'latent_pack']
    assert pack['origins'].dtype == np.int16 and pack['origins'].shape == (n, 3)
    assert np.array_equal(pack['origins'], make_origins(n).astype(np.int16))
    enc = read_ply_ascii(os.path.join(cwd, "rc_enc.ply"))
    dec = read_ply_ascii(os.path.join(cwd, "rc_dec.ply"))
    assert enc.shape == dec.shape and np.array_equal(enc, dec), "encoder and decoder reconstructions must be identical"
