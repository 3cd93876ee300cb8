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


# the reference's own format strings (/root/reference/NVFPCC.py:261 and :371), turned into parsers: whatever reads the
# reference's training log reads ours
REF_TRAIN = ('[Epoch %04d TRAIN %.1f seconds] Loss: %.4e PosiPenal: %.4f PosiGain: %.4f Pacc: %.4f Nacc: %.4f S1 Loss: %.4f '
             'S2 Loss: %.4f S1Pacc: %.4f S1Nacc: %.4f S2Pacc: %.4f S2Nacc: %.4f bpp: %.4f b_latent: %.4f  b_net: %.4f '
             'MSE1: %.4f PSNR1: %.4f')
REF_TEST = ('[Epoch %04d TEST %.1f seconds] Loss: %.4e PosiPenal: %.4f PosiGain: %.4f Pacc: %.4f Nacc: %.4f S1 Loss: %.4f '
            'S2 Loss: %.4f S1Pacc: %.4f S1Nacc: %.4f S2Pacc: %.4f S2Nacc: %.4f bpp: %.4f b_latent: %.4f b_net: %.4f '
            'b_all: %.4f MSE1: %.4f PSNR1: %.4f')


def line_parser(fmt):
    import re
    num = r"([-+]?(?:\d+\.\d+(?:e[-+]\d+)?|nan|inf))"
    pat = re.escape(fmt)
    for spec in ("%04d", "%.1f", "%.4e", "%.4f"):
        pat = pat.replace(re.escape(spec), r"(\d{4})" if spec == "%04d" else num)
    return re.compile("^" + pat + "$")


def check_log_lines(out):
    """Every TRAIN / TEST line matches the reference's format exactly (field names, order, spacing, precision), and
    the accuracies are per-mini-batch means of ratios in [0, 1]."""
    tr, te = line_parser(REF_TRAIN), line_parser(REF_TEST)
    trains = [ln for ln in out.splitlines() if " TRAIN " in ln]
    tests = [ln for ln in out.splitlines() if " TEST " in ln]
    assert trains and tests
    for ln in trains:
        m = tr.match(ln)
        assert m, ln
        v = [float(x) for x in m.groups()]
        assert len(v) == 18 and all(0.0 <= x <= 1.0 for x in v[5:7] + v[9:13]), ln      # Pacc Nacc S1P S1N S2P S2N
        assert abs(v[13] - (v[14] + v[15])) < 2e-4                                      # bpp = b_latent + b_net
    for ln in tests:
        m = te.match(ln)
        assert m, ln
        v = [float(x) for x in m.groups()]
        assert len(v) == 19 and all(0.0 <= x <= 1.0 for x in v[5:7] + v[9:13]), ln
        assert v[16] >= v[13] - 1e-4                                                    # b_all includes the side information


CASES = {
    # BASELINE.json configs[1] / configs[4]: narrow and wide decoder, the reference's own command lines (README.md:50-63)
    "S": ["--chanstr", "8,16,8,8", "--ch", "3"],
    "W": ["--chanstr", "16,32,16,16", "--ch", "8", "--wemb", "8"],
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tag", ["S", "W"])
def test_train_encode_decode_roundtrip(tmp_path, tag):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd import latent_codec
    from nvfpcc_amd.recon import read_ply_ascii
    from nvfpcc_amd.synth import write_dataset, make_origins
    cwd = str(tmp_path)
    n = 24
    ch = int(CASES[tag][3])
    c = [int(v) for v in CASES[tag][1].split(",")]
    write_dataset(os.path.join(cwd, "toy"), n)
    cli = os.path.join(ROOT, "NVFPCC.py")
    common = CASES[tag][:4]
    wemb = CASES[tag][5] if tag == "W" else "5"
    out = run([cli, "train", "toy.ply", "--checkpoint_dir", "ckpts", "--batchsize", "8", "--lambda", "200", "--lr",
               "1e-3", "--w1", "10", "--w2", "57", "--wemb", wemb, "--shuffle", "True", "--epochs", "11",
               "--phase_change", "5"] + common, cwd)
    assert "[Epoch 0010 TRAIN" in out and "[Epoch 0010 TEST" in out
    check_log_lines(out)
    sd = torch.load(os.path.join(cwd, "ckpts", "0010.ckpt"), map_location="cpu")
    assert len(sd) == 50 and sd["reconstructor.conv2.kernel"].shape == (c[3], c[3], 4, 4, 4)
    assert sd["reconstructor.up1.kernel"].shape == (c[1], c[2], 5, 5, 5) and sd["entropy_coder.sigma"].shape == (1, ch, 1, 1, 1)
    emb = torch.load(os.path.join(cwd, "ckpts", "0010_emb.ckpt"), map_location="cpu")
    assert emb.shape == (n, ch, 2, 2, 2) and not torch.equal(emb, torch.ones_like(emb))
    first = float(out.split("[Epoch 0000 TRAIN")[1].split("Loss: ")[1].split()[0])
    last = float(out.split("[Epoch 0010 TRAIN")[1].split("Loss: ")[1].split()[0])
    assert last < first, (first, last)
    run([os.path.join(ROOT, "manipulate_weights.py"), "ckpts/0010.ckpt", "q4.ckpt", "16"], cwd)
    q4 = torch.load(os.path.join(cwd, "q4.ckpt"), map_location="cpu")
    assert type(q4) is dict and len(q4) == 28           # manipulate_weights.py:19-32's key set
    run([cli, "encode", "toy.ply", "--batchsize", "5", "--load_weights", "q4.ckpt", "--load_emb",
         "ckpts/0010_emb.ckpt", "--thh", "0.5", "--pack_fn", "pack.pk"] + common, cwd)
    run([cli, "decode", "pack.pk", "--batchsize", "1", "--thh", "0.5", "--N", str(n)] + common, cwd)
    with open(os.path.join(cwd, "pack.pk"), "rb") as f:
        pack = pickle.load(f)
    # pack.pk layout (NVFPCC.py:471-493, util_code_quantized_weights.py:201-209)
    assert list(pack) == ['net_weight_pack', 'origins', 'latent_pack']
    assert pack['origins'].dtype == np.int16 and pack['origins'].shape == (n, 3)
    assert np.array_equal(pack['origins'], make_origins(n).astype(np.int16))
    wp, lp = pack['net_weight_pack'], pack['latent_pack']
    assert sorted(wp) == sorted(['bit_stream', 'inv_codebook', 'element_length', 'shape_list', 'as_is_pool',
                                 'keys_quantize', 'keys_code_as_is'])
    assert len(wp['keys_quantize']) == 7 and len(wp['keys_code_as_is']) == 14 and len(wp['as_is_pool']) == 14
    assert [tuple(sh) for sh in wp['shape_list']] == [tuple(q4[k].shape) for k in wp['keys_quantize']]
    assert wp['element_length'] == sum(q4[k].numel() for k in wp['keys_quantize'])
    assert sorted(lp) == sorted(['shape', 'latent_byte_stream', 'sigma', 'mu', 'length'])
    assert tuple(lp['shape']) == (n, ch, 2, 2, 2) and int(lp['length'][0]) == n * ch * 8
    # the arithmetic-coded latent stream: its length is what nvf_ac_encode gives for the decoded symbols (and the
    # range coder is byte-identical to the reference's executable, tests/test_codec.py), and it decodes losslessly
    latents = latent_codec.arithmetic_dec(lp)
    again = latent_codec.arithmetic_enc(latents, lp['sigma'].cpu(), lp['mu'].detach().cpu())
    assert len(again['latent_byte_stream']) == len(lp['latent_byte_stream'])
    assert again['latent_byte_stream'] == lp['latent_byte_stream']
    enc = read_ply_ascii(os.path.join(cwd, "rc_enc.ply"))
    dec = read_ply_ascii(os.path.join(cwd, "rc_dec.ply"))
    assert enc.shape == dec.shape and np.array_equal(enc, dec), "encoder and decoder reconstructions must be identical"
    assert enc.shape[0] > 0


@pytest.mark.timeout(600)
def test_graph_driven_and_host_driven_training_write_the_same_checkpoint(tmp_path):
    """NVFPCC.py train replays one captured HIP graph per full mini-batch (engine.EpochDriver); NVF_TRAIN_GRAPH=0
    launches every kernel from the host.  21 blocks at batch 8 = two graph-replayed mini-batches + a short host-launched
    one per epoch; both runs must write identical 0010.ckpt / 0010_emb.ckpt (bit for bit) and the same log numbers."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd.synth import write_dataset
    cli = os.path.join(ROOT, "NVFPCC.py")
    outs = {}
    for mode in ("1", "0"):
        cwd = str(tmp_path / f"g{mode}")
        os.makedirs(cwd)
        write_dataset(os.path.join(cwd, "toy"), 21)
        env = dict(os.environ, PYTHONPATH=ROOT, NVF_TRAIN_GRAPH=mode)
        r = subprocess.run([sys.executable, cli, "train", "toy.ply", "--checkpoint_dir", "ckpts", "--batchsize", "8",
                            "--lambda", "200", "--lr", "1e-3", "--w1", "10", "--w2", "57", "--wemb", "5", "--shuffle",
                            "True", "--epochs", "11", "--phase_change", "4", "--chanstr", "8,16,8,8", "--ch", "3"],
                           cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        outs[mode] = (torch.load(os.path.join(cwd, "ckpts", "0010.ckpt"), map_location="cpu"),
                      torch.load(os.path.join(cwd, "ckpts", "0010_emb.ckpt"), map_location="cpu"), r.stdout)
    (sd_g, emb_g, log_g), (sd_h, emb_h, log_h) = outs["1"], outs["0"]
    assert list(sd_g) == list(sd_h)
    for k in sd_g:
        assert torch.equal(sd_g[k], sd_h[k]), k
    assert torch.equal(emb_g, emb_h)
    pick = lambda log: [ln.split("seconds]")[1] for ln in log.splitlines() if "TRAIN" in ln and "Epoch 0010" in ln]
    fields = lambda s: [float(t) for t in s.replace(":", " ").split() if t.replace(".", "").replace("e", "").replace("-", "").replace("+", "").isdigit()]
    a, b = fields(pick(log_g)[0]), fields(pick(log_h)[0])
    assert len(a) == len(b) and np.allclose(a, b, rtol=1e-3, atol=1e-4), (a, b)
