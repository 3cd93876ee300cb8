"""Entropy-coding side, CPU only.

The latent range coder (libnvf_codec.so) is pinned bit for bit to streams produced by the reference's own
executable (tests/golden/ac.npz, made by tools/gen_golden_codec.py) and, where oracle/_ref holds that
executable, cross-checked live in both directions.  The Huffman weight codec is checked for losslessness,
optimal total length and the pack layout of util_code_quantized_weights.py:201-209.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest
import torch

from nvfpcc_amd import latent_codec, weight_codec
from tests.golden_inputs import codec_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "module_arithmeticcoding")


def test_range_coder_streams_equal_the_reference_executable(golden_dir):
    G = np.load(os.path.join(golden_dir, "ac.npz"))
    for name, (sym, mu, sigma) in codec_cases().items():
        stream = latent_codec.encode_symbols(sym, mu, sigma)
        assert len(stream) == int(G[name + "/length"]), name
        assert hashlib.sha256(stream).digest() == G[name + "/sha256"].tobytes(), name
        if name + "/bytes" in G:
            assert stream == G[name + "/bytes"].tobytes()
        assert np.array_equal(latent_codec.decode_symbols(stream, mu, sigma), sym), name


def test_range_coder_rejects_bad_input():
    with pytest.raises(ValueError):
        latent_codec.encode_symbols(np.array([1500], np.int16), np.array([512.0], np.float32), np.array([1.0], np.float32))
    # every symbol keeps a frequency >= 1 (the "+ symbol" term of the table), even 1600 sigma away
    far = latent_codec.encode_symbols(np.array([900], np.int16), np.array([100.0], np.float32), np.array([0.5], np.float32))
    assert latent_codec.decode_symbols(far, np.array([100.0], np.float32), np.array([0.5], np.float32))[0] == 900


@pytest.mark.skipif(not os.path.isfile(REF_EXE), reason="oracle/_ref not built (make -C oracle)")
def test_range_coder_cross_decodes_with_the_reference_executable():
    rng = np.random.default_rng(7)
    for trial in range(5):
        n = int(rng.integers(1, 3000))
        sg = rng.uniform(0.3, 30, n).astype(np.float32)
        mu = rng.uniform(480, 540, n).astype(np.float32)
        sym = np.clip(np.round(rng.normal(mu, sg)), 0, 1023).astype(np.int16)
        mine = latent_codec.encode_symbols(sym, mu, sg)
        head = np.array([n], np.int64).tobytes()
        ref = subprocess.run([REF_EXE, "e", "1", "1"], input=head + sym.tobytes() + mu.tobytes() + sg.tobytes(),
                             stdout=subprocess.PIPE, check=True).stdout
        assert mine == ref
        back = subprocess.run([REF_EXE, "d", "1", "1"], input=head + mu.tobytes() + sg.tobytes() + mine,
                              stdout=subprocess.PIPE, check=True).stdout
        assert np.array_equal(np.frombuffer(back, np.int16), sym)
        assert np.array_equal(latent_codec.decode_symbols(ref, mu, sg), sym)


def test_latent_pack_layout_and_roundtrip():
    g = torch.Generator().manual_seed(3)
    lat = torch.round(2.0 * torch.randn(50, 3, 2, 2, 2, generator=g))
    sigma = torch.tensor([1.5, 2.5, 0.9]).view(1, 3, 1, 1, 1)
    mu = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.0]).view(1, 3, 1, 1, 1))
    pack = latent_codec.arithmetic_enc(lat, sigma, mu)
    assert list(pack) == ['shape', 'latent_byte_stream', 'sigma', 'mu', 'length']
    assert isinstance(pack['shape'], torch.Size) and isinstance(pack['latent_byte_stream'], bytes)
    assert pack['length'].dtype == np.int64 and pack['length'][0] == 50 * 24
    assert torch.equal(latent_codec.arithmetic_dec(pack), lat)
    # rate close to the model's ideal code length
    from math import erf, sqrt, log2
    ideal = 0.0
    cdf = lambda t, m, s: 0.5 * (1 + erf((t - m) / (s * sqrt(2))))
    for c in range(3):
        for v in lat[:, c].reshape(-1).tolist():
            ideal += -log2(max(cdf(v + .5, mu[0, c].item(), sigma[0, c].item()) - cdf(v - .5, mu[0, c].item(), sigma[0, c].item()), 1e-9))
    assert abs(8 * len(pack['latent_byte_stream']) - ideal) < 0.01 * ideal + 64


def _fake_ckpt(tmp_path):
    g = torch.Generator().manual_seed(11)
    ws = {}
    shapes = {'reconstructor.up0.kernel': (3, 8, 5, 5, 5), 'reconstructor.conv0.kernel': (8, 16, 5, 5, 5),
              'reconstructor.up1.kernel': (16, 8, 5, 5, 5), 'reconstructor.conv1.kernel': (8, 8, 4, 4, 4),
              'reconstructor.up2.kernel': (8, 8, 5, 5, 5), 'reconstructor.conv2.kernel': (8, 8, 4, 4, 4),
              'reconstructor.conv2_cls.kernel': (1, 8, 3, 3, 3)}
    for k, s in shapes.items():
        ws[k] = torch.round(0.08 * torch.randn(s, generator=g) * 16) / 16
    for k in weight_codec.keys_code_as_is:
        ws[k] = torch.randn(3, generator=g)
    fn = str(tmp_path / "q.ckpt")
    torch.save(ws, fn)
    return fn, ws


def test_huffman_weight_pack(tmp_path):
    fn, ws = _fake_ckpt(tmp_path)
    pack = weight_codec.enc_dec_from_file(fn)
    assert list(pack) == ['bit_stream', 'inv_codebook', 'element_length', 'shape_list', 'as_is_pool',
                          'keys_quantize', 'keys_code_as_is']
    assert pack['element_length'] == 51408 and len(pack['shape_list']) == 7 and len(pack['as_is_pool']) == 14
    dec = weight_codec.entropy_decode(pack['bit_stream'], pack['inv_codebook'], pack['element_length'], pack['shape_list'])
    for k, t in zip(weight_codec.keys_quantize, dec):
        assert np.array_equal(t / 16, ws[k].numpy())
    # optimal prefix code: total length equals the sum over merges of the merged weights (in symbols)
    eles = np.concatenate([ws[k].numpy().reshape(-1) * 16 for k in weight_codec.keys_quantize])
    vals, counts = np.unique(np.round(eles).astype(int), return_counts=True)
    import heapq
    h = list(counts.astype(np.int64))
    heapq.heapify(h)
    total = 0
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        total += a + b
        heapq.heappush(h, a + b)
    assert len(pack['bit_stream']) == (total + 7) // 8
    # prefix-free
    words = sorted(pack['inv_codebook'])
    assert not any(b.startswith(a) for a, b in zip(words, words[1:]))


def test_non_discrete_checkpoint_is_rejected(tmp_path):
    fn, ws = _fake_ckpt(tmp_path)
    ws['reconstructor.conv2.kernel'] = ws['reconstructor.conv2.kernel'] + 0.013
    torch.save(ws, fn)
    with pytest.raises(ValueError):
        weight_codec.enc_dec_from_file(fn)


def test_single_valued_kernels_use_the_empty_codeword(tmp_path):
    fn, ws = _fake_ckpt(tmp_path)
    for k in weight_codec.keys_quantize:
        ws[k] = torch.zeros_like(ws[k])
    torch.save(ws, fn)
    pack = weight_codec.enc_dec_from_file(fn)
    assert pack['bit_stream'] == b'' and list(pack['inv_codebook']) == ['']
    dec = weight_codec.entropy_decode(pack['bit_stream'], pack['inv_codebook'], pack['element_length'], pack['shape_list'])
    assert all(not t.any() for t in dec)


@pytest.mark.parametrize("name", ["laplace", "ties", "gap", "single"])
def test_huffman_codebook_equals_the_reference_functions(name, golden_dir):
    """tests/golden/huffman.npz holds what the reference's own get_pdf / get_huffman_codebook / est_rate
    (util_code_quantized_weights.py:53-105) return for these pools (tools/gen_golden_codec.py imports them): the
    same pdf, the same bit string for every symbol -- tie-breaking included -- the same expected length and the
    same stream length in bytes.  What stays unpinned is only the order of bits INSIDE a byte, which belongs to
    the PyPI `bitstream` package."""
    from tests.golden_inputs import huffman_cases
    g = np.load(os.path.join(golden_dir, "huffman.npz"))
    eles = huffman_cases()[name]
    pdf, bins = weight_codec.get_pdf(eles)
    assert np.array_equal(bins, g[name + "/bins"]) and np.array_equal(pdf, g[name + "/pdf"])
    codebook, inv = weight_codec.get_huffman_codebook(pdf, bins)
    want = {str(w): int(s) for w, s in zip(g[name + "/words"], g[name + "/symbols"])}
    assert {k: int(v) for k, v in inv.items()} == want
    for word, sym in want.items():
        assert "".join("1" if b else "0" for b in codebook[sym]) == word
    assert weight_codec.est_rate(pdf, bins, codebook) == pytest.approx(float(g[name + "/rate"]), rel=1e-12, abs=0)
    stream, shapes = weight_codec.entropy_encode([eles], codebook)
    assert len(stream) == int(g[name + "/nbytes"])
    back = weight_codec.entropy_decode(stream, inv, eles.size, shapes)
    assert np.array_equal(back[0], eles)
