#!/usr/bin/env python3
"""Golden vectors of the latent arithmetic coder from the REAL reference executable
(oracle/_ref/module_arithmeticcoding, built by `make -C oracle` from /root/reference/module_arithmeticcoding.cpp).
Stores, per seeded case, the stream's byte length and SHA-256 (and the bytes of the small cases)."""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_inputs import codec_cases  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "module_arithmeticcoding")


def ref_encode(sym, mu, sigma):
    n = np.array([sym.shape[0]], np.int64)
    blob = n.tobytes() + sym.astype(np.int16).tobytes() + mu.astype(np.float32).tobytes() + sigma.astype(np.float32).tobytes()
    return subprocess.run([EXE, "e", "1", "1"], input=blob, stdout=subprocess.PIPE, check=True).stdout


def ref_decode(stream, mu, sigma):
    n = np.array([mu.shape[0]], np.int64)
    blob = n.tobytes() + mu.astype(np.float32).tobytes() + sigma.astype(np.float32).tobytes() + stream
    out = subprocess.run([EXE, "d", "1", "1"], input=blob, stdout=subprocess.PIPE, check=True).stdout
    return np.frombuffer(out, np.int16)


def gen_huffman():
    """Huffman codebooks from the reference's OWN get_pdf / get_huffman_codebook / est_rate
    (util_code_quantized_weights.py:53-105: pure numpy; the module's `import bitstream` is satisfied by an empty
    stand-in because these three functions never touch it)."""
    import types
    sys.modules.setdefault("bitstream", types.ModuleType("bitstream"))
    sys.path.insert(0, "/root/reference")
    import util_code_quantized_weights as ref
    from tests.golden_inputs import huffman_cases
    g = {}
    for name, eles in huffman_cases().items():
        pdf, bins = ref.get_pdf(eles)
        codebook, inv = ref.get_huffman_codebook(pdf, bins)
        words = sorted(inv)                                     # bit strings ('' for the one-symbol pool)
        g[name + "/words"] = np.array(words, dtype="U64")
        g[name + "/symbols"] = np.array([int(inv[w]) for w in words], np.int64)
        g[name + "/pdf"] = np.asarray(pdf, np.float64)
        g[name + "/bins"] = np.asarray(bins, np.int64)
        g[name + "/rate"] = np.float64(ref.est_rate(pdf, bins, codebook))
        nbits = sum(len(codebook[int(v)]) for v in eles)
        g[name + "/nbytes"] = np.int64((nbits + 7) // 8)        # :119-126: zero-padded to whole bytes
        print("huffman", name, "symbols", len(words), "rate", g[name + "/rate"], "bytes", g[name + "/nbytes"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "huffman.npz"), **g)


def main():
    gen_huffman()
    g = {}
    for name, (sym, mu, sigma) in codec_cases().items():
        stream = ref_encode(sym, mu, sigma)
        assert np.array_equal(ref_decode(stream, mu, sigma), sym)
        g[name + "/length"] = np.int64(len(stream))
        g[name + "/sha256"] = np.frombuffer(hashlib.sha256(stream).digest(), np.uint8)
        if len(stream) <= 4096:
            g[name + "/bytes"] = np.frombuffer(stream, np.uint8)
        ideal = 0.0
        print(name, "symbols", sym.shape[0], "bytes", len(stream))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ac.npz"), **g)


if __name__ == "__main__":
    main()
