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


def main():
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
