"""Seed stream behind the decoder's frozen ``*_init`` buffers.

The reference reads ``SEED3.npy`` from the working directory at import time
(/root/reference/utils/network.py:20-22) and every layer constructor consumes a
slice of it in construction order (network.py:4605-4607, 4671-4751).  The
author's file is not redistributable (README.md:10, Google Drive), so this
build owns a deterministic stand-in: ``numpy.random.default_rng(3).random(n)``,
float64 in [0, 1).  A real ``SEED3.npy`` dropped next to the CLI wins.
"""
import os
import numpy as np

_SYNTH_LEN = 262144  # >= 210 683 values consumed by ch=8, chanstr=16,32,16,16


def synthetic_seed(n=_SYNTH_LEN):
    return np.random.default_rng(3).random(n)


def load_seed(path="SEED3.npy"):
    """Return the float64 seed vector: the file if present, else the stand-in."""
    if path and os.path.isfile(path):
        seed = np.load(path)
        return np.asarray(seed, dtype=np.float64).reshape(-1)
    return synthetic_seed()


class SeedCursor:
    """Sequential reader; mirrors the reference's module-global ``seed_ptr``."""

    def __init__(self, seed=None):
        self.seed = load_seed() if seed is None else np.asarray(seed, np.float64).reshape(-1)
        self.ptr = 0

    def take(self, n):
        if self.ptr + n > self.seed.shape[0]:
            raise ValueError(
                f"seed stream exhausted: need {self.ptr + n} values, have {self.seed.shape[0]}")
        out = self.seed[self.ptr:self.ptr + n]
        self.ptr += n
        return out
