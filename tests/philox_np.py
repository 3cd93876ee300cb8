"""NumPy restatement of the counter RNG behind the kernels' noise (nvfpcc_amd/csrc/nvf_common.h:24-52: Philox4x32-10,
key = seed, counter = (index, stream id); one U[0,1) float of 24 mantissa bits per element).  Test infrastructure: lets
tools/gen_golden.py feed the REAL reference -- on a CPU-only host -- exactly the noise the HIP engine draws, so whole
training trajectories can be compared; tests/test_gpu_ops.py holds it to the device generator bit for bit."""
import numpy as np

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def philox4x32(seed, stream_id, counter):
    """counter: uint64 array -> uint32 [n, 4]."""
    counter = np.asarray(counter, np.uint64)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    c0 = (counter & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (counter >> np.uint64(32)).astype(np.uint32)
    sid = np.asarray(stream_id, np.uint64) * np.ones_like(counter)
    c2 = (sid & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c3 = (sid >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = np.uint64(0xD2511F53) * c0.astype(np.uint64)
            p1 = np.uint64(0xCD9E8D57) * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return np.stack([c0, c1, c2, c3], -1)


def uniform01(seed, stream_id, n):
    """float32 [n]: element i = word (i & 3) of philox(seed, stream, i >> 2), top 24 bits / 2^24."""
    i = np.arange(n, dtype=np.uint64)
    r = philox4x32(int(seed) & M64, np.uint64(int(stream_id) & M64), i >> np.uint64(2))
    v = r[np.arange(n), (i & np.uint64(3)).astype(np.int64)]
    return (v >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def weight_noise(seed, step, layer_id, shape):
    """q = 1 weight noise of one layer (pointwise.hip: stream = step << 8 | layer id, index = C-order element)."""
    return uniform01(seed, ((int(step) << 8) | int(layer_id)) & M64, int(np.prod(shape))).reshape(shape)


def latent_noise(seed, step, block_ids, ch, spatial=8):
    """Rate-proxy noise of the latents of the given (global) block ids: [len(ids), ch, 2, 2, 2]
    (latent_tail.h: stream = (block << 20) ^ step * 0x9E3779B97F4A7C15, index = channel * 8 + position)."""
    out = np.empty((len(block_ids), ch, 2, 2, 2), np.float32)
    for j, b in enumerate(block_ids):
        sid = ((int(b) << 20) ^ ((int(step) * GOLDEN) & M64)) & M64
        out[j] = uniform01(seed, sid, ch * spatial).reshape(ch, 2, 2, 2)
    return out
