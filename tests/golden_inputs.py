"""Seeded inputs shared by tools/gen_golden.py (which runs the real reference) and the tests.

Everything here is deterministic torch-CPU / numpy RNG, so the generator and
the tests see identical inputs without the inputs being stored in the fixtures.
"""
import numpy as np
import torch

CONFIGS = {
    # S: configs 1-4 of BASELINE.json; W: config 5 (wide decoder)
    "S": dict(ch=3, channels=(8, 16, 8, 8), batch=2, param_seed=101, emb_seed=202, noise_seed=303),
    "W": dict(ch=8, channels=(16, 32, 16, 16), batch=1, param_seed=111, emb_seed=212, noise_seed=313),
}
HYPER = dict(lmbda=200.0, w1=10.0, w2=57.0, lr=1e-3, wemb=5.0, n_points=917 * 936.0)


# Training trajectory (tools/gen_golden.py:gen_trajectory): NVFPCC.py:105-254 run for 3 epochs on 14 blocks at
# batch 4 (three full mini-batches + a short one per epoch), q = 1 in epoch 0, q = 2 from --phase_change = 1
TRAJ = dict(tag="S", n_blocks=14, batch=4, epochs=3, phase_change=1, noise_seed=5, order_seed=909)


def traj_order(epoch):
    """Block ids in the order epoch `epoch` visits them (what the DataLoader + dataset permutation would yield)."""
    return np.random.default_rng(TRAJ["order_seed"] + epoch).permutation(TRAJ["n_blocks"]).astype(np.int64)


def perturb_state_(sd, seed):
    """Move every trainable tensor off its trivial initial value, in state-dict order."""
    g = torch.Generator().manual_seed(seed)
    for k, v in sd.items():
        if k.endswith("_init") or k.endswith("pedestal"):
            continue
        r = torch.randn(v.shape, generator=g)
        if k.endswith(".kernel"):
            v.add_(0.05 * r)
        elif k.endswith(".b"):
            v.add_(0.02 * r)
        elif k.endswith("beta") or k.endswith("gamma"):
            v.add_(0.01 * r)
        elif k.endswith("sigma"):
            v.add_(0.1 * r)
        elif k.endswith("mu"):
            v.add_(0.1 * r)
        else:
            raise KeyError(k)
    return sd


def make_emb(batch, ch, seed):
    g = torch.Generator().manual_seed(seed)
    return 1.0 + 0.5 * torch.randn(batch, ch, 2, 2, 2, generator=g)


def noise_stream(seed):
    """Yields callables shape -> U[0,1) tensor, one per rand_like call, in call order."""
    g = torch.Generator().manual_seed(seed)
    while True:
        yield lambda shape: torch.rand(tuple(shape), generator=g)


def sample_index(n, k):
    rng = np.random.default_rng(n * 7919 + k)
    return torch.from_numpy(np.sort(rng.choice(n, size=min(k, n), replace=False)))


def loss_case_inputs():
    g = torch.Generator().manual_seed(77)
    cases = {}
    shape = (2, 1, 8, 16, 16)
    p = torch.rand(shape, generator=g)
    gt = (torch.rand(shape, generator=g) < 0.05).float()
    dist = torch.rand(shape, generator=g) * 8 * (1 - gt)
    cases["random"] = (p, gt, dist)
    # saturated predictions: exact 0 / 1 and values beyond the 1e-9 clamp
    p2 = p.clone()
    flat = p2.view(-1)
    flat[0::7] = 0.0
    flat[1::7] = 1.0
    flat[2::7] = 1e-12
    flat[3::7] = 1.0 - 1e-7
    cases["saturated"] = (p2, gt, dist)
    # empty ground truth (no occupied voxel) and full ground truth
    cases["empty_gt"] = (p, torch.zeros(shape), dist + 1.0)
    cases["full_gt"] = (p, torch.ones(shape), torch.zeros(shape))
    return cases


def gdn_case_inputs():
    g = torch.Generator().manual_seed(88)
    cases = {}
    for name, c in (("c3", 3), ("c8", 8)):
        x = torch.randn(2, c, 4, 4, 4, generator=g)
        beta = torch.sqrt(torch.ones(c) + 2.0 ** -36) + 0.05 * torch.randn(c, generator=g)
        gamma = torch.sqrt(0.1 * torch.eye(c) + 2.0 ** -36) + 0.02 * torch.randn(c, c, generator=g).abs()
        gy = torch.randn(2, c, 4, 4, 4, generator=g)
        cases[name] = (x, beta, gamma, gy)
    # parameters below their floors: pins the LowerBound gradient rule
    c = 4
    x = torch.randn(2, c, 2, 2, 2, generator=g)
    beta = torch.tensor([1e-4, 0.5, 1e-5, 1.0])
    gamma = torch.sqrt(0.1 * torch.eye(c) + 2.0 ** -36)
    gamma[0, 1] = 1e-7
    gamma[2, 3] = -0.3
    gamma[1, 0] = 2e-6
    gy = torch.randn(2, c, 2, 2, 2, generator=g)
    cases["below_bound"] = (x, beta, gamma, gy)
    return cases


def codec_cases():
    """(symbols int16 in [0,1023], mu f32, sigma f32) triples for the latent arithmetic coder."""
    rng = np.random.default_rng(4242)
    cases = {}
    # a latent table like longdress l5: 917 blocks x 3 channels x 8 positions, symbols = round(latent) + 512
    n_blk, ch = 917, 3
    mu_c = np.array([0.21, -0.4, 0.05], np.float32)
    sg_c = np.array([1.7, 2.9, 0.8], np.float32)
    lat = np.round(rng.normal(mu_c[None, :, None], sg_c[None, :, None], (n_blk, ch, 8)))
    mu = np.broadcast_to(mu_c[None, :, None], lat.shape).reshape(-1).astype(np.float32) + 512
    sg = np.broadcast_to(sg_c[None, :, None], lat.shape).reshape(-1).astype(np.float32)
    cases["latents917"] = ((lat.reshape(-1) + 512).astype(np.int16), mu, sg)
    # five symbols
    cases["tiny"] = (np.array([512, 511, 515, 512, 509], np.int16), np.full(5, 512.3, np.float32),
                     np.full(5, 1.25, np.float32))
    # wide model, symbols across the alphabet, per-symbol parameters
    n = 1000
    sg2 = rng.uniform(20, 200, n).astype(np.float32)
    mu2 = rng.uniform(300, 700, n).astype(np.float32)
    sym2 = np.clip(np.round(rng.normal(mu2, sg2)), 0, 1023).astype(np.int16)
    cases["wide"] = (sym2, mu2, sg2)
    # empty message: only the terminator is coded
    cases["empty"] = (np.zeros(0, np.int16), np.zeros(0, np.float32), np.zeros(0, np.float32))
    return cases


def huffman_cases():
    """Integer weight pools (float32 arrays of whole numbers, as `ws[k].numpy() * qp` gives them,
    util_code_quantized_weights.py:37-51) for the Huffman codebook goldens."""
    rng = np.random.default_rng(777)
    cases = {}
    # a 4-bit trained decoder looks like this: a narrow two-sided geometric around 0
    cases["laplace"] = np.round(rng.laplace(0.0, 1.6, 52000)).clip(-9, 11).astype(np.float32)
    # exact ties everywhere: every value 0..7 occurs 64 times, -1 occurs 128 and 9 occurs 32 times
    t = np.concatenate([np.repeat(np.arange(8), 64), np.full(128, -1), np.full(32, 9)])
    cases["ties"] = rng.permutation(t).astype(np.float32)
    # two symbols, and a gap in the value range (zero-frequency bins are dropped)
    cases["gap"] = np.array([-3] * 5 + [4] * 11 + [5] * 11 + [7], np.float32)
    # one distinct value: the codebook is the empty word
    cases["single"] = np.full(37, 2, np.float32)
    return cases


def synthetic_cloud(seed=99, n_dir=40000, radius=70.0, center=(500.0, 530.0, 470.0)):
    """A 10-bit voxelised ellipsoid shell (~30 k points, ~100 level-5 cubes) for the pre-processing tests."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_dir, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = np.asarray(center) + d * np.array([radius, 0.8 * radius, 1.3 * radius])
    return np.unique(np.round(p).astype(np.int64), axis=0)


def write_cloud_ply(path, pts):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(pts))
        for x, y, z in pts.tolist():
            f.write(f"{x} {y} {z} 128 128 128\n")
