"""GPU parity of the whole operator stack (Net forward / backward / Adam) against golden vectors
generated from the real reference (tests/golden/net_*.npz).

Stated tolerances: probabilities <= 1e-5 abs; occupancy identical on every voxel with
|p - thh| > 2e-6; scalar rates/loss <= 2e-5 relative; gradients: per decoder and golden, 3 x the
worst case measured on MI355X (TOLS below: <= 1.1e-4 of the tensor's max magnitude; fp32 sums in a
different, fixed order than oneDNN's).
"""
import os

import numpy as np
import pytest
import torch

from nvfpcc_amd.seeds import synthetic_seed
from nvfpcc_amd.synth import make_blocks
from tests.golden_inputs import CONFIGS, HYPER, perturb_state_, make_emb, noise_stream, sample_index

pytestmark = pytest.mark.gpu
TRUNK_ORDER = ["up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls"]
# element-wise relative tolerance on gradient entries above 1e-3 of their tensor's maximum (measured worst case on
# MI355X: 1.1e-4 narrow decoder, all entries; 3.2e-3 wide decoder, 256 sampled entries per tensor)
REL_TOL = 1e-2
REL_SEEN = []
# per decoder and comparison: (tensor-max-relative bound, element-wise relative bound) = 3 x the worst case measured on
# MI355X (printed by every run as "[grad_close <tag>]"; SURVEY 8(c) asks for <= 1e-4 relative on gradients: the fp32
# sums of up to 5e5 products run in another fixed order than oneDNN's, measured below)
# measured (r05, GPUTEST log): S/golden_q2 (3.35e-5, 1.09e-4), W/golden_q2 (3.52e-5, 3.16e-3 on 256 sampled entries per tensor),
# S/golden_q1 (3.35e-5, 3.35e-5), W/golden_q1 (1.26e-5, 1.82e-5).  These goldens are float32 numbers of the real reference
# (oneDNN), ReLU-mask disagreements included; the arithmetic itself is held to 8e-6 against the float64 oracle with the
# masks imposed in tests/test_gpu_measured_path.py
TOLS = {"S/golden_q2": (1e-4, 3.3e-4), "W/golden_q2": (1.1e-4, 1e-2), "S/golden_q1": (1e-4, 1e-4),
        "W/golden_q1": (4e-5, 6e-5)}


def summary(t, n=64):
    a = t.detach().double().cpu().reshape(-1)
    idx = sample_index(a.numel(), n)
    return np.concatenate([[a.mean().item(), a.abs().sum().item()], a[idx].numpy()])


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda")


def build(tag, gpu, golden_dir):
    from nvfpcc_amd import network
    from nvfpcc_amd.model import Net
    cfg = CONFIGS[tag]
    G = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    network.reset_seed(synthetic_seed())
    net = Net(None, "Gaussian", cfg["ch"], ",".join(str(c) for c in cfg["channels"]), verbose=False)
    sd = net.state_dict()
    perturb_state_(sd, cfg["param_seed"])
    net.load_state_dict(sd)
    net = net.to(gpu)
    emb = make_emb(cfg["batch"], cfg["ch"], cfg["emb_seed"]).to(gpu).requires_grad_(True)
    gts, dists = make_blocks(cfg["batch"])
    gt = torch.from_numpy(gts).float().to(gpu)
    dist = torch.from_numpy(dists).float().to(gpu)
    return cfg, G, net, emb, gt, dist


def full_loss(net, emb, gt, dist, mode, q, **kw):
    from nvfpcc_amd.loss import get_focal_dense, get_surf_focal_dense
    from nvfpcc_amd.model import MultiscaleProcessor
    out, cls, nbits, lbits = net(emb, mode, q, **kw)
    pyr = MultiscaleProcessor()(gt)
    b_latent = lbits.sum() / gt.sum()
    b_net = nbits.sum() / HYPER["n_points"]
    loss = (get_surf_focal_dense(out, gt, dist, beta=1, alpha=0.9) + get_focal_dense(cls[0], pyr[0], alpha=0.85)
            + get_focal_dense(cls[1], pyr[1], alpha=0.85)
            + HYPER["lmbda"] * (b_latent * HYPER["w1"] + b_net * HYPER["w2"]))
    return loss, out, cls, nbits, lbits


SEEN = {}       # tag -> [worst err / tensor max, worst element-wise relative error]: printed by the tests (drift is visible)


def grad_close(mine, ref, tol=2e-4, rtol=REL_TOL, tag=None):
    """Two statements: (1) every entry within `tol` of the tensor's largest magnitude; (2) element by element, every
    entry above 1e-3 of that magnitude within `rtol` of ITS OWN value -- so small-but-significant entries are checked
    too (fp32 sums of up to 5e5 products in another order than oneDNN's; measured worst case in the comment at REL_TOL)."""
    mine = np.asarray(mine, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    scale = max(np.abs(ref).max(), 1e-9)
    err = np.abs(mine - ref).max() / scale
    seen = SEEN.setdefault(tag, [0.0, 0.0])
    seen[0] = max(seen[0], err)
    assert err < tol, (tag, err, tol)
    big = np.abs(ref) > 1e-3 * scale
    if big.any():
        rel = (np.abs(mine - ref)[big] / np.abs(ref)[big]).max()
        REL_SEEN.append(rel)
        seen[1] = max(seen[1], rel)
        assert rel < rtol, (tag, rel, rtol)


def report(tag):
    e, r = SEEN.get(tag, (0.0, 0.0))
    print(f"[grad_close {tag}] worst |err| / tensor max = {e:.2e}, worst element-wise relative (entries > 1e-3 max) = {r:.2e}")


@pytest.mark.parametrize("tag", ["S", "W"])
def test_forward_eval_matches_reference(tag, gpu, golden_dir):
    cfg, G, net, emb, gt, dist = build(tag, gpu, golden_dir)
    with torch.no_grad():
        out, cls, nbits, lbits = net(emb, "eval", 2)
        rounded = net.entropy_coder(net.latent_gen(emb), "eval")[0]
    ref = G["fwd_q2/out"]
    o = out.cpu().numpy()
    assert np.abs(o - ref).max() < 1e-5
    assert np.abs(cls[0].cpu().numpy() - G["fwd_q2/cls0"]).max() < 1e-5
    assert np.abs(cls[1].cpu().numpy() - G["fwd_q2/cls1"]).max() < 1e-5
    assert np.array_equal(rounded.cpu().numpy(), G["fwd_q2/latent_rounded"])
    for thh in (0.5, 0.6, 0.64, 0.65):
        decided = np.abs(ref - thh) > 2e-6
        assert np.array_equal((o > thh)[decided], (ref > thh)[decided])
        print(f"[{tag}] thh={thh}: {int((~decided).sum())} voxels within 2e-6 of the threshold, "
              f"{int(((o > thh) != (ref > thh)).sum())} flips")
    np.testing.assert_allclose(nbits.cpu().numpy(), G["fwd_q2/net_bits"], rtol=2e-5)
    np.testing.assert_allclose(lbits.cpu().numpy(), G["fwd_q2/latent_bits"], rtol=2e-5)
    with torch.no_grad():
        out0, _, nbits0, _ = net(emb, "eval", 0)
    np.testing.assert_allclose(summary(out0), G["fwd_q0/out"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["S", "W"])
def test_forward_is_batch_invariant(tag, gpu, golden_dir):
    cfg, G, net, emb, gt, dist = build(tag, gpu, golden_dir)
    e = torch.cat([emb.detach(), emb.detach() * 0.7, emb.detach() + 0.3], 0)[:3]
    with torch.no_grad():
        full = net(e, "eval", 2)[0]
        singles = torch.cat([net(e[i:i + 1].contiguous(), "eval", 2)[0] for i in range(e.shape[0])], 0)
    assert torch.equal(full, singles), "encode at batch B must equal decode at batch 1 bit for bit"


@pytest.mark.parametrize("tag", ["S", "W"])
def test_gradients_match_reference(tag, gpu, golden_dir):
    cfg, G, net, emb, gt, dist = build(tag, gpu, golden_dir)
    loss, *_ = full_loss(net, emb, gt, dist, "eval", 2)
    loss.backward()
    assert abs(loss.item() - float(G["grad_q2/loss"])) < 2e-5 * abs(float(G["grad_q2/loss"]))
    t = tag + "/golden_q2"
    grad_close(emb.grad.cpu().numpy(), G["grad_q2/emb"], *TOLS[t], tag=t)
    for k, p in net.named_parameters():
        ref = G["grad_q2/" + k]
        assert p.grad is not None, k
        if tag == "S":
            grad_close(p.grad.cpu().numpy(), ref, *TOLS[t], tag=t)
        else:
            grad_close(summary(p.grad, 256)[2:], ref[2:], *TOLS[t], tag=t)
    report(t)
    print(f"[{tag}] worst element-wise relative gradient error (entries > 1e-3 max): {max(REL_SEEN):.2e}")


@pytest.mark.parametrize("tag", ["S", "W"])
def test_train_step_with_seeded_noise_and_adam(tag, gpu, golden_dir):
    from nvfpcc_amd import ops
    cfg, G, net, emb, gt, dist = build(tag, gpu, golden_dir)
    stream = noise_stream(cfg["noise_seed"])
    u_latent = next(stream)(emb.shape).to(gpu)
    u_w = {n: next(stream)(getattr(net.reconstructor, n).kernel.shape).to(gpu) for n in TRUNK_ORDER}
    loss, out, cls, nbits, lbits = full_loss(net, emb, gt, dist, "train", 1, u_latent=u_latent, u_w=u_w)
    loss.backward()
    assert abs(loss.item() - float(G["train_q1/loss"])) < 2e-5 * abs(float(G["train_q1/loss"]))
    np.testing.assert_allclose(summary(out), G["train_q1/out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lbits.detach().cpu().numpy(), G["train_q1/latent_bits"], rtol=2e-5)
    np.testing.assert_allclose(nbits.detach().cpu().numpy(), G["train_q1/net_bits"], rtol=2e-5)
    t = tag + "/golden_q1"
    grad_close(emb.grad.cpu().numpy(), G["train_q1/grad_emb"], *TOLS[t], tag=t)
    for k, p in net.named_parameters():
        grad_close(summary(p.grad, 48)[2:], G["train_q1/grad/" + k][2:], *TOLS[t], tag=t)
    report(t)
    # fused Adam (step 1) on every tensor, then compare with torch.optim.Adam run by the reference
    with torch.no_grad():
        for p in list(net.parameters()) + [emb]:
            lr = HYPER["lr"] * (HYPER["wemb"] if p is emb else 1.0)
            flat = p.detach().reshape(-1)
            ops.adam_step(flat, p.grad.reshape(-1).contiguous(), torch.zeros_like(flat), torch.zeros_like(flat), lr, 1)
    np.testing.assert_allclose(emb.detach().cpu().numpy(), G["train_q1/emb_after"], rtol=1e-5, atol=2e-6)
    for k, p in net.named_parameters():
        np.testing.assert_allclose(summary(p, 48), G["train_q1/after/" + k], rtol=1e-4, atol=2e-6)
