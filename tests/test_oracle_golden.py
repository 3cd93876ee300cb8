"""Pins oracle/nvf_oracle.py to outputs of the real reference (tests/golden/*.npz).

The fixtures were produced by tools/gen_golden.py importing /root/reference's
utils/network.py, gdn_3d.py and utils/loss.py; inputs are regenerated here from
the same seeds (tests/golden_inputs.py).  Tolerances: the oracle issues the same
aten CPU calls as the reference, so values agree to rounding (<= 1e-6 relative).
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from nvfpcc_amd.seeds import synthetic_seed
from nvfpcc_amd.synth import make_blocks
from oracle import nvf_oracle as O
from tests.golden_inputs import (CONFIGS, HYPER, perturb_state_, make_emb, noise_stream, sample_index,
                                 loss_case_inputs, gdn_case_inputs)

TRUNK_ORDER = ["up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls"]


def summary(t, n=64):
    a = t.detach().double().reshape(-1)
    idx = sample_index(a.numel(), n)
    return np.concatenate([[a.mean().item(), a.abs().sum().item()], a[idx].numpy()])


def close(a, b, rtol=2e-6, atol=1e-7):
    a = a.detach() if isinstance(a, torch.Tensor) else a
    b = b.detach() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def load_case(golden_dir, tag):
    cfg = CONFIGS[tag]
    G = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    P, used = O.build_state(cfg["ch"], cfg["channels"], synthetic_seed())
    fresh = {k: v.clone() for k, v in P.items()}
    perturb_state_(P, cfg["param_seed"])
    emb = make_emb(cfg["batch"], cfg["ch"], cfg["emb_seed"])
    gts, dists = make_blocks(cfg["batch"])
    return cfg, G, P, fresh, used, emb, gts, dists


@pytest.mark.parametrize("tag", ["S", "W"])
def test_seed_init_and_state_layout(golden_dir, tag):
    cfg, G, P, fresh, used, emb, gts, dists = load_case(golden_dir, tag)
    assert used == int(G["seed_used"])
    assert used == {"S": 52127, "W": 210683}[tag]
    for k, v in fresh.items():
        if k.endswith("_init"):
            close(summary(v, 32), G["init/" + k])
    assert len(O.trainable_keys(P)) == 28
    assert len(P) == 50
    assert hashlib.sha256(gts.tobytes()).digest() == G["gt_sha"].tobytes()
    assert dists.sum() == float(G["dist_sum"])


@pytest.mark.parametrize("tag", ["S", "W"])
def test_forward_eval(golden_dir, tag):
    cfg, G, P, fresh, used, emb, gts, dists = load_case(golden_dir, tag)
    for q in (2, 0):
        keep = {}
        with torch.no_grad():
            out, cls, nbits, lbits = O.net_forward(P, emb, "eval", q, keep=keep)
        p = f"fwd_q{q}/"
        if q == 2:
            close(out, G[p + "out"], atol=2e-7)
            close(cls[0], G[p + "cls0"], atol=2e-7)
            close(cls[1], G[p + "cls1"], atol=2e-7)
            close(keep["latent"], G[p + "latent"])
            assert np.array_equal(keep["latent_rounded"].numpy(), G[p + "latent_rounded"])
            # occupancy decisions identical at every threshold the CLI uses
            for thh in (0.5, 0.6, 0.64, 0.65):
                assert np.array_equal(out.numpy() > thh, G[p + "out"] > thh)
        else:
            close(summary(out), G[p + "out"], rtol=1e-5)
        close(nbits, G[p + "net_bits"], rtol=1e-6)
        close(lbits, G[p + "latent_bits"], rtol=1e-6)
        # per-layer summaries: reference hooks see the conv outputs before ReLU/sigmoid
        pre = {"activation": keep["igdn"]}
        close(summary(pre["activation"]), G[p + "act/activation"], rtol=1e-5)
        for name, post in (("conv0", "conv0"), ("up1", "up1"), ("conv1", "conv1"), ("up2", "up2"),
                           ("conv2", "conv2")):
            ref_pre = G[p + "act/" + name]
            # compare through the ReLU on the sampled entries
            mine = summary(keep[post])
            np.testing.assert_allclose(mine[2:], np.maximum(ref_pre[2:], 0), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["S", "W"])
def test_gradients_eval_q2(golden_dir, tag):
    cfg, G, P, fresh, used, emb, gts, dists = load_case(golden_dir, tag)
    keys = O.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    emb.requires_grad_(True)
    gt, dist = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    loss, *_ = O.rd_loss(P, emb, gt, dist, HYPER["n_points"], HYPER["lmbda"], HYPER["w1"], HYPER["w2"],
                         "eval", 2)
    loss.backward()
    close(loss.item(), G["grad_q2/loss"], rtol=1e-6)
    close(emb.grad, G["grad_q2/emb"], rtol=1e-4, atol=1e-5)
    for k in keys:
        g = P[k].grad
        ref = G["grad_q2/" + k]
        if tag == "S":
            scale = max(np.abs(ref).max(), 1e-6)
            np.testing.assert_allclose(g.numpy(), ref, rtol=1e-4, atol=1e-5 * scale)
        else:
            mine = summary(g, 256)
            scale = max(np.abs(ref[2:]).max(), 1e-6)
            np.testing.assert_allclose(mine, ref, rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("tag", ["S", "W"])
def test_train_step_q1_seeded_noise(golden_dir, tag):
    cfg, G, P, fresh, used, emb, gts, dists = load_case(golden_dir, tag)
    keys = O.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    emb.requires_grad_(True)
    gt, dist = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    stream = noise_stream(cfg["noise_seed"])
    u_latent = next(stream)(emb.shape)
    u_w = {n: next(stream)(P[f"reconstructor.{n}.kernel"].shape) for n in TRUNK_ORDER}
    loss, out, cls, nbits, lbits = O.rd_loss(P, emb, gt, dist, HYPER["n_points"], HYPER["lmbda"],
                                             HYPER["w1"], HYPER["w2"], "train", 1, u_latent, u_w)
    loss.backward()
    close(loss.item(), G["train_q1/loss"], rtol=1e-6)
    close(summary(out), G["train_q1/out"], rtol=1e-5)
    close(lbits, G["train_q1/latent_bits"], rtol=1e-6)
    close(nbits, G["train_q1/net_bits"], rtol=1e-6)
    close(emb.grad, G["train_q1/grad_emb"], rtol=1e-4, atol=1e-5)
    for k in keys:
        ref = G["train_q1/grad/" + k]
        scale = max(np.abs(ref[2:]).max(), 1e-6)
        np.testing.assert_allclose(summary(P[k].grad, 48), ref, rtol=1e-4, atol=1e-5 * scale)
    # one Adam step each, restated (A.7) -- compare with torch.optim.Adam run by the reference
    with torch.no_grad():
        for k in keys:
            m, v = torch.zeros_like(P[k]), torch.zeros_like(P[k])
            O.adam_update(P[k], P[k].grad, m, v, 1, HYPER["lr"])
        m, v = torch.zeros_like(emb), torch.zeros_like(emb)
        O.adam_update(emb, emb.grad, m, v, 1, HYPER["lr"] * HYPER["wemb"])
    close(emb, G["train_q1/emb_after"], rtol=1e-6, atol=1e-7)
    for k in keys:
        close(summary(P[k], 48), G["train_q1/after/" + k], rtol=1e-5, atol=1e-6)


def test_losses_and_metrics(golden_dir):
    G = np.load(os.path.join(golden_dir, "loss.npz"))
    for name, (p, gt, dist) in loss_case_inputs().items():
        p = p.clone().requires_grad_(True)
        f = O.focal_dense(p, gt, alpha=0.85)
        f.backward()
        close(f.item(), G[name + "/focal"], rtol=1e-6)
        close(p.grad, G[name + "/focal_grad"], rtol=1e-6, atol=1e-7)
        p.grad = None
        s = O.surf_focal_dense(p, gt, dist, beta=1, alpha=0.9)
        s.backward()
        close(s.item(), G[name + "/surf"], rtol=1e-6)
        close(p.grad, G[name + "/surf_grad"], rtol=1e-6, atol=1e-7)
        tpr, tnr = O.acc_dense(p.detach(), gt, 0.5)
        np.testing.assert_array_equal(np.array([tpr.item(), tnr.item()]), G[name + "/acc"])  # nan==nan ok
        sse, den = O.sse1(p.detach(), dist, 0.6)
        close([sse.item(), den.item()], G[name + "/sse1"], rtol=1e-6)
        se = O.squared_error_map(p.detach(), dist, 0.6)            # get_se (loss.py:123-128), encode-side only
        assert se.shape == (p.shape[0], 2) + tuple(p.shape[2:])
        close(summary(se), G[name + "/se"], rtol=1e-6)


def test_gdn_forward_backward(golden_dir):
    G = np.load(os.path.join(golden_dir, "gdn.npz"))
    for name, (x, beta, gamma, gy) in gdn_case_inputs().items():
        for inv in (False, True):
            xi = x.clone().requires_grad_(True)
            b = beta.clone().requires_grad_(True)
            g = gamma.clone().requires_grad_(True)
            y = O.gdn3d(xi, b, g, inverse=inv)
            y.backward(gy)
            p = f"{name}/{'igdn' if inv else 'gdn'}/"
            close(y, G[p + "y"], rtol=1e-6, atol=1e-7)
            close(xi.grad, G[p + "dx"], rtol=1e-5, atol=1e-6)
            close(b.grad, G[p + "dbeta"], rtol=1e-5, atol=1e-6)
            close(g.grad, G[p + "dgamma"], rtol=1e-5, atol=1e-6)


def test_lr_schedule_quirk(golden_dir):
    table = np.load(os.path.join(golden_dir, "schedule.npz"))["table"]
    for epoch, lr_dec, lr_emb in table:
        close(O.lr_at_epoch(1e-3, int(epoch)), lr_dec, rtol=1e-9, atol=0)
        close(5e-3, lr_emb, rtol=1e-12, atol=0)
    assert table[-1, 1] < 1.1e-9  # 1e-3 * 0.01^3


def test_cli_lr_schedule_is_the_reference_table(golden_dir):
    """NVFPCC.lr_at_epoch (what the command line feeds the engine every epoch) against the table the reference's two
    MultiStepLR objects produce (tools/gen_golden.py, NVFPCC.py:117,126,253-254): x0.01 per milestone."""
    import NVFPCC
    table = np.load(os.path.join(golden_dir, "schedule.npz"))["table"]
    for epoch, lr_dec, _ in table:
        close(NVFPCC.lr_at_epoch(1e-3, int(epoch)), lr_dec, rtol=1e-9, atol=0)
    for e in (0, 1, 299, 300, 301, 399, 400, 449, 450, 500):
        assert NVFPCC.lr_at_epoch(2e-4, e) == O.lr_at_epoch(2e-4, e)


def test_quantised_checkpoint_has_the_reference_key_set():
    """manipulate_weights.quantise keeps exactly the 28 keys of the reference's bypass_key_list + key_list
    (manipulate_weights.py:19-32): the latent generator incl. its *_init buffers, the entropy coder, the IGDN, the seven
    trunk layers' kernel + bias, the weight likelihood model -- and neither the two coarse heads nor the trunk's *_init
    buffers (the decoder re-creates those from its seed file)."""
    import manipulate_weights as MW
    P, _ = O.build_state(3, (8, 16, 8, 8), synthetic_seed())
    perturb_state_(P, 5)
    qd, lo, hi = MW.quantise(P, 16)
    trunk = ["up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls"]
    expect = (["latent_gen.h_analysis_2." + s for s in ("kernel", "b", "kernel_init", "b_init")]
              + ["latent_gen.gdn_2." + s for s in ("beta", "gamma", "pedestal")]
              + ["entropy_coder.sigma", "entropy_coder.mu"]
              + ["reconstructor.activation." + s for s in ("beta", "gamma", "pedestal")]
              + [f"reconstructor.{n}.{s}" for n in trunk for s in ("kernel", "b")]
              + ["reconstructor.likelihood_model.sigma", "reconstructor.likelihood_model.mu"])
    assert len(expect) == 28 and sorted(qd) == sorted(expect)
    assert type(qd) is dict                                   # a plain dict, as the reference saves it
    for n in trunk:
        k = f"reconstructor.{n}.kernel"
        assert torch.equal(qd[k], torch.round(P[k] * 16) / 16)
        assert torch.equal(qd[k] * 16, torch.round(qd[k] * 16))
        assert torch.equal(qd[f"reconstructor.{n}.b"], P[f"reconstructor.{n}.b"])
    assert lo == min(torch.round(P[f"reconstructor.{n}.kernel"] * 16).min().item() for n in trunk)
    assert hi == max(torch.round(P[f"reconstructor.{n}.kernel"] * 16).max().item() for n in trunk)


def test_adam_restatement_matches_torch():
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(1000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p = p0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(1000, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        O.adam_update(p, grad, m, v, step, 1e-3)
    close(p, p_ref.detach(), rtol=1e-6, atol=1e-7)


# ---- the training trajectory: pins OracleTrainer (the object bench.py's cpu_baseline times) ---------------------------
def traj_noise_fn(ch):
    from tests import philox_np
    from tests.golden_inputs import TRAJ

    def fn(step, ids, q):
        u_lat = torch.from_numpy(philox_np.latent_noise(TRAJ["noise_seed"], step, list(ids), ch))
        u_w = {}
        if q == 1:
            shapes = {n: s for (n, _, s, _) in [(t[0].split(".")[-1], t[1], t[2], t[3])
                                                for t in O.layer_table(ch, CONFIGS[TRAJ["tag"]]["channels"])]}
            u_w = {n: torch.from_numpy(philox_np.weight_noise(TRAJ["noise_seed"], step, i + 1, shapes[n]))
                   for i, n in enumerate(TRUNK_ORDER)}
        return u_lat, u_w
    return fn


def traj_compare_params(G, prefix, get, rtol_abs):
    """get(key) -> CPU tensor; golden holds small tensors whole, large ones as summary(256).  Returns the worst
    |difference| / max|golden| over all tensors."""
    worst = 0.0
    for name in [k[len(prefix):] for k in G.files if k.startswith(prefix)]:
        want = G[prefix + name]
        t = get(name)
        got = t.detach().double().reshape(-1).numpy() if t.numel() <= 1024 else summary(t, 256)
        want = np.asarray(want, np.float64).reshape(-1)
        scale = max(np.abs(want).max(), 1e-3)
        err = np.abs(got - want).max() / scale
        assert err <= rtol_abs, (name, err)
        worst = max(worst, err)
    return worst


def test_oracle_trainer_reproduces_the_reference_trajectory(golden_dir):
    """NVFPCC.py:105-254 for three epochs (q = 1, then q = 2 twice) on 14 blocks at batch 4: Adam moment accumulation
    across steps, the zero_grad / backward / step order of both optimisers, the latent phase on the post-mini-batch
    decoder, all with the reference's own loop as the generator (tools/gen_golden.py:gen_trajectory)."""
    from tests.golden_inputs import TRAJ, traj_order
    G = np.load(os.path.join(golden_dir, "trajectory.npz"))
    cfg = CONFIGS[TRAJ["tag"]]
    n, B = TRAJ["n_blocks"], TRAJ["batch"]
    gts, dists = make_blocks(n)
    gt, dist = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    assert float(gts.sum()) == float(G["n_points"])
    torch.set_num_threads(8)
    tr = O.OracleTrainer(cfg["ch"], cfg["channels"], synthetic_seed(), n_leaf=n, n_points=float(gts.sum()),
                         lr=HYPER["lr"], wemb=HYPER["wemb"], lmbda=HYPER["lmbda"], w1=HYPER["w1"], w2=HYPER["w2"],
                         noise_fn=traj_noise_fn(cfg["ch"]))
    for epoch in range(TRAJ["epochs"]):
        q = 1 if epoch < TRAJ["phase_change"] else 2
        tr.set_epoch(epoch)
        order = traj_order(epoch)
        assert np.array_equal(order, G[f"epoch{epoch}/order"])
        losses = [tr.train_step(torch.from_numpy(order[s:s + B]), gt[order[s:s + B]], dist[order[s:s + B]], q)
                  for s in range(0, n, B)]
        close(np.array(losses), G[f"epoch{epoch}/step_loss"], rtol=2e-6)
        close(tr.latent_step(gt, dist, q), G[f"epoch{epoch}/latent_loss"], rtol=2e-6)
        close(tr.emb, G[f"epoch{epoch}/emb"], rtol=1e-5, atol=2e-6)
        traj_compare_params(G, f"epoch{epoch}/param/", lambda k: tr.P[k], 2e-5)
