"""Parity on the path the bench line is measured on: the 16- and 4-step UNROLLED graphs fed by a
device-resident schedule at batch 16 over 917 resident blocks; one batch-16 step directly against the CPU oracle (q = 2
and q = 1 with the engine's own counter-RNG draws fed to the oracle); and decoder configurations other than the two of
BASELINE.json (the reference's argparse default `--ch 8 --chanstr 8,16,8,8`, NVFPCC.py:723-735, and an off-grid string)."""
import numpy as np
import pytest
import torch

from nvfpcc_amd.seeds import synthetic_seed
from nvfpcc_amd.synth import make_blocks
from tests.golden_inputs import perturb_state_, make_emb
from tests.philox_np import latent_noise, weight_noise
from tests.test_gpu_net import grad_close

pytestmark = pytest.mark.gpu
H = dict(lmbda=200.0, w1=10.0, w2=57.0, lr=1e-3, wemb=5.0)
NPTS = 917 * 936.0
TRUNK = ("up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda")


def make(gpu, ch, channels, nblk, distinct=None, param_seed=101, emb_seed=202):
    from nvfpcc_amd import network
    from nvfpcc_amd.engine import TrainEngine
    from nvfpcc_amd.model import Net
    network.reset_seed(synthetic_seed())
    network.set_noise_seed(0, 0)
    net = Net(None, "Gaussian", ch, ",".join(str(c) for c in channels), verbose=False)
    sd = net.state_dict()
    perturb_state_(sd, param_seed)
    net.load_state_dict(sd)
    P = {k: v.clone() for k, v in net.state_dict().items()}          # the oracle's copy (CPU)
    net = net.to(gpu)
    distinct = nblk if distinct is None else min(distinct, nblk)
    gts, dists = make_blocks(distinct)
    reps = (nblk + distinct - 1) // distinct
    gt = torch.from_numpy(np.tile(gts, (reps, 1, 1, 1, 1))[:nblk]).float()
    dist = torch.from_numpy(np.tile(dists, (reps, 1, 1, 1, 1))[:nblk]).float()
    emb = make_emb(nblk, ch, emb_seed)
    eng = TrainEngine(net, gt.to(gpu), dist.to(gpu), n_points_total=NPTS, emb=emb.to(gpu), seed=0, **H)
    return net, eng, P, gt, dist, emb


def test_unrolled_schedule_replay_equals_host_steps(gpu):
    """bench.py and NVFPCC.py train drive GraphedTrainStep.load_schedule((ids, n_pts)) + replay_all(): 20 steps = one
    16-step graph + one 4-step graph, the last-arriving workgroup of every step's final launch handing the step buffer
    over to the next schedule row INSIDE the graph.  Against 20 host-launched train_steps from the same state: parameters,
    both Adam moments, the latent table and the epoch accumulators bit for bit (q = 1: weight and latent noise on)."""
    from nvfpcc_amd.engine import GraphedTrainStep
    net, eng, P, gt, dist, emb = make(gpu, 3, (8, 16, 8, 8), 917, distinct=24)
    eng.enable_epoch_stats()
    B, K = 16, 20
    order = np.random.default_rng(11).permutation(917)[:K * B].reshape(K, B).astype(np.int64)
    npts = eng.counts[order].sum(axis=1).astype(np.float64)
    snap = lambda: (eng.flat_p.clone(), eng.flat_m.clone(), eng.flat_v.clone(), eng.noise_step, eng.opt_step)

    def restore(s):
        eng.flat_p.copy_(s[0]); eng.flat_m.copy_(s[1]); eng.flat_v.copy_(s[2])
        eng.noise_step, eng.opt_step = s[3], s[4]
        eng.epoch_acc.zero_()
    s0 = snap()
    for k in range(K):
        eng.train_step(order[k], 1, n_pts=float(npts[k]))
    torch.cuda.synchronize()
    ref = (eng.flat_p.clone(), eng.flat_m.clone(), eng.flat_v.clone(), eng.emb.clone(), eng.epoch_acc.clone(), eng.flat_g.clone())
    assert float((ref[0] - s0[0]).abs().max()) > 0
    restore(s0)
    g = GraphedTrainStep(eng, B, 1)                      # captures run the body: restore what they touched
    assert sorted(g.graphs_u) == [2, 4, 8, 16]
    restore(s0)
    g.load_schedule((order, npts))
    g.replay_all()
    torch.cuda.synchronize()
    assert not g.pending and eng.opt_step == s0[4] + K and eng.noise_step == s0[3] + K
    assert torch.equal(eng.flat_p, ref[0]), float((eng.flat_p - ref[0]).abs().max())
    assert torch.equal(eng.flat_m, ref[1]) and torch.equal(eng.flat_v, ref[2])
    assert torch.equal(eng.emb, ref[3])
    assert torch.equal(eng.epoch_acc, ref[4]), (eng.epoch_acc, ref[4])
    assert torch.equal(eng.flat_g, ref[5])               # the last step's gradients
    assert float(eng.epoch_acc[7]) == K


def _oracle_step(P, emb, gt, dist, ids, q, n_pts, noise_step, seed=0, layer_ids=None):
    """The oracle's objective and gradients for the mini-batch `ids` with the engine's own noise draws."""
    from oracle import nvf_oracle as O
    P = {k: v.clone() for k, v in P.items()}
    keys = O.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    e = emb[ids].clone().requires_grad_(True)
    ch = emb.shape[1]
    u_latent = torch.from_numpy(latent_noise(seed, noise_step, ids, ch))
    u_w = None
    if q == 1:
        u_w = {n: torch.from_numpy(weight_noise(seed, noise_step, layer_ids[n], tuple(P["reconstructor." + n + ".kernel"].shape)))
               for n in TRUNK}
    out, cls, nbits, lbits = O.net_forward(P, e, "train", q, u_latent, u_w)
    g = gt[ids]
    pyr = O.gt_pyramid(g)
    loss = (O.surf_focal_dense(out, g, dist[ids], beta=1, alpha=0.9) + O.focal_dense(cls[0], pyr[0], alpha=0.85)
            + O.focal_dense(cls[1], pyr[1], alpha=0.85)
            + H["lmbda"] * (lbits.sum() / n_pts * H["w1"] + nbits.sum() / NPTS * H["w2"]))
    loss.backward()
    return loss.item(), out.detach(), [c.detach() for c in cls], {k: P[k].grad for k in keys}, e.grad


def _layer_ids(net):
    rec = net.reconstructor
    return {n: getattr(rec, n).layer_id for n in TRUNK}


def _check_against_oracle(eng, net, P, gt, dist, emb, ids, q, tol=2e-4, rtol=1e-2, conditioned=False):
    """One mini-batch step of the engine (loss, probabilities, every weight-gradient slice) against the oracle.
    ``conditioned``: the reference is the oracle in FLOAT64 and a slice may be off by max(tol, 10 x the fp32 oracle's own
    distance from float64) of its largest entry -- for parameters behind the latent rate term, whose gradient is a ratio of
    differences of Gaussian CDFs and as ill-conditioned as the latents' (profiles/r04_latent_grad_conditioning.md)."""
    ids = np.asarray(ids, np.int64)
    n_pts = float(eng.counts[ids].sum())
    lids = _layer_ids(net)
    a = eng.train_step(ids, q, update=False)
    loss_ref, out_ref, cls_ref, g_ref, _ = _oracle_step(P, emb, gt, dist, ids, q, n_pts, eng.noise_step, layer_ids=lids)
    assert float((a["p2"].cpu() - out_ref).abs().max()) < 1e-5
    assert float((a["p0"].cpu() - cls_ref[0]).abs().max()) < 1e-5 and float((a["p1"].cpu() - cls_ref[1]).abs().max()) < 1e-5
    assert abs(eng.loss_value() - loss_ref) < 2e-5 * abs(loss_ref), (eng.loss_value(), loss_ref)
    if conditioned:
        g64 = _oracle_step({k: v.double() for k, v in P.items()}, emb.double(), gt.double(), dist.double(), ids, q, n_pts,
                           eng.noise_step, layer_ids=lids)[3]
    for name, (off, n) in eng.slices.items():
        mine = eng.flat_g[off:off + n].cpu().numpy()
        if not conditioned:
            grad_close(mine, g_ref[name].numpy(), tol=tol, rtol=rtol)
            continue
        r64 = g64[name].numpy().reshape(-1)
        scale = max(np.abs(r64).max(), 1e-9)
        cond = np.abs(g_ref[name].double().numpy().reshape(-1) - r64).max() / scale
        err = np.abs(mine.astype(np.float64).reshape(-1) - r64).max() / scale
        assert err < max(tol, 10 * cond), (name, err, cond)
    return loss_ref


@pytest.mark.parametrize("q", [2, 1])
def test_batch16_train_step_matches_the_oracle(q, gpu):
    """The bench configuration's batch (16 blocks of 917 resident): loss, probabilities, all 28 gradient slices against the
    CPU oracle on the same 16 blocks -- slab counts, workgroup caps and the eight-wave kernel variants depend on the
    batch, and the golden vectors stop at batch 4.  q = 1 feeds the oracle the engine's counter-RNG draws (Philox
    restated in NumPy, tests/philox_np.py).  The latent gradient comes from a 16-block latent pass."""
    net, eng, P, gt, dist, emb = make(gpu, 3, (8, 16, 8, 8), 40)
    ids = np.random.default_rng(3).permutation(40)[:16]
    _check_against_oracle(eng, net, P, gt, dist, emb, ids, q)
    # latent gradient: the full-batch latent pass over 16 resident blocks (n_pts = all of them, NVFPCC.py:233-250)
    net2, eng2, P2, gt2, dist2, emb2 = make(gpu, 3, (8, 16, 8, 8), 16)
    a, de = eng2.latent_step(q, update=False)
    n_all = float(eng2.counts.sum())
    _, _, _, _, de_ref = _oracle_step(P2, emb2, gt2, dist2, np.arange(16), q, n_all, eng2.noise_step, layer_ids=_layer_ids(net2))
    grad_close(de.cpu().numpy(), de_ref.numpy())


# the decoders the engine has fused / matrix-core launches for; anything else must either run (generic kernels) and
# agree with the oracle, or refuse with a message naming these
CONFIGS3 = [(8, (8, 16, 8, 8)),       # the reference's argparse default: --ch 8 --chanstr 8,16,8,8 (NVFPCC.py:723-735)
            (4, (4, 8, 4, 4)),        # off-grid: no instantiation of the fused stem / one-launch heads / trunk5
            (3, (8, 8, 8, 8))]        # narrow trunk, other stem


@pytest.mark.parametrize("ch,channels", CONFIGS3)
def test_other_channel_strings_match_the_oracle_or_refuse(ch, channels, gpu):
    """`--ch` / `--chanstr` beyond BASELINE.json's two decoders: forward + all gradients through whatever launches the
    engine picks, against the oracle -- or an explicit NotImplementedError that names the supported strings
    ("Decoder configurations")."""
    try:
        net, eng, P, gt, dist, emb = make(gpu, ch, channels, 6)
    except NotImplementedError as e:
        assert "8,16,8,8" in str(e) and "16,32,16,16" in str(e), str(e)
        return
    for q in (2, 1):
        _check_against_oracle(eng, net, P, gt, dist, emb, [4, 1, 3, 0], q, conditioned=True)
    # latent gradient: against the oracle in FLOAT64.  The rate term's gradient is a ratio of differences of Gaussian CDFs
    # (network.py:145-161); where a latent sits in a tail, one ulp of erf is a %-level change of it, and with these
    # perturbed parameters the fp32 ORACLE itself is off by 4e-5 (ch = 8) to 2e-3 (chanstr 8,8,8,8) of the largest entry
    # (measured: profiles/r04_latent_grad_conditioning.md).  Allowance: 2e-4 of the largest entry, or ten times the fp32
    # oracle's own distance from float64 where that is larger
    a, de = eng.latent_step(2, update=False)
    args = (np.arange(6), 2, float(eng.counts.sum()), eng.noise_step)
    de32 = _oracle_step(P, emb, gt, dist, *args, layer_ids=_layer_ids(net))[4].double()
    de64 = _oracle_step({k: v.double() for k, v in P.items()}, emb.double(), gt.double(), dist.double(), *args,
                        layer_ids=_layer_ids(net))[4]
    scale = float(de64.abs().max())
    cond = float((de32 - de64).abs().max()) / scale
    err = float((de.cpu().double() - de64).abs().max()) / scale
    print(f"latent gradient ch={ch} {channels}: HIP vs fp64 {err:.2e}, fp32 oracle vs fp64 {cond:.2e}")
    assert err < max(2e-4, 10 * cond), (err, cond)
    # eval forward is batch-invariant bit for bit here as well (rc_enc.ply == rc_dec.ply)
    p_all = eng.eval_forward(q=2)["p2"]
    one = eng.eval_forward(lo=3, hi=4, q=2)["p2"]
    assert torch.equal(one[0], p_all[3])


def test_winograd_step_equals_the_direct_step(gpu, monkeypatch):
    """Round 4 runs the 4^3 layers of a TRAINING step in a reduced-multiplication (Winograd) form.  With the engine's
    switch off (NVF_WINO=0: conv2's forward and backward-data and conv1's backward-data through the direct fixed-order
    kernels of rounds 1-3) the same step must give the same probabilities, loss and gradients to rounding -- and the EVAL
    forward must not depend on the switch at all, bit for bit (it never uses the Winograd kernels).  (conv2's weight
    gradient inside the five-gradient launch is switched by NVF_WGRAD_WINO, read once per process by the library: its two
    forms are compared by tests/test_gpu_ops.py::test_wgrad_k4_wino and ::test_three_mfma_weight_gradients_in_one_launch.)"""
    from nvfpcc_amd import engine as E
    got = {}
    for wino in (True, False):
        monkeypatch.setattr(E, "_WINO", wino)
        net, eng, P, gt, dist, emb = make(gpu, 3, (8, 16, 8, 8), 24)
        assert (eng.layers["conv2"].wp_w is not None) == wino and (eng.layers["conv2"].wp_wf is not None) == wino
        ids = np.arange(16)
        a = eng.train_step(ids, 1, update=False)
        ev = eng.eval_forward(lo=0, hi=5, q=2)["p2"].clone()
        got[wino] = (a["p2"].clone(), eng.loss_value(), eng.flat_g.clone(), ev)
    (p_w, l_w, g_w, e_w), (p_d, l_d, g_d, e_d) = got[True], got[False]
    assert torch.equal(e_w, e_d)
    assert float((p_w - p_d).abs().max()) < 1e-5
    assert abs(l_w - l_d) < 2e-5 * abs(l_d)
    for name, (off, n) in eng.slices.items():
        ref = g_d[off:off + n]
        err = float((g_w[off:off + n] - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        assert err < 2e-5, (name, err)


def test_wide_winograd_step_equals_the_direct_step(gpu, monkeypatch):
    """The wide decoder (BASELINE configs[4]: --ch 8 --chanstr 16,32,16,16) runs conv2 / conv1 of a TRAINING step through
    conv16_wino.hip (forward, backward-data) and wgrad16_wino.hip (weight gradient).  With the engine's switch off
    (TrainEngine(winograd=False): the direct 16-row kernels of rounds 2-3, which the gradient goldens were pinned with)
    the same batch-16 step gives the same probabilities, loss and all gradients to rounding; the EVAL forward does not
    depend on the switch, bit for bit."""
    from nvfpcc_amd import engine as E
    got = {}
    for wino in (True, False):
        monkeypatch.setattr(E, "_WINO", wino)
        net, eng, P, gt, dist, emb = make(gpu, 8, (16, 32, 16, 16), 20)
        assert eng.wide and (eng.layers["conv2"].wp_w is not None) == wino and (eng.layers["conv1"].wp_wf is not None) == wino
        ids = np.arange(16)
        a = eng.train_step(ids, 1, update=False)
        ev = eng.eval_forward(lo=0, hi=5, q=2)["p2"].clone()
        got[wino] = (a["p2"].clone(), eng.loss_value(), eng.flat_g.clone(), ev)
    (p_w, l_w, g_w, e_w), (p_d, l_d, g_d, e_d) = got[True], got[False]
    assert torch.equal(e_w, e_d)
    assert float((p_w - p_d).abs().max()) < 1e-5
    assert abs(l_w - l_d) < 2e-5 * abs(l_d)
    for name, (off, n) in eng.slices.items():
        ref = g_d[off:off + n]
        err = float((g_w[off:off + n] - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        # (two Winograd layers, forward and backward, above the deepest parameters: measured 3.0e-5 at up0's kernel, 1e-5
        # or less elsewhere; the gradient goldens are held to 2e-4)
        assert err < 1e-4, (name, err)
