"""The hand-sequenced engine (nvfpcc_amd/engine.py) against the autograd operator path, which
tests/test_gpu_net.py pins to the reference goldens.  Same kernels, different sequencing (fused
masks / loss gradient / flat buffers), so agreement is to fp32 rounding of a few scalar coefficients."""
import numpy as np
import pytest
import torch

from nvfpcc_amd.seeds import synthetic_seed
from nvfpcc_amd.synth import make_blocks
from tests.golden_inputs import CONFIGS, perturb_state_, make_emb

pytestmark = pytest.mark.gpu
H = dict(lmbda=200.0, w1=10.0, w2=57.0, lr=1e-3, wemb=5.0)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda")


def make(tag, gpu, nblk=6):
    from nvfpcc_amd import network
    from nvfpcc_amd.engine import TrainEngine
    from nvfpcc_amd.model import Net
    cfg = CONFIGS[tag]
    network.reset_seed(synthetic_seed())
    network.set_noise_seed(0, 0)
    net = Net(None, "Gaussian", cfg["ch"], ",".join(str(c) for c in cfg["channels"]), verbose=False)
    sd = net.state_dict()
    perturb_state_(sd, cfg["param_seed"])
    net.load_state_dict(sd)
    net = net.to(gpu)
    gts, dists = make_blocks(nblk)
    gt = torch.from_numpy(gts).float().to(gpu)
    dist = torch.from_numpy(dists).float().to(gpu)
    emb = make_emb(nblk, cfg["ch"], cfg["emb_seed"]).to(gpu)
    eng = TrainEngine(net, gt, dist, n_points_total=917 * 936.0, emb=emb, seed=0, **H)
    return net, eng, gt, dist, emb


def module_grads(net, eng, emb, gt, dist, idx, n_pts):
    from nvfpcc_amd.loss import get_focal_dense, get_surf_focal_dense
    from nvfpcc_amd.model import MultiscaleProcessor
    eng.flat_g.zero_()
    e = emb.clone().requires_grad_(True)
    ids = torch.as_tensor(idx, device=emb.device)
    out, cls, nbits, lbits = net(e[ids].contiguous(), "train", 2, block_ids=ids)
    pyr = MultiscaleProcessor()(gt[ids].contiguous())
    loss = (get_surf_focal_dense(out, gt[ids].contiguous(), dist[ids].contiguous(), beta=1, alpha=0.9)
            + get_focal_dense(cls[0], pyr[0], alpha=0.85) + get_focal_dense(cls[1], pyr[1], alpha=0.85)
            + H["lmbda"] * (lbits.sum() / n_pts * H["w1"] + nbits.sum() / eng.n_points_total * H["w2"]))
    loss.backward()
    return loss.item(), eng.flat_g.clone(), e.grad.clone()


def close(a, b, tol=2e-5):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
    assert err < tol, err


@pytest.mark.parametrize("tag", ["S", "W"])
def test_train_step_gradients_equal_autograd_path(tag, gpu):
    net, eng, gt, dist, emb = make(tag, gpu)
    idx = [4, 1, 3]
    n_pts = float(eng.counts[idx].sum())
    loss_ref, g_ref, _ = module_grads(net, eng, emb, gt, dist, idx, n_pts)
    eng.flat_g.zero_()
    eng.train_step(idx, 2, update=False)
    assert abs(eng.loss_value() - loss_ref) < 2e-5 * abs(loss_ref)
    for name, (off, n) in eng.slices.items():
        close(eng.flat_g[off:off + n], g_ref[off:off + n])


def test_latent_step_gradient_and_adam(gpu):
    net, eng, gt, dist, emb = make("S", gpu)
    idx = list(range(eng.N_leaf))
    n_pts = float(eng.counts.sum())
    _, _, de_ref = module_grads(net, eng, emb, gt, dist, idx, n_pts)
    before = eng.flat_p.clone()
    a, de = eng.latent_step(2, update=False)
    close(de, de_ref)
    # Adam on the latents = torch.optim.Adam on the same gradient
    e_ref = eng.emb.clone().requires_grad_(True)
    opt = torch.optim.Adam([e_ref], lr=H["lr"] * H["wemb"])
    for _ in range(3):
        a, de = eng.latent_step(2, update=True)
        e_ref.grad = de.clone()
        opt.step()
    assert torch.allclose(eng.emb, e_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(eng.flat_p, before), "the latent phase must not touch the decoder"


def test_decoder_update_matches_torch_adam(gpu):
    net, eng, gt, dist, emb = make("S", gpu)
    p_ref = eng.flat_p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=H["lr"])
    for i in range(3):
        eng.train_step([i, i + 1], 2, update=False)
        p_ref.grad = eng.flat_g.clone()
        # re-run the same step with the update enabled: identical gradient, then fused Adam
        eng.train_step([i, i + 1], 2, update=True)
        opt.step()
        assert torch.allclose(eng.flat_p, p_ref.detach(), rtol=2e-5, atol=1e-6)
        # keep the reference weights in lock-step (rounding differences must not compound into the test)
        with torch.no_grad():
            p_ref.copy_(eng.flat_p)


def test_two_rank_sharding_sums_to_the_single_rank_gradient(gpu):
    """Emulates W = 2 on one GPU: per-rank gradients with the global n_pts and the weight-rate term
    scaled by 1/W, summed, equal the gradient of the whole mini-batch."""
    net, eng, gt, dist, emb = make("S", gpu)
    whole = [5, 0, 2, 3]
    n_pts = float(eng.counts[whole].sum())
    eng.train_step(whole, 2, update=False, n_pts=n_pts)
    g_all = eng.flat_g.clone()
    eng.rate_grad_scale = 0.5
    parts = []
    for r in range(2):
        eng.train_step(whole[r::2], 2, update=False, n_pts=n_pts)
        parts.append(eng.flat_g.clone())
    eng.rate_grad_scale = 1.0
    close(parts[0] + parts[1], g_all, tol=1e-5)


def test_graph_replay_equals_the_host_launched_step(gpu):
    """bench.py times GraphedTrainStep (the step body replayed from one HIP graph, block ids / noise step / rate
    coefficient read from device memory): its gradients, loss terms and Adam-updated parameters equal those of the
    host-launched train_step bit for bit, step after step (q = 1: weight and latent noise on)."""
    from nvfpcc_amd.engine import GraphedTrainStep
    net, eng, gt, dist, emb = make("S", gpu, nblk=24)
    B = 16
    rng = np.random.default_rng(5)
    batches = [rng.permutation(24)[:B] for _ in range(3)]
    state = lambda: (eng.flat_p.clone(), eng.flat_m.clone(), eng.flat_v.clone(), eng.noise_step, eng.opt_step)
    s0 = state()
    ref = []
    for ids in batches:
        eng.train_step(ids, 1)
        ref.append((eng.flat_g.clone(), eng.flat_p.clone(), eng.loss_value()))
    eng.flat_p.copy_(s0[0]); eng.flat_m.copy_(s0[1]); eng.flat_v.copy_(s0[2])
    eng.noise_step, eng.opt_step = s0[3], s0[4]
    graphed = GraphedTrainStep(eng, B, 1)                # capture runs the body: restore the state it touched
    eng.flat_p.copy_(s0[0]); eng.flat_m.copy_(s0[1]); eng.flat_v.copy_(s0[2])
    eng.noise_step, eng.opt_step = s0[3], s0[4]
    for ids, (g_ref, p_ref, loss_ref) in zip(batches, ref):
        graphed(ids)
        torch.cuda.synchronize()
        assert torch.equal(eng.flat_g, g_ref)
        assert torch.equal(eng.flat_p, p_ref)
        assert eng.loss_value() == loss_ref
    # the host may run ahead of the GPU: six more steps without a sync in between, through a two-slot staging ring
    more = [rng.permutation(24)[:B] for _ in range(6)]
    s1 = state()
    for ids in more:
        eng.train_step(ids, 1)
    p_end = eng.flat_p.clone()
    eng.flat_p.copy_(s1[0]); eng.flat_m.copy_(s1[1]); eng.flat_v.copy_(s1[2])
    eng.noise_step, eng.opt_step = s1[3], s1[4]
    graphed2 = GraphedTrainStep(eng, B, 1, ring=2)
    eng.flat_p.copy_(s1[0]); eng.flat_m.copy_(s1[1]); eng.flat_v.copy_(s1[2])
    eng.noise_step, eng.opt_step = s1[3], s1[4]
    torch.cuda.synchronize()
    for ids in more:
        graphed2(ids)
    torch.cuda.synchronize()
    assert torch.equal(eng.flat_p, p_end)


def test_full_size_step_is_the_sum_of_its_mini_batches(gpu):
    """Size-independent property at the large-batch configuration (B = 256: conv2's backward-data takes the VALU tile
    kernel above batch 64, the loss its multi-launch form above batch 32): every loss term is a SUM over blocks, so the
    gradient of the 256-block step equals the sum of the gradients of its sixteen 16-block mini-batches -- the
    configuration the golden vectors pin -- with the weight-rate term counted once (rate_grad_scale = 1/16)."""
    net, eng, gt, dist, emb = make("S", gpu, nblk=256)
    whole = list(range(256))
    n_pts = float(eng.counts[whole].sum())
    eng.train_step(whole, 2, update=False, n_pts=n_pts)
    g_all, loss_all = eng.flat_g.clone(), eng.loss_value()
    assert torch.isfinite(g_all).all()
    eng.rate_grad_scale = 1.0 / 16
    acc = torch.zeros_like(g_all, dtype=torch.float64)
    for r in range(16):
        eng.train_step(whole[16 * r:16 * (r + 1)], 2, update=False, n_pts=n_pts)
        acc += eng.flat_g.double()
    eng.rate_grad_scale = 1.0
    for name, (off, n) in eng.slices.items():
        close(g_all[off:off + n], acc[off:off + n], tol=2e-4)
    assert np.isfinite(loss_all)


def test_weight_noise_and_latent_noise_are_reproducible(gpu):
    net, eng, gt, dist, emb = make("S", gpu)
    eng.train_step([0, 1, 2], 1, update=False)
    g1, s1 = eng.flat_g.clone(), eng.noise_step
    eng.noise_step = s1 - 1
    eng.train_step([0, 1, 2], 1, update=False)
    assert torch.equal(eng.flat_g, g1)
    eng.train_step([0, 1, 2], 1, update=False)      # next step: different noise
    assert not torch.equal(eng.flat_g, g1)


def test_fused_stem_equals_the_per_layer_kernels(gpu):
    """Forward bit-identical (same accumulation order); gradients to rounding."""
    net, eng, gt, dist, emb = make("S", gpu)
    assert eng.fused_stem
    idx = [0, 2, 5, 1]
    a = eng.train_step(idx, 2, update=False)
    g_fused, y1, h0 = eng.flat_g.clone(), a["y1"].clone(), a["h0"].clone()
    eng.fused_stem = False
    a = eng.train_step(idx, 2, update=False)
    assert torch.equal(a["y1"], y1) and torch.equal(a["h0"], h0)
    for name, (off, n) in eng.slices.items():
        close(g_fused[off:off + n], eng.flat_g[off:off + n], tol=1e-5)
    eng.fused_stem = True
    _, de = eng.latent_step(2, update=False)
    eng.fused_stem = False
    _, de2 = eng.latent_step(2, update=False)
    close(de, de2, tol=1e-5)


def test_latent_generator_and_stem_in_one_launch(gpu):
    """nvf_stem_latent_fwd against nvf_latent_fwd + nvf_stem_fwd: every saved activation bit for bit, train (noise in
    the rate) and eval mode; the stem's workgroups recompute their block's rounded latents with the same arithmetic."""
    net, eng, gt, dist, emb = make("S", gpu)
    assert eng.fused_stem and eng.fused_latent_stem
    ids = torch.tensor([0, 2, 5, 1, 3], device=gpu)
    e = (emb[ids] * 2.5).contiguous()                    # spread the latents over several integers
    for mode, q in (("train", 1), ("eval", 2)):
        eng.prepare_weights(q)
        eng.fused_latent_stem = True
        a = eng.forward(e, mode, ids)
        eng.fused_latent_stem = False
        b = eng.forward(e, mode, ids)
        torch.cuda.synchronize()
        for k in ("h", "lat", "x0", "lbits", "a0", "h0", "y1", "p0", "p2"):
            assert torch.equal(a[k], b[k]), (mode, k)
        assert a["x0"].abs().max().item() >= 1.0
    eng.fused_latent_stem = True


@pytest.mark.parametrize("tag", ["S", "W"])
def test_one_launch_step_head_equals_the_three_launches(tag, gpu):
    """nvf_step_head (effective weights + MFMA packings + mini-batch gather in one launch; the packings recompute
    their effective weights from the raw kernels) against nvf_gather_rows_multi + nvf_prepare_weights +
    nvf_pack_mfma_all: every prepared layout, packed fragment array and gathered row, bit for bit, for every
    quantisation mode."""
    net, eng, gt, dist, emb = make(tag, gpu)
    idx = torch.tensor([4, 1, 5, 1], device=gpu)
    bufs = lambda: [t for L in eng.layers.values() for t in (L.w_fwd, L.w_bwd, L.b_eff, L.wp_f, L.wp_b, L.wp_t, L.wp_s)
                    if t is not None]
    for q in (0, 1, 2):
        eng.noise_step = 7 + q
        for t in bufs():
            t.fill_(float("nan"))
        rows_ref = eng._batch(idx)
        eng.prepare_weights(q)
        ref = [t.clone() for t in bufs()]
        for t in bufs():
            t.fill_(float("nan"))
        rows = eng.batch_and_prepare(idx, q)
        torch.cuda.synchronize()
        assert len(ref) > 0 and all(torch.equal(a, b) for a, b in zip(bufs(), ref)), (tag, q)
        assert all(torch.equal(a, b) for a, b in zip(rows, rows_ref))
        assert not any(torch.isnan(t).any().item() for t in bufs())
