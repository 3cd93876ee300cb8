#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference in this container.

Runs only where /root/reference exists (the build container).  It imports the
reference's live operator modules (utils/network.py, gdn_3d.py, utils/loss.py)
with empty stand-ins for the optional packages the live subset never touches
(open3d, MinkowskiEngine, IPython, bitstream, torchvision) and a synthetic
SEED3.npy, evaluates them on seeded inputs, and stores inputs' seeds plus the
expected outputs.  Nothing from the reference is copied: the fixtures are data.

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from nvfpcc_amd.seeds import synthetic_seed          # noqa: E402
from nvfpcc_amd.synth import make_blocks             # noqa: E402
from tests import philox_np                           # noqa: E402
from tests.golden_inputs import (TRAJ, traj_order,    # noqa: E402
                                 CONFIGS, HYPER, perturb_state_, make_emb, noise_stream,   # noqa: E402
                                 sample_index, loss_case_inputs, gdn_case_inputs)


def import_reference():
    for name in ["open3d", "MinkowskiEngine", "IPython", "bitstream", "torchvision", "torchvision.utils"]:
        sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.datasets = tv.transforms = None
    tv.utils = sys.modules["torchvision.utils"]
    tv.utils.save_image = None
    work = tempfile.mkdtemp(prefix="nvf_ref_")
    np.save(os.path.join(work, "SEED3.npy"), synthetic_seed())
    np.save(os.path.join(work, "SEED4_Gaussian.npy"), np.zeros(4))
    cwd = os.getcwd()
    os.chdir(work)
    sys.path.insert(0, REF)
    try:
        import utils.network as net_mod
        import utils.loss as loss_mod
        import gdn_3d as gdn_mod
    finally:
        os.chdir(cwd)
    return net_mod, loss_mod, gdn_mod


class RefNet(torch.nn.Module):
    """latent_gen -> entropy_coder -> reconstructor, as NVFPCC.py:32-45 wires them."""

    def __init__(self, net_mod, ch, channels):
        super().__init__()
        net_mod.seed_ptr = 0
        self.latent_gen = net_mod.SingleLayerLatentGen(in_channels=ch, out_channels=ch)
        self.entropy_coder = net_mod.QuantGaussianLikelihood(in_channels=ch)
        self.reconstructor = net_mod.CompDecoder(None, "Gaussian", useIGDN=True, in_channels=ch,
                                                 channels=channels)
        self.seed_used = net_mod.seed_ptr

    def forward(self, emb, mode, q):
        lat = self.latent_gen(emb)
        rounded, bits = self.entropy_coder(lat, mode)
        out, cls, nbits = self.reconstructor(rounded, q)
        return out, cls, nbits, bits, lat, rounded


def summary(t, n=64):
    a = t.detach().double().reshape(-1)
    idx = sample_index(a.numel(), n)
    return np.concatenate([[a.mean().item(), a.abs().sum().item()], a[idx].numpy()])


def ref_loss(loss_mod, net, emb, gt, dist, mode, q):
    out, cls, nbits, lbits, lat, rounded = net(emb, mode, q)
    x1 = torch.nn.functional.max_pool3d(gt, 2, 2)
    x2 = torch.nn.functional.max_pool3d(x1, 2, 2)
    b_latent = lbits.sum() / gt.sum()
    b_net = nbits.sum() / HYPER["n_points"]
    loss = (loss_mod.get_surf_focal_dense(out, gt, dist, beta=1, alpha=0.9)
            + loss_mod.get_focal_dense(cls[0], x2, alpha=0.85)
            + loss_mod.get_focal_dense(cls[1], x1, alpha=0.85)
            + HYPER["lmbda"] * (b_latent * HYPER["w1"] + b_net * HYPER["w2"]))
    return loss, out, cls, nbits, lbits, lat, rounded


def gen_net_cases(net_mod, loss_mod, tag):
    cfg = CONFIGS[tag]
    ch, channels, B = cfg["ch"], cfg["channels"], cfg["batch"]
    net = RefNet(net_mod, ch, channels)
    fresh = {k: v.clone() for k, v in net.state_dict().items()}
    g = {"seed_used": np.int64(net.seed_used)}
    # G9: frozen seed-derived buffers of the untouched network
    for k, v in fresh.items():
        if k.endswith("_init"):
            g["init/" + k] = summary(v, 32)
    sd = net.state_dict()
    perturb_state_(sd, cfg["param_seed"])
    net.load_state_dict(sd)
    emb = make_emb(B, ch, cfg["emb_seed"]).requires_grad_(True)
    gts, dists = make_blocks(B)
    gt = torch.from_numpy(gts).float()
    dist = torch.from_numpy(dists).float()
    g["gt_sha"] = np.frombuffer(hashlib.sha256(gts.tobytes()).digest(), np.uint8)
    g["dist_sum"] = np.float64(dists.sum())

    # hooks for per-layer summaries
    acts = {}
    rec = net.reconstructor
    hooks = [getattr(rec, n).register_forward_hook(lambda m, i, o, n=n: acts.__setitem__(n, o.detach()))
             for n in ["up0", "activation", "conv0", "conv0_cls", "up1", "conv1", "conv1_cls",
                       "up2", "conv2", "conv2_cls"]]

    # G1: deterministic forward, eval mode, q=2 and q=0
    for q in (2, 0):
        with torch.no_grad():
            out, cls, nbits, lbits, lat, rounded = net(emb, "eval", q)
        p = f"fwd_q{q}/"
        if q == 2:
            g[p + "out"] = out.numpy()
            g[p + "cls0"] = cls[0].numpy()
            g[p + "cls1"] = cls[1].numpy()
            g[p + "latent"] = lat.numpy()
            g[p + "latent_rounded"] = rounded.numpy()
        else:
            g[p + "out"] = summary(out)
        g[p + "net_bits"] = nbits.numpy()
        g[p + "latent_bits"] = lbits.numpy()
        for n, t in acts.items():
            g[p + "act/" + n] = summary(t)
    for h in hooks:
        h.remove()

    # G2: gradients of the full objective (eval mode, q=2; deterministic)
    net.zero_grad()
    loss, *_ = ref_loss(loss_mod, net, emb, gt, dist, "eval", 2)
    loss.backward()
    g["grad_q2/loss"] = np.float64(loss.item())
    g["grad_q2/emb"] = emb.grad.numpy().copy()
    for k, p_ in net.named_parameters():
        # full gradients for the narrow config; 256-sample summaries for the wide one (size)
        g["grad_q2/" + k] = p_.grad.numpy().copy() if tag == "S" else summary(p_.grad, 256)

    # G3: train mode, q=1, with torch.rand_like replaced by a seeded stream
    # (call order: entropy_coder, then up0, conv0, up1, conv1, up2, conv2, conv2_cls)
    stream = noise_stream(cfg["noise_seed"])
    real_rand_like = torch.rand_like
    torch.rand_like = lambda t, *a, **k: next(stream)(t.shape)
    try:
        emb.grad = None
        net.zero_grad()
        opt = torch.optim.Adam(net.parameters(), lr=HYPER["lr"])
        opt_emb = torch.optim.Adam([emb], lr=HYPER["lr"] * HYPER["wemb"])
        loss, out, cls, nbits, lbits, lat, rounded = ref_loss(loss_mod, net, emb, gt, dist, "train", 1)
        loss.backward()
    finally:
        torch.rand_like = real_rand_like
    g["train_q1/loss"] = np.float64(loss.item())
    g["train_q1/out"] = summary(out)
    g["train_q1/latent_bits"] = lbits.detach().numpy()
    g["train_q1/net_bits"] = nbits.detach().numpy()
    g["train_q1/grad_emb"] = emb.grad.numpy().copy()
    for k, p_ in net.named_parameters():
        g["train_q1/grad/" + k] = summary(p_.grad, 48)
    opt.step()
    opt_emb.step()
    g["train_q1/emb_after"] = emb.detach().numpy().copy()
    for k, p_ in net.named_parameters():
        g["train_q1/after/" + k] = summary(p_, 48)
    np.savez_compressed(os.path.join(OUT, f"net_{tag}.npz"), **g)
    print(tag, "seed_used", net.seed_used, "loss", g["grad_q2/loss"], g["train_q1/loss"])


def gen_trajectory(net_mod, loss_mod):
    """The reference's train() loop itself (NVFPCC.py:105-254), statement for statement where it touches the
    optimisers, on 14 synthetic blocks: fresh Net, emb = ones, Adam + both MultiStepLR objects wired as :116-126,
    zero_grad / backward / step order of :150-222 and :226-250, the latent phase on the post-mini-batch decoder, the
    per-epoch log sums of :256-281.  torch.rand_like is replaced by the counter RNG of the HIP engine (tests/philox_np),
    keyed exactly as the engine keys it (one noise step per train-mode forward), so that the engine -- which cannot be
    fed foreign noise -- can be held to this trajectory directly."""
    cfg = CONFIGS[TRAJ["tag"]]
    ch, channels = cfg["ch"], cfg["channels"]
    n, B = TRAJ["n_blocks"], TRAJ["batch"]
    net = RefNet(net_mod, ch, channels)
    gts, dists = make_blocks(n)
    gt_all, dist_all = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    N = float(gts.sum())                                   # train_data.N (dataloader.py:160)
    lmbda, w1, w2, lr, wemb = (HYPER[k] for k in ("lmbda", "w1", "w2", "lr", "wemb"))
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [300, 400, 450], 0.1)
    emb = torch.ones((n, ch, 2, 2, 2), dtype=torch.float32).requires_grad_(True)
    opt_emb = torch.optim.Adam([emb], lr=lr * wemb)
    sch_emb = torch.optim.lr_scheduler.MultiStepLR(opt, [300, 400, 450], 0.1)
    pool = torch.nn.MaxPool3d(2, 2)
    ms = lambda x: [pool(pool(x)), pool(x), x]
    noise = {"step": 0, "ids": None, "call": 0}

    def rand_like(t, *a, **k):
        c = noise["call"]
        noise["call"] += 1
        if c == 0:       # QuantGaussianLikelihood.forward (network.py:4516)
            return torch.from_numpy(philox_np.latent_noise(TRAJ["noise_seed"], noise["step"], noise["ids"], ch))
        return torch.from_numpy(philox_np.weight_noise(TRAJ["noise_seed"], noise["step"], c, tuple(t.shape)))

    def forward(ids, q):
        noise["step"] += 1
        noise["ids"], noise["call"] = list(ids), 0
        out = net(emb[torch.as_tensor(ids)], "train", q)
        assert noise["call"] == (8 if q == 1 else 1), noise["call"]
        return out

    g = {"n_points": np.float64(N)}
    real_rand_like = torch.rand_like
    torch.rand_like = rand_like
    try:
        for epoch in range(TRAJ["epochs"]):
            q = 1 if epoch < TRAJ["phase_change"] else 2
            order = traj_order(epoch)
            rows, sse_l, den_l, losses = [], [], [], []
            for s in range(0, n, B):
                ids = order[s:s + B]
                opt.zero_grad()
                x, dist = gt_all[ids], dist_all[ids]
                n_pts = x.sum()
                gl = ms(x)
                out, cls, nbits, lbits, _, _ = forward(ids, q)
                b_latent = lbits.sum() / n_pts
                b_net = nbits.sum() / N
                f = [loss_mod.get_focal_dense(cls[0], gl[0], alpha=0.85), loss_mod.get_focal_dense(cls[1], gl[1], alpha=0.85)]
                acc = [loss_mod.get_acc_dense(cls[0], gl[0], thh=0.5), loss_mod.get_acc_dense(cls[1], gl[1], thh=0.5)]
                bce = loss_mod.get_surf_focal_dense(out, x, dist, beta=1, alpha=0.9)
                sse, den = loss_mod.get_sse1(out, x, dist, 0.6)
                loss = bce + f[0] + f[1] + lmbda * (b_latent * w1 + b_net * w2)
                loss.backward()
                pacc, nacc = loss_mod.get_acc_dense(out, x)
                rows.append([loss.item(), pacc.item(), nacc.item(), f[0].item(), f[1].item(), acc[0][0].item(),
                             acc[0][1].item(), acc[1][0].item(), acc[1][1].item(), (b_latent + b_net).item(),
                             b_latent.item(), b_net.item()])
                sse_l.append(sse.item())
                den_l.append(den.item())
                opt.step()
            # latent update (NVFPCC.py:225-251)
            opt_emb.zero_grad()
            gl = ms(gt_all)
            out, cls, nbits, lbits, _, _ = forward(np.arange(n), q)
            b_latent = lbits.sum() / gt_all.sum()
            b_net = nbits.sum() / N
            loss = (loss_mod.get_surf_focal_dense(out, gt_all, dist_all, beta=1, alpha=0.9)
                    + loss_mod.get_focal_dense(cls[0], gl[0], alpha=0.85) + loss_mod.get_focal_dense(cls[1], gl[1], alpha=0.85)
                    + lmbda * (b_latent * w1 + b_net * w2))
            loss.backward()
            opt_emb.step()
            sch_emb.step()
            sch.step()
            r = np.array(rows)
            mse1 = np.sum(sse_l) / np.sum(den_l)
            p = f"epoch{epoch}/"
            g[p + "order"] = order
            g[p + "step_loss"] = r[:, 0]
            g[p + "latent_loss"] = np.float64(loss.item())
            # the numbers of the TRAIN log line in its order (NVFPCC.py:261-281) after "seconds]": Loss, PosiPenal,
            # PosiGain, Pacc, Nacc, S1 Loss, S2 Loss, S1Pacc, S1Nacc, S2Pacc, S2Nacc, bpp, b_latent, b_net, MSE1, PSNR1
            cnt = len(rows)
            g[p + "log"] = np.concatenate([[r[:, 0].sum() / cnt, 0.0, 0.0], r[:, 1:].sum(0) / cnt,
                                           [mse1, 20 * np.log10(1023 / np.sqrt(mse1 / 3))]])
            g[p + "emb"] = emb.detach().numpy().copy()
            for k, p_ in net.named_parameters():
                g[p + "param/" + k] = p_.detach().numpy().copy() if p_.numel() <= 1024 else summary(p_, 256)
            print("trajectory epoch", epoch, "q", q, "losses", r[:, 0], "latent", loss.item())
    finally:
        torch.rand_like = real_rand_like
    g["lr_after"] = np.float64(opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **g)


def gen_loss_cases(loss_mod):
    g = {}
    for name, (p, gt, dist) in loss_case_inputs().items():
        p = p.clone().requires_grad_(True)
        f = loss_mod.get_focal_dense(p, gt, alpha=0.85)
        f.backward()
        g[name + "/focal"] = np.float64(f.item())
        g[name + "/focal_grad"] = p.grad.numpy().copy()
        p.grad = None
        s = loss_mod.get_surf_focal_dense(p, gt, dist, beta=1, alpha=0.9)
        s.backward()
        g[name + "/surf"] = np.float64(s.item())
        g[name + "/surf_grad"] = p.grad.numpy().copy()
        tpr, tnr = loss_mod.get_acc_dense(p.detach(), gt, thh=0.5)
        g[name + "/acc"] = np.array([tpr.item(), tnr.item()])
        sse, den = loss_mod.get_sse1(p.detach(), gt, dist, 0.6)
        g[name + "/sse1"] = np.array([sse.item(), den.item()])
        se = loss_mod.get_se(p.detach(), dist, 0.6)
        g[name + "/se"] = summary(se)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **g)


def gen_gdn_cases(gdn_mod):
    g = {}
    for name, (x, beta, gamma, gy) in gdn_case_inputs().items():
        for inv, cls in ((False, gdn_mod.GDN3d), (True, gdn_mod.IGDN3d)):
            m = cls(x.shape[1])
            with torch.no_grad():
                m.beta.copy_(beta)
                m.gamma.copy_(gamma)
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            y.backward(gy)
            p = f"{name}/{'igdn' if inv else 'gdn'}/"
            g[p + "y"] = y.detach().numpy()
            g[p + "dx"] = xi.grad.numpy()
            g[p + "dbeta"] = m.beta.grad.numpy()
            g[p + "dgamma"] = m.gamma.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "gdn.npz"), **g)


def gen_schedule():
    """Quirk Q1: both MultiStepLR objects drive the decoder optimiser (NVFPCC.py:116-126,253-254)."""
    w = torch.nn.Parameter(torch.zeros(1))
    e = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [300, 400, 450], 0.1)
    opt_emb = torch.optim.Adam([e], lr=1e-3 * 5)
    sch_emb = torch.optim.lr_scheduler.MultiStepLR(opt, [300, 400, 450], 0.1)
    rows = []
    for epoch in range(501):
        if epoch in (0, 299, 300, 400, 450, 500):
            rows.append([epoch, opt.param_groups[0]["lr"], opt_emb.param_groups[0]["lr"]])
        opt.step()
        opt_emb.step()
        sch_emb.step()
        sch.step()
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), table=np.array(rows, np.float64))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    net_mod, loss_mod, gdn_mod = import_reference()
    if "--trajectory" in sys.argv:
        return gen_trajectory(net_mod, loss_mod)
    gen_trajectory(net_mod, loss_mod)
    for tag in CONFIGS:
        gen_net_cases(net_mod, loss_mod, tag)
    gen_loss_cases(loss_mod)
    gen_gdn_cases(gdn_mod)
    gen_schedule()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
