#!/usr/bin/env python3
"""One training run done twice -- by the HIP engine and by the CPU oracle (the reference's aten ops, pinned to the
reference's own loop by tests/golden/trajectory.npz) -- on the same tiny synthetic cloud with the same seeds and the
SAME noise (the oracle is fed the engine's counter-RNG draws, tests/philox_np.py), then quantised and evaluated the
same way: rate (latent bits from the Gaussian model + Huffman-coded 4-bit weights) and the reference's one-sided
PSNR1 at the same threshold.  SURVEY.md section 8 row f4: "an oracle-CPU-trained point beside the HIP one".
Rounding differences are amplified by ~60 x 4 Adam steps, so the comparison is statistical, not bit-wise.

    python tools/rd_cpu_point.py --blocks 12 --epochs 301 --seeds 5 --out profiles/r04_rd_cpu_vs_hip.md      (on the GPU box)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRUNK = ["up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls"]


def evaluate(P, emb, gt, dist, n_points, thh):
    """Quantise the trunk kernels to 1/16 (manipulate_weights.py), then NVFPCC.py's TEST numbers with the oracle."""
    from oracle import nvf_oracle as O
    from nvfpcc_amd import weight_codec
    Q = {k: v.detach().clone() for k, v in P.items()}
    for n in TRUNK:
        Q[f"reconstructor.{n}.kernel"] = torch.round(Q[f"reconstructor.{n}.kernel"] * 16) / 16
    with torch.no_grad():
        out, cls, nbits, lbits = O.net_forward(Q, emb.detach(), "eval", 2)
        sse, den = O.sse1(out, dist, thh)
        tpr, tnr = O.acc_dense(out, gt, 0.5)
    eles = np.concatenate([(Q[f"reconstructor.{n}.kernel"].numpy() * 16).reshape(-1) for n in TRUNK])
    pdf, bins = weight_codec.get_pdf(eles)
    cb, _ = weight_codec.get_huffman_codebook(pdf, bins)
    net_bits = sum(len(cb[int(v)]) for v in np.round(eles).astype(int))
    mse1 = float(sse) / max(float(den), 1e-30)
    return {"bpp_latent": float(lbits) / n_points, "bpp_weights": net_bits / n_points,
            "PSNR1": 20 * np.log10(1023 / np.sqrt(mse1 / 3)) if mse1 > 0 else float("inf"),
            "Pacc": float(tpr), "Nacc": float(tnr), "points": int((out > thh).sum())}


def run_one(a, seed):
    """HIP engine and CPU oracle on the same cloud, the same epoch orders and the same noise draws (all derived from
    `seed`).  Returns (hip metrics, cpu metrics, t_hip, t_cpu, max parameter difference, max latent difference)."""
    from nvfpcc_amd import network
    from nvfpcc_amd.engine import TrainEngine, EpochDriver
    from nvfpcc_amd.model import Net
    from nvfpcc_amd.seeds import synthetic_seed
    from nvfpcc_amd.synth import make_blocks
    from oracle import nvf_oracle as O
    from tests import philox_np
    ch, channels = (8, (16, 32, 16, 16)) if a.wide else (3, (8, 16, 8, 8))      # BASELINE configs[4] / [1]
    H = dict(lmbda=a.lmbda, w1=10.0, w2=57.0, lr=1e-3, wemb=8.0 if a.wide else 5.0)
    gts, dists = make_blocks(a.blocks)
    gt, dist = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    n_points = float(gts.sum())
    orders = [np.random.default_rng(1000 * seed + e).permutation(a.blocks) for e in range(a.epochs)]

    # ---- HIP engine
    dev = torch.device("cuda", 0)
    network.reset_seed(synthetic_seed())
    net = Net(None, "Gaussian", ch, ",".join(map(str, channels)), verbose=False).to(dev)
    eng = TrainEngine(net, gt.to(dev), dist.to(dev), n_points_total=n_points, seed=seed, **H)
    drv = EpochDriver(eng, a.batch, use_graph=True)
    t0 = time.time()
    for e in range(a.epochs):
        q = 1 if e < a.phase_change else 2
        drv.run(orders[e], q)
        eng.latent_step(q)
        eng.read_epoch_stats()
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    P_hip = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    hip = evaluate(P_hip, eng.emb.cpu(), gt, dist, n_points, a.thh)

    # ---- CPU oracle, same noise
    shapes = {t[0].split(".")[-1]: t[2] for t in O.layer_table(ch, channels)}

    def noise(step, ids, q):
        u_lat = torch.from_numpy(philox_np.latent_noise(seed, step, list(ids), ch))
        u_w = {n: torch.from_numpy(philox_np.weight_noise(seed, step, i + 1, shapes[n])) for i, n in enumerate(TRUNK)} \
            if q == 1 else {}
        return u_lat, u_w
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    tr = O.OracleTrainer(ch, channels, synthetic_seed(), n_leaf=a.blocks, n_points=n_points, noise_fn=noise, **H)
    t0 = time.time()
    for e in range(a.epochs):
        q = 1 if e < a.phase_change else 2
        tr.set_epoch(e)
        for s in range(0, a.blocks, a.batch):
            ids = orders[e][s:s + a.batch]
            tr.train_step(torch.from_numpy(ids), gt[ids], dist[ids], q)
        tr.latent_step(gt, dist, q)
    t_cpu = time.time() - t0
    cpu = evaluate({k: v.detach() for k, v in tr.P.items()}, tr.emb, gt, dist, n_points, a.thh)
    dp = max((P_hip[k] - tr.P[k].detach()).abs().max().item() for k in tr.keys)
    de = (eng.emb.cpu() - tr.emb.detach()).abs().max().item()
    return hip, cpu, t_hip, t_cpu, dp, de, n_points


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=61)
    ap.add_argument("--phase_change", type=int, default=20)
    ap.add_argument("--lmbda", type=float, default=200.0)
    ap.add_argument("--thh", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--seeds", type=int, default=1, help="number of (noise, epoch-order) seeds: seed, seed + 1, ...")
    ap.add_argument("--out", default="")
    ap.add_argument("--wide", action="store_true", help="the wide decoder: --ch 8 --chanstr 16,32,16,16 --wemb 8")
    a = ap.parse_args()
    runs = []
    for k in range(a.seeds):
        r = run_one(a, a.seed + k)
        runs.append(r)
        print(f"seed {a.seed + k}: HIP PSNR1 {r[0]['PSNR1']:.2f} bpp {r[0]['bpp_latent'] + r[0]['bpp_weights']:.4f} | "
              f"CPU PSNR1 {r[1]['PSNR1']:.2f} bpp {r[1]['bpp_latent'] + r[1]['bpp_weights']:.4f} | "
              f"{r[2]:.1f} s / {r[3]:.1f} s", flush=True)
    n_points = runs[0][6]
    head = (f"HIP engine vs CPU oracle, the same training runs: {a.blocks} synthetic blocks ({int(n_points)} points), batch "
            f"{a.batch}, {a.epochs} epochs (phase change {a.phase_change}), lambda {a.lmbda:g}, w1 10, w2 57, lr 1e-3, "
            f"{'ch 8, chanstr 16,32,16,16, wemb 8' if a.wide else 'ch 3, chanstr 8,16,8,8, wemb 5'}; "
            f"{a.seeds} seed(s) from {a.seed} (a seed fixes the noise draws AND the epoch orders; the oracle is fed the engine's "
            f"counter-RNG draws); kernels rounded to 1/16, evaluation by the oracle's eval forward at thh {a.thh}.  Divergence "
            f"of the two trajectories (rounding amplified by {a.epochs} x {(a.blocks + a.batch - 1) // a.batch + 1} Adam steps): max "
            f"|parameter difference| {max(r[4] for r in runs):.2e}, latent table {max(r[5] for r in runs):.2e}.\n\n")
    rows = ["| seed | trained by | seconds | bpp latents | bpp weights (Huffman) | bpp | PSNR1 (dB) | Pacc | Nacc | points > thh |",
            "|---|---|---|---|---|---|---|---|---|---|"]
    for k, (hip, cpu, t_hip, t_cpu, dp, de, _) in enumerate(runs):
        for name, t, r in (("HIP engine (1 x MI355X)", t_hip, hip), ("CPU oracle (aten, 16 threads)", t_cpu, cpu)):
            rows.append(f"| {a.seed + k} | {name} | {t:.1f} | {r['bpp_latent']:.4f} | {r['bpp_weights']:.4f} | "
                        f"{r['bpp_latent'] + r['bpp_weights']:.4f} | {r['PSNR1']:.2f} | {r['Pacc']:.4f} | {r['Nacc']:.4f} | {r['points']} |")
    text = head + "\n".join(rows) + "\n"
    if a.seeds > 1:
        def stat(idx, key):
            v = np.array([(r[idx]['bpp_latent'] + r[idx]['bpp_weights']) if key == 'bpp' else r[idx][key] for r in runs], np.float64)
            return v.mean(), v.std(ddof=1)
        srows = ["| metric | HIP mean +- sd | CPU oracle mean +- sd | HIP mean - CPU mean | within one oracle sd |", "|---|---|---|---|---|"]
        ok_all = True
        for key in ("PSNR1", "bpp", "Pacc", "Nacc", "points"):
            (mh, sh), (mc, sc) = stat(0, key), stat(1, key)
            ok = abs(mh - mc) <= sc
            if key in ("PSNR1", "bpp"):
                ok_all &= ok
            srows.append(f"| {key} | {mh:.4f} +- {sh:.4f} | {mc:.4f} +- {sc:.4f} | {mh - mc:+.4f} | {'yes' if ok else 'NO'} |")
        text += (f"\nOver the {a.seeds} seeds (sample sd, n - 1).  Pass criterion (VERDICT r3 item 5): the HIP mean of PSNR1 and of "
                 f"bpp within one oracle sd -- **{'met' if ok_all else 'NOT met'}**.\n\n" + "\n".join(srows) + "\n")
    print(text)
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
