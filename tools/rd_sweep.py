#!/usr/bin/env python3
"""Rate-distortion sweep of the whole codec on a synthetic 10-bit surface (README
steps 1-3 of the reference end to end): pre-process -> train -> 4-bit weight quantisation -> encode -> decode,
for several lambda.  Reports bpp from the REAL stream lengths (NVFPCC.py:542-547), the reference's one-sided
PSNR1 (NVFPCC.py:259-260) and a symmetric D1 PSNR (point-to-point, peak 1023) computed with a KD-tree.

    python tools/rd_sweep.py --lambdas 50,200,800 --epochs 301 --out profiles/r01_rd_sweep.md
    python tools/rd_sweep.py --lambdas 200 --qps 16,8,32 --thhs 0.5,0.6,0.65     # + the README's --qp / --thh knobs
    python tools/rd_sweep.py --lambdas 25,50,100,200,400 --auto-thh              # one operating point per lambda (the curve)

One training run per lambda; per (lambda, qp) one weight quantisation (manipulate_weights.py, README step 3a) and per
(lambda, qp, thh) one encode + decode (README steps 3b-3c).
"""
import argparse
import os
import pickle
import subprocess
import sys
import tempfile
import time

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_cloud(seed, radius, n_dir):
    """Bumpy ellipsoid shell, 10-bit coordinates."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_dir, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    bump = 1.0 + 0.08 * np.sin(5 * d[:, 0]) * np.cos(4 * d[:, 1]) + 0.05 * np.sin(9 * d[:, 2])
    p = np.array([512.0, 512.0, 512.0]) + d * bump[:, None] * np.array([radius, 0.85 * radius, 1.2 * radius])
    return np.unique(np.clip(np.round(p), 0, 1023).astype(np.int64), axis=0)


def d1_psnr(a, b, peak=1023.0):
    da, _ = cKDTree(b).query(a)
    db, _ = cKDTree(a).query(b)
    mse = max(np.mean(da ** 2), np.mean(db ** 2))
    return 10 * np.log10(3 * peak ** 2 / max(mse, 1e-12))


def pick_threshold(ckpt, emb_fn, chanstr, ch, n_pts):
    """--auto-thh: the occupancy threshold of one RD point, by the reference's own rule of thumb -- the README's 0.64 when
    the decoder then returns between 2/3 and 3/2 of the input's point count, otherwise the smallest threshold whose decoded
    point count is within 1.5 x of the input (a lambda that trains a flatter occupancy needs a lower one).  The counts
    come from ONE eval forward of the quantised decoder over all blocks (what encode thresholds); returns (thh, {thh: count})."""
    import torch
    sys.argv = [sys.argv[0]]
    import NVFPCC as cli
    args = cli.build_parser().parse_args(["encode", "cloud.ply", "--chanstr", chanstr, "--ch", str(ch)])
    dev = torch.device("cuda")
    net = cli._build_net(args, dev)
    d = torch.load(ckpt, map_location=dev)
    net.load_state_dict({k: v for k, v in d.items() if "init_coords" not in k}, strict=False)
    emb = torch.load(emb_fn, map_location=dev).to(dev).float().contiguous()
    cands = sorted(set([round(0.20 + 0.02 * i, 2) for i in range(31)] + [0.64, 0.65]))
    counts = {t: 0 for t in cands}
    with torch.no_grad():
        lat = net.get_latent_code(emb)["quantized_latent"]
        for i in range(0, lat.shape[0], 64):
            out = net.reconstruct(lat[i:i + 64].contiguous(), 2)
            for t in cands:
                counts[t] += int((out > t).sum().item())
    del net
    torch.cuda.empty_cache()
    if 2 / 3 * n_pts <= counts[0.64] <= 1.5 * n_pts:
        return 0.64, counts
    ok = [t for t in cands if 0 < counts[t] <= 1.5 * n_pts]
    return (ok[0] if ok else cands[-1]), counts


def run(cmd, cwd, log):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    log.write(r.stdout)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:])
    return r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lambdas", default="50,200,800")
    ap.add_argument("--epochs", type=int, default=301)
    ap.add_argument("--phase_change", type=int, default=100)
    ap.add_argument("--radius", type=float, default=150.0)
    ap.add_argument("--n_dir", type=int, default=1500000)
    ap.add_argument("--chanstr", default="8,16,8,8")
    ap.add_argument("--ch", type=int, default=3)
    ap.add_argument("--thh", type=float, default=0.6)
    ap.add_argument("--qps", default="16", help="weight quantisation parameters (manipulate_weights.py's third argument, --qp)")
    ap.add_argument("--thhs", default="", help="occupancy thresholds (--thh of encode / decode); default: --thh only")
    ap.add_argument("--auto-thh", action="store_true", help="one threshold per (lambda, qp), chosen by pick_threshold")
    ap.add_argument("--wemb", type=float, default=5.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--workdir", default="")
    a = ap.parse_args()
    from nvfpcc_amd.preprocess import preprocess
    from nvfpcc_amd.recon import read_ply_ascii
    from tests.golden_inputs import write_cloud_ply
    wd = a.workdir or tempfile.mkdtemp(prefix="nvf_rd_")
    os.makedirs(wd, exist_ok=True)
    os.chdir(wd)
    log = open("rd_sweep.log", "w")
    pts = make_cloud(7, a.radius, a.n_dir)
    write_cloud_ply("cloud.ply", pts)
    t0 = time.time()
    origins, gt, dist = preprocess("cloud.ply")
    t_pre = time.time() - t0
    n_pts, n_blk = len(pts), len(origins)
    print(f"cloud: {n_pts} points, {n_blk} level-5 cubes, occupancy {100 * n_pts / (n_blk * 32768):.2f} %, "
          f"pre-processing {t_pre:.1f} s")
    cli = os.path.join(ROOT, "NVFPCC.py")
    common = ["--chanstr", a.chanstr, "--ch", str(a.ch)]
    rows = []
    jobs = [(float(v), 0) for v in a.lambdas.split(",")]
    while jobs:
        lam, seed = jobs.pop(0)
        ck = f"ckpt_{lam:g}_s{seed}"
        t0 = time.time()
        run([cli, "train", "cloud.ply", "--checkpoint_dir", ck, "--batchsize", "16", "--lambda", str(lam), "--lr", "1e-3",
             "--w1", "10", "--w2", "57", "--wemb", str(a.wemb), "--shuffle", "True", "--epochs", str(a.epochs), "--phase_change",
             str(a.phase_change), "--seed", str(seed)] + common, wd, log)
        t_train = time.time() - t0
        last = (a.epochs - 1) // 10 * 10
        collapsed = False
        for qp in [int(v) for v in a.qps.split(",")]:
            tagq = f"{lam:g}_q{qp}"
            run([os.path.join(ROOT, "manipulate_weights.py"), f"{ck}/{last:04d}.ckpt", f"q_{tagq}.ckpt", str(qp)], wd, log)
            thh_list = [float(v) for v in a.thhs.split(",")] if a.thhs else [a.thh]
            auto_note = ""
            if a.auto_thh:
                t_auto, cnts = pick_threshold(f"q_{tagq}.ckpt", f"{ck}/{last:04d}_emb.ckpt", a.chanstr, a.ch, n_pts)
                thh_list = [t_auto]
                auto_note = "thh chosen automatically (points at 0.5 / 0.6 / 0.64: %d / %d / %d)" % (cnts[0.5], cnts[0.6], cnts[0.64])
            for thh in thh_list:
                out = run([cli, "encode", "cloud.ply", "--batchsize", "64", "--load_weights", f"q_{tagq}.ckpt", "--load_emb",
                           f"{ck}/{last:04d}_emb.ckpt", "--thh", str(thh), "--qp", str(qp), "--pack_fn", f"pack_{tagq}.pk"]
                          + common, wd, log)
                psnr1 = float(out.split("PSNR1: ")[-1].split()[0])
                run([cli, "decode", f"pack_{tagq}.pk", "--batchsize", "64", "--thh", str(thh), "--qp", str(qp), "--N",
                     str(n_blk)] + common, wd, log)
                enc, dec = read_ply_ascii("rc_enc.ply"), read_ply_ascii("rc_dec.ply")
                same = enc.shape == dec.shape and np.array_equal(enc, dec)
                pack = pickle.load(open(f"pack_{tagq}.pk", "rb"))
                bits_lat = 8 * len(pack["latent_pack"]["latent_byte_stream"])
                bits_net = 8 * len(pack["net_weight_pack"]["bit_stream"])
                # a collapsed run (every latent codes to the same symbol -- the stream is a few bytes: the rate term won the
                # first epochs and the decoder learned a constant) is a property of (lambda, noise seed), not an operating
                # point: say so, and train that lambda once more with the next noise seed
                if bits_lat / n_pts < 1e-3 and not collapsed:
                    collapsed = True
                    if seed < 2:
                        jobs.insert(0, (lam, seed + 1))
                # what a degenerate row is: said in the table, not left to the reader
                if qp != 16:
                    note = ("qp != 16: the decoder's forward re-rounds every kernel to the 1/16 grid (network.py:611-620, "
                            "q = 2) whatever grid the checkpoint was stored on, so it runs weights it was not trained "
                            "with -- collapse expected, the reference behaves the same")
                elif len(dec) == 0:
                    note = "no probability above thh: this lambda trains a flatter occupancy, the threshold is too high for it"
                elif len(dec) > 1.5 * n_pts:
                    note = f"over-decoded ({len(dec) / n_pts:.1f} x the input points): thh too low for this lambda"
                else:
                    note = ""
                note = (note + "; " if note and auto_note else note) + auto_note
                if collapsed:
                    note = ("COLLAPSED RUN (noise seed %d: the latent table ended constant -- no information in the latents; "
                            "not an operating point%s); " % (seed, ", retrained below with the next seed" if seed < 2 else "")) + note
                elif seed:
                    note = ("noise seed %d; " % seed) + note
                rows.append((lam, qp, thh, (bits_lat + bits_net) / n_pts, bits_lat / n_pts, bits_net / n_pts, psnr1,
                             d1_psnr(pts, dec) if len(dec) else float("nan"), len(dec), same, t_train, note))
                print(rows[-1])
    lines = ["| lambda | qp | thh | bpp | bpp latents | bpp weights | PSNR1 (dB) | D1 PSNR sym. (dB) | decoded points | rc_enc == rc_dec | train s | note |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r[0]:g} | {r[1]} | {r[2]:g} | {r[3]:.4f} | {r[4]:.4f} | {r[5]:.4f} | {r[6]:.2f} | {r[7]:.2f} | {r[8]} | "
                     f"{r[9]} | {r[10]:.0f} | {r[11]} |")
    table = "\n".join(lines)
    head = (f"RD sweep on a synthetic 10-bit surface: {n_pts} points, {n_blk} level-5 cubes; ch={a.ch}, chanstr={a.chanstr}, "
            f"{a.epochs} epochs (phase change {a.phase_change}), batch 16, lr 1e-3, w1 10, w2 57, wemb {a.wemb:g}, weights on "
            f"the 1/qp grid (qp 16 = the reference's 4-bit setting); 1 x MI355X.\n\n")
    print(head + table)
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w") as f:
            f.write(head + table + "\n")


if __name__ == "__main__":
    main()
