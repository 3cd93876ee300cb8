"""Diagnostic (not a test): are the 1e-4 .. 1e-3 distances of some batch-16 gradient slices from the float64 oracle ReLU-mask
disagreements (a pre-activation within rounding of zero switches a gradient entry on or off) rather than arithmetic?
For each decoder / q: (1) the ReLU masks of the HIP step against the float64 oracle's, with the size of the disagreeing
activations; (2) every gradient slice against the float64 oracle with its OWN masks and with the HIP step's masks imposed."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_measured_path as T
gpu = torch.device("cuda")
LAYER = {"conv0": "y1", "up1": "y2", "conv1": "y3", "up2": "y4", "conv2": "y5"}
for dec in (sys.argv[1:] or ["W", "S"]):
    ch, chans = (3, (8, 16, 8, 8)) if dec == "S" else (8, (16, 32, 16, 16))
    for q in (1, 2):
        net, eng, P, gt, dist, emb = T.make(gpu, ch, chans, 40)
        ids = np.random.default_rng(3).permutation(40)[:16].astype(np.int64)
        n_pts = float(eng.counts[ids].sum())
        lids = T._layer_ids(net)
        a = eng.train_step(ids, q, update=False)
        P64 = {k: v.double() for k, v in P.items()}
        args64 = (P64, emb.double(), gt.double(), dist.double(), ids, q, n_pts, eng.noise_step)
        keep = {}
        g_own = T._oracle_step(*args64, layer_ids=lids, keep=keep)[3]
        g32 = T._oracle_step(P, emb, gt, dist, ids, q, n_pts, eng.noise_step, layer_ids=lids)[3]
        masks = {n: (a[k] > 0).cpu() for n, k in LAYER.items()}
        print(f"== decoder {dec} q={q}")
        for n, k in LAYER.items():
            ref = keep[n].detach()
            mine = a[k].cpu().double()
            flip = (mine > 0) != (ref > 0)
            big = max(float(ref.abs().max()), 1e-30)
            worst = float(torch.maximum(mine.abs(), ref.abs())[flip].max()) / big if flip.any() else 0.0
            print(f"   ReLU after {n:6s}: {int(flip.sum()):5d} of {flip.numel():9d} mask entries differ from the fp64 oracle's; "
                  f"largest |activation| among them / layer max = {worst:.1e}; max |HIP - fp64| / layer max = {float((mine - ref).abs().max()) / big:.1e}")
        g_imp = T._oracle_step(*args64, layer_ids=lids, relu_masks=masks)[3]
        print("   slice | fp32 oracle vs fp64 | HIP vs fp64 (own masks) | HIP vs fp64 (HIP's masks imposed)")
        for name, (off, n) in eng.slices.items():
            mine = eng.flat_g[off:off + n].cpu().numpy().astype(np.float64)
            r = g_own[name].numpy().reshape(-1)
            sc = max(np.abs(r).max(), 1e-30)
            e_own = np.abs(mine - r).max() / sc
            e_imp = np.abs(mine - g_imp[name].numpy().reshape(-1)).max() / sc
            e32 = np.abs(g32[name].double().numpy().reshape(-1) - r).max() / sc
            print(f"   {name:44s} {e32:9.2e} {e_own:9.2e} {e_imp:9.2e}{' <--' if e_imp > 2e-5 else ''}")
