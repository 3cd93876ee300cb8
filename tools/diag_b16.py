"""Diagnostic (not a test): batch-16 train step of both BASELINE decoders, default (Winograd) and direct engines, against
the oracle in float32 and float64 -- per gradient slice: HIP vs fp64, fp32 oracle vs fp64 (how well conditioned the
comparison is), as max |diff| / max |gradient| of the slice."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_measured_path as T
gpu = torch.device("cuda")
which = sys.argv[1:] or ["W", "S"]
for dec in which:
    ch, chans = (3, (8, 16, 8, 8)) if dec == "S" else (8, (16, 32, 16, 16))
    for q in (2, 1):
        rows = {}
        for wino in (True, False):
            from nvfpcc_amd import engine as E
            E._WINO = wino
            net, eng, P, gt, dist, emb = T.make(gpu, ch, chans, 40)
            ids = np.random.default_rng(3).permutation(40)[:16].astype(np.int64)
            n_pts = float(eng.counts[ids].sum())
            lids = T._layer_ids(net)
            eng.train_step(ids, q, update=False)
            if wino:
                g32 = T._oracle_step(P, emb, gt, dist, ids, q, n_pts, eng.noise_step, layer_ids=lids)[3]
                g64 = T._oracle_step({k: v.double() for k, v in P.items()}, emb.double(), gt.double(), dist.double(), ids, q,
                                     n_pts, eng.noise_step, layer_ids=lids)[3]
            for name, (off, n) in eng.slices.items():
                r64 = g64[name].numpy().reshape(-1)
                sc = max(np.abs(r64).max(), 1e-30)
                mine = eng.flat_g[off:off + n].cpu().numpy().astype(np.float64)
                rows.setdefault(name, [sc, np.abs(g32[name].double().numpy().reshape(-1) - r64).max() / sc])
                rows[name].append(np.abs(mine - r64).max() / sc)
                rows[name].append(np.abs(mine - g32[name].double().numpy().reshape(-1)).max() / sc)
        print(f"== decoder {dec} q={q}: slice | max |g| | fp32 oracle vs fp64 | HIP(wino) vs fp64 | HIP(wino) vs fp32 | HIP(direct) vs fp64 | HIP(direct) vs fp32")
        for name, r in rows.items():
            flag = " <--" if max(r[2], r[4]) > 1e-4 else ""
            print(f"{name:44s} {r[0]:10.3e} {r[1]:9.2e} {r[2]:9.2e} {r[3]:9.2e} {r[4]:9.2e} {r[5]:9.2e}{flag}")
