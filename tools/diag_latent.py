"""Diagnostic (not a test): latent gradient of the ch = 8 narrow decoder, HIP vs the oracle in fp32 and fp64."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_measured_path as T
gpu = torch.device("cuda")
for ch, chans in ((8, (8, 16, 8, 8)), (3, (8, 16, 8, 8)), (3, (8, 8, 8, 8))):
    net, eng, P, gt, dist, emb = T.make(gpu, ch, chans, 6)
    a, de = eng.latent_step(2, update=False)
    n_all = float(eng.counts.sum())
    r32 = T._oracle_step(P, emb, gt, dist, np.arange(6), 2, n_all, eng.noise_step, layer_ids=T._layer_ids(net))[4]
    P64 = {k: v.double() for k, v in P.items()}
    r64 = T._oracle_step(P64, emb.double(), gt.double(), dist.double(), np.arange(6), 2, n_all, eng.noise_step, layer_ids=T._layer_ids(net))[4]
    s = r64.abs().max().item()
    print(ch, chans, "scale %.3g  hip-vs-64 %.2e  oracle32-vs-64 %.2e  hip-vs-32 %.2e" % (
        s, (de.cpu().double() - r64).abs().max().item() / s, (r32.double() - r64).abs().max().item() / s,
        (de.cpu() - r32).abs().max().item() / s))
