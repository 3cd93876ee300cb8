"""Diagnostic (not a test): WHERE does the ch = 8 latent gradient of the HIP path leave float64?  The three stages of the
latent backward (rate gradient + straight-through addend -> GDN backward -> 1x1x1 backward-data) are captured from one
eng.latent_step and each is recomputed in float64 torch FROM THE HIP PATH'S OWN INPUTS, so a stage's error is its own."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_measured_path as T
from nvfpcc_amd import ops

gpu = torch.device("cuda")
cap = {}
_lr, _gb, _cg = ops.latent_rate, ops.gdn_bwd, ops.conv3d_gather


def lr(x, sigma, mu, mode, **kw):
    out = _lr(x, sigma, mu, mode, **kw)
    if kw.get("want_grad"):
        cap["rate"] = (x.clone(), sigma.clone(), mu.clone(), mode, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()},
                       [o.clone() if torch.is_tensor(o) else o for o in out])
    return out


def gb(x, beta, gamma, dy, inverse, *a, **kw):
    out = _gb(x, beta, gamma, dy, inverse, *a, **kw)
    if not inverse:
        cap["gdn"] = (x.clone(), beta.clone(), gamma.clone(), dy.clone(), out[0].clone())
    return out


_sb = ops.stem_bwd


def sb(g1, x0, a0, w0b, wub, beta, gamma, *a, **kw):
    out = _sb(g1, x0, a0, w0b, wub, beta, gamma, *a, **kw)
    cap["stem"] = (g1.clone(), x0.clone(), a0.clone(), beta.clone(), gamma.clone(), out[0].clone(), out[1].clone())
    return out


ops.latent_rate, ops.gdn_bwd, ops.stem_bwd = lr, gb, sb


def rate_grad64(x, sigma, mu, u, g, addend):
    """d/dx of g * sum -log2(max(Phi((v - mu + .5)/s) - Phi((v - mu - .5)/s), 1e-8)), v = x + u - .5 ... in float64"""
    x = x.double().clone().requires_grad_(True)
    s = sigma.double().abs().view(1, -1, 1, 1, 1)
    m = mu.double().view(1, -1, 1, 1, 1)
    v = x + (u.double() - 0.5)
    nd = torch.distributions.normal.Normal(0.0, 1.0)
    lik = nd.cdf((v - m + 0.5) / s) - nd.cdf((v - m - 0.5) / s)
    lik = torch.clamp(lik, min=1e-8)       # (no entry sits at the bound in these runs; checked below)
    bits = (-torch.log(lik) / np.log(2)).sum()
    (g * bits).backward()
    return x.grad + addend.double(), float((lik <= 1e-8).sum())


for ch, chans in ((8, (8, 16, 8, 8)), (3, (8, 16, 8, 8))):
    net, eng, P, gt, dist, emb = T.make(gpu, ch, chans, 6)
    a, de = eng.latent_step(2, update=False)
    x, sigma, mu, mode, kw, out = cap["rate"]
    from tests.philox_np import latent_noise
    ids = np.arange(6)
    u = torch.from_numpy(latent_noise(0, eng.noise_step, ids, ch)).to(gpu)
    g = kw.get("g_host", 1.0)
    ref, nb = rate_grad64(x, sigma, mu, u, g, kw["dx_addend"])
    dlat = out[2].double()
    sc = ref.abs().max().item()
    print(f"ch={ch}: stage 1 (rate gradient + addend): max err / max = {(dlat - ref).abs().max().item() / sc:.2e} (scale {sc:.3g}, at bound {nb})")
    k = (dlat - ref).abs().argmax().item()
    xi = x.reshape(-1)[k].item()
    c = (k // 8) % ch
    print(f"   worst entry: x={xi:.6f} sigma={abs(sigma[c].item()):.5f} mu={mu[c].item():.5f} hip={dlat.reshape(-1)[k].item():.6e} ref={ref.reshape(-1)[k].item():.6e} addend={kw['dx_addend'].reshape(-1)[k].item():.6e}")
    # stage 2: GDN backward (forward direction, not inverse) in float64 from the HIP dlat
    h, beta_hat, gamma_hat, dy, dh = cap["gdn"]
    hh = h.double().clone().requires_grad_(True)
    ped = 2.0 ** -36
    beta = torch.clamp(beta_hat.double(), min=(1e-6 + ped) ** 0.5) ** 2 - ped
    gamma = torch.clamp(gamma_hat.double(), min=2.0 ** -18) ** 2 - ped
    norm = torch.sqrt(torch.nn.functional.conv3d(hh ** 2, gamma.view(ch, ch, 1, 1, 1), beta))
    y = hh / norm
    (y * dy.double()).sum().backward()
    sc = hh.grad.abs().max().item()
    print(f"ch={ch}: stage 2 (GDN backward): max err / max = {(dh.double() - hh.grad).abs().max().item() / sc:.2e} (scale {sc:.3g})")
    # stage 3: 1x1x1 backward-data
    w = (eng.layers["latent"].w_bwd if False else net.latent_gen.h_analysis_2.kernel + net.latent_gen.h_analysis_2.kernel_init).double()
    de64 = torch.nn.functional.conv_transpose3d(dh.double(), w)
    sc = de64.abs().max().item()
    print(f"ch={ch}: stage 3 (1x1x1 backward-data): max err / max = {(de.double() - de64).abs().max().item() / sc:.2e} (scale {sc:.3g})")
    # stage 0: the stem's backward (conv0^T -> IGDN' -> up0^T) in float64 from the HIP path's own g1 / a0
    g1, x0s, a0, ibeta_hat, igamma_hat, da0, dx0 = cap["stem"]
    rec = net.reconstructor
    W0 = (rec.conv0.kernel + rec.conv0.kernel_init).double()        # q = 2: round16(kernel) + init
    W0 = (torch.round(rec.conv0.kernel * 16) / 16 + rec.conv0.kernel_init).double()
    Wu = (torch.round(rec.up0.kernel * 16) / 16 + rec.up0.kernel_init).double()
    F = torch.nn.functional
    dh0 = F.conv3d(g1.double(), W0, stride=2, padding=2)
    aa = a0.double().clone().requires_grad_(True)
    ib = torch.clamp(ibeta_hat.double(), min=(1e-6 + ped) ** 0.5) ** 2 - ped
    igm = torch.clamp(igamma_hat.double(), min=2.0 ** -18) ** 2 - ped
    c0 = a0.shape[1]
    nrm = torch.sqrt(F.conv3d(aa ** 2, igm.view(c0, c0, 1, 1, 1), ib))
    ((aa * nrm) * dh0).sum().backward()
    sc = aa.grad.abs().max().item()
    print(f"ch={ch}: stage 0a+0b (conv0^T, IGDN backward): da0 max err / max = {(da0.double() - aa.grad).abs().max().item() / sc:.2e} (scale {sc:.3g})")
    dx64 = F.conv3d(da0.double(), Wu, stride=2, padding=2)
    sc = dx64.abs().max().item()
    k = (dx0.double() - dx64).abs().argmax().item()
    mag = F.conv3d(da0.double().abs(), Wu.abs(), stride=2, padding=2)
    print(f"ch={ch}: stage 0c (up0^T): dx0 max err / max = {(dx0.double() - dx64).abs().max().item() / sc:.2e} (scale {sc:.3g}); "
          f"sum|terms| / |sum| at the worst entry = {mag.reshape(-1)[k].item() / max(abs(dx64.reshape(-1)[k].item()), 1e-30):.3g}, "
          f"max over entries of sum|terms| / max|dx0| = {mag.max().item() / sc:.3g}")
    # end to end, for reference
    n_all = float(eng.counts.sum())
    P64 = {k: v.double() for k, v in P.items()}
    r64 = T._oracle_step(P64, emb.double(), gt.double(), dist.double(), ids, 2, n_all, eng.noise_step, layer_ids=T._layer_ids(net))[4]
    s = r64.abs().max().item()
    print(f"ch={ch}: end to end HIP vs fp64 oracle {(de.cpu().double() - r64).abs().max().item() / s:.2e}")
