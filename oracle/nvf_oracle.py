"""ORACLE -- test infrastructure only.  Never imported by the product path.

CPU restatement (plain torch CPU tensor ops) of NVFPCC's per-block neural
volumetric field hot path, written from the algorithm, not from the source
text.  Every function cites the reference lines it follows
(paths relative to /root/reference).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker / the reported CPU baseline.

Parity pin: ``tests/test_oracle_golden.py`` checks this file against the golden
vectors in ``tests/golden/*.npz`` which ``tools/gen_golden.py`` produced by
importing the real reference (utils/network.py, gdn_3d.py, utils/loss.py) in
the build container.

The network state is an ordered ``dict`` keyed exactly like the reference's
``Net.state_dict()`` (NVFPCC.py:32-39): ``latent_gen.*``, ``entropy_coder.*``,
``reconstructor.*``.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LOG2 = math.log(2.0)
REPARAM_OFFSET = 2.0 ** -18          # gdn_3d.py:41
PEDESTAL = REPARAM_OFFSET ** 2       # gdn_3d.py:51
BETA_BOUND = (1e-6 + PEDESTAL) ** 0.5  # gdn_3d.py:52
GAMMA_BOUND = REPARAM_OFFSET         # gdn_3d.py:53

TRUNK = ("up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls")  # network.py:4781-4792
HEADS = ("conv1_cls", "conv0_cls")


# --------------------------------------------------------------------------
# straight-through pieces
# --------------------------------------------------------------------------
class _Floor(torch.autograd.Function):
    """max(x, bound) whose gradient also passes when it pushes x upward.

    gdn_3d.py:13-29 / utils/network.py:56-72: backward keeps g where
    ``x >= bound`` or ``g < 0``.
    """

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = float(bound)
        return x.clamp(min=float(bound))

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        keep = (x >= ctx.bound) | (g < 0)
        return g * keep, None


class _RoundSTE(torch.autograd.Function):
    """round(x*s)/s forward, identity backward (network.py:25-50)."""

    @staticmethod
    def forward(ctx, x, scale):
        if scale == 1:
            return torch.round(x)
        return torch.round(x * scale) / scale

    @staticmethod
    def backward(ctx, g):
        return g, None


def floor_ste(x, bound):
    return _Floor.apply(x, bound)


def round_ste(x, scale=1):
    return _RoundSTE.apply(x, scale)


# --------------------------------------------------------------------------
# seed-derived frozen init (network.py:377-400, 564-742)
# --------------------------------------------------------------------------
def seeded_kernel_init(shape, u):
    """(u-.5)*2*sqrt(6/fan_in), fan_in = shape[1]*k^3 (network.py:377-400).

    For transposed convs shape[1] is Cout -- the reference's quirk, kept.
    """
    fan_in = shape[1] * int(np.prod(shape[2:]))
    bound = math.sqrt(3.0) * (math.sqrt(2.0) / math.sqrt(fan_in))
    seed = torch.from_numpy(np.asarray(u, np.float64).reshape(shape)).float()
    return (seed - 0.5) * 2 * bound


def seeded_bias_init(n, u, fan):
    seed = torch.from_numpy(np.asarray(u, np.float64).reshape(n)).float()
    return (seed - 0.5) * 2 * (1 / np.sqrt(fan))


def layer_table(ch, channels):
    """(name, kind, weight shape, bias fan) in seed-consumption order.

    Order: latent_gen (network.py:4597-4606), then up0, conv0, up1, conv1, up2,
    conv2, conv2_cls, conv1_cls, conv0_cls (network.py:4664-4751).
    """
    c0, c1, c2, c3 = channels
    return [
        ("latent_gen.h_analysis_2", "iconv", (ch, ch, 1, 1, 1), ch),
        ("reconstructor.up0", "qconvT", (ch, c0, 5, 5, 5), ch),
        ("reconstructor.conv0", "qconvT", (c0, c1, 5, 5, 5), c0),
        ("reconstructor.up1", "qconvT", (c1, c2, 5, 5, 5), c1),
        ("reconstructor.conv1", "qconv", (c2, c2, 4, 4, 4), c2),
        ("reconstructor.up2", "qconvT", (c2, c3, 5, 5, 5), c2),
        ("reconstructor.conv2", "qconv", (c3, c3, 4, 4, 4), c3),
        ("reconstructor.conv2_cls", "qconv", (1, c3, 3, 3, 3), c3),
        ("reconstructor.conv1_cls", "iconv", (1, c2, 3, 3, 3), c2),
        ("reconstructor.conv0_cls", "iconv", (1, c1, 3, 3, 3), c1),
    ]


def _gdn_init(ch):
    beta = torch.sqrt(torch.ones(ch) + PEDESTAL)                    # gdn_3d.py:56
    gamma = torch.sqrt(0.1 * torch.eye(ch) + PEDESTAL)              # gdn_3d.py:61-64
    return beta, gamma, torch.FloatTensor([PEDESTAL])               # gdn_3d.py:68


def build_state(ch, channels, seed):
    """Fresh network state in the reference's state_dict order.

    ``seed`` is the float64 SEED3 vector; returns (state, values consumed).
    """
    seed = np.asarray(seed, np.float64).reshape(-1)
    ptr = 0
    inits = {}
    for name, kind, shape, fan in layer_table(ch, channels):
        nk = int(np.prod(shape))
        nb = shape[1] if kind == "qconvT" else shape[0]
        inits[name] = (seeded_kernel_init(shape, seed[ptr:ptr + nk]),
                       seeded_bias_init(nb, seed[ptr + nk:ptr + nk + nb], fan), shape, nb)
        ptr += nk + nb

    st = OrderedDict()

    def put_conv(name):
        ki, bi, shape, nb = inits[name]
        st[name + ".kernel"] = torch.zeros(shape)
        st[name + ".b"] = torch.zeros(nb)
        st[name + ".kernel_init"] = ki
        st[name + ".b_init"] = bi

    put_conv("latent_gen.h_analysis_2")
    b, g, p = _gdn_init(ch)
    st["latent_gen.gdn_2.beta"], st["latent_gen.gdn_2.gamma"], st["latent_gen.gdn_2.pedestal"] = b, g, p
    st["entropy_coder.sigma"] = torch.ones(1, ch, 1, 1, 1)          # network.py:4504-4506
    st["entropy_coder.mu"] = torch.zeros(1, ch, 1, 1, 1)            # network.py:4507-4512
    b, g, p = _gdn_init(channels[0])
    st["reconstructor.activation.beta"] = b
    st["reconstructor.activation.gamma"] = g
    st["reconstructor.activation.pedestal"] = p
    for n in TRUNK + HEADS:
        put_conv("reconstructor." + n)
    st["reconstructor.likelihood_model.sigma"] = torch.ones(1)      # network.py:291-296
    st["reconstructor.likelihood_model.mu"] = torch.zeros(1)
    return st, ptr


def trainable_keys(state):
    """The 28 keys ``net.parameters()`` yields, in that order."""
    return [k for k in state if not (k.endswith("_init") or k.endswith("pedestal"))]


# --------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------
def gdn3d(x, beta_hat, gamma_hat, inverse):
    """GDN3d / IGDN3d forward (gdn_3d.py:72-95, 137-159)."""
    c = x.shape[1]
    beta = floor_ste(beta_hat, BETA_BOUND) ** 2 - PEDESTAL
    gamma = floor_ste(gamma_hat, GAMMA_BOUND) ** 2 - PEDESTAL
    norm = torch.sqrt(F.conv3d(x ** 2, gamma.view(c, c, 1, 1, 1), beta))
    return x * norm if inverse else x / norm


def std_normal_cdf(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2)))                 # torch Normal(0,1).cdf


def gaussian_bits(v, sigma, mu, half):
    """sum -log2(max(Phi((v-mu+h)/s) - Phi((v-mu-h)/s), 1e-8)) (network.py:145-161)."""
    like = std_normal_cdf((v - mu + half) / sigma) - std_normal_cdf((v - mu - half) / sigma)
    like = floor_ste(like, 1e-8)
    return (-1 * torch.log(like) / np.log(2)).sum()


def effective_kernel(k, k_init, q, u=None):
    """network.py:611-620 / 677-686: q=1 adds (U-.5)/16, q=2 rounds to 1/16 (STE)."""
    if q == 1:
        u = torch.rand_like(k) if u is None else u
        k = k + (u - 0.5) * (1 / 16)
    elif q == 2:
        k = round_ste(k, 16)
    return k + k_init


def latent_gen(P, emb):
    """SingleLayerLatentGen (network.py:4610-4612): GDN(1x1x1 conv)."""
    p = "latent_gen.h_analysis_2."
    h = F.conv3d(emb, P[p + "kernel"] + P[p + "kernel_init"], P[p + "b"] + P[p + "b_init"])
    return gdn3d(h, P["latent_gen.gdn_2.beta"], P["latent_gen.gdn_2.gamma"], inverse=False)


def entropy_coder(P, latent, mode, u=None):
    """QuantGaussianLikelihood.forward (network.py:4514-4539)."""
    u = torch.rand_like(latent) if u is None else u
    noisy = latent + (u - 0.5)
    rounded = round_ste(latent, 1)
    v = noisy if mode == "train" else rounded
    bits = gaussian_bits(v, torch.abs(P["entropy_coder.sigma"]), P["entropy_coder.mu"], 0.5)
    return rounded, bits


def decoder(P, x, q, u_w=None, keep=None, relu_masks=None):
    """CompDecoder.forward, live definition (network.py:4758-4779).

    ``u_w``: optional dict layer-name -> uniform sample for the q=1 weight noise.
    ``keep``: optional dict that receives every intermediate activation.
    ``relu_masks`` (checker only; None = the reference's F.relu): dict layer-name -> bool tensor; that layer's ReLU becomes
    ``pre_activation * mask`` -- the subgradient choice of ANOTHER fp32 evaluation imposed on this one.  ReLU is
    discontinuous in its derivative at 0: a pre-activation that two correctly rounded evaluations place on either side of
    zero (|x| ~ 1e-7 of the layer's range) switches a whole gradient entry on or off, which no tolerance on rounding
    covers.  With the masks imposed, what is compared is arithmetic; the masks themselves are compared separately.
    """
    r = "reconstructor."
    u_w = u_w or {}

    def eff(n):
        return (effective_kernel(P[r + n + ".kernel"], P[r + n + ".kernel_init"], q, u_w.get(n)),
                P[r + n + ".b"] + P[r + n + ".b_init"])

    def head(n, t):
        k = P[r + n + ".kernel"] + P[r + n + ".kernel_init"]           # IConv3d, network.py:735-742
        return torch.sigmoid(F.conv3d(t, k, P[r + n + ".b"] + P[r + n + ".b_init"], 1, 1))

    def note(n, t):
        if keep is not None:
            keep[n] = t
        return t

    def relu(n, t):
        if keep is not None:
            keep[n + ".pre"] = t
        if relu_masks is not None and n in relu_masks:
            return t * relu_masks[n].to(t.dtype)
        return F.relu(t)

    w, b = eff("up0")
    t = note("up0", F.conv_transpose3d(x, w, b, 2, 2, 1))
    t = note("igdn", gdn3d(t, P[r + "activation.beta"], P[r + "activation.gamma"], inverse=True))
    w, b = eff("conv0")
    t = note("conv0", relu("conv0", F.conv_transpose3d(t, w, b, 2, 2, 1)))
    cls0 = note("cls0", head("conv0_cls", t))
    w, b = eff("up1")
    t = note("up1", relu("up1", F.conv_transpose3d(t, w, b, 2, 0, 0)))
    w, b = eff("conv1")
    t = note("conv1", relu("conv1", F.conv3d(t, w, b, 1, 0)))
    cls1 = note("cls1", head("conv1_cls", t))
    w, b = eff("up2")
    t = note("up2", relu("up2", F.conv_transpose3d(t, w, b, 2, 0, 0)))
    w, b = eff("conv2")
    t = note("conv2", relu("conv2", F.conv3d(t, w, b, 1, 0)))
    w, b = eff("conv2_cls")
    out = note("out", torch.sigmoid(F.conv3d(t, w, b, 1, 1)))
    return out, [cls0, cls1, out], weight_bits(P)


def weight_bits(P):
    """Per-kernel rate of the 7 quantised trunk kernels (network.py:4777-4778, 301-305)."""
    r = "reconstructor."
    s = torch.abs(P[r + "likelihood_model.sigma"])
    m = P[r + "likelihood_model.mu"]
    return torch.stack([
        gaussian_bits(round_ste(P[r + n + ".kernel"], 16).reshape(-1, 1), s, m, 0.5 / 16)
        for n in TRUNK])


def net_forward(P, emb, mode, q, u_latent=None, u_w=None, keep=None, relu_masks=None):
    """Net.forward (NVFPCC.py:41-45).  ``relu_masks``: see decoder (checker only)."""
    lat = latent_gen(P, emb)
    rounded, lbits = entropy_coder(P, lat, mode, u_latent)
    if keep is not None:
        keep["latent"], keep["latent_rounded"] = lat, rounded
    out, cls, nbits = decoder(P, rounded, q, u_w, keep, relu_masks)
    return out, cls, nbits, lbits


def decoder_aux_bits(channels):
    """CompDecoder.get_bits side-information term (network.py:4797)."""
    return sum(channels[i] * 2 for i in (1, 2, 3)) * 32 + 32 + (channels[1] ** 2 + channels[1]) * 32


def latent_header_bits(P):
    """QuantGaussianLikelihood.get_bits (network.py:4541-4545)."""
    return 32 * (P["entropy_coder.sigma"].numel() + P["entropy_coder.mu"].numel())


# --------------------------------------------------------------------------
# losses / metrics (utils/loss.py)
# --------------------------------------------------------------------------
def gt_pyramid(x):
    """MultiscaleProcessor (NVFPCC.py:81-88): [pool(pool(x)), pool(x), x]."""
    x1 = F.max_pool3d(x, 2, 2)
    return [F.max_pool3d(x1, 2, 2), x1, x]


def _focal_terms(p, gt, alpha):
    m = gt.bool()
    pt = torch.where(m, p, 1 - p)
    a = torch.full_like(p, alpha)
    at = torch.where(m, a, 1 - a)          # fp32 "-alpha + 1", as loss.py:66-67 evaluates it
    pt = pt.clamp(min=1e-9)
    return at, pt


def focal_dense(p, gt, alpha=0.97, gamma=2):
    """get_focal_dense (loss.py:61-72)."""
    at, pt = _focal_terms(p, gt, alpha)
    return (-1 * at * (1 - pt) ** gamma * torch.log(pt)).sum()


def surf_focal_dense(p, gt, dist, beta=1, alpha=0.97, gamma=2):
    """get_surf_focal_dense (loss.py:94-111): focal term weighted by dist + gt*beta."""
    at, pt = _focal_terms(p, gt, alpha)
    w = dist + gt.bool() * beta
    return (-1 * at * (1 - pt) ** gamma * w * torch.log(pt)).sum()


def acc_dense(p, gt, thh=0.5):
    """get_acc_dense (loss.py:74-84) -> (TPR, TNR)."""
    m = gt.bool()
    return ((p > thh) & m).sum() / m.sum(), ((p <= thh) & ~m).sum() / (~m).sum()


def sse1(p, dist, thh):
    """get_sse1 (loss.py:113-121) -> (sum((p>thh)*dist)^2, count(p>thh))."""
    pred = (p > thh).float()
    return torch.square(pred * dist).sum(), pred.sum()


def squared_error_map(p, dist, thh):
    """get_se (loss.py:123-128) -> [B,2,...]: ((p>thh)*dist)^2 stacked with p along the channel axis."""
    pred = (p > thh).float()
    return torch.cat([torch.square(pred * dist), p], 1)


def rd_loss(P, emb, gt, dist, n_points_total, lmbda, w1, w2, mode, q,
            u_latent=None, u_w=None, focal_alpha=0.9):
    """The training objective (NVFPCC.py:154-196)."""
    out, cls, nbits, lbits = net_forward(P, emb, mode, q, u_latent, u_w)
    pyr = gt_pyramid(gt)
    b_latent = lbits.sum() / gt.sum()
    b_net = nbits.sum() / n_points_total
    loss = (surf_focal_dense(out, gt, dist, beta=1, alpha=focal_alpha)
            + focal_dense(cls[0], pyr[0], alpha=0.85)
            + focal_dense(cls[1], pyr[1], alpha=0.85)
            + lmbda * (b_latent * w1 + b_net * w2))
    return loss, out, cls, nbits, lbits


# --------------------------------------------------------------------------
# optimiser / schedule (NVFPCC.py:116-126, 253-254)
# --------------------------------------------------------------------------
def adam_update(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam step with default hyper-parameters, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def lr_at_epoch(base_lr, epoch):
    """Decoder LR at the start of ``epoch``.

    Both MultiStepLR([300,400,450], 0.1) schedulers are bound to the decoder
    optimiser (NVFPCC.py:117,126) and both step once per epoch (:253-254), so
    each milestone multiplies the decoder LR by 0.01; the latent LR never moves.
    """
    k = sum(epoch >= ms for ms in (300, 400, 450))
    return base_lr * (0.01 ** k)


class OracleTrainer:
    """CPU restatement of train()'s optimisation loop (NVFPCC.py:105-254): one mini-batch step (:149-223), one latent
    step per epoch (:225-251), the learning-rate schedule (:117,126,253-254).  Pinned by tests/golden/trajectory.npz
    (tests/test_oracle_golden.py).  ``noise_fn(step, ids, q) -> (u_latent, u_w)`` replaces torch.rand_like (``step``
    counts train-mode forwards from 1); None draws from torch's global RNG as the reference does."""

    def __init__(self, ch, channels, seed, n_leaf, n_points, lr=1e-3, wemb=5.0,
                 lmbda=200.0, w1=10.0, w2=57.0, noise_fn=None):
        self.P, _ = build_state(ch, channels, seed)
        self.keys = trainable_keys(self.P)
        for k in self.keys:
            self.P[k].requires_grad_(True)
        self.emb = torch.ones(n_leaf, ch, 2, 2, 2, requires_grad=True)   # NVFPCC.py:120-123
        self.base_lr = lr
        self.opt = torch.optim.Adam([self.P[k] for k in self.keys], lr=lr)
        self.opt_emb = torch.optim.Adam([self.emb], lr=lr * wemb)
        self.n_points = n_points
        self.h = dict(lmbda=lmbda, w1=w1, w2=w2)
        self.noise_fn = noise_fn
        self.forwards = 0

    def set_epoch(self, epoch):
        """What sch.step() + sch_emb.step() have done to the decoder LR by the start of ``epoch``."""
        for grp in self.opt.param_groups:
            grp["lr"] = lr_at_epoch(self.base_lr, epoch)

    def _noise(self, ids, q):
        self.forwards += 1
        if self.noise_fn is None:
            return {}
        u_latent, u_w = self.noise_fn(self.forwards, ids, q)
        return dict(u_latent=u_latent, u_w=u_w)

    def train_step(self, idx, gt, dist, q=1):
        self.opt.zero_grad()
        loss, *_ = rd_loss(self.P, self.emb[idx], gt, dist, self.n_points, mode="train", q=q, **self.h,
                           **self._noise(idx, q))
        loss.backward()
        self.opt.step()
        return float(loss.detach())

    def latent_step(self, gt, dist, q=1):
        self.opt_emb.zero_grad()
        loss, *_ = rd_loss(self.P, self.emb, gt, dist, self.n_points, mode="train", q=q, **self.h,
                           **self._noise(range(self.emb.shape[0]), q))
        loss.backward()
        self.opt_emb.step()
        return float(loss.detach())
