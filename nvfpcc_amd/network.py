"""Operator classes of the NVF codec with the reference's names, constructor arguments, forward
signatures and state-dict keys (live subset of /root/reference/utils/network.py), running on
hand-written gfx950 kernels.

    SingleLayerLatentGen   network.py:4592-4612     IConv3d 1x1x1 -> GDN3d
    QuantGaussianLikelihood network.py:4490-4552    round / noise / Gaussian rate of the latents
    CompDecoder            network.py:4648-4799     the conv decoder (live, second definition)
    QConvTranspose3d / QConv3d / IConv3d            network.py:564-742
    GaussianLikelihoodModel                         network.py:287-305

Construction happens on the CPU (seed-derived frozen buffers, network.py:377-400); forward needs
the modules on a HIP device -- there is no CPU compute path.
"""
import numpy as np
import torch
from torch import nn

from . import functional as NF
from . import seeds
from .gdn_3d import GDN3d, IGDN3d  # noqa: F401  (re-exported like the reference does, network.py:18)

# ---------------------------------------------------------------------------------------------
# seed stream: module-global like the reference's SEED2 / seed_ptr (network.py:20-22); it keeps
# advancing when a second network is built in the same process.
# ---------------------------------------------------------------------------------------------
SEED2 = None
seed_ptr = 0


def reset_seed(seed=None):
    """(Re)load the seed vector and rewind the cursor (the reference never rewinds; tests and the
    CLI call this once per process)."""
    global SEED2, seed_ptr
    SEED2 = seeds.load_seed() if seed is None else np.asarray(seed, np.float64).reshape(-1)
    seed_ptr = 0


def _seed_tail():
    if SEED2 is None:
        reset_seed()
    return SEED2[seed_ptr:]


def _advance(n):
    global seed_ptr
    seed_ptr += n


# ---------------------------------------------------------------------------------------------
# weight-noise bookkeeping for q == 1: every rank must draw the SAME weight noise, so it is a
# counter RNG keyed by (seed, step, layer) instead of the device generator.
# ---------------------------------------------------------------------------------------------
class NoiseState:
    seed = 0
    step = 0          # weight noise: advances on every q == 1 decoder forward (network.py:612,678 draw per call)
    latent_step = 0   # latent rate-proxy noise: advances on every mode='train' forward, whatever q (network.py:4516)


def set_noise_seed(seed, step=0, latent_step=None):
    NoiseState.seed, NoiseState.step = int(seed), int(step)
    NoiseState.latent_step = int(step if latent_step is None else latent_step)


def get_kaiming_init_from_seed(w, seed):
    """(seed - .5) * 2 * sqrt(6 / fan_in) with fan_in = w.size(1) * k^3 (network.py:377-400).
    For transposed convs size(1) is Cout: the reference's quirk, kept because it fixes the decoder."""
    fan_in = w.size(1) * (w[0][0].numel() if w.dim() > 2 else 1)
    bound = np.sqrt(3.0) * (np.sqrt(2.0) / np.sqrt(fan_in))
    return (seed - 0.5) * 2 * bound


class _SeededConv(nn.Module):
    """kernel/b start at zero; kernel_init/b_init are frozen seed-derived buffers (state-dict entries)."""

    def _build(self, shape, nbias, bias_fan, SEED, zero_bias=False):
        kernel = nn.Parameter(torch.zeros(shape))
        b = nn.Parameter(torch.zeros(nbias))
        n = kernel.numel()
        k_seed = torch.from_numpy(np.asarray(SEED[:n]).reshape(shape)).float()
        self.register_buffer("kernel_init", get_kaiming_init_from_seed(kernel, k_seed))
        b_seed = torch.from_numpy(np.asarray(SEED[n:n + nbias]).reshape(nbias)).float()
        b_init = torch.zeros_like(b_seed) if zero_bias else (b_seed - 0.5) * 2 * (1 / np.sqrt(bias_fan))
        self.register_buffer("b_init", b_init)
        self.register_parameter("kernel", kernel)
        self.register_parameter("b", b)
        self.offset = n + nbias
        self.layer_id = 0

    def _effective(self, q, u=None):
        sid = (NoiseState.step << 8) | self.layer_id
        return NF.EffectiveParams.apply(self.kernel, self.kernel_init, self.b, self.b_init, q, u, NoiseState.seed, sid)


class QConvTranspose3d(_SeededConv):
    """network.py:564-622.  Only the k=5, stride=2 geometry the decoder uses has a kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias, padding=0, output_padding=0, iQ=16,
                 SEED=None, groups=1, zero_bias=False):
        super().__init__()
        if kernel_size != 5 or stride != 2 or groups != 1 or not bias or iQ != 16:
            raise NotImplementedError("HIP path implements kernel_size=5, stride=2, groups=1, bias=True, iQ=16")
        if (padding, output_padding) not in ((0, 0), (2, 1)):
            raise NotImplementedError("HIP path implements (padding, output_padding) = (0,0) or (2,1)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.bias = kernel_size, stride, bias
        self.padding, self.output_padding, self.groups = padding, output_padding, groups
        self.Q = 1 / iQ
        self._build((in_channels, out_channels, 5, 5, 5), out_channels, in_channels, SEED, zero_bias)

    def forward(self, x, q, act=NF.ACT_NONE, u=None):
        w, b = self._effective(q, u)
        return NF.conv_transpose3d_k5s2(x, w, b, self.padding, act)


class QConv3d(_SeededConv):
    """network.py:624-688 (stride 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias, padding=0, iQ=16, SEED=None, groups=1,
                 zero_bias=False):
        super().__init__()
        if stride != 1 or groups != 1 or not bias or iQ != 16:
            raise NotImplementedError("HIP path implements stride=1, groups=1, bias=True, iQ=16")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.bias, self.padding, self.groups = kernel_size, stride, bias, padding, groups
        self.Q = 1 / iQ
        self._build((out_channels, in_channels, kernel_size, kernel_size, kernel_size), out_channels, in_channels,
                    SEED, zero_bias)

    def forward(self, x, q, act=NF.ACT_NONE, u=None):
        w, b = self._effective(q, u)
        return NF.conv3d(x, w, b, self.padding, act)


class IConv3d(_SeededConv):
    """network.py:690-742: never quantised (kernel + kernel_init)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias, padding=0, SEED=None, groups=1,
                 zero_bias=False):
        super().__init__()
        if stride != 1 or groups != 1 or not bias:
            raise NotImplementedError("HIP path implements stride=1, groups=1, bias=True")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.bias, self.padding, self.groups = kernel_size, stride, bias, padding, groups
        self._build((out_channels, in_channels, kernel_size, kernel_size, kernel_size), out_channels, in_channels,
                    SEED, zero_bias)

    def forward(self, x, act=NF.ACT_NONE):
        w, b = self._effective(0)
        return NF.conv3d(x, w, b, self.padding, act)


class GaussianLikelihoodModel(nn.Module):
    """network.py:287-305 with step_size = 1/16: rate of one 1/16-quantised kernel."""

    def __init__(self, step_size=1):
        super().__init__()
        if step_size != 1 / 16:
            raise NotImplementedError("HIP weight-rate kernel implements step_size = 1/16")
        self.sigma = nn.Parameter(torch.ones(1))
        self.mu = nn.Parameter(torch.zeros(1))
        self.step_size, self.half_step_size = step_size, step_size / 2

    def forward(self, x):
        # x is a kernel (rounded to 1/16 or not: the HIP kernel rounds, which is idempotent)
        return NF.WeightRate.apply(x, self.sigma, self.mu)


class QuantGaussianLikelihood(nn.Module):
    """Entropy model of the latents, signalled parameters (network.py:4490-4552)."""

    def __init__(self, in_channels, step_size=1, iQ=1, assume_zero_mean=False):
        super().__init__()
        if step_size != 1 or iQ != 1 or assume_zero_mean:
            raise NotImplementedError("HIP latent-rate kernel implements step_size=1, iQ=1, learned mean")
        self.Q = 1 / iQ
        self.assume_zero_mean = assume_zero_mean
        self.sigma = nn.Parameter(torch.ones(1, in_channels, 1, 1, 1))
        self.mu = nn.Parameter(torch.zeros(1, in_channels, 1, 1, 1))

    def forward(self, x, mode='train', u=None, block_ids=None):
        if mode not in ('train', 'eval'):
            raise ValueError(mode)
        if mode == 'train':
            NoiseState.latent_step += 1
        return NF.LatentRate.apply(x, self.sigma, self.mu, mode, u, block_ids, NoiseState.seed,
                                   NoiseState.latent_step)

    def get_bits(self):
        return int(np.prod(self.sigma.shape) * 32 + np.prod(self.mu.shape) * 32)


class SingleLayerLatentGen(nn.Module):
    """1x1x1 IConv3d followed by GDN3d (network.py:4592-4612)."""

    def __init__(self, in_channels=8, out_channels=4):
        super().__init__()
        self.h_analysis_2 = IConv3d(in_channels, out_channels, kernel_size=1, stride=1, bias=True, padding=0,
                                    SEED=_seed_tail())
        _advance(self.h_analysis_2.offset)
        self.gdn_2 = GDN3d(out_channels)

    def forward(self, x):
        return self.gdn_2(self.h_analysis_2(x))


class CompDecoder(nn.Module):
    """latent [B,ch,2,2,2] -> occupancy probability [B,1,32,32,32] plus two coarse heads
    (network.py:4648-4799).  ``args`` and ``param_model`` are accepted and ignored, as in the reference."""

    _order = ("up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls", "conv1_cls", "conv0_cls")

    def __init__(self, args, param_model, in_channels=4, useIGDN=False, channels=(8, 16, 8, 8)):
        super().__init__()
        if not useIGDN:
            raise NotImplementedError("the HIP decoder implements the useIGDN=True configuration NVFPCC.py uses")
        c = tuple(int(v) for v in channels)
        self.channels, self.useIGDN = c, useIGDN
        self.activation = IGDN3d(c[0])

        def add(name, mod):
            setattr(self, name, mod)
            _advance(mod.offset)
            mod.layer_id = self._order.index(name) + 1

        add("up0", QConvTranspose3d(in_channels, c[0], 5, 2, True, padding=2, output_padding=1, SEED=_seed_tail()))
        add("conv0", QConvTranspose3d(c[0], c[1], 5, 2, True, padding=2, output_padding=1, SEED=_seed_tail()))
        add("up1", QConvTranspose3d(c[1], c[2], 5, 2, True, SEED=_seed_tail()))
        add("conv1", QConv3d(c[2], c[2], 4, 1, True, padding=0, SEED=_seed_tail()))
        add("up2", QConvTranspose3d(c[2], c[3], 5, 2, True, SEED=_seed_tail()))
        add("conv2", QConv3d(c[3], c[3], 4, 1, True, padding=0, SEED=_seed_tail()))
        add("conv2_cls", QConv3d(c[3], 1, 3, 1, True, padding=1, SEED=_seed_tail()))
        add("conv1_cls", IConv3d(c[2], 1, 3, 1, True, padding=1, SEED=_seed_tail()))
        add("conv0_cls", IConv3d(c[1], 1, 3, 1, True, padding=1, SEED=_seed_tail()))
        self.likelihood_model = GaussianLikelihoodModel(step_size=1 / 16)

    def forward(self, x, q, u_w=None):
        """Returns (out, [cls0, cls1, out], net_bits[7]).  ``u_w``: optional {layer: uniform sample}
        replacing the counter RNG for the q=1 weight noise (tests)."""
        u_w = u_w or {}
        R, S = NF.ACT_RELU, NF.ACT_SIGMOID
        if q == 1:
            NoiseState.step += 1
        t = self.activation(self.up0(x, q, u=u_w.get("up0")))
        t = self.conv0(t, q, R, u_w.get("conv0"))
        cls0 = self.conv0_cls(t, S)
        t = self.up1(t, q, R, u_w.get("up1"))
        t = self.conv1(t, q, R, u_w.get("conv1"))
        cls1 = self.conv1_cls(t, S)
        t = self.up2(t, q, R, u_w.get("up2"))
        t = self.conv2(t, q, R, u_w.get("conv2"))
        out = self.conv2_cls(t, q, S, u_w.get("conv2_cls"))
        net_bits = torch.stack([self.likelihood_model(p) for p in self.get_q_params()])
        return out, [cls0, cls1, out], net_bits

    def get_q_params(self):
        return [getattr(self, n).kernel for n in self._order[:7]]

    def get_bits(self):
        net_bits = torch.stack([self.likelihood_model(p) for p in self.get_q_params()])
        c = self.channels
        aux_bits = sum(c[i] * 2 for i in (1, 2, 3)) * 32 + 32 + (c[1] ** 2 + c[1]) * 32
        return net_bits.sum().item() + aux_bits
