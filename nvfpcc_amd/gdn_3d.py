"""GDN3d / IGDN3d with the reference's constructor, parameter names and state-dict entries
(/root/reference/gdn_3d.py:32-159), computed by the fused gfx950 kernels nvf_gdn_fwd / nvf_gdn_bwd.

    y_c = x_c / sqrt(beta_c + sum_j gamma_cj x_j^2)      (GDN3d)
    y_c = x_c * sqrt(beta_c + sum_j gamma_cj x_j^2)      (IGDN3d)

with beta = max(beta_hat, sqrt(beta_min + 2^-36))^2 - 2^-36 and gamma = max(gamma_hat, 2^-18)^2 - 2^-36.
The kernels hard-wire the reference's defaults (beta_min 1e-6, reparam_offset 2^-18).
"""
import torch
from torch import nn

from . import functional as NF

_OFFSET = 2.0 ** -18
_PEDESTAL = _OFFSET ** 2


class _GdnBase(nn.Module):
    _inverse = False

    def __init__(self, ch, inverse=False, beta_min=1e-6, gamma_init=.1, reparam_offset=_OFFSET):
        super().__init__()
        if beta_min != 1e-6 or reparam_offset != _OFFSET:
            raise NotImplementedError("the HIP GDN kernels implement beta_min=1e-6, reparam_offset=2**-18 only")
        self.inverse = inverse           # kept for signature parity; the class decides the direction
        self.beta_min, self.gamma_init, self.reparam_offset = beta_min, gamma_init, reparam_offset
        self.pedestal_data = _PEDESTAL
        self.beta_bound = (beta_min + _PEDESTAL) ** .5
        self.gamma_bound = reparam_offset
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + _PEDESTAL))
        self.gamma = nn.Parameter(torch.sqrt(gamma_init * torch.eye(ch) + _PEDESTAL))
        self.register_buffer("pedestal", torch.FloatTensor([_PEDESTAL]))

    def forward(self, inputs):
        return NF.Gdn.apply(inputs, self.beta, self.gamma, self._inverse)


class GDN3d(_GdnBase):
    _inverse = False


class IGDN3d(_GdnBase):
    _inverse = True
