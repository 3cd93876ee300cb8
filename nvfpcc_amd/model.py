"""Model assembly (mirror of /root/reference/NVFPCC.py:32-88): latent_gen -> entropy_coder -> reconstructor."""
import numpy as np
import torch
from torch import nn

from . import ops
from .network import SingleLayerLatentGen, QuantGaussianLikelihood, CompDecoder


class Net(nn.Module):
    def __init__(self, args, param_model, ch=4, channel_str='8,16,8,8', verbose=True):
        super().__init__()
        if verbose:
            print(f'[Net] Building model with latent channel {ch} and channel string {channel_str}')
        channels = tuple(int(v) for v in str(channel_str).split(','))
        self.latent_gen = SingleLayerLatentGen(in_channels=ch, out_channels=ch)
        self.entropy_coder = QuantGaussianLikelihood(in_channels=ch)
        self.reconstructor = CompDecoder(args, param_model, useIGDN=True, in_channels=ch, channels=channels)

    def forward(self, emb, mode, q, block_ids=None, u_latent=None, u_w=None):
        latent = self.latent_gen(emb)
        latent_rounded, latent_likelihood = self.entropy_coder(latent, mode, u_latent, block_ids)
        out, out_cls_list, net_bits = self.reconstructor(latent_rounded, q, u_w)
        return out, out_cls_list, net_bits, latent_likelihood

    def reconstruct(self, latent, q):
        out, _, _ = self.reconstructor(latent, q)
        return out

    def get_network_bits(self):
        return self.entropy_coder.get_bits() + self.reconstructor.get_bits()

    def get_latent_bits(self, all_emb):
        _, like = self.entropy_coder(self.latent_gen(all_emb), mode='eval')
        return like.sum()

    def get_latent_code(self, all_emb):
        q_latent, like = self.entropy_coder(self.latent_gen(all_emb), mode='eval')
        return {'quantized_latent': q_latent, 'sigma': torch.abs(self.entropy_coder.sigma),
                'mu': self.entropy_coder.mu, 'latent_likelihood': like}

    def get_bits(self, all_emb):
        return self.get_latent_bits(all_emb), self.get_network_bits()


class MultiscaleProcessor(nn.Module):
    """[pool(pool(x)), pool(x), x] with 2x2x2 max pooling (NVFPCC.py:76-88)."""

    def forward(self, x):
        x1 = ops.maxpool2(x.contiguous())
        x2 = ops.maxpool2(x1)
        return [x2, x1, x]
