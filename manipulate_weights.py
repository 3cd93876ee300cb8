#!/usr/bin/env python3
"""Post-training weight quantiser with the reference's command line
(/root/reference/manipulate_weights.py; README step 3a):

    python manipulate_weights.py ckpts/0500.ckpt 0500_quantized_q4.ckpt 16

Rounds the seven trunk kernels to multiples of 1/iqp, copies the other coded parameters, and drops what the
decoder re-creates from its own seed file (`*_init` buffers of the reconstructor) or never needs
(conv0_cls / conv1_cls heads), exactly the key set of the reference (manipulate_weights.py:19-32)."""
import sys

import torch

LATENT_KEYS = ['latent_gen.h_analysis_2.kernel', 'latent_gen.h_analysis_2.b', 'latent_gen.h_analysis_2.kernel_init',
               'latent_gen.h_analysis_2.b_init', 'latent_gen.gdn_2.beta', 'latent_gen.gdn_2.gamma',
               'latent_gen.gdn_2.pedestal']
TRUNK = ['up0', 'conv0', 'up1', 'conv1', 'up2', 'conv2', 'conv2_cls']
OTHER_KEYS = (['entropy_coder.sigma', 'entropy_coder.mu', 'reconstructor.activation.beta',
               'reconstructor.activation.gamma', 'reconstructor.activation.pedestal']
              + [f'reconstructor.{n}.{p}' for n in TRUNK for p in ('kernel', 'b')]
              + ['reconstructor.likelihood_model.sigma', 'reconstructor.likelihood_model.mu'])


def quantise(state, iqp):
    out, lo, hi = {}, 100.0, -100.0
    with torch.no_grad():
        for k in LATENT_KEYS:
            out[k] = state[k].clone()
        for k in OTHER_KEYS:
            if k.endswith('.kernel'):
                steps = torch.round(state[k] * iqp)
                lo, hi = min(lo, steps.min().item()), max(hi, steps.max().item())
                out[k] = steps / iqp
            else:
                out[k] = state[k].clone()
    return out, lo, hi


if __name__ == '__main__':
    src, dst, iqp = sys.argv[1], sys.argv[2], int(sys.argv[3])
    q, lo, hi = quantise(torch.load(src, map_location=torch.device('cpu')), iqp)
    print(f'min: {lo}  max: {hi}')
    torch.save(q, dst)
