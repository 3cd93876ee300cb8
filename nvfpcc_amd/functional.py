"""torch.autograd glue: each Function's forward and backward are hand-written gfx950 kernels
(nvfpcc_amd.ops -> libnvf_hip.so); autograd only strings them together, so the reference's own
call pattern (``net(emb, mode, q)`` ... ``loss.backward()``, NVFPCC.py:160-197) keeps working.
"""
import torch
from torch.autograd import Function

from . import ops

ACT_NONE, ACT_RELU, ACT_SIGMOID = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_SIGMOID


def _act_bwd(gy, y, act):
    gy = gy.contiguous()
    if act == ACT_RELU:
        return ops.relu_bwd(gy, y)
    if act == ACT_SIGMOID:
        return ops.sigmoid_bwd(gy, y)
    return gy


class EffectiveParams(Function):
    """w_eff = f_q(kernel) + kernel_init, b_eff = b + b_init; straight-through gradients
    (utils/network.py:611-620, 677-686, 735-740 of the reference)."""

    @staticmethod
    def forward(ctx, kernel, kernel_init, b, b_init, q, u, seed, stream_id):
        w, be = ops.effective_params(kernel.contiguous(), kernel_init, b.contiguous(), b_init, q, u, seed, stream_id)
        return w, be

    @staticmethod
    def backward(ctx, gw, gb):
        return gw, None, gb, None, None, None, None, None


class Conv3d(Function):
    """act(F.conv3d(x, w, b, 1, pad)) -- utils/network.py:687, 741."""

    @staticmethod
    def forward(ctx, x, w, b, pad, act):
        x = x.contiguous()
        cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
        wf, wb = ops.pack_conv_weight(w.contiguous())
        osz = tuple(s + 2 * pad - k + 1 for s in x.shape[2:])
        # the trunk's 4^3 convolutions run on the matrix cores (same kernels as the step engine; the choice depends on
        # the layer shape only, never on the batch: encode at any batch == decode at batch 1)
        if k == 4 and pad == 0 and cin == 8 and cout == 8 and x.shape[-1] in (19, 35):
            y = ops.conv3d_k4_mfma(x, ops.pack_mfma_k4(wf, 8, 0), b, 0, 0, act)
        elif k == 4 and pad == 0 and cin == 16 and cout == 16 and x.shape[-1] in (19, 35):
            y = ops.conv3d_g16_mfma(x, ops.pack_g16_mfma(wf, 16, 16, 4), b, 16, 4, 1, 0, osz, act)
        else:
            y = ops.conv3d_gather(x, wf, b, cout, k, 1, pad, osz, act)
        ctx.save_for_backward(x, y, wb)
        ctx.cfg = (cin, cout, k, pad, act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, wb = ctx.saved_tensors
        cin, cout, k, pad, act = ctx.cfg
        gp = _act_bwd(gy, y, act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d_gather(gp, wb, None, cin, k, 1, k - 1 - pad, tuple(x.shape[2:]))
        if ctx.needs_input_grad[1]:
            dw = ops.wgrad(gp, x, k, 1, pad, out_mode=0)
        if ctx.needs_input_grad[2]:
            db = ops.channel_sum(gp)
        return dx, dw, db, None, None


class ConvTranspose3dK5S2(Function):
    """act(F.conv_transpose3d(x, w, b, 2, pad, output_padding)) for k=5 -- utils/network.py:621."""

    @staticmethod
    def forward(ctx, x, w, b, pad, act):
        x = x.contiguous()
        cin, cout = w.shape[0], w.shape[1]
        wf, wb = ops.pack_convT_weight(w.contiguous())
        if pad == 0 and cout == 8 and (cin, x.shape[-1]) in ((16, 8), (8, 16)):        # up1 / up2, narrow decoder
            y = ops.convT3d_k5s2_mfma(x, ops.pack_convT_mfma(wf, cin), b, act)
        elif pad == 0 and cout == 16 and (cin, x.shape[-1]) in ((32, 8), (16, 16)):    # up1 / up2, wide decoder
            y = ops.convT3d_k5s2_mfma16(x, ops.pack_convT16_mfma(wf, cin, 16), b, act)
        else:
            y = ops.convT3d_k5s2_fwd(x, wf, b, cout, pad, act)
        ctx.save_for_backward(x, y, wb)
        ctx.cfg = (cin, cout, pad, act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, wb = ctx.saved_tensors
        cin, cout, pad, act = ctx.cfg
        gp = _act_bwd(gy, y, act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d_gather(gp, wb, None, cin, 5, 2, pad, tuple(x.shape[2:]))
        if ctx.needs_input_grad[1]:
            dw = ops.wgrad(x, gp, 5, 2, pad, out_mode=0)
        if ctx.needs_input_grad[2]:
            db = ops.channel_sum(gp)
        return dx, dw, db, None, None


class Gdn(Function):
    """GDN3d / IGDN3d with their re-parametrised beta/gamma (gdn_3d.py:72-95, 137-159)."""

    @staticmethod
    def forward(ctx, x, beta_hat, gamma_hat, inverse):
        x = x.contiguous()
        ctx.save_for_backward(x, beta_hat, gamma_hat)
        ctx.inverse = inverse
        return ops.gdn_fwd(x, beta_hat.contiguous(), gamma_hat.contiguous(), inverse)

    @staticmethod
    def backward(ctx, gy):
        x, beta_hat, gamma_hat = ctx.saved_tensors
        dx, db, dg = ops.gdn_bwd(x, beta_hat.contiguous(), gamma_hat.contiguous(), gy.contiguous(), ctx.inverse)
        return dx, db, dg, None


class LatentRate(Function):
    """round + noise + Gaussian rate of the latents (utils/network.py:4514-4539)."""

    @staticmethod
    def forward(ctx, x, sigma, mu, mode, u, block_ids, seed, step):
        x = x.contiguous()
        s, m = sigma.reshape(-1).contiguous(), mu.reshape(-1).contiguous()
        xr, bits, _, _, _ = ops.latent_rate(x, s, m, mode, u=u, block_ids=block_ids, seed=seed, step=step)
        ctx.save_for_backward(x, s, m, u, block_ids)
        ctx.cfg = (mode, seed, step, sigma.shape, mu.shape)
        return xr, bits.reshape(())

    @staticmethod
    def backward(ctx, g_rounded, g_bits):
        x, s, m, u, block_ids = ctx.saved_tensors
        mode, seed, step, sshape, mshape = ctx.cfg
        _, _, dx, ds, dm = ops.latent_rate(x, s, m, mode, u=u, block_ids=block_ids, want_grad=True,
                                           g_dev=g_bits.reshape(1).contiguous(), seed=seed, step=step,
                                           dx_addend=g_rounded.contiguous())
        return dx, ds.reshape(sshape), dm.reshape(mshape), None, None, None, None, None


class WeightRate(Function):
    """bits of one 1/16-quantised kernel under N(mu, |sigma|) (utils/network.py:301-305, 4777-4778)."""

    @staticmethod
    def forward(ctx, kernel, sigma, mu):
        kernel = kernel.contiguous()
        ctx.save_for_backward(kernel, sigma, mu)
        return ops.weight_rate(kernel, sigma.contiguous(), mu.contiguous()).reshape(())

    @staticmethod
    def backward(ctx, g):
        kernel, sigma, mu = ctx.saved_tensors
        dk = torch.empty_like(kernel)
        ds = torch.empty_like(sigma)
        dm = torch.empty_like(mu)
        ops.weight_rate(kernel, sigma.contiguous(), mu.contiguous(), dk=dk, dsigma=ds, dmu=dm,
                        g_dev=g.reshape(1).contiguous())
        return dk, ds, dm


class FocalLoss(Function):
    """get_focal_dense / get_surf_focal_dense (utils/loss.py:61-72, 94-111), SUM reduction."""

    @staticmethod
    def forward(ctx, p, gt, dist, alpha, beta):
        p, gt = p.contiguous(), gt.contiguous()
        dist = dist.contiguous() if dist is not None else None
        loss, _ = ops.focal_loss(p, gt, dist, alpha, beta)
        ctx.save_for_backward(p, gt, dist)
        ctx.cfg = (alpha, beta)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        p, gt, dist = ctx.saved_tensors
        alpha, beta = ctx.cfg
        _, dp = ops.focal_loss(p, gt, dist, alpha, beta, want_grad=True, g_dev=g.reshape(1).contiguous())
        return dp, None, None, None, None


class GatherRows(Function):
    """emb[indices] (NVFPCC.py:158) with its scatter-add backward."""

    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.shape = src.shape
        return ops.gather_rows(src.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, device=g.device)
        ops.scatter_add_rows(g.contiguous(), idx, out)
        return out, None


def conv3d(x, w, b, pad, act=ACT_NONE):
    return Conv3d.apply(x, w, b, pad, act)


def conv_transpose3d_k5s2(x, w, b, pad, act=ACT_NONE):
    return ConvTranspose3dK5S2.apply(x, w, b, pad, act)
