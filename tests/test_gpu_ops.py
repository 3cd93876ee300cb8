"""GPU parity of every C-ABI kernel against the oracle (CPU aten ops = what the reference runs).

Tolerances (fp32, different but fixed summation order): forward activations <= 1e-5 abs on
O(1) values; gradients <= 1e-4 relative to the tensor's max magnitude.  Tiled kernels must equal
the one-thread-per-output kernels bit for bit (same fmaf chain).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nvf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from nvfpcc_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda().contiguous()


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(b.abs().max().item(), 1e-12))


def gen(seed):
    return torch.Generator().manual_seed(seed)


# (cin, cout, k, pad, spatial_in) -- every stride-1 conv of both decoder configs
CONV_CASES = [
    (8, 8, 4, 0, 35), (8, 8, 4, 0, 19), (8, 1, 3, 1, 32), (8, 1, 3, 1, 16), (16, 1, 3, 1, 8),
    (16, 16, 4, 0, 35), (16, 16, 4, 0, 19), (16, 1, 3, 1, 32), (16, 1, 3, 1, 16), (32, 1, 3, 1, 8),
    (3, 3, 1, 0, 2), (8, 8, 1, 0, 2),
]


@pytest.mark.parametrize("cin,cout,k,pad,n", CONV_CASES)
def test_conv3d_forward_and_backward_data(ops, cin, cout, k, pad, n):
    g = gen(cin * 1000 + cout * 10 + k)
    B = 2
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, generator=g)
    x.requires_grad_(True)
    y_ref = F.conv3d(x, w, b, 1, pad)
    no = y_ref.shape[-1]
    wf, wb = ops.pack_conv_weight(dev(w))
    for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, F.relu), (ops.ACT_SIGMOID, torch.sigmoid)):
        ops.set_naive(False)
        y = ops.conv3d_gather(dev(x.detach()), wf, dev(b), cout, k, 1, pad, (no, no, no), act)
        ops.set_naive(True)
        y_naive = ops.conv3d_gather(dev(x.detach()), wf, dev(b), cout, k, 1, pad, (no, no, no), act)
        ops.set_naive(False)
        assert torch.equal(y, y_naive), "tiled and naive kernels must agree bit for bit"
        assert (y.cpu() - fn(y_ref.detach())).abs().max() < 2e-5
    # backward-data: dX = gather-conv of dY with flipped/transposed weights, pad' = k-1-pad
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    add = torch.randn(x.shape, generator=g)
    dx = ops.conv3d_gather(dev(gy), wb, None, cin, k, 1, k - 1 - pad, (n, n, n))
    assert rel_err(dx, x.grad) < 1e-5
    dx2 = ops.conv3d_gather(dev(gy), wb, None, cin, k, 1, k - 1 - pad, (n, n, n), addend=dev(add), mask=dev(mask))
    ops.set_naive(True)
    dx2n = ops.conv3d_gather(dev(gy), wb, None, cin, k, 1, k - 1 - pad, (n, n, n), addend=dev(add), mask=dev(mask))
    ops.set_naive(False)
    assert torch.equal(dx2, dx2n)
    assert rel_err(dx2, (x.grad + add) * (mask > 0)) < 1e-5


# (cin, cout, pad, outpad, spatial_in) -- every transposed conv of both configs
CONVT_CASES = [(3, 8, 2, 1, 2), (8, 16, 2, 1, 4), (16, 8, 0, 0, 8), (8, 8, 0, 0, 16),
               (8, 16, 2, 1, 2), (16, 32, 2, 1, 4), (32, 16, 0, 0, 8), (16, 16, 0, 0, 16)]


@pytest.mark.parametrize("cin,n,B", [(8, 16, 2), (16, 8, 3), (8, 16, 1)])
def test_conv_transpose_k5s2_mfma_forward(ops, cin, n, B):
    """Matrix-core form of up1 / up2 (padding 0, 8 output channels) against torch's conv_transpose3d; batch
    invariance bit for bit."""
    g = gen(5000 + cin + n + B)
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cin, 8, 5, 5, 5, generator=g) / (cin * 125 / 8) ** 0.5
    b = torch.randn(8, generator=g)
    y_ref = F.conv_transpose3d(x, w, b, stride=2)
    wf, _ = ops.pack_convT_weight(dev(w))
    wp = ops.pack_convT_mfma(wf, cin)
    xd = dev(x)
    y = ops.convT3d_k5s2_mfma(xd, wp, dev(b), ops.ACT_RELU)
    assert y.shape == y_ref.shape
    assert (y.cpu() - F.relu(y_ref)).abs().max() < 2e-5
    y_lin = ops.convT3d_k5s2_mfma(xd, wp, dev(b), ops.ACT_NONE)
    assert (y_lin.cpu() - y_ref).abs().max() < 2e-5
    for i in range(B):
        yi = ops.convT3d_k5s2_mfma(xd[i:i + 1].contiguous(), wp, dev(b), ops.ACT_RELU)
        assert torch.equal(yi[0], y[i])


@pytest.mark.parametrize("cin,n,B", [(8, 16, 1), (8, 16, 5), (8, 16, 16), (16, 8, 1), (16, 8, 5), (16, 8, 16)])
def test_conv_transpose_k5s2_mfma_forward_training_form(ops, cin, n, B):
    """The training-step form of up1 / up2 (variant 15, pack kind 12: the kx = 4 taps of the even x outputs run on rows
    (co, ey) and are added at the end of their sums) against torch's float64 conv_transpose3d.  Odd x outputs have no kx = 4
    tap: they are the BITS of the evaluation form (variant 5); even ones differ by a re-association only.  Batch invariant."""
    g = gen(5100 + cin + n + B)
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cin, 8, 5, 5, 5, generator=g) / (cin * 125 / 8) ** 0.5
    b = torch.randn(8, generator=g)
    y_ref = F.conv_transpose3d(x.double(), w.double(), b.double(), stride=2)
    wf, _ = ops.pack_convT_weight(dev(w))
    wp, wpr = ops.pack_convT_mfma(wf, cin), ops.pack_convT_mfma(wf, cin, edge_rows=True)
    xd = dev(x)
    y5 = ops.convT3d_k5s2_mfma(xd, wp, dev(b), ops.ACT_NONE, variant=5)
    y15 = ops.convT3d_k5s2_mfma(xd, wpr, dev(b), ops.ACT_NONE, variant=15)
    scale = y_ref.abs().max().item()
    e5, e15 = (y5.cpu().double() - y_ref).abs().max().item() / scale, (y15.cpu().double() - y_ref).abs().max().item() / scale
    print(f"convT training form cin={cin} B={B}: max err / max vs float64: evaluation form {e5:.2e}, training form {e15:.2e}")
    assert e15 < 1e-6 and e15 < 2 * e5 + 1e-7
    assert torch.equal(y15[..., 1::2], y5[..., 1::2])
    assert (y15[..., 0::2] - y5[..., 0::2]).abs().max().item() / scale < 1e-6
    yr = ops.convT3d_k5s2_mfma(xd, wpr, dev(b), ops.ACT_RELU, variant=15)
    assert torch.equal(yr, torch.relu(y15))
    for i in (0, B - 1):
        yi = ops.convT3d_k5s2_mfma(xd[i:i + 1].contiguous(), wpr, dev(b), ops.ACT_NONE, variant=15)
        assert torch.equal(yi[0], y15[i])


@pytest.mark.parametrize("cin,n,B", [(8, 16, 2), (16, 8, 3)])
def test_conv_transpose_k5s2_mfma_backward_data(ops, cin, n, B):
    """Matrix-core backward-data of up1 / up2 against torch's autograd (with the fused addend and ReLU mask)."""
    g = gen(6000 + cin + n + B)
    x = torch.randn(B, cin, n, n, n, generator=g, requires_grad=True)
    w = torch.randn(cin, 8, 5, 5, 5, generator=g) / (cin * 125 / 8) ** 0.5
    y = F.conv_transpose3d(x, w, None, stride=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    add = torch.randn(x.shape, generator=g)
    _, wb = ops.pack_convT_weight(dev(w))
    wp = ops.pack_s2k5_mfma(wb, 8, cin)
    dx = ops.conv3d_s2k5_mfma(dev(gy), wp, cin)
    assert rel_err(dx, x.grad) < 1e-5
    dx2 = ops.conv3d_s2k5_mfma(dev(gy), wp, cin, addend=dev(add), mask=dev(mask))
    assert rel_err(dx2, (x.grad + add) * (mask > 0)) < 1e-5
    one = ops.conv3d_s2k5_mfma(dev(gy)[1:2].contiguous(), wp, cin)
    assert torch.equal(one[0], dx[1])


@pytest.mark.parametrize("B", [1, 5, 16])
def test_up1_backward_data_with_channel_groups_on_different_waves(ops, B):
    """Variant 7 of nvf_conv3d_s2k5_mfma (up1's backward-data in training steps: the two channel groups of g on different
    waves, their sums added through LDS) against torch's float64 autograd, with the fused addend and ReLU mask; against the
    fixed-order kernel (variant 0) it differs by that one re-association only.  Batch invariant."""
    g = gen(6100 + B)
    x = torch.randn(B, 16, 8, 8, 8, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(16, 8, 5, 5, 5, generator=g) / (16 * 125 / 8) ** 0.5
    y = F.conv_transpose3d(x, w.double(), None, stride=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    mask, add = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
    _, wb = ops.pack_convT_weight(dev(w))
    wp = ops.pack_s2k5_mfma(wb, 8, 16)
    ref = ((x.grad + add.double()) * (mask > 0)).float()
    d0 = ops.conv3d_s2k5_mfma(dev(gy), wp, 16, addend=dev(add), mask=dev(mask), variant=0)
    d7 = ops.conv3d_s2k5_mfma(dev(gy), wp, 16, addend=dev(add), mask=dev(mask), variant=7)
    e0, e7 = rel_err(d0, ref), rel_err(d7, ref)
    print(f"up1 backward-data B={B}: rel err vs float64: fixed order {e0:.2e}, channel groups on different waves {e7:.2e}")
    assert e7 < 3e-6 and e7 < 2 * e0 + 1e-7
    assert rel_err(d7, d0.cpu()) < 3e-6
    plain = ops.conv3d_s2k5_mfma(dev(gy), wp, 16, variant=7)
    assert rel_err(plain, x.grad.float()) < 3e-6
    one = ops.conv3d_s2k5_mfma(dev(gy)[B - 1:B].contiguous(), wp, 16, addend=dev(add)[B - 1:B].contiguous(),
                               mask=dev(mask)[B - 1:B].contiguous(), variant=7)
    assert torch.equal(one[0], d7[B - 1])


HEAD_TUPLES = {"narrow": [(16, 8), (8, 16), (8, 32)], "wide": [(32, 8), (16, 16), (16, 32)]}


@pytest.mark.parametrize("which", ["narrow", "wide"])
def test_three_head_launches_equal_the_single_head_kernels(ops, which):
    """nvf_heads3_* run the same kernel bodies as the per-head calls: forward and backward-data must agree bit for bit
    -- for the heads of both decoders (chanstr 8,16,8,8 and 16,32,16,16).  The weight gradients run on the matrix cores
    (heads_wgrad_mfma.hip: positions are the K index, another summation order; the wide decoder's 32-channel head as
    two groups of 16 rows) and are held to the fp64 sum and to the VALU kernels to rounding."""
    g = gen(7000)
    B = 3
    shapes = HEAD_TUPLES[which]
    xs = [dev(torch.randn(B, c, s, s, s, generator=g)) for c, s in shapes]
    ws = [torch.randn(1, c, 3, 3, 3, generator=g) * 0.1 for c, s in shapes]
    bs = [dev(torch.randn(1, generator=g)) for _ in shapes]
    packed = [ops.pack_conv_weight(dev(w)) for w in ws]
    ps = ops.heads3_fwd(xs, [p[0] for p in packed], bs)
    for x, (wf, wb), b, p, (c, s) in zip(xs, packed, bs, ps, shapes):
        assert torch.equal(p, ops.conv3d_gather(x, wf, b, 1, 3, 1, 1, (s, s, s), ops.ACT_SIGMOID))
    dls = [dev(torch.randn(B, 1, s, s, s, generator=g)) for c, s in shapes]
    masks = [None, None, xs[2]]
    dxs = ops.heads3_bwd_data(dls, [p[1] for p in packed], [c for c, s in shapes], masks)
    for dl, (wf, wb), dx, m, (c, s) in zip(dls, packed, dxs, masks, shapes):
        assert torch.equal(dx, ops.conv3d_gather(dl, wb, None, c, 3, 1, 1, (s, s, s), mask=m))
    outs = [torch.empty(1, c, 3, 3, 3, device=xs[0].device) for c, s in shapes]
    wg = ops.WgradBatch(xs[0].device, nbytes=64 << 20)
    wg.add_heads3(dls, xs, outs)
    wg.finish()
    for dl, x, o, (c, s) in zip(dls, xs, outs, shapes):
        ref = ops.wgrad(dl, x, 3, 1, 1, out_mode=0)
        if True:
            xp = torch.nn.functional.pad(x.double().cpu(), (1, 1, 1, 1, 1, 1))
            d64 = dl.double().cpu()
            want = torch.stack([(xp[:, :, kz:kz + s, ky:ky + s, kx:kx + s] * d64).sum(dim=(0, 2, 3, 4))
                                for kz in range(3) for ky in range(3) for kx in range(3)], 1).reshape(1, c, 3, 3, 3)
            scale = want.abs().max().item()
            assert (o.double().cpu() - want).abs().max().item() < 2e-6 * scale * np.sqrt(B * s ** 3 / 512.0)
            assert (o - ref).abs().max().item() < 1e-5 * scale


def test_three_mfma_weight_gradients_in_one_launch(ops):
    """nvf_wgrad_mfma3_partial (conv2, up2, conv1 of the narrow trunk) equals three nvf_wgrad calls bit for bit."""
    g = gen(7100)
    B = 2
    R = lambda *s: dev(torch.randn(*s, generator=g))
    g5, y4 = R(B, 8, 32, 32, 32), R(B, 8, 35, 35, 35)
    y3, g4 = R(B, 8, 16, 16, 16), R(B, 8, 35, 35, 35)
    g3, y2 = R(B, 8, 16, 16, 16), R(B, 8, 19, 19, 19)
    outs = [torch.empty(8, 8, k, k, k, device=g5.device) for k in (4, 5, 4)]
    wg = ops.WgradBatch(g5.device, nbytes=128 << 20)
    wg.add_mfma3([g5, y3, g3], [y4, g4, y2], outs)
    wg.finish()
    # conv2's gradient runs in the Winograd (y, x) form inside the merged launches (wgrad_wino.h): another arithmetic,
    # agreement to rounding (a context with set_direct restores the direct form and bit-equality, below)
    ref0 = ops.wgrad(g5, y4, 4, 1, 0, out_mode=0)
    assert (outs[0] - ref0).abs().max().item() < 2e-5 * ref0.abs().max().item()
    assert torch.equal(outs[1], ops.wgrad(y3, g4, 5, 2, 0, out_mode=0))
    ref2 = ops.wgrad(g3, y2, 4, 1, 0, out_mode=0)                  # conv1: direct form by default
    assert torch.equal(outs[2], ref2)
    # the forms are the CALLER's choice, carried by its context (nvf_step_ctx_set_direct / _set_wgrad_forms): the library
    # reads no environment variable.  direct: all three bit-equal to the per-layer kernels; (z split 2, conv1 Winograd):
    # both 4^3 gradients to rounding, up2 untouched
    cd = ops.StepCtx()
    cd.set_direct(True)
    outs_d = [torch.empty_like(o) for o in outs]
    wgd = ops.WgradBatch(g5.device, nbytes=128 << 20, ctx=cd)
    wgd.add_mfma3([g5, y3, g3], [y4, g4, y2], outs_d)
    wgd.finish()
    assert torch.equal(outs_d[0], ref0) and torch.equal(outs_d[1], outs[1]) and torch.equal(outs_d[2], ref2)
    cw = ops.StepCtx()
    cw.set_wgrad_forms(2, True)
    outs_w = [torch.empty_like(o) for o in outs]
    wgw = ops.WgradBatch(g5.device, nbytes=128 << 20, ctx=cw)
    wgw.add_mfma3([g5, y3, g3], [y4, g4, y2], outs_w)
    wgw.finish()
    assert (outs_w[0] - ref0).abs().max().item() < 2e-5 * ref0.abs().max().item()
    assert torch.equal(outs_w[1], outs[1])
    assert not torch.equal(outs_w[2], ref2) and (outs_w[2] - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()
    with pytest.raises(RuntimeError):
        cw.set_wgrad_forms(9, False)
    # ... and so do up1 + conv0
    y1, g2 = R(B, 16, 8, 8, 8), R(B, 8, 19, 19, 19)
    h0, g1 = R(B, 8, 4, 4, 4), R(B, 16, 8, 8, 8)
    outs2 = [torch.empty(16, 8, 5, 5, 5, device=g5.device), torch.empty(8, 16, 5, 5, 5, device=g5.device)]
    wg.add_up1_conv0([y1, h0], [g2, g1], outs2)
    wg.finish()
    assert torch.equal(outs2[0], ops.wgrad(y1, g2, 5, 2, 0, out_mode=0))
    assert torch.equal(outs2[1], ops.wgrad(h0, g1, 5, 2, 2, out_mode=0))
    # ... and all five in one launch (nvf_wgrad_trunk5_partial: the two VALU jobs run as 256-thread workgroups)
    outs5 = [torch.empty_like(o) for o in outs + outs2]
    wg.add_trunk5([g5, y3, g3, y1, h0], [y4, g4, y2, g2, g1], outs5)
    wg.finish()
    for i, (got, ref) in enumerate(zip(outs5, outs + outs2)):
        if i in (0, 2, 3):  # conv2, conv1 (Winograd form: the same arithmetic in both launches, but other work items per
            # workgroup) and up1 (on the matrix cores inside the five-gradient launch): another summation order
            assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
        else:
            assert torch.equal(got, ref)


@pytest.mark.parametrize("mode,c", [("train", 3), ("eval", 8)])
def test_latent_fwd_equals_the_three_kernels(ops, mode, c):
    """nvf_latent_fwd = conv1x1 + GDN + quantiser/rate: every output bit-identical to the separate launches."""
    g = gen(7200 + c)
    B = 5
    e = dev(1.0 + 0.5 * torch.randn(B, c, 2, 2, 2, generator=g))
    w = torch.randn(c, c, 1, 1, 1, generator=g) * 0.5
    b = dev(torch.randn(c, generator=g) * 0.1)
    beta = dev(1.0 + 0.1 * torch.rand(c, generator=g))
    gamma = dev(0.3 * torch.rand(c, c, generator=g))
    sigma, mu = dev(0.5 + torch.rand(c, generator=g)), dev(torch.randn(c, generator=g) * 0.1)
    wf, _ = ops.pack_conv_weight(dev(w))
    ids = dev(torch.tensor([7, 3, 11, 0, 5]))
    h, lat, xr, bits = ops.latent_fwd(e, wf, b, beta, gamma, sigma, mu, mode, block_ids=ids, seed=9, step=4)
    h2 = ops.conv3d_gather(e, wf, b, c, 1, 1, 0, (2, 2, 2))
    lat2 = ops.gdn_fwd(h2, beta, gamma, False)
    xr2, bits2, _, _, _ = ops.latent_rate(lat2, sigma, mu, mode, block_ids=ids, seed=9, step=4)
    assert torch.equal(h, h2) and torch.equal(lat, lat2) and torch.equal(xr, xr2) and torch.equal(bits, bits2)


# matrix-core (MFMA) form of the 4^3, 8 -> 8 channel convolutions: (spatial_in, batch)
@pytest.mark.parametrize("n,B", [(35, 2), (19, 3), (35, 1), (19, 5)])
def test_conv3d_k4_mfma_forward_and_backward_data(ops, n, B):
    """Forward (pair axis x) and backward-data (pair axis z) against torch's conv3d / its autograd: the MFMA is an
    exact fp32 fmaf chain in its own fixed order, so the tolerance is the one of the VALU kernels; every variant
    must also be invariant to the batch it runs in, bit for bit (encode at any batch == decode at batch 1)."""
    g = gen(4000 + n + B)
    x = torch.randn(B, 8, n, n, n, generator=g)
    w = torch.randn(8, 8, 4, 4, 4, generator=g) / 512 ** 0.5
    b = torch.randn(8, generator=g)
    x.requires_grad_(True)
    y_ref = F.conv3d(x, w, b)
    wf, wb = ops.pack_conv_weight(dev(w))
    wpf, wpb = ops.pack_mfma_k4(wf, 8, 0), ops.pack_mfma_k4(wb, 8, 2)
    xd = dev(x.detach())
    y = ops.conv3d_k4_mfma(xd, wpf, dev(b), 0, 0, ops.ACT_RELU)
    assert (y.cpu() - F.relu(y_ref.detach())).abs().max() < 2e-5
    y_lin = ops.conv3d_k4_mfma(xd, wpf, dev(b), 0, 0, ops.ACT_NONE)          # generic epilogue
    assert (y_lin.cpu() - y_ref.detach()).abs().max() < 2e-5
    for i in range(B):                                                       # batch invariance, bit for bit
        yi = ops.conv3d_k4_mfma(xd[i:i + 1].contiguous(), wpf, dev(b), 0, 0, ops.ACT_RELU)
        assert torch.equal(yi[0], y[i])
    for var in ((2, 3, 4, 5, 6) if n == 35 else (2, 3)):                     # every tile shape: the same bits
        assert torch.equal(ops.conv3d_k4_mfma(xd, wpf, dev(b), 0, 0, ops.ACT_RELU, variant=var), y), var
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    add = torch.randn(x.shape, generator=g)
    dx = ops.conv3d_k4_mfma(dev(gy), wpb, None, 3, 2, ops.ACT_NONE, mask=dev(mask))
    assert rel_err(dx, x.grad * (mask > 0)) < 1e-5
    dx2 = ops.conv3d_k4_mfma(dev(gy), wpb, None, 3, 2, ops.ACT_NONE, addend=dev(add), mask=dev(mask))
    assert rel_err(dx2, (x.grad + add) * (mask > 0)) < 1e-5
    dx3 = ops.conv3d_k4_mfma(dev(gy), wpb, None, 3, 2, ops.ACT_NONE)
    assert rel_err(dx3, x.grad) < 1e-5
    # the x-pair mapping with flattened 18- / 10-cell rows (weights packed with pair axis 0 from w_bwd)
    wpbx = ops.pack_mfma_k4(wb, 8, 0)
    dx4 = ops.conv3d_k4_mfma(dev(gy), wpbx, None, 3, 0, ops.ACT_NONE, mask=dev(mask))
    assert rel_err(dx4, x.grad * (mask > 0)) < 1e-5
    dx5 = ops.conv3d_k4_mfma(dev(gy), wpbx, None, 3, 0, ops.ACT_NONE, addend=dev(add))
    assert rel_err(dx5, x.grad + add) < 1e-5
    # the multi-pack launch produces the same fragments
    wpf2, wpb2 = torch.empty_like(wpf), torch.empty_like(wpb)
    ops.pack_mfma_k4_multi([(wf, 8, 0, wpf2), (wb, 8, 2, wpb2)])
    assert torch.equal(wpf, wpf2) and torch.equal(wpb, wpb2)
    # ... and so does the all-kinds launch the step engine uses (here with a transposed-conv weight as well)
    wt = torch.randn(16, 8, 5, 5, 5, generator=g) * 0.05
    wtf, wtb = ops.pack_convT_weight(dev(wt))
    ref_t, ref_s = ops.pack_convT_mfma(wtf, 16), ops.pack_s2k5_mfma(wtb, 8, 16)
    outs = [torch.empty_like(t) for t in (wpf, wpb, ref_t, ref_s)]
    ops.pack_mfma_all([(wf, outs[0], 0, 8, 8), (wb, outs[1], 2, 8, 8), (wtf, outs[2], 10, 16, 8), (wtb, outs[3], 20, 8, 16)])
    for got, ref in zip(outs, (wpf, wpb, ref_t, ref_s)):
        assert torch.equal(got, ref)


@pytest.mark.parametrize("n,B", [(35, 2), (19, 3)])
def test_conv3d_g16_mfma_forward_and_backward_data(ops, n, B):
    """Wide decoder (chanstr 16,32,16,16): conv1 / conv2 on the matrix cores with 16 output channels as the MFMA rows
    (conv16_mfma.hip) against torch's conv3d and its autograd backward-data (network.py:687); every tile variant gives
    the same bits, and the results do not depend on the batch they run in."""
    g = gen(4600 + n + B)
    x = torch.randn(B, 16, n, n, n, generator=g)
    w = torch.randn(16, 16, 4, 4, 4, generator=g) / 1024 ** 0.5
    b = torch.randn(16, generator=g)
    x.requires_grad_(True)
    y_ref = F.conv3d(x, w, b)
    no = n - 3
    wf, wb = ops.pack_conv_weight(dev(w))
    wpf, wpb = ops.pack_g16_mfma(wf, 16, 16, 4), ops.pack_g16_mfma(wb, 16, 16, 4)
    xd = dev(x.detach())
    y = ops.conv3d_g16_mfma(xd, wpf, dev(b), 16, 4, 1, 0, (no, no, no), ops.ACT_RELU)
    assert (y.cpu() - F.relu(y_ref).detach()).abs().max() < 2e-5
    for v in (2, 3, 4):
        try:
            yv = ops.conv3d_g16_mfma(xd, wpf, dev(b), 16, 4, 1, 0, (no, no, no), ops.ACT_RELU, variant=v)
        except RuntimeError:
            continue
        assert torch.equal(yv, y), v
    y1 = ops.conv3d_g16_mfma(xd[1:2].contiguous(), wpf, dev(b), 16, 4, 1, 0, (no, no, no), ops.ACT_RELU)
    assert torch.equal(y1[0], y[1])
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    add = torch.randn(x.shape, generator=g)
    dx = ops.conv3d_g16_mfma(dev(gy), wpb, None, 16, 4, 1, 3, (n, n, n), addend=dev(add), mask=dev(mask))
    ref = (x.grad + add) * (mask > 0)
    assert (dx.cpu() - ref).abs().max() < 2e-5 * max(ref.abs().max().item(), 1.0)
    for v in (2, 3, 4):
        try:
            dv = ops.conv3d_g16_mfma(dev(gy), wpb, None, 16, 4, 1, 3, (n, n, n), addend=dev(add), mask=dev(mask), variant=v)
        except RuntimeError:
            continue
        assert torch.equal(dv, dx), v
    # the VALU tile kernel computes the same sums in another order
    dxv = ops.conv3d_gather(dev(gy), wb, None, 16, 4, 1, 3, (n, n, n), addend=dev(add), mask=dev(mask))
    assert (dx - dxv).abs().max().item() < 2e-5 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("cin,cout,pad,n,B", [(16, 16, 0, 16, 2), (32, 16, 0, 8, 3), (16, 16, 0, 16, 1), (16, 32, 2, 4, 3),
                                              (8, 16, 2, 2, 5)])
def test_conv_transpose_k5s2_mfma16_forward(ops, cin, cout, pad, n, B):
    """Wide decoder's four transposed convolutions on the matrix cores (16 / 32 output channels = MFMA rows, eight
    parity classes, convt16_mfma.hip): up2 / up1 (padding 0) and conv0 / up0 (padding 2, output_padding 1) against
    torch's conv_transpose3d; tile variants and batch sizes give the same bits; the packed fragments equal those of
    the one-launch packer."""
    g = gen(5200 + cin + n + B)
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cin, cout, 5, 5, 5, generator=g) / (cin * 125 / 8) ** 0.5
    b = torch.randn(cout, generator=g)
    y_ref = F.relu(F.conv_transpose3d(x, w, b, stride=2, padding=pad, output_padding=1 if pad else 0))
    wf, _ = ops.pack_convT_weight(dev(w))
    wp = ops.pack_convT16_mfma(wf, cin, cout)
    wp2 = torch.empty_like(wp)
    ops.pack_mfma_all([(wf, wp2, 11, cin, cout)])
    assert torch.equal(wp, wp2)
    y = ops.convT3d_k5s2_mfma16(dev(x), wp, dev(b), ops.ACT_RELU, cout=cout, pad=pad)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert (y.cpu() - y_ref).abs().max() < 2e-5 * max(y_ref.abs().max().item(), 1.0)
    for v in (2, 3):
        try:
            yv = ops.convT3d_k5s2_mfma16(dev(x), wp, dev(b), ops.ACT_RELU, variant=v, cout=cout, pad=pad)
        except RuntimeError:
            continue
        assert torch.equal(yv, y), v
    y1 = ops.convT3d_k5s2_mfma16(dev(x[B - 1:]), wp, dev(b), ops.ACT_RELU, cout=cout, pad=pad)
    assert torch.equal(y1[0], y[B - 1])
    yv = ops.convT3d_k5s2_fwd(dev(x), wf, dev(b), cout, pad, ops.ACT_RELU)
    assert (y - yv).abs().max().item() < 2e-5 * max(y_ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("cin,cout,n,pad", [(16, 16, 16, 0), (32, 16, 8, 0), (16, 32, 4, 2), (8, 16, 2, 2)])
def test_conv_transpose_backward_data_g16_mfma(ops, cin, cout, n, pad):
    """Backward-data of the wide decoder's up2 (16 -> 16, 16^3 -> 35^3), up1 (32 -> 16, 8^3 -> 19^3), conv0 (16 -> 32,
    padding 2) and up0 (8 -> 16, padding 2: 8 output channels, rows 8..15 of the tile zero): a stride-2 gather
    convolution with `cin` output channels on the matrix cores, against torch's autograd."""
    g = gen(4700 + cin + n)
    B = 2
    x = torch.randn(B, cin, n, n, n, generator=g, requires_grad=True)
    w = torch.randn(cin, cout, 5, 5, 5, generator=g) / (cout * 125 / 8) ** 0.5
    y_ref = F.conv_transpose3d(x, w, None, 2, pad, 1 if pad else 0)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    wf, wb = ops.pack_convT_weight(dev(w))
    wpb = ops.pack_g16_mfma(wb, cout, cin, 5)              # gather form: `cout` input channels -> `cin` outputs
    mask = torch.randn(x.shape, generator=g)
    add = torch.randn(x.shape, generator=g)
    dx = ops.conv3d_g16_mfma(dev(gy), wpb, None, cin, 5, 2, pad, (n, n, n), addend=dev(add), mask=dev(mask))
    ref = (x.grad + add) * (mask > 0)
    assert (dx.cpu() - ref).abs().max() < 2e-5 * max(ref.abs().max().item(), 1.0)
    for v in (2, 3):
        try:
            dv = ops.conv3d_g16_mfma(dev(gy), wpb, None, cin, 5, 2, pad, (n, n, n), addend=dev(add), mask=dev(mask), variant=v)
        except RuntimeError:
            continue
        assert torch.equal(dv, dx), v
    d1 = ops.conv3d_g16_mfma(dev(gy[1:2]), wpb, None, cin, 5, 2, pad, (n, n, n), addend=dev(add[1:2]), mask=dev(mask[1:2]))
    assert torch.equal(d1[0], dx[1])


@pytest.mark.parametrize("cin,cout,pad,opad,n", CONVT_CASES)
def test_conv_transpose_forward_and_backward_data(ops, cin, cout, pad, opad, n):
    g = gen(cin * 100 + cout + pad)
    B = 2
    x = torch.randn(B, cin, n, n, n, generator=g, requires_grad=True)
    w = torch.randn(cin, cout, 5, 5, 5, generator=g) / (cin * 125 / 8) ** 0.5
    b = torch.randn(cout, generator=g)
    y_ref = F.conv_transpose3d(x, w, b, 2, pad, opad)
    wf, wb = ops.pack_convT_weight(dev(w))
    y = ops.convT3d_k5s2_fwd(dev(x.detach()), wf, dev(b), cout, pad, ops.ACT_RELU)
    ops.set_naive(True)
    yn = ops.convT3d_k5s2_fwd(dev(x.detach()), wf, dev(b), cout, pad, ops.ACT_RELU)
    ops.set_naive(False)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert torch.equal(y, yn)
    assert (y.cpu() - F.relu(y_ref.detach())).abs().max() < 2e-5
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    dx = ops.conv3d_gather(dev(gy), wb, None, cin, 5, 2, pad, (n, n, n))
    ops.set_naive(True)
    dxn = ops.conv3d_gather(dev(gy), wb, None, cin, 5, 2, pad, (n, n, n))
    ops.set_naive(False)
    assert torch.equal(dx, dxn)
    assert rel_err(dx, x.grad) < 1e-5


@pytest.mark.parametrize("cin,cout,k,pad,n", [c for c in CONV_CASES if c[2] > 1])
def test_conv_weight_gradient(ops, cin, cout, k, pad, n):
    g = gen(7 + cin + cout * 3 + k)
    B = 3
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g, requires_grad=True)
    y = F.conv3d(x, w, None, 1, pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dw = ops.wgrad(dev(gy), dev(x), k, 1, pad, out_mode=0)
    ops.set_naive(True)
    dwn = ops.wgrad(dev(gy), dev(x), k, 1, pad, out_mode=0)
    ops.set_naive(False)
    if cout == 1:   # the transposed orientation (p = X, q = dlogit, flipped taps) gives the same tensor
        dwf = ops.wgrad(dev(x), dev(gy), k, 1, k - 1 - pad, out_mode=1)
        assert rel_err(dwf, w.grad) < 2e-5
    assert tuple(dw.shape) == tuple(w.shape)
    assert rel_err(dwn, w.grad) < 2e-5
    assert rel_err(dw, w.grad) < 2e-5
    acc = ops.wgrad(dev(gy), dev(x), k, 1, pad, out_mode=0, out=dw.clone(), accumulate=True)
    assert rel_err(acc, 2 * w.grad) < 2e-5


@pytest.mark.parametrize("cin,cout,pad,opad,n", CONVT_CASES)
def test_conv_transpose_weight_gradient(ops, cin, cout, pad, opad, n):
    g = gen(11 + cin + cout * 3 + pad)
    B = 3
    x = torch.randn(B, cin, n, n, n, generator=g)
    w = torch.randn(cin, cout, 5, 5, 5, generator=g, requires_grad=True)
    y = F.conv_transpose3d(x, w, None, 2, pad, opad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dw = ops.wgrad(dev(x), dev(gy), 5, 2, pad, out_mode=0)
    assert tuple(dw.shape) == tuple(w.shape)
    assert rel_err(dw, w.grad) < 2e-5
    ops.set_naive(True)
    dwn = ops.wgrad(dev(x), dev(gy), 5, 2, pad, out_mode=0)
    ops.set_naive(False)
    assert rel_err(dwn, w.grad) < 2e-5


def test_channel_sum(ops):
    g = gen(3)
    for shape in ((3, 8, 35, 35, 35), (2, 16, 8, 8, 8), (5, 1, 32, 32, 32), (1, 3, 2, 2, 2)):
        x = torch.randn(shape, generator=g)
        ref = x.double().sum(dim=(0, 2, 3, 4))
        out = ops.channel_sum(dev(x))
        assert rel_err(out, ref) < 1e-5
        out2 = ops.channel_sum(dev(x), out=out.clone(), accumulate=True)
        assert rel_err(out2, 2 * ref) < 1e-5


@pytest.mark.parametrize("c,n,inverse", [(3, 2, False), (8, 4, True), (16, 4, True), (8, 2, False), (4, 2, True)])
def test_gdn_forward_backward(ops, c, n, inverse):
    g = gen(c * 10 + n)
    B = 5
    x = torch.randn(B, c, n, n, n, generator=g, requires_grad=True)
    beta = (torch.sqrt(torch.ones(c) + 2.0 ** -36) + 0.05 * torch.randn(c, generator=g)).requires_grad_(True)
    gamma = (torch.sqrt(0.1 * torch.eye(c) + 2.0 ** -36) + 0.02 * torch.randn(c, c, generator=g).abs())
    with torch.no_grad():
        if c == 4:   # below-bound parameters: pins the LowerBound gradient rule
            beta[0] = 1e-4
            gamma[0, 1] = 1e-7
            gamma[2, 3] = -0.3
    gamma.requires_grad_(True)
    gy = torch.randn(B, c, n, n, n, generator=g)
    y_ref = O.gdn3d(x, beta, gamma, inverse)
    y_ref.backward(gy)
    y = ops.gdn_fwd(dev(x.detach()), dev(beta.detach()), dev(gamma.detach()), inverse)
    assert (y.cpu() - y_ref.detach()).abs().max() < 1e-5
    dx, db, dg = ops.gdn_bwd(dev(x.detach()), dev(beta.detach()), dev(gamma.detach()), dev(gy), inverse)
    assert rel_err(dx, x.grad) < 1e-5
    assert rel_err(db, beta.grad) < 1e-4
    assert rel_err(dg, gamma.grad) < 1e-4
    assert torch.equal(db.cpu() == 0, beta.grad == 0)
    assert torch.equal(dg.cpu() == 0, gamma.grad == 0)


def _latent_likelihood64(x, sigma, mu, mode, u):
    """Per-element likelihood Phi(up) - Phi(lo) in float64 (classifies bulk / tail elements; network.py:145-161)."""
    v = (x.detach() + (u - 0.5)) if mode == "train" else torch.round(x.detach())
    s, m = sigma.detach().abs().double(), mu.detach().double()
    cdf = lambda z: 0.5 * (1 + torch.erf(z / 2 ** 0.5))
    return cdf((v.double() - m + 0.5) / s) - cdf((v.double() - m - 0.5) / s)


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("c", [3, 8])
def test_latent_rate_bulk(ops, mode, c):
    """Every latent within ~2.2 sigma of its channel mean (likelihood >= 1e-2): value 1e-5, ALL gradients -- d/dx per
    element, d/dsigma, d/dmu -- to the 1e-4 SURVEY.md 8(c) states for gradients.  The tail allowance of
    test_latent_rate cannot hide an error here."""
    g = gen(140 + c)
    B = 9
    sigma = (0.6 + 0.8 * torch.rand(1, c, 1, 1, 1, generator=g))
    with torch.no_grad():
        sigma.view(-1)[0] = -sigma.view(-1)[0]        # abs() path
    mu = 0.3 * torch.randn(1, c, 1, 1, 1, generator=g)
    x = (mu + sigma.abs() * torch.clamp(torch.randn(B, c, 2, 2, 2, generator=g), -2.2, 2.2)).requires_grad_(True)
    sigma.requires_grad_(True)
    mu.requires_grad_(True)
    u = torch.rand(B, c, 2, 2, 2, generator=g)
    assert _latent_likelihood64(x, sigma, mu, mode, u).min().item() > 5e-3
    P = {"entropy_coder.sigma": sigma, "entropy_coder.mu": mu}
    rounded_ref, bits_ref = O.entropy_coder(P, x, mode, u)
    coef = 0.37
    (coef * bits_ref).backward()
    xr, bits, dx, ds, dm = ops.latent_rate(dev(x.detach()), dev(sigma.detach().reshape(-1)),
                                           dev(mu.detach().reshape(-1)), mode, u=dev(u), want_grad=True, g_host=coef)
    assert torch.equal(xr.cpu(), rounded_ref.detach())
    assert abs(bits.item() - bits_ref.item()) < 1e-5 * abs(bits_ref.item())
    scale = x.grad.abs().max().item()
    assert torch.allclose(dx.cpu(), x.grad, rtol=1e-4, atol=1e-5 * scale)
    assert rel_err(ds, sigma.grad.reshape(-1)) < 1e-4
    assert rel_err(dm, mu.grad.reshape(-1)) < 1e-4


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("c", [3, 8])
def test_latent_rate(ops, mode, c):
    """Bulk AND tails in one batch (floor, half-to-even, abs(sigma)).  The reference formulates the likelihood as
    Phi(up) - Phi(lo), a difference of fp32 values near 1 (network.py:145-161): one ulp of erf (6e-8) is a relative
    error of ~1.2e-7 / L in a likelihood L and in every gradient divided by it.  So elements with L >= 2e-3 are held
    to 1e-4; an element in the tail to 4e-7 / L (documented allowance: the oracle's CPU erf and the device's erff
    differ by an ulp); the sums d/dsigma, d/dmu, which include the tail elements, to 2e-3."""
    g = gen(40 + c)
    B = 7
    x = (1.2 * torch.randn(B, c, 2, 2, 2, generator=g)).requires_grad_(True)
    with torch.no_grad():
        x.view(-1)[0] = 6.6    # tail: likelihood under the 1e-8 floor
        x.view(-1)[1] = 0.5    # round-half-to-even
        x.view(-1)[2] = 1.5
        x.view(-1)[3] = -2.5
    sigma = (1 + 0.3 * torch.randn(1, c, 1, 1, 1, generator=g))
    with torch.no_grad():
        sigma.view(-1)[0] = -abs(sigma.view(-1)[0])   # abs() path
    sigma.requires_grad_(True)
    mu = (0.2 * torch.randn(1, c, 1, 1, 1, generator=g)).requires_grad_(True)
    u = torch.rand(B, c, 2, 2, 2, generator=g)
    L = _latent_likelihood64(x, sigma, mu, mode, u)
    bulk = L >= 2e-3
    assert bulk.float().mean() > 0.8 and (~bulk).any()
    # relative allowance per element: 1e-4 in the bulk, 4e-7 / L in the tail, 1e-3 under the floor (no 1/L there)
    tol = torch.where(bulk, torch.full_like(L, 1e-4), torch.where(L >= 1e-8, (4e-7 / L).clamp(max=0.5),
                                                                  torch.full_like(L, 1e-3)))

    def check_dx(got, ref):
        err = (got.double().cpu() - ref.double()).abs()
        assert bool((err <= tol * ref.double().abs() + 1e-6).all()), (err / (ref.double().abs() + 1e-30))[~bulk]

    P = {"entropy_coder.sigma": sigma, "entropy_coder.mu": mu}
    rounded_ref, bits_ref = O.entropy_coder(P, x, mode, u)
    coef = 0.37
    (coef * bits_ref).backward()
    xr, bits, dx, ds, dm = ops.latent_rate(dev(x.detach()), dev(sigma.detach().reshape(-1)),
                                           dev(mu.detach().reshape(-1)), mode, u=dev(u), want_grad=True,
                                           g_host=coef)
    assert torch.equal(xr.cpu(), rounded_ref.detach())
    assert abs(bits.item() - bits_ref.item()) < 1e-4 * abs(bits_ref.item())
    check_dx(dx, x.grad)
    assert rel_err(ds, sigma.grad.reshape(-1)) < 2e-3
    assert rel_err(dm, mu.grad.reshape(-1)) < 2e-3
    # upstream gradient from a device scalar, negative sign: floor blocks the tail element
    x.grad = None
    rounded_ref, bits_ref = O.entropy_coder(P, x, mode, u)
    (-1.0 * bits_ref).backward()
    gdev = dev(torch.tensor([-1.0]))
    _, _, dx2, _, _ = ops.latent_rate(dev(x.detach()), dev(sigma.detach().reshape(-1)), dev(mu.detach().reshape(-1)),
                                      mode, u=dev(u), want_grad=True, g_dev=gdev)
    check_dx(dx2, x.grad)
    # where the likelihood sits under the floor a positive incoming gradient is blocked on both sides
    assert torch.equal(dx2.cpu() == 0, x.grad == 0)


def test_latent_noise_is_per_block(ops):
    x = torch.zeros(6, 3, 2, 2, 2)
    s, m = dev(torch.ones(3)), dev(torch.zeros(3))
    ids = torch.tensor([5, 9, 2, 7, 1, 0])
    _, b_all, dx_all, _, _ = ops.latent_rate(dev(x), s, m, "train", block_ids=dev(ids), want_grad=True, seed=11, step=3)
    _, b_a, dx_a, _, _ = ops.latent_rate(dev(x[:2]), s, m, "train", block_ids=dev(ids[:2]), want_grad=True, seed=11, step=3)
    _, b_b, dx_b, _, _ = ops.latent_rate(dev(x[2:]), s, m, "train", block_ids=dev(ids[2:]), want_grad=True, seed=11, step=3)
    assert torch.equal(dx_all[:2], dx_a) and torch.equal(dx_all[2:], dx_b)
    assert abs(b_all.item() - (b_a.item() + b_b.item())) < 1e-3


def test_weight_rate(ops):
    g = gen(50)
    k = (0.2 * torch.randn(8, 8, 4, 4, 4, generator=g)).requires_grad_(True)
    sigma = torch.tensor([0.35], requires_grad=True)
    mu = torch.tensor([0.02], requires_grad=True)
    bits_ref = O.gaussian_bits(O.round_ste(k, 16).reshape(-1, 1), torch.abs(sigma), mu, 0.5 / 16)
    (0.01 * bits_ref).backward()
    dk = torch.zeros_like(k).cuda()
    ds = torch.zeros(1).cuda()
    dm = torch.zeros(1).cuda()
    bits = ops.weight_rate(dev(k.detach()), dev(sigma.detach()), dev(mu.detach()), dk=dk, dsigma=ds, dmu=dm,
                           g_host=0.01)
    assert abs(bits.item() - bits_ref.item()) < 1e-5 * bits_ref.item()
    assert rel_err(dk, k.grad) < 1e-4
    assert rel_err(ds, sigma.grad) < 1e-4
    assert rel_err(dm, mu.grad) < 1e-4


def test_focal_losses_match_reference_goldens(ops, golden_dir):
    import os
    from tests.golden_inputs import loss_case_inputs
    G = np.load(os.path.join(golden_dir, "loss.npz"))
    for name, (p, gt, dist) in loss_case_inputs().items():
        loss, dp = ops.focal_loss(dev(p), dev(gt), None, 0.85, want_grad=True)
        ref = float(G[name + "/focal"])
        assert abs(loss.item() - ref) <= 2e-5 * max(abs(ref), 1.0), name
        np.testing.assert_allclose(dp.cpu().numpy(), G[name + "/focal_grad"], rtol=2e-5, atol=1e-6)
        loss, dp = ops.focal_loss(dev(p), dev(gt), dev(dist), 0.9, beta=1.0, want_grad=True)
        ref = float(G[name + "/surf"])
        assert abs(loss.item() - ref) <= 2e-5 * max(abs(ref), 1.0), name
        np.testing.assert_allclose(dp.cpu().numpy(), G[name + "/surf_grad"], rtol=2e-5, atol=1e-6)
        m = ops.metrics(dev(p), dev(gt), dev(dist), 0.5, 0.6).cpu().numpy()
        acc = G[name + "/acc"]
        with np.errstate(invalid="ignore", divide="ignore"):
            np.testing.assert_allclose(np.array([m[0] / m[1], m[2] / m[3]], np.float32), acc.astype(np.float32),
                                       rtol=1e-6)
        np.testing.assert_allclose(m[4:6], G[name + "/sse1"], rtol=1e-5)


def test_three_focal_terms_in_one_launch(ops, golden_dir):
    """nvf_focal_loss_multi (float4 groups) against the single-term kernel and the reference goldens: the gradients
    are elementwise and must be identical, the sums agree to fp32 summation-order tolerance."""
    import os
    from tests.golden_inputs import loss_case_inputs
    G = np.load(os.path.join(golden_dir, "loss.npz"))
    for name, (p, gt, dist) in loss_case_inputs().items():
        for chain in (False, True):
            out = torch.empty(4, device="cuda")
            dps = ops.focal_loss_multi([(dev(p), dev(gt), dev(dist), 0.9, 1.0), (dev(p), dev(gt), None, 0.85, 0.0)],
                                       out, chain_sigmoid=chain)
            s1, d1 = ops.focal_loss(dev(p), dev(gt), dev(dist), 0.9, beta=1.0, want_grad=True, chain_sigmoid=chain)
            s0, d0 = ops.focal_loss(dev(p), dev(gt), None, 0.85, want_grad=True, chain_sigmoid=chain)
            assert torch.equal(dps[0], d1) and torch.equal(dps[1], d0), name
            for got, ref in ((out[0].item(), float(G[name + "/surf"])), (out[1].item(), float(G[name + "/focal"]))):
                assert abs(got - ref) <= 2e-5 * max(abs(ref), 1.0), name
    # a length that is not a multiple of four goes through the scalar tail
    n = 4 * 1000 + 3
    p = torch.rand(n + 1, device="cuda")[:n].contiguous()
    gt = (torch.rand(n, device="cuda") > 0.7).float()
    out = torch.empty(4, device="cuda")
    dps = ops.focal_loss_multi([(p, gt, None, 0.85, 0.0)], out, chain_sigmoid=False)
    s0, d0 = ops.focal_loss(p, gt, None, 0.85, want_grad=True)
    assert torch.equal(dps[0], d0) and abs(out[0].item() - s0.item()) <= 2e-5 * max(abs(s0.item()), 1.0)


def test_deferred_final_passes_equal_immediate_ones(ops):
    """StepCtx.begin / flush (nvf_finals_begin / nvf_finals_flush on a caller-owned NvfStepCtx): the focal, bias-sum,
    weight-rate and metrics final passes queued into one launch give bit for bit what the separate launches give;
    nothing is written before the flush."""
    torch.manual_seed(3)
    B = 4
    p = torch.rand(B, 1, 16, 16, 16, device="cuda")
    gt = (torch.rand_like(p) > 0.8).float()
    dist = torch.rand_like(p)
    xs = [torch.randn(B, 8, 12, 12, 12, device="cuda"), torch.randn(B, 16, 5, 5, 5, device="cuda")]
    ks = [torch.randn(8, 8, 4, 4, 4, device="cuda") * 0.2, torch.randn(16, 8, 5, 5, 5, device="cuda") * 0.2]
    sigma, mu = torch.tensor([0.3], device="cuda"), torch.tensor([0.01], device="cuda")

    ctx = ops.StepCtx()

    def run(defer):
        c = ctx if defer else None
        loss = torch.full((4,), -7.0, device="cuda")
        macc = torch.full((6,), 2.0, device="cuda")
        outs = [torch.full((x.shape[1],), -7.0, device="cuda") for x in xs]
        dks = [torch.zeros_like(k) for k in ks]
        bits = torch.full((2,), -7.0, device="cuda")
        ds, dm = torch.full((1,), -7.0, device="cuda"), torch.full((1,), -7.0, device="cuda")
        if defer:
            ctx.begin()
        dps = ops.focal_loss_multi([(p, gt, dist, 0.9, 1.0), (p, gt, None, 0.85, 0.0)], loss, ctx=c)
        ops.multi_channel_sum(xs, outs, ctx=c)
        ops.weight_rate_batch(ks, dks, sigma, mu, bits, ds, dm, g_host=0.5, ctx=c)
        ops.metrics(p, gt, dist, 0.5, 0.6, out=macc, accumulate=True, ctx=c)
        if defer:
            torch.cuda.synchronize()
            assert loss[0].item() == -7.0 and bits[0].item() == -7.0 and outs[0][0].item() == -7.0
            assert macc[1].item() == 2.0
            ctx.flush()
        torch.cuda.synchronize()
        return [loss[:2].clone(), *outs, bits, ds, dm, macc, *dps, *dks]

    for got, ref in zip(run(True), run(False)):
        assert torch.equal(got, ref)
    # outside begin/flush nothing is queued, and a flush with an empty queue launches nothing
    ctx.flush()
    # a queue cannot be opened twice on one context (NVF_EINVAL), and cancel() closes it without launching
    ctx.begin()
    with pytest.raises(RuntimeError):
        ctx.begin()
    ctx.cancel()
    ctx.begin()
    ctx.flush()


def test_two_step_contexts_do_not_interfere(ops):
    """The library holds no queue of its own (no hidden global state, re-entrant): two caller-owned
    contexts interleave their deferred final passes and latent tails call by call -- as two engines or two threads
    in one process would -- and each delivers exactly its own results."""
    torch.manual_seed(8)
    B, c = 4, 3
    mk = lambda: dict(p=torch.rand(B, 1, 16, 16, 16, device="cuda"), gt=(torch.rand(B, 1, 16, 16, 16, device="cuda") > 0.8).float(),
                      x=torch.randn(B, 8, 6, 6, 6, device="cuda"), lat=torch.randn(B, c, 2, 2, 2, device="cuda") * 3,
                      h=torch.randn(B, c, 2, 2, 2, device="cuda"), e=torch.randn(B, c, 2, 2, 2, device="cuda"),
                      dx0=torch.randn(B, c, 2, 2, 2, device="cuda") * 0.1)
    sigma, mu = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
    beta, gamma = torch.rand(c, device="cuda") + 0.5, torch.rand(c, c, device="cuda") * 0.2
    ids = torch.arange(B, device="cuda")
    data = [mk(), mk()]

    def outputs():
        return dict(loss=torch.full((4,), -7.0, device="cuda"), bsum=torch.full((8,), -7.0, device="cuda"),
                    dlat=torch.full((B, c, 2, 2, 2), float("nan"), device="cuda"),
                    dh=torch.full((B, c, 2, 2, 2), float("nan"), device="cuda"),
                    ds=torch.empty(c, device="cuda"), dm=torch.empty(c, device="cuda"), dbeta=torch.empty_like(beta),
                    dgamma=torch.empty_like(gamma), dw=torch.empty(c, c, 1, 1, 1, device="cuda"),
                    db=torch.empty(c, device="cuda"))

    def stages(d, o, ctx):
        """The calls of one 'step', as a generator so that two steps can be interleaved."""
        wb = ops.WgradBatch(torch.device("cuda"), nbytes=8 << 20, ctx=ctx)
        if ctx is not None:
            ctx.begin()
        yield
        o["dp"] = ops.focal_loss_multi([(d["p"], d["gt"], None, 0.85, 0.0)], o["loss"], ctx=ctx)[0]
        yield
        if ctx is not None:
            ops.latent_tail_queue(ctx, d["lat"], sigma, mu, "eval", ids, d["dx0"], o["dlat"], o["ds"], o["dm"], None, 1.5,
                                  9, 4, None, d["h"], beta, gamma, o["dh"], o["dbeta"], o["dgamma"], d["e"], o["dw"],
                                  o["db"])
        else:
            _, _, dl, ds, dm = ops.latent_rate(d["lat"], sigma, mu, "eval", block_ids=ids, want_grad=True, g_host=1.5,
                                               seed=9, step=4, dx_addend=d["dx0"])
            o["dlat"], o["ds"], o["dm"] = dl, ds, dm
            o["dh"], o["dbeta"], o["dgamma"] = ops.gdn_bwd(d["h"], beta, gamma, dl, False)
            o["dw"] = ops.wgrad(o["dh"], d["e"], 1, 1, 0)
        yield
        wb.add(d["x"], d["x"], 1, 1, 0, 0, torch.empty(8, 8, 1, 1, 1, device="cuda"))
        wb.finish_with_sums([d["x"]], [o["bsum"]])
        yield
        if ctx is not None:
            ctx.flush()
        yield

    ref = [outputs(), outputs()]
    for d, o in zip(data, ref):
        for _ in stages(d, o, None):
            pass
    got = [outputs(), outputs()]
    ctxs = [ops.StepCtx(), ops.StepCtx()]
    gens = [stages(d, o, c_) for d, o, c_ in zip(data, got, ctxs)]
    for _ in range(5):                         # A1 B1 A2 B2 ...: every call of A is followed by the same call of B
        for g in gens:
            next(g)
    torch.cuda.synchronize()
    assert not ctxs[0].tail_pending() and not ctxs[1].tail_pending()
    for o, r in zip(got, ref):
        for k in ("loss", "bsum", "dp", "dlat", "ds", "dm", "dh", "dbeta", "dgamma", "dw"):
            a, b = o[k], r[k]
            assert torch.equal(a[:1] if k == "loss" else a, b[:1] if k == "loss" else b), k
    assert not torch.equal(got[0]["loss"][:1], got[1]["loss"][:1])


@pytest.mark.parametrize("which", ["narrow", "wide"])
def test_heads_loss_and_backward_data_in_one_launch(ops, which):
    """nvf_heads3_loss_bwd_data against nvf_focal_loss_multi + nvf_heads3_bwd_data: the logit gradients and the
    heads' input gradients are the same bits (same elementwise code, same stencil order); the three loss sums agree
    to fp32 summation order."""
    torch.manual_seed(11)
    B = 3
    shapes = HEAD_TUPLES[which]
    ps = [torch.rand(B, 1, s, s, s, device="cuda") for c, s in shapes]
    ps[2][0, 0, :2] = 0.0                      # saturated probabilities hit the 1e-9 clamp
    ps[1][1, 0, 3] = 1.0
    gts = [(torch.rand(B, 1, s, s, s, device="cuda") > 0.75).float() for c, s in shapes]
    dist = torch.rand(B, 1, 32, 32, 32, device="cuda")
    ws = [torch.randn(1, c, 3, 3, 3, device="cuda") for c, s in shapes]
    wbs = [ops.pack_conv_weight(w)[1] for w in ws]
    mask = torch.randn(B, shapes[2][0], 32, 32, 32, device="cuda")
    cs = [c for c, s in shapes]
    loss_ref = torch.empty(4, device="cuda")
    dl2, dl0, dl1 = ops.focal_loss_multi([(ps[2], gts[2], dist, 0.9, 1.0), (ps[0], gts[0], None, 0.85, 0.0),
                                          (ps[1], gts[1], None, 0.85, 0.0)], loss_ref)
    dx_ref = ops.heads3_bwd_data([dl0, dl1, dl2], wbs, cs, [None, None, mask])
    loss = torch.empty(4, device="cuda")
    dls, dxs = ops.heads3_loss_bwd_data(ps, gts, [None, None, dist], [0.85, 0.85, 0.9], [0.0, 0.0, 1.0], [1, 2, 0],
                                        loss, wbs, cs, [None, None, mask])
    for got, ref in zip(dls, (dl0, dl1, dl2)):
        assert torch.equal(got, ref)
    for got, ref in zip(dxs, dx_ref):
        assert torch.equal(got, ref)
    np.testing.assert_allclose(loss[:3].cpu().numpy(), loss_ref[:3].cpu().numpy(), rtol=2e-6)
    # ... and with bias_outs: the heads' bias gradients (sum of the logit gradients) from the same launch, directly and
    # through the deferred final passes; everything else unchanged
    for deferred in (False, True):
        gbs = [torch.full((1,), 7.0, device="cuda") for _ in range(3)]
        ctx = ops.StepCtx() if deferred else None
        if deferred:
            ctx.begin()
        dls2, dxs2 = ops.heads3_loss_bwd_data(ps, gts, [None, None, dist], [0.85, 0.85, 0.9], [0.0, 0.0, 1.0],
                                              [1, 2, 0], loss, wbs, cs, [None, None, mask], ctx=ctx, bias_outs=gbs)
        if deferred:
            assert all(float(g) == 7.0 for g in gbs)
            ctx.flush()
        for got, ref, gb in zip(dls2, (dl0, dl1, dl2), gbs):
            assert torch.equal(got, ref)
            want = float(ref.double().sum())
            assert abs(float(gb) - want) <= 2e-6 * float(ref.double().abs().sum()) + 1e-12, (float(gb), want)
        for got, ref in zip(dxs2, dx_ref):
            assert torch.equal(got, ref)
    with pytest.raises(RuntimeError):          # one loss partial per workgroup: batch <= 32
        big = [torch.rand(33, 1, s, s, s, device="cuda") for c, s in shapes]
        ops.heads3_loss_bwd_data(big, big, [None, None, None], [0.85, 0.85, 0.9], [0.0, 0.0, 1.0], [1, 2, 0], loss,
                                 wbs, cs, [None, None, None])


@pytest.mark.parametrize("which,B", [("narrow", 16), ("narrow", 5), ("narrow", 32), ("wide", 16), ("wide", 3)])
def test_heads_forward_loss_and_backward_data_in_one_launch(ops, which, B):
    """nvf_heads3_fwd_loss_bwd_data (the heads' forward workgroups hand p to their block's loss / backward-data workgroups
    INSIDE the launch: device-scope stores, arrival counters) against nvf_heads3_fwd + nvf_heads3_loss_bwd_data_bias: p,
    the logit gradients, the input gradients, the loss terms and the bias gradients are the same BITS -- on every one of
    several back-to-back calls with fresh inputs (the last consumer of a block resets its counters for the next call)."""
    g = gen(7100 + B)
    shapes = HEAD_TUPLES[which]
    cs = [c for c, s in shapes]
    ws = [torch.randn(1, c, 3, 3, 3, generator=g) * 0.1 for c, s in shapes]
    bs = [dev(torch.randn(1, generator=g)) for _ in shapes]
    packed = [ops.pack_conv_weight(dev(w)) for w in ws]
    ctx = ops.StepCtx()
    args = ([0.85, 0.85, 0.9], [0.0, 0.0, 1.0], [1, 2, 0])
    for rep in range(4):
        xs = [dev(torch.randn(B, c, s, s, s, generator=g)) for c, s in shapes]
        gts = [dev((torch.rand(B, 1, s, s, s, generator=g) > 0.75).float()) for c, s in shapes]
        dist = dev(torch.rand(B, 1, 32, 32, 32, generator=g))
        masks = [None, None, xs[2]]
        ps_ref = ops.heads3_fwd(xs, [p[0] for p in packed], bs)
        loss_ref, gb_ref = torch.empty(4, device="cuda"), [torch.zeros(1, device="cuda") for _ in range(3)]
        dls_ref, dxs_ref = ops.heads3_loss_bwd_data(ps_ref, gts, [None, None, dist], *args, loss_ref,
                                                    [p[1] for p in packed], cs, masks, bias_outs=gb_ref)
        loss, gbs = torch.empty(4, device="cuda"), [torch.zeros(1, device="cuda") for _ in range(3)]
        ctx.begin()
        ps, dls, dxs = ops.heads3_fwd_loss_bwd_data(xs, [p[0] for p in packed], bs, gts, [None, None, dist], *args, loss,
                                                    [p[1] for p in packed], masks, ctx, bias_outs=gbs)
        ctx.flush()
        torch.cuda.synchronize()
        for name, got, ref in (("p", ps, ps_ref), ("dl", dls, dls_ref), ("dx", dxs, dxs_ref), ("bias", gbs, gb_ref)):
            for h in range(3):
                assert torch.equal(got[h], ref[h]), (name, h, rep)
        assert torch.equal(loss[:3], loss_ref[:3]), rep
        assert int(ctx._ws["heads_flags"].abs().sum()) == 0          # every counter back at zero
    with pytest.raises(RuntimeError):          # one loss partial per workgroup: batch <= 32
        big = [torch.rand(33, c, s, s, s, device="cuda") for c, s in shapes]
        bg = [torch.rand(33, 1, s, s, s, device="cuda") for c, s in shapes]
        ops.heads3_fwd_loss_bwd_data(big, [p[0] for p in packed], bs, bg, [None, None, None], *args, loss,
                                     [p[1] for p in packed], [None, None, None], ctx)


@pytest.mark.parametrize("c,B", [(3, 16), (8, 5), (3, 40)])
def test_latent_tail_inside_the_slab_reduction_launch(ops, c, B):
    """nvf_latent_tail_queue (one workgroup of the nvf_wgrad_reduce_multi_and_sums launch) against nvf_latent_rate +
    nvf_gdn_bwd + nvf_wgrad: every gradient bit for bit (same arithmetic, same order); the bias gradient to fp32
    summation order.  40 blocks x 8 voxels = three 128-voxel chunks of the GDN stage."""
    torch.manual_seed(5 + c)
    dev_ = "cuda"
    lat = torch.randn(B, c, 2, 2, 2, device=dev_) * 3
    h = torch.randn(B, c, 2, 2, 2, device=dev_)
    e = torch.randn(B, c, 2, 2, 2, device=dev_)
    dx0 = torch.randn(B, c, 2, 2, 2, device=dev_) * 0.1
    sigma, mu = torch.rand(c, device=dev_) + 0.5, torch.randn(c, device=dev_) * 0.1
    beta, gamma = torch.rand(c, device=dev_) + 0.5, torch.rand(c, c, device=dev_) * 0.2
    ids = torch.arange(B, device=dev_) * 3 + 1
    g_dev = torch.tensor([0.37], device=dev_)
    for mode in ("train", "eval"):
        _, _, dlat_r, ds_r, dm_r = ops.latent_rate(lat, sigma, mu, mode, block_ids=ids, want_grad=True, g_dev=g_dev,
                                                   g_host=1.5, seed=9, step=4, dx_addend=dx0)
        dh_r, db_r, dg_r = ops.gdn_bwd(h, beta, gamma, dlat_r, False)
        dw_r = ops.wgrad(dh_r, e, 1, 1, 0)
        bias_r = dh_r.sum(dim=(0, 2, 3, 4))
        # the tail rides on a slab reduction: give it one real weight-gradient job and one bias-sum job
        p, q = torch.randn(B, 8, 16, 16, 16, device=dev_), torch.randn(B, 8, 19, 19, 19, device=dev_)
        dw_other_r = ops.wgrad(p, q, 4, 1, 0)
        ctx = ops.StepCtx()
        wb = ops.WgradBatch(torch.device(dev_), ctx=ctx)
        dw_other, pb = torch.empty_like(dw_other_r), torch.empty(8, device=dev_)
        wb.add(p, q, 4, 1, 0, 0, dw_other)
        dlat, dh = torch.full_like(lat, float("nan")), torch.full_like(h, float("nan"))
        ds, dm = torch.empty(c, device=dev_), torch.empty(c, device=dev_)
        dbeta, dgamma = torch.empty_like(beta), torch.empty_like(gamma)
        dw, dbias = torch.empty(c, c, 1, 1, 1, device=dev_), torch.empty(c, device=dev_)
        ops.latent_tail_queue(ctx, lat, sigma, mu, mode, ids, dx0, dlat, ds, dm, g_dev, 1.5, 9, 4, None, h, beta, gamma,
                              dh, dbeta, dgamma, e, dw, dbias)
        wb.finish_with_sums([p], [pb])
        torch.cuda.synchronize()
        for got, ref in ((dlat, dlat_r), (ds, ds_r), (dm, dm_r), (dh, dh_r), (dbeta, db_r), (dgamma, dg_r), (dw, dw_r),
                         (dw_other, dw_other_r)):
            assert torch.equal(got, ref), mode
        np.testing.assert_allclose(dbias.cpu().numpy(), bias_r.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(pb.cpu().numpy(), p.sum(dim=(0, 2, 3, 4)).cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_squared_error_map_matches_reference_goldens(ops, golden_dir):
    """get_se (utils/loss.py:123-128; encode-side only, NVFPCC.py:524): nvf_squared_error_map against the oracle on
    every element and against the reference's own outputs (tests/golden/loss.npz */se, tools/gen_golden.py)."""
    import os
    from tests.golden_inputs import loss_case_inputs, sample_index
    G = np.load(os.path.join(golden_dir, "loss.npz"))
    for name, (p, gt, dist) in loss_case_inputs().items():
        se = ops.squared_error_map(dev(p), dev(dist), 0.6).cpu()
        ref = O.squared_error_map(p, dist, 0.6)
        assert se.shape == ref.shape and torch.equal(se, ref), name       # compare + multiply + square: exact
        a = se.double().reshape(-1)
        summ = np.concatenate([[a.mean().item(), a.abs().sum().item()], a[sample_index(a.numel(), 64)].numpy()])
        np.testing.assert_allclose(summ, G[name + "/se"], rtol=1e-6, atol=1e-7)


def test_small_elementwise(ops):
    g = gen(60)
    p = torch.rand(3, 1, 8, 8, 8, generator=g)
    dp = torch.randn(3, 1, 8, 8, 8, generator=g)
    assert torch.allclose(ops.sigmoid_bwd(dev(dp), dev(p)).cpu(), dp * p * (1 - p), rtol=1e-6, atol=1e-7)
    x = (torch.rand(4, 1, 32, 32, 32, generator=g) < 0.03).float()
    y = ops.maxpool2(dev(x))
    assert torch.equal(y.cpu(), F.max_pool3d(x, 2, 2))
    assert torch.equal(ops.maxpool2(y).cpu(), F.max_pool3d(F.max_pool3d(x, 2, 2), 2, 2))
    emb = torch.randn(10, 3, 2, 2, 2, generator=g)
    idx = torch.tensor([7, 0, 3, 9])
    assert torch.equal(ops.gather_rows(dev(emb), dev(idx)).cpu(), emb[idx])
    dst = torch.zeros(10, 3, 2, 2, 2)
    out = ops.scatter_add_rows(dev(emb[:4].contiguous()), dev(idx), dev(dst))
    ref = dst.clone()
    ref[idx] += emb[:4]
    assert torch.equal(out.cpu(), ref)


def test_adam_matches_torch(ops):
    g = gen(70)
    p0 = torch.randn(5000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p, m, v = dev(p0), dev(torch.zeros(5000)), dev(torch.zeros(5000))
    for step in range(1, 6):
        grad = torch.randn(5000, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, dev(grad), m, v, 1e-3, step)
    assert torch.allclose(p.cpu(), p_ref.detach(), rtol=1e-6, atol=1e-7)


def test_uniform_stream(ops):
    u = ops.uniform((1 << 16,), torch.device("cuda"), seed=3, stream_id=9).cpu()
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean().item() - 0.5) < 0.01 and abs(u.var().item() - 1 / 12) < 0.005
    u2 = ops.uniform((1 << 16,), torch.device("cuda"), seed=3, stream_id=9).cpu()
    u3 = ops.uniform((1 << 16,), torch.device("cuda"), seed=3, stream_id=10).cpu()
    assert torch.equal(u, u2) and not torch.equal(u, u3)
    k = torch.zeros(4096).cuda()
    ki = torch.zeros(4096).cuda()
    w1, _ = ops.effective_params(k, ki, None, None, 1, seed=3, stream_id=9)
    assert torch.allclose(w1.cpu(), (u[:4096] - 0.5) / 16, atol=1e-9)


def test_effective_params(ops):
    g = gen(80)
    k = 0.3 * torch.randn(8, 8, 4, 4, 4, generator=g)
    ki = 0.1 * torch.randn(8, 8, 4, 4, 4, generator=g)
    b, bi = torch.randn(8, generator=g), torch.randn(8, generator=g)
    u = torch.rand(k.shape, generator=g)
    for q in (0, 1, 2):
        w_ref = O.effective_kernel(k, ki, q, u)
        w, be = ops.effective_params(dev(k), dev(ki), dev(b), dev(bi), q, u=dev(u))
        assert torch.equal(w.cpu(), w_ref), q
        assert torch.equal(be.cpu(), b + bi)


def test_threshold_points_match_nonzero(ops):
    g = gen(90)
    p = torch.rand(5, 1, 32, 32, 32, generator=g)
    p[3] = 0.0   # empty block
    origins = torch.randint(0, 1024, (5, 3), generator=g, dtype=torch.int32)
    for thh in (0.5, 0.97, 0.9999):
        pts, counts = ops.threshold_points(dev(p), thh, origins)
        nz = torch.nonzero(p[:, 0] > thh)
        ref = nz[:, 1:] + origins[nz[:, 0]].long()
        assert torch.equal(pts.cpu().long(), ref)
        assert torch.equal(counts.cpu().long(), (p[:, 0] > thh).flatten(1).sum(1))


@pytest.mark.parametrize("n,B,ppc", [(35, 1, 0), (35, 5, 0), (35, 16, 0), (35, 2, 2), (35, 3, 16), (35, 16, 4), (35, 3, 65537),
                                      (35, 2, 65544), (19, 1, 0), (19, 5, 0), (19, 16, 4)])
def test_conv3d_k4_wino_forward(ops, n, B, ppc):
    """The training-step forward of conv2 / conv1 in the Winograd form against torch's conv3d on the CPU (float64) and
    against the direct fixed-order matrix-core kernel: 1e-5 of max |y| (measured 1e-6), zeros of the ReLU in the same
    places up to rounding at the kink."""
    g = gen(4900 + n + B)
    x = torch.relu(torch.randn(B, 8, n, n, n, generator=g) * 0.7)
    w = torch.round(torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08 * 16) / 16 + 0.02 * torch.randn(8, 8, 4, 4, 4, generator=g)
    b = torch.randn(8, generator=g) * 0.3
    ref = F.relu(F.conv3d(x.double(), w.double(), b.double()))
    wf, _ = ops.pack_conv_weight(dev(w))
    y = ops.conv3d_k4_wino_fwd(dev(x), ops.pack_wino_k4(wf), dev(b), ppc=ppc)
    assert rel_err(y, ref) < 1e-5, rel_err(y, ref)
    y_direct = ops.conv3d_k4_mfma(dev(x), ops.pack_mfma_k4(wf, 8, 0), dev(b), 0, 0, ops.ACT_RELU)
    assert rel_err(y, y_direct.cpu()) < 1e-5


@pytest.mark.parametrize("n,B,ppc", [(35, 1, 0), (35, 5, 0), (35, 16, 0), (35, 2, 1), (35, 3, 16), (19, 1, 0), (19, 5, 0), (19, 16, 3)])
def test_conv3d_k4_wino16_forward(ops, n, B, ppc):
    """The wide decoder's conv2 / conv1 (16 -> 16 channels) training-step forward in the Winograd form (conv16_wino.hip:
    rows = the 16 output channels, two output planes in flight) against torch's conv3d on the CPU (float64) and against
    the direct 16-row matrix-core kernel: 1e-5 of max |y|."""
    g = gen(5100 + n + B)
    x = torch.relu(torch.randn(B, 16, n, n, n, generator=g) * 0.7)
    w = torch.round(torch.randn(16, 16, 4, 4, 4, generator=g) * 0.06 * 16) / 16 + 0.02 * torch.randn(16, 16, 4, 4, 4, generator=g)
    b = torch.randn(16, generator=g) * 0.3
    ref = F.relu(F.conv3d(x.double(), w.double(), b.double()))
    wf, _ = ops.pack_conv_weight(dev(w))
    y = ops.conv3d_k4_wino16_fwd(dev(x), ops.pack_wino16_k4(wf), dev(b), ppc=ppc)
    assert rel_err(y, ref) < 1e-5, rel_err(y, ref)
    no = n - 3
    y_direct = ops.conv3d_g16_mfma(dev(x), ops.pack_g16_mfma(wf, 16, 16, 4), dev(b), 16, 4, 1, 0, (no, no, no), ops.ACT_RELU)
    assert rel_err(y, y_direct.cpu()) < 1e-5


@pytest.mark.parametrize("n,B,ppc", [(35, 1, 0), (35, 5, 0), (35, 16, 0), (35, 2, 1), (35, 3, 18), (35, 16, 6), (35, 3, 65539),
                                      (19, 1, 0), (19, 5, 0), (19, 16, 5)])
def test_conv3d_k4_wino16_backward_data(ops, n, B, ppc):
    """... and their backward-data through the ReLU mask of the layer below, against torch's autograd on the CPU; zeros
    exactly where the mask is not positive."""
    g = gen(5300 + B + n)
    x = torch.randn(B, 16, n, n, n, generator=g)
    w = torch.round(torch.randn(16, 16, 4, 4, 4, generator=g) * 0.06 * 16) / 16 + 0.02 * torch.randn(16, 16, 4, 4, 4, generator=g)
    x.requires_grad_(True)
    y_ref = F.conv3d(x, w)
    gy = torch.randn(y_ref.shape, generator=g) * (torch.rand(y_ref.shape, generator=g) < 0.6)
    y_ref.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    _, wb = ops.pack_conv_weight(dev(w))
    dx = ops.conv3d_k4_wino16_bwd(dev(gy), ops.pack_wino16_k4(wb), dev(mask), ppc=ppc)
    ref = x.grad * (mask > 0)
    assert rel_err(dx, ref) < 1e-5, rel_err(dx, ref)
    assert bool(((dx.cpu() == 0) >= (mask <= 0)).all())
    dx_direct = ops.conv3d_g16_mfma(dev(gy), ops.pack_g16_mfma(wb, 16, 16, 4), None, 16, 4, 1, 3, (n, n, n), mask=dev(mask))
    assert rel_err(dx, dx_direct.cpu()) < 1e-5
    if n == 35:      # conv2: the one-plane kernel (conv16_wino1.hip, the default without bias sums) and the two-plane kernel: same bits
        for other in (6, 65536 + 7):
            assert torch.equal(ops.conv3d_k4_wino16_bwd(dev(gy), ops.pack_wino16_k4(wb), dev(mask), ppc=other), dx)
    if ppc >> 16:
        return       # (the one-plane kernel leaves no bias sums: NVF_EINVAL with bias_part)
    # the channel sums it leaves per work unit (the bias gradient of the layer below): slabs of 16 floats
    slabs = torch.full((8192 * 16,), float("nan"), device=dx.device)
    dx2, nparts = ops.conv3d_k4_wino16_bwd(dev(gy), ops.pack_wino16_k4(wb), dev(mask), ppc=ppc, bias_part=slabs.data_ptr())
    assert torch.equal(dx2, dx) and 0 < nparts <= 8192
    sums = slabs[:16 * nparts].view(nparts, 16).double().sum(0).cpu()
    want = dx.double().sum(dim=(0, 2, 3, 4)).cpu()
    assert float((sums - want).abs().max()) < 1e-4 * float(want.abs().max() + dx.abs().max().cpu() * 100)


@pytest.mark.parametrize("w,B,zsplit", [(32, 1, 0), (32, 5, 0), (32, 16, 0), (32, 3, 2), (32, 20, 0), (16, 1, 0), (16, 5, 0), (16, 16, 0), (16, 3, 1)])
def test_wgrad16_k4_wino(ops, w, B, zsplit):
    """The wide decoder's conv2 / conv1 weight gradient in the Winograd (y, x) form (wgrad16_wino.hip) + the fixed-order slab
    reduction against torch's autograd of F.conv3d on the CPU (float64): 2e-5 of max |dW| (measured 4e-6; the gradient
    goldens are held to 2e-4); more items than workgroups (batch 20) and a split z range."""
    g = gen(5500 + B + w)
    x = torch.relu(torch.randn(B, 16, w + 3, w + 3, w + 3, generator=g) * 0.7)
    gy = torch.randn(B, 16, w, w, w, generator=g) * (torch.rand(B, 16, w, w, w, generator=g) < 0.6)
    wt = torch.zeros(16, 16, 4, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), wt).backward(gy.double())
    wb = ops.WgradBatch(torch.device("cuda"))
    base = wb.reserve(256 * 16384 * 4)
    n = ops.wgrad16_k4_wino_partial(dev(gy), dev(x), base, zsplit=zsplit)
    assert 0 < n <= 256
    dw = torch.full((16 * 16 * 64,), float("nan"), device="cuda")
    wb.add_job(base, dw, n, 16384)
    wb.finish()
    assert rel_err(dw.view(16, 16, 4, 4, 4), wt.grad) < 2e-5, rel_err(dw.view(16, 16, 4, 4, 4), wt.grad)


@pytest.mark.parametrize("B,ppc", [(1, 0), (5, 0), (16, 0), (2, 2), (3, 18), (16, 6), (3, 65537), (2, 65545)])
def test_conv3d_k4_wino_backward_data(ops, B, ppc):
    """conv2's backward-data in the reduced-multiplication form (conv_wino.hip: Winograd F(2x2, 4x4) over (y, x), direct
    over z on the matrix cores) against torch's autograd of F.conv3d on the CPU, through the ReLU mask of the layer below,
    at batches 1, 5 and 16 and with other chunkings of the z pairs; the channel sums it leaves (up2's bias gradient)
    against the sums of what it wrote.  Tolerance: 1e-5 of max |dx| (measured 1e-6; the direct MFMA form is held to the
    same figure above)."""
    _wino_bwd_case(ops, 35, B, ppc)


@pytest.mark.parametrize("B,ppc", [(1, 0), (5, 0), (16, 0), (3, 4), (2, 10), (16, 2), (3, 65538)])
def test_conv3d_k4_wino_backward_data_conv1(ops, B, ppc):
    """... and conv1's (16^3 -> 19^3: 10 x 10 tiles, three tile rows per group of sixteen)."""
    _wino_bwd_case(ops, 19, B, ppc)


def test_conv2_winograd_kernels_give_the_same_bits(ops):
    """conv2's default Winograd kernel keeps one accumulator set per wave and runs two waves per SIMD (conv_wino1.hip; ppc 0
    or bit 16 of ppc); the two-set kernel of conv_wino.hip (an explicit ppc) walks every plane once.  Every output's sum
    has the same order in both -- taps 0..4, channel group 0 then 1 -- so forward and backward-data agree bit for bit, at
    any chunking of the z pairs."""
    g = gen(4450)
    for B in (1, 5, 16):
        x = torch.relu(torch.randn(B, 8, 35, 35, 35, generator=g) * 0.7)
        gy = torch.randn(B, 8, 32, 32, 32, generator=g) * (torch.rand(B, 8, 32, 32, 32, generator=g) < 0.6)
        w = torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08
        b = torch.randn(8, generator=g) * 0.3
        wf, wb = ops.pack_conv_weight(dev(w))
        wpf, wpb = ops.pack_wino_k4(wf), ops.pack_wino_k4(wb)
        y = ops.conv3d_k4_wino_fwd(dev(x), wpf, dev(b))                      # default: conv_wino1.hip
        for ppc in (4, 2, 65537, 65540):
            assert torch.equal(ops.conv3d_k4_wino_fwd(dev(x), wpf, dev(b), ppc=ppc), y), (B, ppc)
        dx = ops.conv3d_k4_wino_bwd(dev(gy), wpb, dev(x))
        for ppc in (6, 2, 65537, 65542):
            assert torch.equal(ops.conv3d_k4_wino_bwd(dev(gy), wpb, dev(x), ppc=ppc), dx), (B, ppc)
        # conv1's backward-data (16^3 -> 19^3): the default is the one-set kernel with six waves per workgroup
        x1 = torch.randn(B, 8, 19, 19, 19, generator=g)
        gy1 = torch.randn(B, 8, 16, 16, 16, generator=g)
        dx1 = ops.conv3d_k4_wino_bwd(dev(gy1), wpb, dev(x1))
        for ppc in (2, 4, 65537, 65539):
            assert torch.equal(ops.conv3d_k4_wino_bwd(dev(gy1), wpb, dev(x1), ppc=ppc), dx1), (B, ppc)


def _wino_bwd_case(ops, n, B, ppc):
    g = gen(4400 + B + n)
    x = torch.randn(B, 8, n, n, n, generator=g)
    w = torch.round(torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08 * 16) / 16 + 0.02 * torch.randn(8, 8, 4, 4, 4, generator=g)
    x.requires_grad_(True)
    y_ref = F.conv3d(x, w)
    gy = torch.randn(y_ref.shape, generator=g) * (torch.rand(y_ref.shape, generator=g) < 0.6)
    y_ref.backward(gy)
    mask = torch.randn(x.shape, generator=g)
    _, wb = ops.pack_conv_weight(dev(w))
    wp = ops.pack_wino_k4(wb)
    dx = ops.conv3d_k4_wino_bwd(dev(gy), wp, dev(mask), ppc=ppc)
    ref = x.grad * (mask > 0)
    assert rel_err(dx, ref) < 1e-5, rel_err(dx, ref)
    assert torch.equal(dx.cpu() == 0, (ref == 0) | (dx.cpu() == 0)) and bool(((dx.cpu() == 0) >= (mask <= 0)).all())
    # channel sums per work unit: slabs of 8 floats
    slabs = torch.full((4096 * 8,), float("nan"), device=dx.device)
    dx2, n = ops.conv3d_k4_wino_bwd(dev(gy), wp, dev(mask), ppc=ppc, bias_part=slabs.data_ptr())
    assert torch.equal(dx2, dx) and 0 < n <= 4096
    sums = slabs[:8 * n].view(n, 8).double().sum(0).cpu()
    want = dx.double().sum(dim=(0, 2, 3, 4)).cpu()
    assert float((sums - want).abs().max()) < 1e-4 * float(want.abs().max() + dx.abs().max().cpu() * 100)
    # the direct matrix-core form agrees (same contract, different arithmetic)
    wpbx = ops.pack_mfma_k4(wb, 8, 0)
    dx_direct = ops.conv3d_k4_mfma(dev(gy), wpbx, None, 3, 0, ops.ACT_NONE, mask=dev(mask))
    assert rel_err(dx, dx_direct.cpu()) < 1e-5


@pytest.mark.parametrize("B,zsplit", [(1, 1), (5, 1), (16, 1), (3, 2), (2, 5)])
def test_wgrad_k4_wino(ops, B, zsplit):
    """conv2's weight gradient in the Winograd (y, x) form (wgrad_wino.h) against torch's autograd of F.conv3d on the CPU
    (float64 reference), at batches 1, 5, 16 and with the z steps split over several work items; the bias gradient it leaves
    (channel sums of dy).  Tolerance 2e-5 of max |dW| (measured 4e-6; the gradient goldens are held to 2e-4)."""
    g = gen(4700 + B)
    x = torch.relu(torch.randn(B, 8, 35, 35, 35, generator=g) * 0.7)
    gy = torch.randn(B, 8, 32, 32, 32, generator=g) * (torch.rand(B, 8, 32, 32, 32, generator=g) < 0.6)
    w = torch.zeros(8, 8, 4, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), w).backward(gy.double())
    dw, db = ops.wgrad_k4_wino(dev(gy), dev(x), zsplit=zsplit, want_bias=True)
    assert rel_err(dw, w.grad) < 2e-5, rel_err(dw, w.grad)
    assert rel_err(db, gy.double().sum(dim=(0, 2, 3, 4))) < 2e-5
    # deterministic: the same bits on a second run
    dw2 = ops.wgrad_k4_wino(dev(gy), dev(x), zsplit=zsplit)
    assert torch.equal(dw, dw2)
    # the direct matrix-core form agrees
    dw_direct = ops.wgrad(dev(gy), dev(x), 4, 1, 0, out_mode=0)
    assert rel_err(dw, dw_direct.cpu().reshape(8, 8, 4, 4, 4)) < 2e-5
