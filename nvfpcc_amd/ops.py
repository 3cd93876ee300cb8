"""Tensor-level wrappers over the C ABI (include/nvf_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every value is computed by a
hand-written gfx950 kernel in libnvf_hip.so.  All tensors must be fp32, contiguous and on a
HIP device; anything else raises (no CPU fallback).
"""
import os

import torch

from ._lib import lib, check

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
# kernel variant passed to the conv / wgrad entry points: 0 = tuned LDS-tiled kernels, 1 = the
# one-thread-per-output kernels (debug cross-check), >= 2 = alternates kept for tuning (tools/kbench.py)
_NAIVE = int(os.environ.get("NVF_VARIANT", "0"))


def set_naive(flag):
    global _NAIVE
    _NAIVE = int(bool(flag))


def set_variant(v):
    global _NAIVE
    _NAIVE = int(v)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("nvfpcc_amd ops run on the HIP device only (got a CPU tensor); "
                               "there is no CPU fallback for the NVF hot path")
        if not t.is_contiguous():
            raise RuntimeError("nvfpcc_amd ops need contiguous tensors")


def _f32(*tensors):
    _chk(*tensors)
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"expected float32, got {t.dtype}")


class StepCtx:
    """Caller-owned NvfStepCtx (include/nvf_hip.h): the queue of deferred final passes and the queued latent tail of ONE
    step in flight.  The library keeps no state of its own; each engine owns one of these."""

    def __init__(self):
        import ctypes
        self._mem = ctypes.create_string_buffer(int(lib().nvf_step_ctx_bytes()) + 16)
        addr = ctypes.addressof(self._mem)
        self.ptr = (addr + 15) // 16 * 16
        self._ws = {}      # workspaces of the passes deferred in this context: their partial sums wait for flush()
        self._retired = []  # outgrown workspaces (see workspace())
        check(lib().nvf_step_ctx_init(self.ptr), "nvf_step_ctx_init")

    def set_direct(self, on=True):
        """Launches given this context keep the direct summation order (nvf_step_ctx_set_direct)."""
        check(lib().nvf_step_ctx_set_direct(self.ptr, int(bool(on))), "nvf_step_ctx_set_direct")

    def set_wgrad_forms(self, conv2_zsplit=1, conv1_wino=False):
        """Forms of the merged weight-gradient launches given this context (nvf_step_ctx_set_wgrad_forms): z work items of
        conv2's Winograd gradient, and whether conv1's gradient takes the Winograd form too (another summation order)."""
        check(lib().nvf_step_ctx_set_wgrad_forms(self.ptr, int(conv2_zsplit), int(bool(conv1_wino))),
              "nvf_step_ctx_set_wgrad_forms")

    def begin(self):
        """Queue the final passes of focal_loss_multi / heads3_loss_bwd_data / WgradBatch.finish_with_sums /
        weight_rate_batch / metrics issued with this context (their outputs exist only after flush())."""
        check(lib().nvf_finals_begin(self.ptr), "nvf_finals_begin")

    def flush(self):
        """Run the queued final passes in one launch on the current stream."""
        check(lib().nvf_finals_flush(self.ptr, _stream()), "nvf_finals_flush")

    def flush_tail(self, tail, ranges):
        """flush() + the optimiser / epoch statistics / schedule hand-over of the step in the same launch
        (nvf_finals_flush_tail): ``tail`` is the NvfStepTail of ops.step_tail_args, ``ranges`` the [lo, hi) index ranges
        of gradient elements no fused launch covers."""
        import ctypes
        flat = [int(v) for r in ranges for v in r]
        arr = (ctypes.c_int64 * max(len(flat), 1))(*flat)
        check(lib().nvf_finals_flush_tail(self.ptr, ctypes.byref(tail), arr, len(ranges), _stream()),
              "nvf_finals_flush_tail")

    def cancel(self):
        lib().nvf_finals_cancel(self.ptr)
        lib().nvf_latent_tail_cancel(self.ptr)

    def tail_pending(self):
        return bool(lib().nvf_latent_tail_pending(self.ptr))

    def stem_pending(self):
        return bool(lib().nvf_stem_bwd_pending(self.ptr))


def _ctx(ctx):
    return None if ctx is None else ctx.ptr


_ws_cache = {}
_ws_retired = []


def workspace(nbytes, device, tag="ws", ctx=None):
    """Grow-only scratch buffer per (device, tag) -- per StepCtx when one is given: partial sums parked there until
    the context's flush must not be overwritten by another engine's step."""
    cache = _ws_cache if ctx is None else ctx._ws
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # an outgrown buffer stays alive: a captured HIP graph (engine.GraphedTrainStep) or a kernel in flight on
            # another stream may still hold its address
            (_ws_retired if ctx is None else ctx._retired).append(buf)
        buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


# ---------------------------------------------------------------- weights
def pack_conv_weight(w, want_fwd=True, want_bwd=True):
    """w [Co,Ci,K,K,K] -> (w_fwd [Ci,K^3,Co], w_bwd [Co,K^3 flipped,Ci])."""
    _f32(w)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    wf = torch.empty(ci * k ** 3 * co, device=w.device) if want_fwd else None
    wb = torch.empty(ci * k ** 3 * co, device=w.device) if want_bwd else None
    check(lib().nvf_pack_conv_weight(_ptr(w), co, ci, k, _ptr(wf), _ptr(wb), _stream()), "nvf_pack_conv_weight")
    return wf, wb


def pack_convT_weight(w, want_fwd=True, want_bwd=True):
    """w [Ci,Co,K,K,K] -> (w_fwd [Ci,K^3,Co], w_bwd [Co,K^3,Ci])."""
    _f32(w)
    ci, co, k = w.shape[0], w.shape[1], w.shape[2]
    wf = torch.empty(ci * k ** 3 * co, device=w.device) if want_fwd else None
    wb = torch.empty(ci * k ** 3 * co, device=w.device) if want_bwd else None
    check(lib().nvf_pack_convT_weight(_ptr(w), ci, co, k, _ptr(wf), _ptr(wb), _stream()), "nvf_pack_convT_weight")
    return wf, wb


def effective_params(kernel, kernel_init, b, b_init, q, u=None, seed=0, stream_id=0):
    _f32(kernel, kernel_init, b, b_init, u)
    w_eff = torch.empty_like(kernel)
    b_eff = torch.empty_like(b) if b is not None else None
    nb = b.numel() if b is not None else 0
    check(lib().nvf_effective_params(_ptr(kernel), _ptr(kernel_init), _ptr(u), _ptr(w_eff), kernel.numel(), _ptr(b),
                                     _ptr(b_init), _ptr(b_eff), nb, int(q), int(seed), int(stream_id), _stream()),
          "nvf_effective_params")
    return w_eff, b_eff


# ---------------------------------------------------------------- convolutions
def conv3d_gather(x, w_packed, bias, cout, k, stride, pad, out_spatial, act=ACT_NONE, addend=None, mask=None,
                  out=None):
    """y[b,co,o] = act(bias + sum x[b,ci,stride*o - pad + k] w[ci][k][co]) (+addend) (*(mask>0))."""
    _f32(x, w_packed, bias, addend, mask)
    B, cin, di, hi, wi = x.shape
    do, ho, wo = out_spatial
    if w_packed.numel() != cin * k ** 3 * cout:
        raise RuntimeError("packed weight size does not match (cin, k, cout)")
    y = out if out is not None else torch.empty((B, cout, do, ho, wo), device=x.device)
    for t in (addend, mask):
        if t is not None and t.shape != y.shape:
            raise RuntimeError("addend/mask shape must equal the output shape")
    check(lib().nvf_conv3d_gather(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(y), _ptr(addend), _ptr(mask), B, cin,
                                  cout, k, stride, pad, di, hi, wi, do, ho, wo, act, _NAIVE, _stream()),
          "nvf_conv3d_gather")
    return y


def pack_mfma_k4(gather_w, cin, pair_axis, out=None):
    """MFMA A-fragments of a 4^3, 8-output-channel gather weight [cin][64][8] (w_fwd, or w_bwd for backward-data)."""
    _f32(gather_w, out)
    n = int(lib().nvf_pack_mfma_k4_floats(cin, pair_axis))
    wp = out if out is not None else torch.empty(n, device=gather_w.device)
    if gather_w.numel() != cin * 64 * 8 or wp.numel() != n:
        raise RuntimeError("pack_mfma_k4: weight size does not match (cin, 4, 8)")
    check(lib().nvf_pack_mfma_k4(_ptr(gather_w), cin, 8, pair_axis, _ptr(wp), _stream()), "nvf_pack_mfma_k4")
    return wp


def pack_mfma_k4_multi(jobs):
    """jobs: list of (gather_w, cin, pair_axis, wp_out): all packed by one launch."""
    import ctypes
    n = len(jobs)
    _f32(*[j[0] for j in jobs])
    _f32(*[j[3] for j in jobs])
    check(lib().nvf_pack_mfma_k4_multi((ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs]),
                                       (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in jobs]),
                                       (ctypes.c_int * n)(*[j[1] for j in jobs]),
                                       (ctypes.c_int * n)(*[j[2] for j in jobs]), n, _stream()),
          "nvf_pack_mfma_k4_multi")


def pack_mfma_all(jobs):
    """jobs: list of (src, dst, kind, c0, c1) -- see nvf_pack_mfma_all; one launch for all of them."""
    import ctypes
    n = len(jobs)
    _f32(*[j[0] for j in jobs])
    _f32(*[j[1] for j in jobs])
    check(lib().nvf_pack_mfma_all((ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs]),
                                  (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in jobs]),
                                  (ctypes.c_int * n)(*[j[2] for j in jobs]), (ctypes.c_int * n)(*[j[3] for j in jobs]),
                                  (ctypes.c_int * n)(*[j[4] for j in jobs]), n, _stream()), "nvf_pack_mfma_all")


def conv3d_k4_mfma(x, wp, bias, pad, pair_axis, act=ACT_NONE, addend=None, mask=None, out=None, variant=None,
                   bias_part=None):
    """Matrix-core 4^3 convolution, 8 output channels: same contract as conv3d_gather(k=4, stride=1).
    ``bias_part`` (device address of >= 2048 x 8 floats; backward-data through a ReLU mask only): the launch also
    leaves per-(workgroup, wave) channel sums of what it stored there; returns (y, number of 8-float slabs)."""
    import ctypes
    _f32(x, wp, bias, addend, mask)
    B, cin, di, hi, wi = x.shape
    do, ho, wo = di + 2 * pad - 3, hi + 2 * pad - 3, wi + 2 * pad - 3
    y = out if out is not None else torch.empty((B, 8, do, ho, wo), device=x.device)
    for t in (addend, mask):
        if t is not None and t.shape != y.shape:
            raise RuntimeError("addend/mask shape must equal the output shape")
    if bias_part is not None:
        nparts = ctypes.c_int(0)
        check(lib().nvf_conv3d_k4_mfma_bias(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(addend), _ptr(mask), B, cin, 8,
                                            pad, pair_axis, di, hi, wi, do, ho, wo, act,
                                            _MFMA_VARIANT if variant is None else int(variant), int(bias_part),
                                            ctypes.byref(nparts), _stream()), "nvf_conv3d_k4_mfma_bias")
        return y, nparts.value
    check(lib().nvf_conv3d_k4_mfma(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(addend), _ptr(mask), B, cin, 8, pad,
                                   pair_axis, di, hi, wi, do, ho, wo, act,
                                   _MFMA_VARIANT if variant is None else int(variant), _stream()),
          "nvf_conv3d_k4_mfma")
    return y


def pack_wino_k4(w_bwd):
    """A fragments of nvf_conv3d_k4_wino_bwd from a gather-form backward weight [8][64][8] (one pack_mfma_all job)."""
    wp = torch.empty(int(lib().nvf_pack_wino_k4_floats()), device=w_bwd.device)
    pack_mfma_all([(w_bwd, wp, 40, 8, 8)])
    return wp


def conv3d_k4_wino_bwd(dy, wp, mask, out=None, bias_part=None, ppc=0):
    """Backward-data of a valid 4^3 convolution (8 -> 8 channels) through the ReLU mask of the layer below, in the
    Winograd (y, x) form: dx = mask > 0 ? conv_full(dy, w) : 0.  ``bias_part``: device address of the slabs that receive
    the channel sums of dx per work unit; returns (dx, number of 8-float slabs) then."""
    import ctypes
    _f32(dy, wp, mask)
    B, c, di = dy.shape[0], dy.shape[1], dy.shape[2]
    shape = (B, 8, di + 3, di + 3, di + 3)
    if c != 8 or tuple(mask.shape) != shape:
        raise RuntimeError("conv3d_k4_wino_bwd: dy [B,8,n^3], mask [B,8,(n+3)^3]")
    dx = out if out is not None else torch.empty(shape, device=dy.device)
    nparts = ctypes.c_int(0)
    check(lib().nvf_conv3d_k4_wino_bwd(_ptr(dy), _ptr(wp), _ptr(dx), _ptr(mask), B, di, int(ppc),
                                       None if bias_part is None else int(bias_part),
                                       ctypes.byref(nparts) if bias_part is not None else None, _stream()),
          "nvf_conv3d_k4_wino_bwd")
    return dx if bias_part is None else (dx, nparts.value)


def conv3d_k4_wino_fwd(x, wp, bias, out=None, ppc=0):
    """relu(conv3d(x, w) + bias) of a valid 4^3 convolution (8 -> 8 channels) in the Winograd (y, x) form -- training steps
    only (rounding-level differences from the direct fixed-order kernel).  wp = pack_wino_k4(w_fwd)."""
    _f32(x, wp, bias)
    B, c, di = x.shape[0], x.shape[1], x.shape[2]
    if c != 8:
        raise RuntimeError("conv3d_k4_wino_fwd: x [B,8,n^3]")
    y = out if out is not None else torch.empty((B, 8, di - 3, di - 3, di - 3), device=x.device)
    check(lib().nvf_conv3d_k4_wino_fwd(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, di, int(ppc), _stream()),
          "nvf_conv3d_k4_wino_fwd")
    return y


def pack_wino16_k4(w):
    """A fragments of nvf_conv3d_k4_wino16_bwd / _fwd from a gather-form weight [16][64][16] (one pack_mfma_all job)."""
    wp = torch.empty(int(lib().nvf_pack_wino16_k4_floats()), device=w.device)
    pack_mfma_all([(w, wp, 41, 16, 16)])
    return wp


def conv3d_k4_wino16_bwd(dy, wp, mask, out=None, ppc=0, bias_part=None):
    """conv3d_k4_wino_bwd for 16 -> 16 channels (the wide decoder): dx = mask > 0 ? conv_full(dy, w) : 0.  ``bias_part``:
    device address of the slabs that receive the 16 channel sums of dx per work unit; returns (dx, number of slabs) then."""
    import ctypes
    _f32(dy, wp, mask)
    B, c, di = dy.shape[0], dy.shape[1], dy.shape[2]
    shape = (B, 16, di + 3, di + 3, di + 3)
    if c != 16 or tuple(mask.shape) != shape:
        raise RuntimeError("conv3d_k4_wino16_bwd: dy [B,16,n^3], mask [B,16,(n+3)^3]")
    dx = out if out is not None else torch.empty(shape, device=dy.device)
    nparts = ctypes.c_int(0)
    check(lib().nvf_conv3d_k4_wino16_bwd(_ptr(dy), _ptr(wp), _ptr(dx), _ptr(mask), B, di, int(ppc),
                                         None if bias_part is None else int(bias_part),
                                         ctypes.byref(nparts) if bias_part is not None else None, _stream()),
          "nvf_conv3d_k4_wino16_bwd")
    return dx if bias_part is None else (dx, nparts.value)


def conv3d_k4_wino16_fwd(x, wp, bias, out=None, ppc=0):
    """conv3d_k4_wino_fwd for 16 -> 16 channels: relu(conv3d(x, w) + bias), training steps only."""
    _f32(x, wp, bias)
    B, c, di = x.shape[0], x.shape[1], x.shape[2]
    if c != 16:
        raise RuntimeError("conv3d_k4_wino16_fwd: x [B,16,n^3]")
    y = out if out is not None else torch.empty((B, 16, di - 3, di - 3, di - 3), device=x.device)
    check(lib().nvf_conv3d_k4_wino16_fwd(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, di, int(ppc), _stream()),
          "nvf_conv3d_k4_wino16_fwd")
    return y


def wgrad16_k4_wino_partial(dy, x, slabs, max_slabs=256, zsplit=0):
    """Partial sums of the weight gradient of a valid 4^3 convolution with 16 -> 16 channels in the Winograd (y, x) form
    (wgrad16_wino.hip): writes <= max_slabs slabs of 16384 floats at device address ``slabs`` (WgradBatch.reserve) and
    returns their number -- a WgradBatch.add_job(slabs, dw, n, 16384) adds them."""
    import ctypes
    _f32(dy, x)
    B, c, w = dy.shape[0], dy.shape[1], dy.shape[2]
    if c != 16 or tuple(x.shape) != (B, 16, w + 3, w + 3, w + 3):
        raise RuntimeError("wgrad16_k4_wino_partial: dy [B,16,w^3], x [B,16,(w+3)^3]")
    n = ctypes.c_int(0)
    check(lib().nvf_wgrad16_k4_wino_partial(_ptr(dy), _ptr(x), int(slabs), B, w, int(zsplit), int(max_slabs),
                                            ctypes.byref(n), _stream()), "nvf_wgrad16_k4_wino_partial")
    return n.value


def wgrad_k4_wino(dy, x, zsplit=1, want_bias=False):
    """conv2's weight gradient [8,8,4,4,4] (and the bias gradient [8]) in the Winograd (y, x) form: one launch of slabs +
    the fixed-order reduction."""
    _f32(dy, x)
    B = dy.shape[0]
    if tuple(dy.shape[1:]) != (8, 32, 32, 32) or tuple(x.shape) != (B, 8, 35, 35, 35):
        raise RuntimeError("wgrad_k4_wino: dy [B,8,32^3], x [B,8,35^3]")
    dw = torch.empty(8, 8, 4, 4, 4, device=dy.device)
    db = torch.empty(8, device=dy.device) if want_bias else None
    ws = workspace(512 * (4096 + 8) * 4, dy.device, tag="wgrad_wino")
    check(lib().nvf_wgrad_k4_wino(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), ws.data_ptr(), ws.numel(), B,
                                  int(zsplit), _stream()), "nvf_wgrad_k4_wino")
    return (dw, db) if want_bias else dw


_MFMA_VARIANT = int(os.environ.get("NVF_MFMA_VARIANT", "0"))
# slabs (= workgroups) of the big head's weight gradient: 512 (matrix-core kernel inside the five-gradient launch:
# 447.7 us of step kernels against 449.6 with 256 and 450.4 with 1024); the VALU kernels did best with 256
_HEADS_SLABS = int(os.environ.get("NVF_HEADS_SLABS", "512"))


def set_mfma_variant(v):
    global _MFMA_VARIANT
    _MFMA_VARIANT = int(v)


def convT3d_k5s2_fwd(x, w_fwd, bias, cout, pad, act=ACT_NONE, out=None):
    _f32(x, w_fwd, bias)
    B, cin, di, hi, wi = x.shape
    extra = 3 if pad == 0 else 0
    do, ho, wo = 2 * di + extra, 2 * hi + extra, 2 * wi + extra
    if w_fwd.numel() != cin * 125 * cout:
        raise RuntimeError("packed weight size does not match (cin, 5, cout)")
    y = out if out is not None else torch.empty((B, cout, do, ho, wo), device=x.device)
    check(lib().nvf_convT3d_k5s2_fwd(_ptr(x), _ptr(w_fwd), _ptr(bias), _ptr(y), B, cin, cout, pad, di, hi, wi, do,
                                     ho, wo, act, _NAIVE, _stream()), "nvf_convT3d_k5s2_fwd")
    return y


def pack_convT_mfma(w_fwd, cin, out=None, edge_rows=False):
    """MFMA A-fragments of a k5 s2 transposed-conv weight in the packed forward layout [cin][125][8].  ``edge_rows``: the
    training-step form (kernel variant 15: the kx = 4 taps on rows (co, ey), 65 fragments per channel group; pack kind 12)."""
    _f32(w_fwd, out)
    if edge_rows:
        n = (cin // 4) * 65 * 64
        wp = out if out is not None else torch.empty(n, device=w_fwd.device)
        if w_fwd.numel() != cin * 125 * 8 or wp.numel() != n:
            raise RuntimeError("pack_convT_mfma: weight size does not match (cin, 5, 8)")
        pack_mfma_all([(w_fwd, wp, 12, cin, 8)])
        return wp
    n = int(lib().nvf_pack_convT_mfma_floats(cin))
    wp = out if out is not None else torch.empty(n, device=w_fwd.device)
    if w_fwd.numel() != cin * 125 * 8 or wp.numel() != n:
        raise RuntimeError("pack_convT_mfma: weight size does not match (cin, 5, 8)")
    check(lib().nvf_pack_convT_mfma(_ptr(w_fwd), cin, 8, _ptr(wp), _stream()), "nvf_pack_convT_mfma")
    return wp


def convT3d_k5s2_mfma(x, wp, bias, act=ACT_NONE, out=None, variant=None):
    """Matrix-core transposed convolution k5 s2 padding 0, 8 output channels."""
    _f32(x, wp, bias)
    B, cin, di = x.shape[0], x.shape[1], x.shape[2]
    do = 2 * di + 3
    y = out if out is not None else torch.empty((B, 8, do, do, do), device=x.device)
    check(lib().nvf_convT3d_k5s2_mfma(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, cin, 8, di, act,
                                      _MFMA_VARIANT if variant is None else int(variant), _stream()),
          "nvf_convT3d_k5s2_mfma")
    return y


def pack_convT16_mfma(w_fwd, cin, cout=16, out=None):
    """MFMA A-fragments of a k5 s2 transposed-conv weight with 16 / 32 output channels, packed forward layout
    [cin][125][cout]."""
    _f32(w_fwd, out)
    n = int(lib().nvf_pack_convT16_mfma_floats(cin, cout))
    wp = out if out is not None else torch.empty(n, device=w_fwd.device)
    if w_fwd.numel() != cin * 125 * cout or wp.numel() != n:
        raise RuntimeError("pack_convT16_mfma: weight size does not match (cin, 5, cout)")
    check(lib().nvf_pack_convT16_mfma(_ptr(w_fwd), cin, cout, _ptr(wp), _stream()), "nvf_pack_convT16_mfma")
    return wp


def convT3d_k5s2_mfma16(x, wp, bias, act=ACT_NONE, out=None, variant=None, cout=16, pad=0):
    """Matrix-core transposed convolution k5 s2 with 16 / 32 output channels (wide decoder); pad 0 or 2 (+ output
    padding 1)."""
    _f32(x, wp, bias)
    B, cin, di = x.shape[0], x.shape[1], x.shape[2]
    do = 2 * di + (3 if pad == 0 else 0)
    y = out if out is not None else torch.empty((B, cout, do, do, do), device=x.device)
    check(lib().nvf_convT3d_k5s2_mfma16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, cin, cout, pad, di, act,
                                        _MFMA_VARIANT if variant is None else int(variant), _stream()),
          "nvf_convT3d_k5s2_mfma16")
    return y


def pack_s2k5_mfma(w_bwd, cig, cog, out=None):
    """MFMA A-fragments for the backward-data of a k5 s2 transposed conv: w_bwd is its [cout][125][cin] layout."""
    _f32(w_bwd, out)
    n = int(lib().nvf_pack_s2k5_mfma_floats(cig, cog))
    wp = out if out is not None else torch.empty(n, device=w_bwd.device)
    if w_bwd.numel() != cig * 125 * cog or wp.numel() != n:
        raise RuntimeError("pack_s2k5_mfma: weight size does not match (cig, 5, cog)")
    check(lib().nvf_pack_s2k5_mfma(_ptr(w_bwd), cig, cog, _ptr(wp), _stream()), "nvf_pack_s2k5_mfma")
    return wp


def conv3d_s2k5_mfma(g, wp, cog, addend=None, mask=None, out=None, variant=None):
    """Matrix-core backward-data of a k5 s2 padding-0 transposed convolution (stride-2 gather conv).
    ``variant``: tile shape (None = the module default, set_mfma_variant)."""
    _f32(g, wp, addend, mask)
    B, cig, di = g.shape[0], g.shape[1], g.shape[2]
    do = (di - 3) // 2
    dx = out if out is not None else torch.empty((B, cog, do, do, do), device=g.device)
    for t in (addend, mask):
        if t is not None and t.shape != dx.shape:
            raise RuntimeError("addend/mask shape must equal the output shape")
    check(lib().nvf_conv3d_s2k5_mfma(_ptr(g), _ptr(wp), _ptr(dx), _ptr(addend), _ptr(mask), B, cig, cog, di, do,
                                     _MFMA_VARIANT if variant is None else int(variant), _stream()),
          "nvf_conv3d_s2k5_mfma")
    return dx


def pack_g16_mfma(gather_w, cin, cout, k, out=None):
    """MFMA A-fragments of a packed gather weight [cin][k^3][cout] with cout a multiple of 16 (conv16_mfma.hip)."""
    _f32(gather_w, out)
    n = int(lib().nvf_pack_g16_mfma_floats(cin, cout, k))
    wp = out if out is not None else torch.empty(n, device=gather_w.device)
    if gather_w.numel() != cin * k ** 3 * cout or wp.numel() != n:
        raise RuntimeError("pack_g16_mfma: weight size does not match (cin, k, cout)")
    check(lib().nvf_pack_g16_mfma(_ptr(gather_w), cin, cout, k, _ptr(wp), _stream()), "nvf_pack_g16_mfma")
    return wp


def conv3d_g16_mfma(x, wp, bias, cout, k, stride, pad, out_spatial, act=ACT_NONE, addend=None, mask=None, out=None,
                    variant=None):
    """Matrix-core gather convolution with 16 / 32 output channels: same contract as conv3d_gather."""
    _f32(x, wp, bias, addend, mask)
    B, cin, di, hi, wi = x.shape
    do, ho, wo = out_spatial
    if wp.numel() != int(lib().nvf_pack_g16_mfma_floats(cin, cout, k)):
        raise RuntimeError("packed weight size does not match (cin, k, cout)")
    y = out if out is not None else torch.empty((B, cout, do, ho, wo), device=x.device)
    for t in (addend, mask):
        if t is not None and t.shape != y.shape:
            raise RuntimeError("addend/mask shape must equal the output shape")
    check(lib().nvf_conv3d_g16_mfma(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(addend), _ptr(mask), B, cin, cout, k,
                                    stride, pad, di, hi, wi, do, ho, wo, act,
                                    _MFMA_VARIANT if variant is None else int(variant), _stream()),
          "nvf_conv3d_g16_mfma")
    return y


def stem_fwd(x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b):
    """Fused up0 -> IGDN -> conv0 + ReLU for chanstr (8, 16, ...) or (16, 32, ...): returns (a0, h0, y1)."""
    _f32(x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b)
    B, ch = x0.shape[0], x0.shape[1]
    c0, c1 = up0_b.numel(), conv0_b.numel()
    a0 = torch.empty((B, c0, 4, 4, 4), device=x0.device)
    h0 = torch.empty((B, c0, 4, 4, 4), device=x0.device)
    y1 = torch.empty((B, c1, 8, 8, 8), device=x0.device)
    check(lib().nvf_stem_fwd(_ptr(x0), _ptr(up0_w_fwd), _ptr(up0_b), _ptr(beta_hat), _ptr(gamma_hat),
                             _ptr(conv0_w_fwd), _ptr(conv0_b), _ptr(a0), _ptr(h0), _ptr(y1), B, ch, c0, c1,
                             _stream()), "nvf_stem_fwd")
    return a0, h0, y1


def stem_latent_fwd(e, lat_w_fwd, lat_bias, lat_beta_hat, lat_gamma_hat, sigma, mu, mode, up0_w_fwd, up0_b, beta_hat,
                    gamma_hat, conv0_w_fwd, conv0_b, block_ids=None, seed=0, step=0, step_dev=None):
    """latent_fwd and stem_fwd in one launch: returns (h, lat, x_rounded, bits[1], a0, h0, y1), bit-identical to the
    two calls."""
    _f32(e, lat_w_fwd, lat_bias, lat_beta_hat, lat_gamma_hat, sigma, mu, up0_w_fwd, up0_b, beta_hat, gamma_hat,
         conv0_w_fwd, conv0_b)
    _chk(block_ids)
    B, ch = e.shape[0], e.shape[1]
    if e[0, 0].numel() != 8:
        raise ValueError("stem_latent_fwd: latents are [B, ch, 2, 2, 2]")
    h, lat, xr = torch.empty_like(e), torch.empty_like(e), torch.empty_like(e)
    bits = torch.empty(1, device=e.device)
    c0, c1 = up0_b.numel(), conv0_b.numel()
    a0 = torch.empty((B, c0, 4, 4, 4), device=e.device)
    h0 = torch.empty((B, c0, 4, 4, 4), device=e.device)
    y1 = torch.empty((B, c1, 8, 8, 8), device=e.device)
    check(lib().nvf_stem_latent_fwd(_ptr(e), _ptr(lat_w_fwd), _ptr(lat_bias), _ptr(lat_beta_hat), _ptr(lat_gamma_hat),
                                    _ptr(block_ids), _ptr(sigma), _ptr(mu), _ptr(h), _ptr(lat), _ptr(xr), _ptr(bits),
                                    0 if mode == "train" else 1, int(seed), int(step), _ptr(step_dev), _ptr(up0_w_fwd),
                                    _ptr(up0_b), _ptr(beta_hat), _ptr(gamma_hat), _ptr(conv0_w_fwd), _ptr(conv0_b),
                                    _ptr(a0), _ptr(h0), _ptr(y1), B, ch, c0, c1, _stream()), "nvf_stem_latent_fwd")
    return h, lat, xr, bits, a0, h0, y1


def stem_bwd(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out=None, dgamma_out=None, dw_up0=None):
    """Fused conv0 backward-data -> IGDN backward -> up0 backward-data (+ IGDN / up0 parameter gradients when
    the three outputs are given).  Returns (da0, dx0)."""
    _f32(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out, dgamma_out, dw_up0)
    B, ch = x0.shape[0], x0.shape[1]
    da0 = torch.empty_like(a0)
    dx0 = torch.empty_like(x0)
    c0, c1 = a0.shape[1], g1.shape[1]
    ws = workspace(lib().nvf_stem_bwd_workspace_for(B, ch, c0, c1), x0.device, "stem")
    check(lib().nvf_stem_bwd(_ptr(g1), _ptr(x0), _ptr(a0), _ptr(conv0_w_bwd), _ptr(up0_w_bwd), _ptr(beta_hat),
                             _ptr(gamma_hat), _ptr(da0), _ptr(dx0), _ptr(dbeta_out), _ptr(dgamma_out),
                             _ptr(dw_up0), _ptr(ws), ws.numel(), B, ch, c0, c1, _stream()), "nvf_stem_bwd")
    return da0, dx0


def stem_bwd_partial(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out, dgamma_out, dw_up0, wg,
                     ctx=None, h0=None, dw_conv0=None):
    """stem_bwd whose final launch is shared: up0's weight-gradient slabs become a job of the WgradBatch ``wg``
    (dw_up0 exists after wg.finish*), the IGDN parameter gradients a deferred final pass.  With ``h0`` and
    ``dw_conv0``: conv0's weight gradient too (one slab per block, another job of ``wg``).  Returns (da0, dx0)."""
    import ctypes
    _f32(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out, dgamma_out, dw_up0, h0, dw_conv0)
    if (h0 is None) != (dw_conv0 is None):
        raise ValueError("stem_bwd_partial: h0 and dw_conv0 go together")
    B, ch = x0.shape[0], x0.shape[1]
    da0 = torch.empty_like(a0)
    dx0 = torch.empty_like(x0)
    c0, c1 = a0.shape[1], g1.shape[1]
    ws = workspace(lib().nvf_stem_bwd_workspace_for(B, ch, c0, c1), x0.device, "stem", ctx)
    slabs, nsl, slabs0 = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_void_p()
    check(lib().nvf_stem_bwd_partial(_ptr(g1), _ptr(x0), _ptr(a0), _ptr(conv0_w_bwd), _ptr(up0_w_bwd), _ptr(beta_hat),
                                     _ptr(gamma_hat), _ptr(da0), _ptr(dx0), _ptr(dbeta_out), _ptr(dgamma_out),
                                     ctypes.byref(slabs), ctypes.byref(nsl), _ptr(ws), ws.numel(), B, ch, c0, c1,
                                     _ptr(h0), ctypes.byref(slabs0) if h0 is not None else None, _ctx(ctx), _stream()),
          "nvf_stem_bwd_partial")
    wg.jobs.append((slabs.value, dw_up0.data_ptr(), nsl.value, dw_up0.numel()))
    if h0 is not None:
        wg.jobs.append((slabs0.value, dw_conv0.data_ptr(), B, dw_conv0.numel()))
    return da0, dx0


def stem_bwd_queue(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out, dgamma_out, dw_up0, db_up0, wg, ctx):
    """stem_bwd_partial with NO launch of its own (nvf_stem_bwd_queue): the work is queued in ``ctx`` and runs as the first
    workgroups of the next WgradBatch.add_trunk5 launch of ``wg`` (which must carry the queued latent tail that consumes
    dx0).  up0's weight gradient and -- from per-block channel sums the stage leaves -- its bias gradient ``db_up0`` become
    reduction jobs of ``wg``.  Returns (da0, dx0): they exist after that launch."""
    import ctypes
    _f32(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, dbeta_out, dgamma_out, dw_up0, db_up0)
    B, ch = x0.shape[0], x0.shape[1]
    da0 = torch.empty_like(a0)
    dx0 = torch.empty_like(x0)
    c0, c1 = a0.shape[1], g1.shape[1]
    ws = workspace(lib().nvf_stem_bwd_workspace_for(B, ch, c0, c1), x0.device, "stem", ctx)
    flags = ctx._ws.get("stem_flags")
    if flags is None:     # one 256-byte line per arrival counter, batch <= 32 (kStemCoopMaxBatch)
        flags = ctx._ws["stem_flags"] = torch.zeros(33 * 64, dtype=torch.int32, device=x0.device)
    slabs, nsl, bias = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_void_p()
    check(lib().nvf_stem_bwd_queue(ctx.ptr, _ptr(g1), _ptr(x0), _ptr(a0), _ptr(conv0_w_bwd), _ptr(up0_w_bwd),
                                   _ptr(beta_hat), _ptr(gamma_hat), _ptr(da0), _ptr(dx0), _ptr(dbeta_out),
                                   _ptr(dgamma_out), ctypes.byref(slabs), ctypes.byref(nsl), ctypes.byref(bias), _ptr(ws),
                                   ws.numel(), _ptr(flags), B, ch, c0, c1, _stream()), "nvf_stem_bwd_queue")
    wg.jobs.append((slabs.value, dw_up0.data_ptr(), nsl.value, dw_up0.numel()))
    wg.jobs.append((bias.value, db_up0.data_ptr(), B, c0))
    return da0, dx0


def wgrad(p, q, k, stride, pad, out_mode=0, out=None, accumulate=False):
    """dw[a][b][k] (out_mode 0) or dw[b][a][flip k] (out_mode 1) = sum p[n,a,i] q[n,b,stride*i-pad+k]."""
    _f32(p, q)
    B, a, dp, hp, wp = p.shape
    b, dq, hq, wq = q.shape[1], q.shape[2], q.shape[3], q.shape[4]
    shape = (a, b, k, k, k) if out_mode == 0 else (b, a, k, k, k)
    dw = out if out is not None else torch.empty(shape, device=p.device)
    nbytes = lib().nvf_wgrad_workspace(B, a, b, k, dp, hp, wp)
    ws = workspace(nbytes, p.device, "wgrad")
    check(lib().nvf_wgrad(_ptr(p), _ptr(q), _ptr(dw), _ptr(ws), ws.numel(), B, a, b, k, stride, pad, dp, hp, wp, dq,
                          hq, wq, out_mode, int(accumulate), _NAIVE, _stream()), "nvf_wgrad")
    return dw


def _parr(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _iarr(vs):
    import ctypes
    return (ctypes.c_int * len(vs))(*[int(v) for v in vs])


def heads3_fwd(xs, w_fwds, biases, act=ACT_SIGMOID):
    """The three classifier heads (inputs [B,16,8^3], [B,8,16^3], [B,8,32^3]) in one launch: returns [p0, p1, p2]."""
    _f32(*xs, *w_fwds, *biases)
    B = xs[0].shape[0]
    ps = [torch.empty((B, 1) + tuple(x.shape[2:]), device=x.device) for x in xs]
    check(lib().nvf_heads3_fwd(_parr(xs), _parr(w_fwds), _parr(biases), _parr(ps), _iarr([x.shape[1] for x in xs]),
                               _iarr([x.shape[-1] for x in xs]), B, act, _stream()), "nvf_heads3_fwd")
    return ps


def heads3_bwd_data(dls, w_bwds, cs, masks):
    """Backward-data of the three heads in one launch: returns [dx0, dx1, dx2] (masks[h] = ReLU mask or None)."""
    _f32(*dls, *w_bwds, *[m for m in masks if m is not None])
    B = dls[0].shape[0]
    dxs = [torch.empty((B, c) + tuple(d.shape[2:]), device=d.device) for d, c in zip(dls, cs)]
    check(lib().nvf_heads3_bwd_data(_parr(dls), _parr(w_bwds), _parr(dxs), _parr(masks), _iarr(cs),
                                    _iarr([d.shape[-1] for d in dls]), B, _stream()), "nvf_heads3_bwd_data")
    return dxs


def heads3_loss_bwd_data(ps, gts, dists, alphas, betas, slots, loss, w_bwds, cs, masks, ctx=None, bias_outs=None):
    """focal_loss_multi (gradients w.r.t. the logits) + heads3_bwd_data in one launch: returns (dls, dxs);
    loss[slots[h]] = focal term of head h (after finals_flush when the final passes are deferred).  batch <= 32.
    ``bias_outs``: three tensors that receive sum(dls[h]) (the heads' bias gradients) from the same launch."""
    import ctypes
    _f32(*ps, *gts, *[d for d in dists if d is not None], *w_bwds, *[m for m in masks if m is not None], loss)
    B = ps[0].shape[0]
    dls = [torch.empty_like(p) for p in ps]
    dxs = [torch.empty((B, c) + tuple(p.shape[2:]), device=p.device) for p, c in zip(ps, cs)]
    ws = workspace(lib().nvf_reduce_workspace(), ps[0].device, "reduce", ctx)
    if bias_outs is not None:
        _f32(*bias_outs)
    check(lib().nvf_heads3_loss_bwd_data_bias(_parr(ps), _parr(gts), _parr(dists), (ctypes.c_float * 3)(*alphas),
                                              (ctypes.c_float * 3)(*betas), _iarr(slots), _ptr(loss), _parr(dls),
                                              _parr(w_bwds), _parr(dxs), _parr(masks), _iarr(cs),
                                              _iarr([p.shape[-1] for p in ps]), B,
                                              None if bias_outs is None else _parr(bias_outs), _ptr(ws), ws.numel(),
                                              _ctx(ctx), _stream()), "nvf_heads3_loss_bwd_data_bias")
    return dls, dxs


def heads3_fwd_loss_bwd_data(xs, w_fwds, biases, gts, dists, alphas, betas, slots, loss, w_bwds, masks, ctx,
                             bias_outs=None, act=ACT_SIGMOID):
    """heads3_fwd + heads3_loss_bwd_data in ONE launch (the forward workgroups hand p to their block's loss workgroups
    inside the launch): returns (ps, dls, dxs), the bits of the two calls.  batch <= 32; ``ctx`` (StepCtx) keeps the
    arrival counters."""
    import ctypes
    _f32(*xs, *w_fwds, *biases, *gts, *[d for d in dists if d is not None], *w_bwds,
         *[m for m in masks if m is not None], loss)
    B = xs[0].shape[0]
    cs = [x.shape[1] for x in xs]
    ps = [torch.empty((B, 1) + tuple(x.shape[2:]), device=x.device) for x in xs]
    dls = [torch.empty_like(p) for p in ps]
    dxs = [torch.empty_like(x) for x in xs]
    ws = workspace(lib().nvf_reduce_workspace(), xs[0].device, "reduce", ctx)
    flags = ctx._ws.get("heads_flags")
    if flags is None:     # one 256-byte line per counter, batch <= 32
        flags = ctx._ws["heads_flags"] = torch.zeros(6 * 32 * 64, dtype=torch.int32, device=xs[0].device)
    if bias_outs is not None:
        _f32(*bias_outs)
    check(lib().nvf_heads3_fwd_loss_bwd_data(
        _parr(xs), _parr(w_fwds), _parr(biases), _parr(ps), int(act), _parr(gts), _parr(dists),
        (ctypes.c_float * 3)(*alphas), (ctypes.c_float * 3)(*betas), _iarr(slots), _ptr(loss), _parr(dls), _parr(w_bwds),
        _parr(dxs), _parr(masks), _iarr(cs), _iarr([x.shape[-1] for x in xs]), B,
        None if bias_outs is None else _parr(bias_outs), _ptr(ws), ws.numel(), flags.data_ptr(), _ctx(ctx), _stream()),
        "nvf_heads3_fwd_loss_bwd_data")
    return ps, dls, dxs


class WgradBatch:
    """Weight gradients of one backward pass with a single reduction launch: ``add`` launches only the partial
    sums (each gradient keeps its own slab region until ``finish``), ``finish`` adds all slabs in one kernel."""

    def __init__(self, device, nbytes=128 << 20, ctx=None):
        self.device, self.jobs, self.offset = device, [], 0
        self.ctx = ctx          # StepCtx: a queued latent tail rides in add_mfma3 / add_trunk5 / finish_with_sums
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._retired = []      # outgrown buffers stay alive: kernels on another stream may still read them

    def reserve(self, nbytes):
        """Device address of ``nbytes`` of slab space that stays untouched until finish() (for partial sums a kernel
        other than the weight-gradient ones leaves: see add_job)."""
        nbytes = (int(nbytes) + 255) // 256 * 256
        if self.offset + nbytes > self.ws.numel():
            if self.jobs:
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(nbytes, 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        base = self.ws.data_ptr() + self.offset
        self.offset += nbytes
        return base

    def add_job(self, base, out, nslab, jtotal):
        """A reduction job over slabs someone else wrote: out[j] = sum of nslab slabs of jtotal floats at ``base``."""
        _f32(out)
        self.jobs.append((int(base), out.data_ptr(), int(nslab), int(jtotal)))

    def add(self, p, q, k, stride, pad, out_mode, out):
        import ctypes
        _f32(p, q, out)
        B, a, dp, hp, wp = p.shape
        b, dq, hq, wq = q.shape[1], q.shape[2], q.shape[3], q.shape[4]
        nbytes = int(lib().nvf_wgrad_workspace(B, a, b, k, dp, hp, wp))
        if self.offset + nbytes > self.ws.numel():
            if self.jobs:    # cannot move slabs that are already in flight: finish them first
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(nbytes, 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        base = self.ws.data_ptr() + self.offset
        nslab = ctypes.c_int(0)
        check(lib().nvf_wgrad_partial(_ptr(p), _ptr(q), _ptr(out), base, nbytes, B, a, b, k, stride, pad, dp, hp, wp,
                                      dq, hq, wq, out_mode, _NAIVE, ctypes.byref(nslab), _stream()),
              "nvf_wgrad_partial")
        self.jobs.append((base, out.data_ptr(), nslab.value, a * b * k ** 3))
        self.offset += (nbytes + 255) // 256 * 256

    def add_mfma3(self, ps, qs, outs):
        """conv2 / up2 / conv1 weight gradients of the narrow trunk: one partial-sum launch, three reduction jobs."""
        import ctypes
        _f32(*ps, *qs, *outs)
        B = ps[0].shape[0]
        jt = (4096, 8000, 4096)
        sizes = [(512 * j * 4 + 255) // 256 * 256 for j in jt]
        if self.offset + sum(sizes) > self.ws.numel():
            if self.jobs:
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(sum(sizes), 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        bases = []
        for sz in sizes:
            bases.append(self.ws.data_ptr() + self.offset)
            self.offset += sz
        nsl = (ctypes.c_int * 3)()
        check(lib().nvf_wgrad_mfma3_partial(_parr(ps), _parr(qs), (ctypes.c_void_p * 3)(*bases), B, nsl, _ctx(self.ctx),
                                            _stream()), "nvf_wgrad_mfma3_partial")
        for h in range(3):
            self.jobs.append((bases[h], outs[h].data_ptr(), nsl[h], jt[h]))

    def _grouped(self, fn, what, ps, qs, outs, jt, max_slabs):
        import ctypes
        _f32(*ps, *qs, *outs)
        n = len(jt)
        sizes = [(max_slabs * j * 4 + 255) // 256 * 256 for j in jt]
        if self.offset + sum(sizes) > self.ws.numel():
            if self.jobs:
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(sum(sizes), 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        bases = []
        for sz in sizes:
            bases.append(self.ws.data_ptr() + self.offset)
            self.offset += sz
        nsl = (ctypes.c_int * n)()
        check(fn(_parr(ps), _parr(qs), (ctypes.c_void_p * n)(*bases), ps[0].shape[0], nsl, _stream()), what)
        for h in range(n):
            self.jobs.append((bases[h], outs[h].data_ptr(), nsl[h], jt[h]))

    def add_up1_conv0(self, ps, qs, outs):
        """up1 / conv0 weight gradients of the narrow trunk: one partial-sum launch, two reduction jobs."""
        self._grouped(lib().nvf_wgrad_up1_conv0_partial, "nvf_wgrad_up1_conv0_partial", ps, qs, outs, (16000, 16000), 512)

    def add_trunk5(self, ps, qs, outs, bias_outs=None, heads=None, sums=None, coef=None):
        """conv2 / up2 / conv1 / up1 / conv0 weight gradients of the narrow trunk: one partial-sum launch (which also
        carries a queued latent tail), five reduction jobs.  ps/qs/outs: add_mfma3's three, then add_up1_conv0's two.
        ``bias_outs`` = (conv2's bias gradient, conv1's): the launch also leaves the channel sums of their dY (two more
        reduction jobs of 8 floats) -- nobody has to read those two tensors again for the bias sums.
        ``heads`` = (dls, xs, outs) of add_heads3 (narrow decoder, needs bias_outs): the heads' weight gradients as
        further workgroups of the same launch.  ``sums`` = (tensors, outs) of multi_channel_sum (needs heads): its first
        pass rides in the launch too, its final pass is queued in the context (self.sums_done is set); ``coef`` =
        (src, live): two floats the launch copies for finish_and_flush_tail's reduction."""
        import ctypes
        _f32(*ps, *qs, *outs)
        B = ps[0].shape[0]
        self.sums_done = False
        jt = (4096, 8000, 4096, 16000, 16000) + ((8, 8) if bias_outs is not None else ())
        sizes = [(512 * j * 4 + 255) // 256 * 256 for j in jt]
        if heads is not None:
            hd, hx, ho = heads
            _f32(*hd, *hx, *ho)
            hcs = [x.shape[1] for x in hx]
            sizes += [(_HEADS_SLABS * c * 27 * 4 + 255) // 256 * 256 for c in hcs]
        if self.offset + sum(sizes) > self.ws.numel():
            if self.jobs:
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(sum(sizes), 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        bases = []
        for sz in sizes:
            bases.append(self.ws.data_ptr() + self.offset)
            self.offset += sz
        nsl = (ctypes.c_int * 5)()
        if heads is not None and sums is not None:
            st, so = sums
            _f32(*st, *so)
            hn = (ctypes.c_int * 3)()
            total = sum(t.shape[1] for t in st)
            ws = workspace(lib().nvf_multi_channel_sum_workspace(total), st[0].device, "mchsum", self.ctx)
            check(lib().nvf_wgrad_trunk5_heads_sums_partial(
                _parr(ps), _parr(qs), (ctypes.c_void_p * 5)(*bases[:5]), (ctypes.c_void_p * 3)(bases[5], None, bases[6]),
                _parr(hd), _parr(hx), (ctypes.c_void_p * 3)(*bases[7:10]), _HEADS_SLABS, _parr(st), _parr(so),
                _iarr([t.shape[1] for t in st]), _iarr([t[0, 0].numel() for t in st]), len(st), _ptr(ws), ws.numel(),
                _ptr(coef[0]) if coef else None, _ptr(coef[1]) if coef else None, B,
                nsl, hn, _ctx(self.ctx), _stream()), "nvf_wgrad_trunk5_heads_sums_partial")
            for h in range(3):
                self.jobs.append((bases[7 + h], ho[h].data_ptr(), hn[h], hcs[h] * 27))
            self.sums_done = True
        elif heads is not None:
            hn = (ctypes.c_int * 3)()
            check(lib().nvf_wgrad_trunk5_heads_partial(
                _parr(ps), _parr(qs), (ctypes.c_void_p * 5)(*bases[:5]), (ctypes.c_void_p * 3)(bases[5], None, bases[6]),
                _parr(hd), _parr(hx), (ctypes.c_void_p * 3)(*bases[7:10]), _HEADS_SLABS, B, nsl, hn, _ctx(self.ctx),
                _stream()), "nvf_wgrad_trunk5_heads_partial")
            for h in range(3):
                self.jobs.append((bases[7 + h], ho[h].data_ptr(), hn[h], hcs[h] * 27))
        elif bias_outs is None:
            check(lib().nvf_wgrad_trunk5_partial(_parr(ps), _parr(qs), (ctypes.c_void_p * 5)(*bases[:5]), B, nsl,
                                                 _ctx(self.ctx), _stream()), "nvf_wgrad_trunk5_partial")
        else:
            _f32(*bias_outs)
            check(lib().nvf_wgrad_trunk5_partial_bias(_parr(ps), _parr(qs), (ctypes.c_void_p * 5)(*bases[:5]),
                                                      (ctypes.c_void_p * 3)(bases[5], None, bases[6]), B, nsl,
                                                      _ctx(self.ctx), _stream()), "nvf_wgrad_trunk5_partial_bias")
        for h in range(5):
            self.jobs.append((bases[h], outs[h].data_ptr(), nsl[h], jt[h]))
        if bias_outs is not None:
            self.jobs.append((bases[5], bias_outs[0].data_ptr(), nsl[0], 8))
            self.jobs.append((bases[6], bias_outs[1].data_ptr(), nsl[2], 8))

    def add_heads3(self, dls, xs, outs, max_slabs=None):
        """Weight gradients of the three classifier heads: one partial-sum launch, three reduction jobs."""
        import ctypes
        if max_slabs is None:
            max_slabs = _HEADS_SLABS
        _f32(*dls, *xs, *outs)
        B = xs[0].shape[0]
        cs = [x.shape[1] for x in xs]
        sizes = [(max_slabs * c * 27 * 4 + 255) // 256 * 256 for c in cs]
        if self.offset + sum(sizes) > self.ws.numel():
            if self.jobs:
                self.finish()
            self._retired.append(self.ws)
            self.ws = torch.empty(max(sum(sizes), 2 * self.ws.numel()), dtype=torch.uint8, device=self.device)
        bases = []
        for sz in sizes:
            bases.append(self.ws.data_ptr() + self.offset)
            self.offset += sz
        nsl = (ctypes.c_int * 3)()
        check(lib().nvf_heads3_wgrad_partial(_parr(dls), _parr(xs), (ctypes.c_void_p * 3)(*bases), _iarr(cs),
                                             _iarr([x.shape[-1] for x in xs]), B, max_slabs, nsl, _stream()),
              "nvf_heads3_wgrad_partial")
        for h in range(3):
            self.jobs.append((bases[h], outs[h].data_ptr(), nsl[h], cs[h] * 27))

    def finish_with_sums(self, tensors, outs, addends=None, adam=None):
        """finish() and multi_channel_sum(tensors, outs) with the reduction and the partial bias sums in one launch.
        ``addends``: {gradient data_ptr: tensor added to that gradient}; ``adam``: NvfAdamFuse applied to every weight
        gradient element written (both need the one-launch form: <= 16 jobs)."""
        import ctypes
        jobs = self.jobs
        if not jobs or len(jobs) > 16 or not tensors:
            if addends or adam is not None:
                raise RuntimeError("finish_with_sums: addends / fused Adam need the one-launch reduction")
            self.finish()
            if tensors:
                multi_channel_sum(tensors, outs, ctx=self.ctx)
            return
        self.jobs, self.offset = [], 0
        _f32(*tensors)
        _f32(*outs)
        n, nt = len(jobs), len(tensors)
        total = sum(t.shape[1] for t in tensors)
        ws = workspace(lib().nvf_multi_channel_sum_workspace(total), tensors[0].device, "mchsum", self.ctx)
        adds = None
        if addends:
            adds = (ctypes.c_void_p * n)(*[(addends[j[1]].data_ptr() if (j[1] in addends and j[2] > 0) else None)
                                           for j in jobs])
        check(lib().nvf_wgrad_reduce_multi_and_sums_fused(
            (ctypes.c_void_p * n)(*[j[0] for j in jobs]), (ctypes.c_void_p * n)(*[j[1] for j in jobs]),
            (ctypes.c_int * n)(*[j[2] for j in jobs]), (ctypes.c_int * n)(*[j[3] for j in jobs]), n, adds,
            None if adam is None else ctypes.byref(adam),
            _parr(tensors), _parr(outs), _iarr([t.shape[1] for t in tensors]), _iarr([t[0, 0].numel() for t in tensors]),
            nt, tensors[0].shape[0], _ptr(ws), ws.numel(), _ctx(self.ctx), _stream()),
            "nvf_wgrad_reduce_multi_and_sums_fused")

    def finish_and_flush_tail(self, addends, adam, tail, ranges):
        """The slab reduction (addends, fused optimiser) and the context's queued final passes + step tail in ONE
        launch (nvf_wgrad_reduce_finals_tail): for steps whose partial bias sums were made earlier (add_trunk5 sums=)."""
        import ctypes
        jobs = self.jobs
        n = len(jobs)
        if not (0 < n <= 16) or self.ctx is None:
            raise RuntimeError("finish_and_flush_tail: 1..16 reduction jobs and a step context are required")
        self.jobs, self.offset = [], 0
        adds = None
        if addends:
            adds = (ctypes.c_void_p * n)(*[(addends[j[1]].data_ptr() if (j[1] in addends and j[2] > 0) else None)
                                           for j in jobs])
        flat = [int(v) for r in ranges for v in r]
        arr = (ctypes.c_int64 * max(len(flat), 1))(*flat)
        check(lib().nvf_wgrad_reduce_finals_tail(
            (ctypes.c_void_p * n)(*[j[0] for j in jobs]), (ctypes.c_void_p * n)(*[j[1] for j in jobs]),
            (ctypes.c_int * n)(*[j[2] for j in jobs]), (ctypes.c_int * n)(*[j[3] for j in jobs]), n, adds,
            ctypes.byref(adam), _ctx(self.ctx), ctypes.byref(tail), arr, len(ranges), _stream()),
            "nvf_wgrad_reduce_finals_tail")

    def finish_and_flush(self, addends):
        """The slab reduction (with addends) and the context's queued final passes in ONE launch, no optimiser
        (nvf_wgrad_reduce_finals): data-parallel steps, whose all-reduce and step tail follow."""
        import ctypes
        jobs = self.jobs
        n = len(jobs)
        if not (0 < n <= 16) or self.ctx is None:
            raise RuntimeError("finish_and_flush: 1..16 reduction jobs and a step context are required")
        self.jobs, self.offset = [], 0
        adds = None
        if addends:
            adds = (ctypes.c_void_p * n)(*[(addends[j[1]].data_ptr() if (j[1] in addends and j[2] > 0) else None)
                                           for j in jobs])
        check(lib().nvf_wgrad_reduce_finals(
            (ctypes.c_void_p * n)(*[j[0] for j in jobs]), (ctypes.c_void_p * n)(*[j[1] for j in jobs]),
            (ctypes.c_int * n)(*[j[2] for j in jobs]), (ctypes.c_int * n)(*[j[3] for j in jobs]), n, adds,
            _ctx(self.ctx), _stream()), "nvf_wgrad_reduce_finals")

    def finish(self):
        import ctypes
        jobs, self.jobs, self.offset = self.jobs, [], 0
        for i in range(0, len(jobs), 16):
            chunk = jobs[i:i + 16]
            n = len(chunk)
            check(lib().nvf_wgrad_reduce_multi((ctypes.c_void_p * n)(*[j[0] for j in chunk]),
                                               (ctypes.c_void_p * n)(*[j[1] for j in chunk]),
                                               (ctypes.c_int * n)(*[j[2] for j in chunk]),
                                               (ctypes.c_int * n)(*[j[3] for j in chunk]), n, _stream()),
                  "nvf_wgrad_reduce_multi")


def channel_sum(x, out=None, accumulate=False):
    _f32(x)
    B, c = x.shape[0], x.shape[1]
    spatial = x[0, 0].numel()
    o = out if out is not None else torch.empty(c, device=x.device)
    ws = workspace(lib().nvf_channel_sum_workspace(c), x.device, "chsum")
    check(lib().nvf_channel_sum(_ptr(x), _ptr(o), _ptr(ws), ws.numel(), B, c, spatial, int(accumulate), _stream()),
          "nvf_channel_sum")
    return o


def multi_channel_sum(tensors, outs, ctx=None):
    """outs[i][c] = sum over batch and space of tensors[i][:, c]; all tensors share the batch size."""
    import ctypes
    _f32(*tensors)
    _f32(*outs)
    n = len(tensors)
    B = tensors[0].shape[0]
    xs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    os_ = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    cs = (ctypes.c_int * n)(*[t.shape[1] for t in tensors])
    sp = (ctypes.c_int * n)(*[t[0, 0].numel() for t in tensors])
    total = sum(t.shape[1] for t in tensors)
    ws = workspace(lib().nvf_multi_channel_sum_workspace(total), tensors[0].device, "mchsum", ctx)
    check(lib().nvf_multi_channel_sum(xs, os_, cs, sp, n, B, _ptr(ws), ws.numel(), _ctx(ctx), _stream()),
          "nvf_multi_channel_sum")


# ---------------------------------------------------------------- GDN
def gdn_fwd(x, beta_hat, gamma_hat, inverse):
    _f32(x, beta_hat, gamma_hat)
    B, c = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    check(lib().nvf_gdn_fwd(_ptr(x), _ptr(beta_hat), _ptr(gamma_hat), _ptr(y), B, c, x[0, 0].numel(), int(inverse),
                            _stream()), "nvf_gdn_fwd")
    return y


def gdn_bwd(x, beta_hat, gamma_hat, dy, inverse, dbeta_out=None, dgamma_out=None):
    _f32(x, beta_hat, gamma_hat, dy, dbeta_out, dgamma_out)
    B, c = x.shape[0], x.shape[1]
    dx = torch.empty_like(x)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty_like(beta_hat)
    dgamma = dgamma_out if dgamma_out is not None else torch.empty_like(gamma_hat)
    ws = workspace(lib().nvf_gdn_bwd_workspace(c), x.device, "gdn")
    check(lib().nvf_gdn_bwd(_ptr(x), _ptr(beta_hat), _ptr(gamma_hat), _ptr(dy), _ptr(dx), _ptr(dbeta), _ptr(dgamma),
                            _ptr(ws), ws.numel(), B, c, x[0, 0].numel(), int(inverse), _stream()), "nvf_gdn_bwd")
    return dx, dbeta, dgamma


# ---------------------------------------------------------------- rates
def latent_fwd(e, w_fwd, bias, beta_hat, gamma_hat, sigma, mu, mode, block_ids=None, seed=0, step=0, step_dev=None):
    """conv1x1 + GDN + round/noise + rate in one launch: returns (h, lat, x_rounded, bits[1])."""
    _f32(e, w_fwd, bias, beta_hat, gamma_hat, sigma, mu)
    _chk(block_ids)
    B, c = e.shape[0], e.shape[1]
    h, lat, xr = torch.empty_like(e), torch.empty_like(e), torch.empty_like(e)
    bits = torch.empty(1, device=e.device)
    check(lib().nvf_latent_fwd(_ptr(e), _ptr(w_fwd), _ptr(bias), _ptr(beta_hat), _ptr(gamma_hat), _ptr(block_ids),
                               _ptr(sigma), _ptr(mu), _ptr(h), _ptr(lat), _ptr(xr), _ptr(bits), B, c, e[0, 0].numel(),
                               0 if mode == "train" else 1, int(seed), int(step), _ptr(step_dev), _stream()),
          "nvf_latent_fwd")
    return h, lat, xr, bits


def latent_rate(x, sigma, mu, mode, u=None, block_ids=None, want_grad=False, g_dev=None, g_host=1.0, seed=0,
                step=0, dx_addend=None, dsigma_out=None, dmu_out=None, step_dev=None):
    """Returns (x_rounded, bits[1], dx, dsigma, dmu); the gradient outputs are None unless want_grad."""
    _f32(x, sigma, mu, u, g_dev, dx_addend)
    _chk(block_ids)
    B, c = x.shape[0], x.shape[1]
    xr = torch.empty_like(x)
    bits = torch.empty(1, device=x.device)
    dx = torch.empty_like(x) if want_grad else None
    ds = dsigma_out if dsigma_out is not None else (torch.empty(c, device=x.device) if want_grad else None)
    dm = dmu_out if dmu_out is not None else (torch.empty(c, device=x.device) if want_grad else None)
    check(lib().nvf_latent_rate(_ptr(x), _ptr(u), _ptr(block_ids), _ptr(sigma), _ptr(mu), _ptr(xr), _ptr(bits),
                                _ptr(dx), _ptr(dx_addend), _ptr(ds), _ptr(dm), _ptr(g_dev), float(g_host), B, c,
                                x[0, 0].numel(),
                                0 if mode == "train" else 1, int(seed), int(step), _ptr(step_dev), _stream()),
          "nvf_latent_rate")
    return xr, bits, dx, ds, dm


def weight_rate(kernel, sigma, mu, bits_out=None, dk=None, dsigma=None, dmu=None, g_dev=None, g_host=1.0,
                accumulate=False):
    _f32(kernel, sigma, mu, bits_out, dk, dsigma, dmu, g_dev)
    bits = bits_out if bits_out is not None else torch.empty(1, device=kernel.device)
    check(lib().nvf_weight_rate(_ptr(kernel), kernel.numel(), _ptr(sigma), _ptr(mu), _ptr(bits), _ptr(dk),
                                _ptr(dsigma), _ptr(dmu), _ptr(g_dev), float(g_host), int(accumulate), _stream()),
          "nvf_weight_rate")
    return bits


def latent_tail_queue(ctx, lat, sigma, mu, mode, block_ids, dx_addend, dlat, dsigma, dmu, g_dev, g_host, seed, step,
                      step_dev, h, beta_hat, gamma_hat, dh, dbeta_hat, dgamma_hat, e, dw, db):
    """latent_rate (gradient) -> gdn_bwd -> 1x1x1 weight / bias gradient as one workgroup of the next
    WgradBatch.add_trunk5 / add_mfma3 / finish_with_sums launch of a WgradBatch that holds the same StepCtx (see
    include/nvf_hip.h).  Outputs are written in place; the caller keeps every tensor alive until that launch has
    been enqueued."""
    _f32(lat, sigma, mu, dx_addend, dlat, dsigma, dmu, g_dev, h, beta_hat, gamma_hat, dh, dbeta_hat, dgamma_hat, e, dw, db)
    _chk(block_ids)
    B, c = lat.shape[0], lat.shape[1]
    check(lib().nvf_latent_tail_queue(ctx.ptr, _ptr(lat), _ptr(block_ids), _ptr(sigma), _ptr(mu), _ptr(dx_addend), _ptr(dlat),
                                      _ptr(dsigma), _ptr(dmu), _ptr(g_dev), float(g_host),
                                      0 if mode == "train" else 1, int(seed), int(step), _ptr(step_dev), _ptr(h),
                                      _ptr(beta_hat), _ptr(gamma_hat), _ptr(dh), _ptr(dbeta_hat), _ptr(dgamma_hat),
                                      _ptr(e), _ptr(dw), _ptr(db), B, c, lat[0, 0].numel()), "nvf_latent_tail_queue")


def weight_rate_batch(kernels, dks, sigma, mu, bits, dsigma=None, dmu=None, g_dev=None, g_host=1.0, ctx=None):
    """bits[l] for every kernel; dks[l] (or None) += g dbits/dk; dsigma/dmu overwritten with the layer sum."""
    import ctypes
    _f32(*kernels)
    n = len(kernels)
    ks = (ctypes.c_void_p * n)(*[k.data_ptr() for k in kernels])
    ds = (ctypes.c_void_p * n)(*[(d.data_ptr() if d is not None else None) for d in dks]) if dks is not None else None
    ns = (ctypes.c_int * n)(*[k.numel() for k in kernels])
    ws = workspace(lib().nvf_weight_rate_batch_workspace(), kernels[0].device, "wrate", ctx)
    check(lib().nvf_weight_rate_batch(ks, ds, ns, n, _ptr(sigma), _ptr(mu), _ptr(bits), _ptr(dsigma), _ptr(dmu),
                                      _ptr(g_dev), float(g_host), _ptr(ws), ws.numel(), _ctx(ctx), _stream()),
          "nvf_weight_rate_batch")
    return bits


# ---------------------------------------------------------------- losses / metrics
def focal_loss(p, gt, dist, alpha, beta=0.0, want_grad=False, g_dev=None, g_host=1.0, loss_out=None,
               accumulate=False, chain_sigmoid=False, dp_out=None):
    _f32(p, gt, dist, g_dev, loss_out)
    loss = loss_out if loss_out is not None else torch.empty(1, device=p.device)
    dp = dp_out if dp_out is not None else (torch.empty_like(p) if want_grad else None)
    ws = workspace(lib().nvf_reduce_workspace(), p.device, "reduce")
    check(lib().nvf_focal_loss(_ptr(p), _ptr(gt), _ptr(dist), float(alpha), float(beta), _ptr(loss), _ptr(dp),
                               _ptr(g_dev), float(g_host), _ptr(ws), ws.numel(), p.numel(), int(accumulate),
                               int(chain_sigmoid), _stream()), "nvf_focal_loss")
    return loss, dp


def focal_loss_multi(terms, loss_out, chain_sigmoid=True, ctx=None):
    """terms: list of (p, gt, dist-or-None, alpha, beta); returns the list of gradients w.r.t. p (or the logits)."""
    import ctypes
    n = len(terms)
    for p, gt, dist, _, _ in terms:
        _f32(p, gt, dist)
    _f32(loss_out)
    dps = [torch.empty_like(t[0]) for t in terms]
    arr = lambda xs: (ctypes.c_void_p * n)(*[(x.data_ptr() if x is not None else None) for x in xs])
    al = (ctypes.c_float * n)(*[float(t[3]) for t in terms])
    be = (ctypes.c_float * n)(*[float(t[4]) for t in terms])
    ns = (ctypes.c_int64 * n)(*[t[0].numel() for t in terms])
    ws = workspace(lib().nvf_reduce_workspace(), terms[0][0].device, "reduce", ctx)
    check(lib().nvf_focal_loss_multi(arr([t[0] for t in terms]), arr([t[1] for t in terms]),
                                     arr([t[2] for t in terms]), arr(dps), al, be, ns, n, _ptr(loss_out),
                                     int(chain_sigmoid), _ptr(ws), ws.numel(), _ctx(ctx), _stream()),
          "nvf_focal_loss_multi")
    return dps


def metrics(p, gt, dist, thh_acc, thh_sse, out=None, accumulate=False, ctx=None):
    """out[0..5] (+)= tp, ap, tn, an at thh_acc; sse, denom at thh_sse (utils/loss.py:74-84, 113-121).  With ``ctx``
    (an open StepCtx.begin) the final pass joins the deferred ones: ``out`` exists after ctx.flush()."""
    _f32(p, gt, dist, out)
    o = out if out is not None else torch.empty(6, device=p.device)
    ws = workspace(lib().nvf_metrics_workspace(), p.device, "metrics", ctx)   # its own buffer: partials may wait for a flush
    check(lib().nvf_metrics(_ptr(p), _ptr(gt), _ptr(dist), float(thh_acc), float(thh_sse), _ptr(o), _ptr(ws),
                            ws.numel(), p.numel(), int(accumulate), _ctx(ctx), _stream()), "nvf_metrics")
    return o


def metrics3(ps, gts, dists, thh_acc, thh_sse, out=None, ctx=None):
    """The six sums of ``metrics`` for up to three (p, gt[, dist]) pairs in ONE launch: out[6 t + k] (overwritten) --
    the main output and the two coarse heads of NVFPCC.py's log lines (:174-179, 214-221)."""
    import ctypes
    n = len(ps)
    dists = list(dists) if dists is not None else [None] * n
    _f32(*ps, *gts, *[d for d in dists if d is not None], out)
    o = out if out is not None else torch.empty(6 * n, device=ps[0].device)
    ws = workspace(lib().nvf_metrics_workspace(), ps[0].device, "metrics", ctx)
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    check(lib().nvf_metrics3(arr(ps), arr(gts), arr(dists), (ctypes.c_int64 * n)(*[p.numel() for p in ps]), n,
                             float(thh_acc), float(thh_sse), _ptr(o), _ptr(ws), ws.numel(), _ctx(ctx), _stream()),
          "nvf_metrics3")
    return o


def sigmoid_bwd(dp, p):
    _f32(dp, p)
    out = torch.empty_like(p)
    check(lib().nvf_sigmoid_bwd(_ptr(dp), _ptr(p), _ptr(out), p.numel(), _stream()), "nvf_sigmoid_bwd")
    return out


def relu_bwd(dy, y):
    _f32(dy, y)
    out = torch.empty_like(y)
    check(lib().nvf_relu_bwd(_ptr(dy), _ptr(y), _ptr(out), y.numel(), _stream()), "nvf_relu_bwd")
    return out


def squared_error_map(p, dist, thh):
    _f32(p, dist)
    B = p.shape[0]
    spatial = p[0].numel()
    out = torch.empty((B, 2) + tuple(p.shape[2:]), device=p.device)
    check(lib().nvf_squared_error_map(_ptr(p), _ptr(dist), float(thh), _ptr(out), B, spatial, _stream()),
          "nvf_squared_error_map")
    return out


def maxpool2(x):
    _f32(x)
    B, c, d, h, w = x.shape
    y = torch.empty((B, c, d // 2, h // 2, w // 2), device=x.device)
    check(lib().nvf_maxpool2(_ptr(x), _ptr(y), B * c, d, h, w, _stream()), "nvf_maxpool2")
    return y


# ---------------------------------------------------------------- optimiser / rows / rng
def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    _f32(p, g, m, v)
    check(lib().nvf_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
                              float(eps), int(step), _stream()), "nvf_adam_step")


def adam_coefficients(lr, step, beta1=0.9, beta2=0.999):
    """The two floats [lr / (1 - beta1^t), sqrt(1 - beta2^t)] step_tail reads from device memory."""
    import ctypes
    c = (ctypes.c_float * 2)()
    check(lib().nvf_adam_coefficients(float(lr), float(beta1), float(beta2), int(step), c), "nvf_adam_coefficients")
    return float(c[0]), float(c[1])


def adam_coefficients_n(lr, step0, n, beta1=0.9, beta2=0.999):
    """adam_coefficients of steps step0 .. step0 + n - 1 as a float32 array [n, 2] (one call)."""
    import numpy as np
    out = np.empty((n, 2), np.float32)
    check(lib().nvf_adam_coefficients_n(float(lr), float(beta1), float(beta2), int(step0), int(n),
                                        out.ctypes.data), "nvf_adam_coefficients_n")
    return out


def step_tail(p, g, m, v, coef_dev=None, coef_host=(0.0, 0.0), loss_terms=None, lbits=None, nbits=None,
              inv_npts_dev=None, inv_npts_host=1.0, nbits_scale=1.0, counts=None, acc=None, done=None, sched=None,
              beta1=0.9, beta2=0.999, eps=1e-8):
    """Adam with its two step-dependent scalars in device memory (graph-capturable), the epoch's running sums and
    non-finite counters in ``acc`` [16], and the hand-over to the next step of a device-resident schedule
    (``sched`` = (buf, rows, cursor, words); see NvfStepTail in include/nvf_hip.h)."""
    import ctypes
    a = step_tail_args(p, g, m, v, coef_dev, coef_host, loss_terms, lbits, nbits, inv_npts_dev, inv_npts_host,
                       nbits_scale, counts, acc, done, sched, beta1, beta2, eps)
    check(lib().nvf_step_tail(ctypes.byref(a), _stream()), "nvf_step_tail")


def step_tail_args(p, g, m, v, coef_dev=None, coef_host=(0.0, 0.0), loss_terms=None, lbits=None, nbits=None,
                   inv_npts_dev=None, inv_npts_host=1.0, nbits_scale=1.0, counts=None, acc=None, done=None, sched=None,
                   beta1=0.9, beta2=0.999, eps=1e-8):
    """The NvfStepTail struct behind step_tail / StepCtx.flush_tail (the tensors must outlive the launch)."""
    from ._lib import NvfStepTail
    _f32(p, g, m, v, coef_dev, loss_terms, lbits, nbits, inv_npts_dev, counts, acc)
    a = NvfStepTail()
    a.p, a.g, a.m, a.v, a.n = _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel()
    a.coef_dev, a.coef0_host, a.coef1_host = _ptr(coef_dev), float(coef_host[0]), float(coef_host[1])
    a.beta1, a.beta2, a.eps = float(beta1), float(beta2), float(eps)
    a.nnb = 0 if nbits is None else nbits.numel()
    a.loss_terms, a.lbits, a.nbits = _ptr(loss_terms), _ptr(lbits), _ptr(nbits)
    a.inv_npts_dev, a.inv_npts_host, a.nbits_scale = _ptr(inv_npts_dev), float(inv_npts_host), float(nbits_scale)
    a.counts, a.acc = _ptr(counts), _ptr(acc)
    if done is not None:
        _chk(done)
        a.done = done.data_ptr()
    if sched is not None:
        buf, rows, cursor, words = sched
        _chk(buf, rows, cursor)
        a.sched_buf, a.sched_rows, a.sched_cursor, a.sched_words = buf.data_ptr(), rows.data_ptr(), cursor.data_ptr(), int(words)
    return a


def adam_fuse_args(tail):
    """NvfAdamFuse with the optimiser state of an NvfStepTail (for WgradBatch.finish_with_sums)."""
    from ._lib import NvfAdamFuse
    a = NvfAdamFuse()
    a.g_base, a.p_base, a.m_base, a.v_base, a.n = tail.g, tail.p, tail.m, tail.v, tail.n
    a.coef_dev, a.coef0_host, a.coef1_host = tail.coef_dev, tail.coef0_host, tail.coef1_host
    a.beta1, a.beta2, a.eps = tail.beta1, tail.beta2, tail.eps
    a.bad_count = (tail.acc + 6 * 4) if tail.acc else None
    return a


def rate_job(kernels, dks, sigma, mu, part, g):
    """NvfRateJob: the weight-rate term's partial pass for nvf_step_head (dks[l] receives g dbits/dk, overwritten)."""
    from ._lib import NvfRateJob
    _f32(*kernels, *[d for d in dks if d is not None], sigma, mu, part)
    j = NvfRateJob()
    for i, k in enumerate(kernels):
        j.kernel[i], j.n[i] = k.data_ptr(), k.numel()
        j.dk[i] = None if dks[i] is None else dks[i].data_ptr()
    j.nlayers = len(kernels)
    j.sigma, j.mu, j.part, j.g = sigma.data_ptr(), mu.data_ptr(), part.data_ptr(), float(g)
    return j


def weight_rate_final(job, bits, dsigma=None, dmu=None, ctx=None):
    """Final pass of a rate job whose partial sums nvf_step_head computed (queued in ``ctx`` while it defers)."""
    import ctypes
    _f32(bits, dsigma, dmu)
    check(lib().nvf_weight_rate_batch_final(ctypes.byref(job), _ptr(bits), _ptr(dsigma), _ptr(dmu), _ctx(ctx), _stream()),
          "nvf_weight_rate_batch_final")


def gather_rows(src, idx):
    _f32(src)
    _chk(idx)
    if idx.dtype != torch.int64:
        raise RuntimeError("row indices must be int64")
    width = src[0].numel()
    dst = torch.empty((idx.numel(),) + tuple(src.shape[1:]), device=src.device)
    check(lib().nvf_gather_rows(_ptr(src), _ptr(idx), _ptr(dst), idx.numel(), width, _stream()), "nvf_gather_rows")
    return dst


def gather_rows_multi(srcs, idx):
    """[src[idx] for src in srcs] in one launch (all sources indexed by the same int64 vector)."""
    import ctypes
    _f32(*srcs)
    _chk(idx)
    n, rows = len(srcs), idx.numel()
    dsts = [torch.empty((rows,) + tuple(s.shape[1:]), device=s.device) for s in srcs]
    sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    wd = (ctypes.c_int * n)(*[s[0].numel() for s in srcs])
    check(lib().nvf_gather_rows_multi(sp, dp, wd, n, _ptr(idx), rows, _stream()), "nvf_gather_rows_multi")
    return dsts


def scatter_add_rows(src, idx, dst):
    _f32(src, dst)
    _chk(idx)
    check(lib().nvf_scatter_add_rows(_ptr(src), _ptr(idx), _ptr(dst), idx.numel(), src[0].numel(), _stream()),
          "nvf_scatter_add_rows")
    return dst


def uniform(shape, device, seed, stream_id):
    out = torch.empty(shape, device=device)
    check(lib().nvf_uniform(_ptr(out), out.numel(), int(seed), int(stream_id), _stream()), "nvf_uniform")
    return out


# ---------------------------------------------------------------- occupancy -> points
def threshold_points(p, thh, origins=None):
    """p [B,1,D,D,D] -> int32 [n,3] points (origin + (z,y,x)) in (b,z,y,x) raster order, plus per-block counts."""
    _f32(p)
    B, dim = p.shape[0], p.shape[-1]
    counts = torch.empty(B, dtype=torch.int32, device=p.device)
    check(lib().nvf_threshold_count(_ptr(p), float(thh), _ptr(counts), B, dim ** 3, _stream()),
          "nvf_threshold_count")
    offsets = (torch.cumsum(counts, 0, dtype=torch.int32) - counts).contiguous()
    total = int(counts.sum().item())
    coords = torch.empty((max(total, 1), 3), dtype=torch.int32, device=p.device)
    if origins is not None:
        origins = origins.to(device=p.device, dtype=torch.int32).contiguous()
    check(lib().nvf_threshold_compact(_ptr(p), float(thh), _ptr(offsets), _ptr(origins), _ptr(coords), B, dim,
                                      _stream()), "nvf_threshold_compact")
    return coords[:total], counts
