"""Hand-sequenced NVF train / latent / eval steps on the gfx950 kernels.

The reference runs one step as ~1 500 aten ops driven by autograd with ~14 host syncs
(NVFPCC.py:149-250).  Here a step is a fixed list of C-ABI launches on one HIP stream, with

  * every leaf block's grids and the latent table resident in HBM (no DataLoader process),
  * all 28 decoder parameter tensors living in ONE flat buffer (and their gradients in another),
    so Adam is a single fused launch and the data-parallel exchange is a single RCCL all-reduce,
  * all ten layers' effective weights produced by one table-driven launch,
  * the ReLU / sigmoid backward, the head gradients' addition and the loss gradient fused into the
    kernels that produce or consume them,
  * no host synchronisation inside a step (rate coefficients come from per-block point counts
    precomputed on the host).

Result-preserving shortcuts taken from the reference's own data flow: the mini-batch phase never
uses the latent gradients (they are zeroed at NVFPCC.py:226) and the latent phase never uses the
decoder weight gradients (zeroed at NVFPCC.py:150), so each phase skips the half it discards.
"""
import os

import numpy as np
import torch

from . import ops
from ._lib import lib, check
from .network import NoiseState

R, S, NONE = ops.ACT_RELU, ops.ACT_SIGMOID, ops.ACT_NONE
_TAIL = os.environ.get("NVF_TAIL", "1") != "0"   # latent backward as one workgroup of a later launch (0: three launches)
_STEM = os.environ.get("NVF_STEM", "1") != "0"   # fused stem launches (0: per-layer kernels)
_G16 = os.environ.get("NVF_G16", "1") != "0"     # matrix-core kernels of the wide decoder (0: the VALU tile kernels)
_VAR = {k: int(os.environ.get("NVF_VAR_" + k, "0")) for k in ("UP1F", "UP2F", "UP1B", "UP2B", "C1F", "C1B")}   # tile variants
_CONV2_FWD_VAR = int(os.environ.get("NVF_CONV2_FWD_VAR", "0"))   # tile-shape variants of conv_k4_mfma (tuning)
_CONV2_BWD_VAR = int(os.environ.get("NVF_CONV2_BWD_VAR", "0"))
_WINO = os.environ.get("NVF_WINO", "1") != "0"   # conv2's / conv1's backward-data in the Winograd (y, x) form (conv_wino.hip)
_WINO_FWD = os.environ.get("NVF_WINO_FWD", "1") != "0"   # ... and conv2's forward in TRAINING steps (never in eval)
_WINO_C1 = os.environ.get("NVF_WINO_C1", "1") != "0"     # conv1's backward-data as well
# (conv1's training FORWARD in that form: measured slower at batch 16 -- 19.1 us two-set / 16.7 us one-set kernel against
# 12.2 us for the direct kernel, r05 A/B -- 16^3 outputs do not amortise the transforms; used above batch 64 only)
_CONVT_EDGE = os.environ.get("NVF_CONVT_EDGE", "1") != "0"   # up1 / up2 training forward: kx = 4 taps on rows (co, ey)


def _wino_bwd_ppc(g_out):
    """Plane pairs per work unit of the Winograd backward-data at a full batch (0: the kernel's batch-16 default; an explicit
    count without bit 16 = the two-set kernel of conv_wino.hip).  Every choice gives the same bits; batch 917: conv2
    2182 -> 1945 us, conv1 569 -> 387 us (tools/wino_ppc_sweep.py, tools/wino_bench.py --batch 917)."""
    if g_out.shape[0] <= 64:
        return 0
    # (conv1: the two-set kernel of conv_wino.hip, all ten plane pairs per unit -- 387 us against 466 for the one-set kernel
    # with 5 at batch 917, tools/wino_bench.py --fwd --batch 917; the same bits)
    # (conv2 likewise: all 18 pairs in the two-set kernel 1945 us, one-set kernel with 9 / 18: 2203 / 2098 in the same run)
    return 18 if g_out.shape[-1] == 32 else 10


_WINO16 = os.environ.get("NVF_WINO16", "1") != "0"       # the wide decoder's 4^3 layers in that form (conv16_wino.hip)
# ... bias sums of the layer below from its backward-data epilogue: measured neutral (the reduction launch 48.9 -> 44.0 us
# without its 51 MB of re-reads, the two epilogues + 2.7 / + 2.1 us): off by default
_WINO16_BIAS = os.environ.get("NVF_WINO16_BIAS", "0") != "0"
_WINO16_WGRAD = tuple(int(v) for v in os.environ.get("NVF_WINO16_WGRAD", "32,16").split(",") if v)   # ... weight gradients (dY extents)
_GRAPH_LAST = os.environ.get("NVF_GRAPH_LAST_BATCH", "1") != "0"     # the short last mini-batch of an epoch as a graph too
_UP1B_KSPLIT = os.environ.get("NVF_UP1B_KSPLIT", "1") != "0"   # up1's backward-data: channel groups on different waves
_HEADS_FWD_IN_LOSS = os.environ.get("NVF_HEADS_FWD_IN_LOSS", "1") != "0"   # heads' forward inside the loss launch
_HEAD_BIAS_IN_LOSS = os.environ.get("NVF_HEAD_BIAS_IN_LOSS", "1") != "0"   # heads' bias gradients from the loss launch
_SUMS_IN_TRUNK5 = os.environ.get("NVF_SUMS_IN_TRUNK5", "1") != "0"   # partial bias sums inside the five-gradient launch
_HEADS_IN_TRUNK5 = os.environ.get("NVF_HEADS_IN_TRUNK5", "1") != "0"   # heads' weight gradients as workgroups of the five-gradient launch
# the stem's backward (conv0^T -> IGDN' -> up0^T, up0's gradients) as the first workgroups of the five-gradient launch
# instead of two launches in front of it (csrc/stem_bwd.h): it needs g1 only and feeds the latent tail only
_STEM_IN_TRUNK5 = os.environ.get("NVF_STEM_IN_TRUNK5", "1") != "0"
# the stem's FORWARD (latent generator + quantiser + up0 / IGDN / conv0) inside the step head's launch: its workgroups
# derive their weights from the raw parameters, so the two latency-bound launches have nothing to wait for in each other
_STEM_IN_HEAD = os.environ.get("NVF_STEM_IN_HEAD", "1") != "0"


def _NAIVE_OFF():
    """The one-launch groupings bypass the variant switch of the per-layer entry points: only with the tuned kernels."""
    return ops._NAIVE == 0
TRUNK = ("up0", "conv0", "up1", "conv1", "up2", "conv2", "conv2_cls")
HEADS = ("conv1_cls", "conv0_cls")
# layers whose backward-data also runs on the matrix cores: name -> (pair axis, largest batch it is used for)
# (above batch 64 the VALU tile kernel is faster for both: conv1 222 vs 367 us, conv2 1167 vs 1234 us at batch 256)
MFMA_BWD = {"conv1": (2, 64), "conv2": (0, 64)}

_DESC = np.dtype([("kernel", "<u8"), ("kernel_init", "<u8"), ("b", "<u8"), ("b_init", "<u8"), ("w_fwd", "<u8"),
                  ("w_bwd", "<u8"), ("b_eff", "<u8"), ("dim0", "<i4"), ("dim1", "<i4"), ("k3", "<i4"),
                  ("kind", "<i4"), ("quantised", "<i4"), ("layer_id", "<i4"), ("nbias", "<i4"), ("pad", "<i4")])


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Layer:
    __slots__ = ("name", "mod", "kind", "k", "w_fwd", "w_bwd", "b_eff", "gk", "gb", "cin", "cout", "pad", "wp_f",
                 "wp_b", "wp_t", "wp_tr", "wp_s", "wp_gf", "wp_gb", "wp_t16", "wp_w", "wp_wf", "bwd_pair", "bwd_max_batch")


class TrainEngine:
    def __init__(self, net, gt_all, dist_all, n_points_total, lmbda, w1, w2, lr, wemb, emb=None, seed=0, winograd=None):
        assert lib().nvf_layer_desc_size() == _DESC.itemsize
        # winograd=False (or NVF_WINO=0): every 4^3 layer of a training step keeps the direct summation order -- 0.41 ms
        # per step instead of 0.35, and the reference's own three-epoch trajectory reproduced to 1e-7 instead of
        # statistically (tests/test_gpu_engine.py, the trajectory golden)
        self.winograd = _WINO if winograd is None else bool(winograd)
        self.net = net
        self.dev = gt_all.device
        self.gt, self.dist = gt_all.contiguous(), dist_all.contiguous()
        self.N_leaf = self.gt.shape[0]
        self.n_points_total = float(n_points_total)
        self.lmbda, self.w1, self.w2 = float(lmbda), float(w1), float(w2)
        self.lr, self.lr_emb = float(lr), float(lr) * float(wemb)
        self.seed = int(seed)
        # Counter behind both noise sources (weight noise at q = 1, latent rate-proxy noise in every train-mode
        # forward): it advances once per train / latent step WHATEVER q is -- the reference draws a fresh
        # torch.rand_like(x) on every mode='train' forward (network.py:4516), also after --phase_change.  Every rank
        # advances it identically (idle ranks of a short last batch included, NVFPCC.py train()).
        self.noise_step = 0
        self.rate_grad_scale = 1.0   # 1/world_size under data parallelism: the weight-rate term is replicated
        self.grad_hook = None        # called on flat_g between backward and Adam (RCCL all-reduce)
        self.opt_step = 0
        self.emb_step = 0
        self.ch = net.entropy_coder.sigma.shape[1]
        self.channels = net.reconstructor.channels
        # GT pyramid and per-block occupied-voxel counts, once (MultiscaleProcessor, NVFPCC.py:76-88)
        self.gt16 = ops.maxpool2(self.gt)
        self.gt8 = ops.maxpool2(self.gt16)
        self.counts = self.gt.sum(dim=(1, 2, 3, 4)).double().cpu().numpy()
        # latent table + its Adam state (NVFPCC.py:120-124)
        self.emb = (torch.ones(self.N_leaf, self.ch, 2, 2, 2, device=self.dev) if emb is None
                    else emb.detach().to(self.dev).float().contiguous().clone())
        self.emb_m = torch.zeros_like(self.emb)
        self.emb_v = torch.zeros_like(self.emb)
        # Decoder classes ("Decoder configurations"): the two of BASELINE.json have matrix-core / fused
        # launches instantiated for their shapes; ANY other --chanstr runs every layer on the shape-generic kernels
        # (conv3d_gather, convT3d_k5s2_fwd, wgrad_tiled; per-layer stem, heads and gradients) -- same results, not tuned
        chans = tuple(net.reconstructor.channels)
        self.narrow = chans == (8, 16, 8, 8)
        self.wide = chans == (16, 32, 16, 16)
        self.generic = not (self.narrow or self.wide)
        self._flatten_parameters()
        self._build_layers()
        self.last = {}
        # per-step scalars live in device memory while a captured HIP graph replays (see GraphedTrainStep)
        self._step_dev = None     # uint64 noise-step offset
        # second HIP stream: weight gradients, head backward-data, bias sums and the weight-rate term do not sit on
        # the backward-data chain, so they overlap with it (at batch 16 one kernel cannot fill 256 CUs by itself)
        self.side = torch.cuda.Stream(device=self.dev)
        # Measured at batch 16 once the kernels were fast (round 1): running the side jobs concurrently no longer
        # shortens the step (0.828 ms either way) -- every kernel fills the chip on its own -- so the default is
        # one stream; NVF_OVERLAP=1 turns the two-stream schedule back on.
        self.allow_overlap = os.environ.get("NVF_OVERLAP", "0") == "1"
        self.fused_stem = ((self.narrow or self.wide) and net.entropy_coder.sigma.shape[1] <= 8 and _STEM)
        self.fused_latent_stem = True      # latent generator + quantiser ride in the stem's forward launch
        self.overlap = True
        self._g_lat_dev = None    # lambda * w1 / n_pts
        self._wg = None
        self.ctx = ops.StepCtx()  # deferred final passes + queued latent tail of the step in flight (caller-owned)
        self.ctx.set_direct(not self.winograd)
        self.epoch_acc = None     # float[16] epoch sums written by nvf_step_tail (enable_epoch_stats)
        self._tail_done = torch.zeros(2, dtype=torch.int32, device=self.dev)   # nvf_step_tail's arrival counter
        self._coef_live = torch.zeros(2, device=self.dev)   # the step's Adam coefficients, staged outside the step buffer
        self._rate = None         # (NvfRateJob, {gk data_ptr: its weight-rate gradient}, rate_grad_scale) of the step head
        self._rate_ready = False  # the step head of the step in flight carried the weight-rate partial pass
        self._rate_retired = []   # outgrown weight-rate buffers stay referenced (kernels in flight)
        self._graphs_captured = 0  # GraphedTrainStep instances that baked this engine's buffers into a graph
        self.tail_done = False    # the last backward pass applied the optimiser itself (fused tail)
        self._stem_gdn_in_finals = False
        self._stem_pre = None     # (idx pointer, mode, tensors) of a stem forward the step head's launch already ran
        self._tail_ranges = {}    # gradient index ranges no fused launch covers, per set of covered intervals
        self.collective_mode = None   # "graph" / "host": where GraphedTrainStep puts the all-reduce (dist.attach)
        # the three classifier heads go through the one-launch kernels (instantiated for the two decoders of BASELINE.json);
        # the one-launch trunk weight gradients exist for the narrow decoder only
        self.heads3 = self.narrow or self.wide

    # ------------------------------------------------------------------ parameters
    def _flatten_parameters(self):
        named = list(self.net.named_parameters())
        total = sum(p.numel() for _, p in named)
        self.flat_p = torch.empty(total, device=self.dev)
        # gradients + 32 floats behind them: [0, 18) = the step's metric counts (ops.metrics3: main output, head 0,
        # head 1).  They ride in the data-parallel all-reduce of the gradients, so the per-step accuracy RATIOS the
        # reference logs (NVFPCC.py:214-221) are those of the whole mini-batch whatever the number of ranks
        self.flat_gx = torch.zeros(total + 32, device=self.dev)
        self.flat_g = self.flat_gx[:total]
        self.step_counts = self.flat_gx[total:total + 18]
        self.flat_m = torch.zeros(total, device=self.dev)
        self.flat_v = torch.zeros(total, device=self.dev)
        self.slices = {}
        off = 0
        for name, p in named:
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            self.slices[name] = (off, n)
            off += n

    def _g(self, name):
        off, n = self.slices[name]
        return self.flat_g[off:off + n]

    def _build_layers(self):
        rec = self.net.reconstructor
        mods = [("latent", self.net.latent_gen.h_analysis_2, "latent_gen.h_analysis_2")]
        mods += [(n, getattr(rec, n), "reconstructor." + n) for n in TRUNK + HEADS]
        self.layers = {}
        table = np.zeros(len(mods), _DESC)
        for i, (name, m, prefix) in enumerate(mods):
            L = _Layer()
            L.name, L.mod = name, m
            L.kind = 1 if m.kernel.shape[2] == 5 else 0      # k=5 layers are the transposed convs
            L.k = m.kernel.shape[2]
            n = m.kernel.numel()
            L.w_fwd = torch.empty(n, device=self.dev)
            L.w_bwd = torch.empty(n, device=self.dev)
            L.b_eff = torch.empty(m.b.numel(), device=self.dev)
            L.gk, L.gb = self._g(prefix + ".kernel").view(m.kernel.shape), self._g(prefix + ".b")
            L.cin, L.cout, L.pad = m.in_channels, m.out_channels, m.padding
            L.wp_f = L.wp_b = L.wp_t = L.wp_tr = L.wp_s = L.wp_gf = L.wp_gb = L.wp_t16 = L.wp_w = L.wp_wf = None
            L.bwd_pair, L.bwd_max_batch = 2, 0
            if self.narrow and L.k == 5 and L.cin % 4 == 0 and L.cout == 8 and L.pad == 0 and name in ("up1", "up2"):
                # matrix-core form of the padding-0 transposed convolutions: forward, and backward-data (a
                # stride-2 gather convolution with cin output channels)
                L.wp_t = torch.empty(int(lib().nvf_pack_convT_mfma_floats(L.cin)), device=self.dev)
                if self.winograd and _CONVT_EDGE:
                    # the forward of TRAINING steps with the kx = 4 taps on rows (co, ey): 65 instead of 75 A fragments per
                    # channel group, another summation order for the even x outputs (pack kind 12, kernel variant 15);
                    # evaluation / encode / decode keep wp_t
                    L.wp_tr = torch.empty((L.cin // 4) * 65 * 64, device=self.dev)
                if L.cin in (8, 16):
                    L.wp_s = torch.empty(int(lib().nvf_pack_s2k5_mfma_floats(L.cout, L.cin)), device=self.dev)
            if self.narrow and L.k == 4 and L.cin % 4 == 0 and L.cout == 8 and L.cin == 8 and L.pad == 0:
                # matrix-core form of the 4^3 convolutions: MFMA A-fragments, re-packed after every weight preparation
                L.wp_f = torch.empty(int(lib().nvf_pack_mfma_k4_floats(L.cin, 0)), device=self.dev)
                if name in MFMA_BWD:
                    # backward-data mapping: rows pair outputs along z with 4x4 patches (conv1) or along x with
                    # flattened 18-cell rows (conv2; faster than the VALU kernel only while the batch is small)
                    L.bwd_pair, L.bwd_max_batch = MFMA_BWD[name]
                    L.wp_b = torch.empty(int(lib().nvf_pack_mfma_k4_floats(L.cout, L.bwd_pair)), device=self.dev)
                if self.winograd and (name == "conv2" or (name == "conv1" and _WINO_C1)):
                    # backward-data in the reduced-multiplication form (Winograd over (y, x), z pairs on the matrix
                    # cores: 52 us against 85 for the direct form at batch 16)
                    L.wp_w = torch.empty(int(lib().nvf_pack_wino_k4_floats()), device=self.dev)
                if self.winograd and _WINO_FWD and name in ("conv2", "conv1"):
                    # ... and the forward of TRAINING steps (mode 'train': NVFPCC.py:160, 234); the eval / encode / decode
                    # forward keeps the direct fixed-order kernel (bit-exact batch invariance, the occupancy contract).
                    # conv1 only above batch 64 (the full-batch latent step: 207 us against 375 for the direct kernel at
                    # batch 917; at batch 16 its 16^3 outputs do not amortise the transforms: 16.7 against 12.2 us)
                    L.wp_wf = torch.empty(int(lib().nvf_pack_wino_k4_floats()), device=self.dev)
            # wide decoder (16 / 32 channels): the output channels are the MFMA rows (conv16_mfma.hip) -- conv1 / conv2
            # forward and backward-data, and the backward-data of up2 / up1 (stride-2 gather with cin output channels)
            if _G16 and self.wide and L.k == 4 and L.cin == 16 and L.cout == 16 and L.pad == 0 and name in ("conv1", "conv2"):
                L.wp_gf = torch.empty(int(lib().nvf_pack_g16_mfma_floats(16, 16, 4)), device=self.dev)
                L.wp_gb = torch.empty(int(lib().nvf_pack_g16_mfma_floats(16, 16, 4)), device=self.dev)
                if self.winograd and _WINO16:
                    # the Winograd (y, x) form with the 16 output channels as MFMA rows (conv16_wino.hip), training steps
                    # only: conv2 backward-data 257 -> 140 us, forward 170 -> 95, conv1 backward-data 59 -> 33 at batch 16.
                    # conv1's FORWARD stays direct (27 -> 25 us would cost the engine == operator-path agreement its 2e-5:
                    # measured 2.004e-5), so only conv2 gets a forward packing -- and a pack-job slot stays free
                    L.wp_w = torch.empty(int(lib().nvf_pack_wino16_k4_floats()), device=self.dev)
                    if name == "conv2":
                        L.wp_wf = torch.empty(int(lib().nvf_pack_wino16_k4_floats()), device=self.dev)
            if _G16 and self.wide and L.k == 5 and L.cout == 16 and L.cin in (16, 32) and L.pad == 0 and name in ("up1", "up2"):
                L.wp_gb = torch.empty(int(lib().nvf_pack_g16_mfma_floats(16, L.cin, 5)), device=self.dev)
            if _G16 and self.wide and L.k == 5 and (name, L.cin, L.cout, L.pad) in (("up1", 32, 16, 0), ("up2", 16, 16, 0),
                                                                      ("conv0", 16, 32, 2), ("up0", 8, 16, 2)):
                L.wp_t16 = torch.empty(int(lib().nvf_pack_convT16_mfma_floats(L.cin, L.cout)), device=self.dev)
            if _G16 and self.wide and name == "up0" and L.cin == 8 and L.cout == 16 and L.pad == 2:     # 16 -> 8 channels, rows 8..15 zero
                L.wp_gb = torch.empty(int(lib().nvf_pack_g16_mfma_floats(16, 8, 5)), device=self.dev)
            if _G16 and self.wide and name == "conv0" and L.cin == 16 and L.cout == 32 and L.pad == 2:
                L.wp_gb = torch.empty(int(lib().nvf_pack_g16_mfma_floats(32, 16, 5)), device=self.dev)
            self.layers[name] = L
            t = table[i]
            t["kernel"], t["kernel_init"] = m.kernel.data_ptr(), m.kernel_init.data_ptr()
            t["b"], t["b_init"] = m.b.data_ptr(), m.b_init.data_ptr()
            t["w_fwd"], t["w_bwd"], t["b_eff"] = L.w_fwd.data_ptr(), L.w_bwd.data_ptr(), L.b_eff.data_ptr()
            t["dim0"], t["dim1"], t["k3"] = m.kernel.shape[0], m.kernel.shape[1], L.k ** 3
            t["kind"] = L.kind
            t["quantised"] = 1 if name in TRUNK else 0
            t["layer_id"] = m.layer_id
            t["nbias"] = m.b.numel()
        row = {name: i for i, (name, _, _) in enumerate(mods)}
        self._rows = row
        named = list(self.layers.items())
        jobs = [(L.w_fwd, L.wp_f, 0, L.cin, 8) for _, L in named if L.wp_f is not None]
        meta = [(row[nm], 0) for nm, L in named if L.wp_f is not None]
        jobs += [(L.w_bwd, L.wp_b, L.bwd_pair, L.cout, 8) for _, L in named if L.wp_b is not None]
        meta += [(row[nm], 1) for nm, L in named if L.wp_b is not None]
        wk, wc = (41, 16) if self.wide else (40, 8)          # Winograd packing of the decoder class
        jobs += [(L.w_bwd, L.wp_w, wk, L.cout, wc) for _, L in named if L.wp_w is not None]
        meta += [(row[nm], 1) for nm, L in named if L.wp_w is not None]
        jobs += [(L.w_fwd, L.wp_wf, wk, L.cin, wc) for _, L in named if L.wp_wf is not None]
        meta += [(row[nm], 0) for nm, L in named if L.wp_wf is not None]
        jobs += [(L.w_fwd, L.wp_t, 10, L.cin, 8) for _, L in named if L.wp_t is not None]
        meta += [(row[nm], 0) for nm, L in named if L.wp_t is not None]
        jobs += [(L.w_fwd, L.wp_tr, 12, L.cin, 8) for _, L in named if L.wp_tr is not None]
        meta += [(row[nm], 0) for nm, L in named if L.wp_tr is not None]
        jobs += [(L.w_bwd, L.wp_s, 20, L.cout, L.cin) for _, L in named if L.wp_s is not None]
        meta += [(row[nm], 1) for nm, L in named if L.wp_s is not None]
        # 16-row gather forms: kind 30 (k = 4) / 31 (k = 5), c0 = input channels of the gather, c1 = its output channels
        jobs += [(L.w_fwd, L.wp_gf, 30, L.cin, L.cout) for _, L in named if L.wp_gf is not None]
        meta += [(row[nm], 0) for nm, L in named if L.wp_gf is not None]
        jobs += [(L.w_bwd, L.wp_gb, 30 if L.k == 4 else 31, L.cout, L.cin) for _, L in named if L.wp_gb is not None]
        meta += [(row[nm], 1) for nm, L in named if L.wp_gb is not None]
        jobs += [(L.w_fwd, L.wp_t16, 11, L.cin, L.cout) for _, L in named if L.wp_t16 is not None]
        meta += [(row[nm], 0) for nm, L in named if L.wp_t16 is not None]
        assert len(jobs) <= 16
        self._mfma_jobs = jobs
        self._mfma_job_layers = meta           # (layer-table row, 0 = w_fwd / 1 = w_bwd) of each job's source
        self._table_host = table
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(self.dev)
        self.nlayers = len(mods)

    def _rate_job(self):
        """The weight-rate term's partial pass as a job of the step head: it reads parameters only.  g is a host
        constant (lambda w2 / N, times 1 / world under data parallelism)."""
        key = (self.rate_grad_scale, self.lmbda, self.w2, self.n_points_total)
        if self._rate is not None and self._rate[2] != key:
            # captured step graphs hold the old buffers' addresses and the old g: a graph replayed after this point would
            # add stale addends into retired memory -- refuse instead (dist.attach runs before any capture)
            if self._graphs_captured:
                raise RuntimeError("weight-rate coefficients changed after a step graph was captured "
                                   "(attach data parallelism / set lambda before building GraphedTrainStep)")
            self._rate_retired.append(self._rate)       # a kernel in flight may still read them
        if self._rate is None or self._rate[2] != key:
            lm = self.net.reconstructor.likelihood_model
            ks = [self.layers[n].mod.kernel for n in TRUNK]
            dk = torch.zeros(sum(k.numel() for k in ks), device=self.dev)
            dks, off = [], 0
            for k in ks:
                dks.append(dk[off:off + k.numel()])
                off += k.numel()
            part = torch.zeros(3 * 512, device=self.dev)
            g = self.lmbda * self.w2 / self.n_points_total * self.rate_grad_scale
            job = ops.rate_job(ks, dks, lm.sigma, lm.mu, part, g)
            add = {self.layers[n].gk.data_ptr(): d for n, d in zip(TRUNK, dks)}
            self._rate = (job, add, key, dk, part)
        return self._rate

    def batch_and_prepare(self, idx_dev, q, with_rate=False, stem_mode=None):
        """_batch(idx_dev) and prepare_weights(q) -- row gather, effective weights, MFMA packings -- as ONE launch
        (``with_rate``: + the weight-rate term's partial sums, consumed by the backward pass of the same step;
        ``stem_mode`` = 'train' / 'eval': + the stem's forward of forward(e, stem_mode, idx_dev), which must be the next
        call -- narrow decoder, nvf_step_head_stem)."""
        import ctypes
        srcs = [self.gt, self.dist, self.gt16, self.gt8, self.emb]
        n, rows = len(srcs), idx_dev.numel()
        dsts = [torch.empty((rows,) + tuple(s.shape[1:]), device=s.device) for s in srcs]
        sd = self._step_dev
        jobs, meta = self._mfma_jobs, self._mfma_job_layers
        npk = len(jobs)
        iarr = lambda xs: (ctypes.c_int * max(len(xs), 1))(*xs)
        args = (self.table.data_ptr(), self.nlayers, int(q), self.seed, 0 if sd is not None else self.noise_step,
                None if sd is None else sd.data_ptr(),
                (ctypes.c_void_p * max(npk, 1))(*[j[1].data_ptr() for j in jobs]), iarr([j[2] for j in jobs]),
                iarr([j[3] for j in jobs]), iarr([j[4] for j in jobs]), iarr([m[0] for m in meta]),
                iarr([m[1] for m in meta]), npk, (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs]),
                (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts]), (ctypes.c_int * n)(*[s[0].numel() for s in srcs]), n,
                idx_dev.data_ptr(), rows, ctypes.byref(self._rate_job()[0]) if with_rate else None)
        self._stem_pre = None
        if (stem_mode is not None and _STEM_IN_HEAD and (self.narrow or self.wide) and self.fused_stem and self.fused_latent_stem
                and self.ch <= 8 and rows <= 32 and _NAIVE_OFF()):
            # ... and the stem's forward of this mini-batch (forward() picks the tensors up instead of launching it)
            from ._lib import NvfStemHead
            net = self.net
            g2, ec, ig = net.latent_gen.gdn_2, net.entropy_coder, net.reconstructor.activation
            dev, ch = self.dev, self.ch
            c0, c1 = self.channels[0], self.channels[1]
            o = {"h": torch.empty(rows, ch, 2, 2, 2, device=dev), "lat": torch.empty(rows, ch, 2, 2, 2, device=dev),
                 "x0": torch.empty(rows, ch, 2, 2, 2, device=dev), "lbits": torch.empty(1, device=dev),
                 "a0": torch.empty(rows, c0, 4, 4, 4, device=dev), "h0": torch.empty(rows, c0, 4, 4, 4, device=dev),
                 "y1": torch.empty(rows, c1, 8, 8, 8, device=dev)}
            sj = NvfStemHead()
            sj.emb, sj.lat_beta_hat, sj.lat_gamma_hat = self.emb.data_ptr(), g2.beta.data_ptr(), g2.gamma.data_ptr()
            sj.sigma, sj.mu = ec.sigma.data_ptr(), ec.mu.data_ptr()
            sj.beta_hat, sj.gamma_hat = ig.beta.data_ptr(), ig.gamma.data_ptr()
            sj.h, sj.lat, sj.x_rounded, sj.bits = (o[k].data_ptr() for k in ("h", "lat", "x0", "lbits"))
            sj.a0, sj.h0, sj.y1 = (o[k].data_ptr() for k in ("a0", "h0", "y1"))
            sj.lat_row, sj.up0_row, sj.conv0_row = self._rows["latent"], self._rows["up0"], self._rows["conv0"]
            sj.mode, sj.ch, sj.c0, sj.c1 = (0 if stem_mode == "train" else 1), ch, c0, c1
            check(lib().nvf_step_head_stem(*args, ctypes.byref(sj), torch.cuda.current_stream().cuda_stream),
                  "nvf_step_head_stem")
            self._stem_pre = (idx_dev.data_ptr(), stem_mode, o)
        else:
            check(lib().nvf_step_head(*args, torch.cuda.current_stream().cuda_stream), "nvf_step_head")
        self._rate_ready = bool(with_rate)
        return dsts

    def prepare_weights(self, q):
        sd = self._step_dev
        check(lib().nvf_prepare_weights(self.table.data_ptr(), self.nlayers, int(q), self.seed,
                                        0 if sd is not None else self.noise_step, None if sd is None else sd.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "nvf_prepare_weights")
        if self._mfma_jobs:
            ops.pack_mfma_all(self._mfma_jobs)      # conv1, conv2, up1, up2: every MFMA weight layout, one launch

    # ------------------------------------------------------------------ forward
    def _convT(self, L, x, act, train=False):
        if L.wp_t16 is not None:
            return ops.convT3d_k5s2_mfma16(x, L.wp_t16, L.b_eff, act, cout=L.cout, pad=L.pad)
        if L.wp_t is not None:
            # eight waves with one column tile each (variant 5: two waves per SIMD; up1 19.3 -> 15.6 us, up2 34.0 -> 30.7 at
            # batch 16 -- and at batch 917 (the full-batch latent step; r05 sweep): up2 1115 -> 916 us, up1 389 -> 310;
            # every variant runs the same per-output fmaf chain: bit-identical)
            var = _VAR["UP1F" if L.cin == 16 else "UP2F"] or 5
            if train and L.wp_tr is not None and var == 5:
                return ops.convT3d_k5s2_mfma(x, L.wp_tr, L.b_eff, act, variant=15)
            return ops.convT3d_k5s2_mfma(x, L.wp_t, L.b_eff, act, variant=var)
        return ops.convT3d_k5s2_fwd(x, L.w_fwd, L.b_eff, L.cout, L.pad, act)

    def _conv(self, L, x, act, train=False):
        if train and L.wp_wf is not None and act == R and (self.wide or x.shape[-1] == 35 or x.shape[0] > 64):
            if self.wide:
                return ops.conv3d_k4_wino16_fwd(x, L.wp_wf, L.b_eff)
            if x.shape[-1] == 19:      # conv1 at a full batch: the two-set kernel, all eight plane pairs per unit
                return ops.conv3d_k4_wino_fwd(x, L.wp_wf, L.b_eff, ppc=8)
            # (a full-batch launch has workgroups to spare: more plane pairs per work unit repeat fewer plane transforms --
            # conv2 at batch 917: 1667 -> 1481 us with 8 pairs per unit; the same bits, tools/wino_ppc_sweep.py)
            # (above batch 64 the two-set kernel with all 16 pairs: 1480 us against 1507 for the one-set kernel with 8)
            return ops.conv3d_k4_wino_fwd(x, L.wp_wf, L.b_eff, ppc=16 if x.shape[0] > 64 else 0)
        if L.wp_gf is not None:
            osz = tuple(s - 3 for s in x.shape[2:])
            return ops.conv3d_g16_mfma(x, L.wp_gf, L.b_eff, L.cout, 4, 1, 0, osz, act)
        if L.wp_f is not None:
            # conv1 at large batch: 8 rows x 4 planes on eight waves (variant 5: 34.2 vs 39.4 us at batch 64, 107 vs 120
            # (variant 2) at 256, 402 vs 443 at 917; at batch 16 the default tile: 12.7 vs 17.6).  Every variant runs
            # the same per-output fmaf chain, so the bits -- and encode-at-any-batch == decode-at-batch-1 -- do not change
            var = 5 if (x.shape[-1] == 19 and x.shape[0] >= 64) else None
            if x.shape[-1] == 35 and _CONV2_FWD_VAR:
                var = _CONV2_FWD_VAR
            if x.shape[-1] == 19 and _VAR["C1F"] and x.shape[0] <= 64:
                var = _VAR["C1F"]
            return ops.conv3d_k4_mfma(x, L.wp_f, L.b_eff, 0, 0, act, variant=var)
        osz = tuple(s + 2 * L.pad - L.k + 1 for s in x.shape[2:])
        return ops.conv3d_gather(x, L.w_fwd, L.b_eff, L.cout, L.k, 1, L.pad, osz, act)

    def forward(self, e, mode, block_ids, defer_heads=False):
        """e [B,ch,2,2,2] latents-before-latent_gen.  Returns the dict of saved activations.  ``defer_heads``: the caller
        runs backward() next -- at mini-batch sizes the heads' forward then rides in the launch of their loss and
        backward-data (a["p0"..] stay None until then)."""
        net, Ls = self.net, self.layers
        a = {"e": e}
        g2 = net.latent_gen.gdn_2
        ec = net.entropy_coder
        sd = self._step_dev
        ig = net.reconstructor.activation
        stem_done = False
        pre, self._stem_pre = getattr(self, "_stem_pre", None), None
        if pre is not None and pre[0] == block_ids.data_ptr() and pre[1] == mode:
            a.update(pre[2])                  # the step head's launch ran the latent generator, quantiser and stem
            stem_done = True
        elif e.shape[1] <= 8 and _NAIVE_OFF() and self.fused_stem and self.fused_latent_stem:   # latent generator,
            # quantiser and stem in one launch
            (a["h"], a["lat"], a["x0"], a["lbits"], a["a0"], a["h0"], a["y1"]) = ops.stem_latent_fwd(
                e, Ls["latent"].w_fwd, Ls["latent"].b_eff, g2.beta, g2.gamma, ec.sigma.reshape(-1), ec.mu.reshape(-1),
                mode, Ls["up0"].w_fwd, Ls["up0"].b_eff, ig.beta, ig.gamma, Ls["conv0"].w_fwd, Ls["conv0"].b_eff,
                block_ids=block_ids, seed=self.seed, step=0 if sd is not None else self.noise_step, step_dev=sd)
            stem_done = True
        elif e.shape[1] <= 8 and _NAIVE_OFF():         # latent generator + quantiser in one launch
            a["h"], a["lat"], a["x0"], a["lbits"] = ops.latent_fwd(
                e, Ls["latent"].w_fwd, Ls["latent"].b_eff, g2.beta, g2.gamma, ec.sigma.reshape(-1), ec.mu.reshape(-1),
                mode, block_ids=block_ids, seed=self.seed, step=0 if sd is not None else self.noise_step, step_dev=sd)
        else:
            a["h"] = self._conv(Ls["latent"], e, NONE)
            a["lat"] = ops.gdn_fwd(a["h"], g2.beta, g2.gamma, False)
            a["x0"], a["lbits"], _, _, _ = ops.latent_rate(a["lat"], ec.sigma.reshape(-1), ec.mu.reshape(-1), mode,
                                                           block_ids=block_ids, seed=self.seed,
                                                           step=0 if sd is not None else self.noise_step, step_dev=sd)
        self.overlap = self.allow_overlap and e.shape[0] <= 64   # large batches fill the chip by themselves
        if stem_done:
            pass
        elif self.fused_stem:
            a["a0"], a["h0"], a["y1"] = ops.stem_fwd(a["x0"], Ls["up0"].w_fwd, Ls["up0"].b_eff, ig.beta, ig.gamma,
                                                     Ls["conv0"].w_fwd, Ls["conv0"].b_eff)
        else:
            a["a0"] = self._convT(Ls["up0"], a["x0"], NONE)
            a["h0"] = ops.gdn_fwd(a["a0"], ig.beta, ig.gamma, True)
            a["y1"] = self._convT(Ls["conv0"], a["h0"], R)
        if not self.heads3:
            self._fork()
            with self._on_side():                   # the two coarse heads run beside the trunk
                a["p0"] = self._conv(Ls["conv0_cls"], a["y1"], S)
        a["y2"] = self._convT(Ls["up1"], a["y1"], R, train=(mode == "train"))
        a["y3"] = self._conv(Ls["conv1"], a["y2"], R, train=(mode == "train"))
        if not self.heads3:
            self._fork()
            with self._on_side():
                a["p1"] = self._conv(Ls["conv1_cls"], a["y3"], S)
        a["y4"] = self._convT(Ls["up2"], a["y3"], R, train=(mode == "train"))
        a["y5"] = self._conv(Ls["conv2"], a["y4"], R, train=(mode == "train"))
        if (self.heads3 and defer_heads and _HEADS_FWD_IN_LOSS and e.shape[0] <= 32 and _NAIVE_OFF()
                and not self.allow_overlap):
            a["p0"] = a["p1"] = a["p2"] = None      # nvf_heads3_fwd_loss_bwd_data (backward)
        elif self.heads3:                           # all three heads in one launch, after the trunk
            hl = [Ls["conv0_cls"], Ls["conv1_cls"], Ls["conv2_cls"]]
            a["p0"], a["p1"], a["p2"] = ops.heads3_fwd([a["y1"], a["y3"], a["y5"]], [L.w_fwd for L in hl],
                                                       [L.b_eff for L in hl])
        else:
            a["p2"] = self._conv(Ls["conv2_cls"], a["y5"], S)
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.side)
        return a

    # ------------------------------------------------------------------ backward
    def _fork(self):
        """Side stream waits for everything issued so far on the current (main) stream."""
        if self.overlap:
            self.side.wait_stream(torch.cuda.current_stream())

    def _on_side(self):
        return torch.cuda.stream(self.side) if self.overlap else _NullCtx()

    def _wgrad_conv(self, L, g_out, x_in):
        if (self.wide and self.winograd and _WINO16 and L.k == 4 and L.cin == 16 and L.cout == 16 and L.pad == 0
                and g_out.shape[-1] in _WINO16_WGRAD):
            # the Winograd (y, x) form (wgrad16_wino.hip): slabs for the common reduction (conv2 172 -> 100 us at batch 16)
            base = self._wg.reserve(256 * 16384 * 4)
            n = ops.wgrad16_k4_wino_partial(g_out, x_in, base, zsplit=0 if g_out.shape[-1] == 32 else 8)
            self._wg.add_job(base, L.gk, n, 16384)
        else:
            self._wg.add(g_out, x_in, L.k, 1, L.pad, 0, L.gk)
        self._bias_jobs.append((g_out, L.gb))

    def _wgrad_convT(self, L, g_out, x_in, bias=True):
        self._wg.add(x_in, g_out, 5, 2, L.pad, 0, L.gk)
        if bias:        # (False: the kernel that wrote g_out left its channel sums as slabs, see _dx_conv)
            self._bias_jobs.append((g_out, L.gb))

    def _dx_conv(self, L, g_out, x_in, mask=None, addend=None, bias_out=None):
        """Backward-data of a 4^3 convolution.  ``bias_out``: the bias gradient of the layer BELOW (whose masked output
        gradient this pass writes): the matrix-core kernel leaves its channel sums as slabs for the reduction launch;
        returns (dx, True) then, (dx, False) when the caller has to sum dx itself."""
        if self.wide and L.wp_w is not None and mask is not None and addend is None and g_out.shape[-1] in (32, 16):
            if bias_out is not None and g_out.shape[0] <= 64 and _WINO16_BIAS:
                # the channel sums of dx (the bias gradient of the layer below) leave with the kernel's stores instead of a
                # second pass over the 44 MB it wrote (conv2 at batch 16)
                base = self._wg.reserve(8192 * 16 * 4)
                dx, nparts = ops.conv3d_k4_wino16_bwd(g_out, L.wp_w, mask, bias_part=base)
                self._wg.add_job(base, bias_out, nparts, 16)
                return dx, True
            dx = ops.conv3d_k4_wino16_bwd(g_out, L.wp_w, mask)
            return dx if bias_out is None else (dx, False)
        if L.wp_gb is not None:
            dx = ops.conv3d_g16_mfma(g_out, L.wp_gb, None, L.cin, 4, 1, 3, tuple(x_in.shape[2:]), addend=addend,
                                     mask=mask)
            return dx if bias_out is None else (dx, False)
        if L.wp_w is not None and mask is not None and addend is None and g_out.shape[-1] in (32, 16):
            if bias_out is not None:
                # (one slab of 8 sums per work unit: 126 units per block in conv2's default kernel, conv_wino1.hip)
                base = self._wg.reserve(16384 * 8 * 4) if g_out.shape[0] <= 64 else None
                if base is not None:
                    dx, nparts = ops.conv3d_k4_wino_bwd(g_out, L.wp_w, mask, bias_part=base)
                    self._wg.add_job(base, bias_out, nparts, 8)
                    return dx, True
                return ops.conv3d_k4_wino_bwd(g_out, L.wp_w, mask, ppc=_wino_bwd_ppc(g_out)), False
            return ops.conv3d_k4_wino_bwd(g_out, L.wp_w, mask, ppc=_wino_bwd_ppc(g_out))
        if L.wp_b is not None and g_out.shape[0] <= L.bwd_max_batch:
            var = _CONV2_BWD_VAR if (g_out.shape[-1] == 32 and _CONV2_BWD_VAR) else None
            if g_out.shape[-1] == 16 and _VAR["C1B"]:
                var = _VAR["C1B"]
            if bias_out is not None and mask is not None and addend is None:
                base = self._wg.reserve(4096 * 8 * 4)
                dx, nparts = ops.conv3d_k4_mfma(g_out, L.wp_b, None, 3, L.bwd_pair, NONE, mask=mask, bias_part=base,
                                                variant=var)
                self._wg.add_job(base, bias_out, nparts, 8)
                return dx, True
            dx = ops.conv3d_k4_mfma(g_out, L.wp_b, None, 3, L.bwd_pair, NONE, addend=addend, mask=mask, variant=var)
            return dx if bias_out is None else (dx, False)
        if bias_out is not None:
            return ops.conv3d_gather(g_out, L.w_bwd, None, L.cin, L.k, 1, L.k - 1 - L.pad, tuple(x_in.shape[2:]),
                                     addend=addend, mask=mask), False
        return ops.conv3d_gather(g_out, L.w_bwd, None, L.cin, L.k, 1, L.k - 1 - L.pad, tuple(x_in.shape[2:]),
                                 addend=addend, mask=mask)

    def _dx_convT(self, L, g_out, x_in, mask=None, addend=None):
        if L.wp_gb is not None:
            return ops.conv3d_g16_mfma(g_out, L.wp_gb, None, L.cin, 5, 2, L.pad, tuple(x_in.shape[2:]), addend=addend,
                                       mask=mask)
        if L.wp_s is not None:
            # up1 at large batch: two planes per wave (variant 2: 55 vs 63 us at batch 256; 173 vs 184 at 917)
            # up2 at batch <= 64: 8 rows x 2 planes on eight waves (variant 6: 25.3 -> 22.8 us at batch 16; bit-identical);
            # above: 4 rows x 4 planes on eight waves (variant 5: 1027 vs 1149 us at batch 917)
            # up1 in training steps of the default engine: the two channel groups of g on different waves, 256 workgroups of
            # 2 rows x 2 planes (variant 7: another summation order, so not in the strict-trajectory engine)
            var = 2 if (L.cin == 16 and g_out.shape[0] > 64) else (
                _VAR["UP1B" if L.cin == 16 else "UP2B"] or ((6 if g_out.shape[0] <= 64 else 5) if L.cin == 8 else
                                                            (7 if (self.winograd and _UP1B_KSPLIT) else None)))
            return ops.conv3d_s2k5_mfma(g_out, L.wp_s, L.cin, addend=addend, mask=mask, variant=var)
        return ops.conv3d_gather(g_out, L.w_bwd, None, L.cin, 5, 2, L.pad, tuple(x_in.shape[2:]), addend=addend,
                                 mask=mask)

    def backward(self, *args, **kw):
        try:
            return self._backward(*args, **kw)
        except BaseException:
            self.ctx.cancel()              # never leave final passes or a latent tail queued in the context
            raise

    def _backward(self, a, gt, dist, gt16, gt8, n_pts, mode, block_ids, want_w, want_emb, fuse=None):
        """Loss (NVFPCC.py:161-196) and its gradients.  Weight grads land in self.flat_g.

        Main stream: the backward-data chain.  Side stream: head backward-data (t0, t1), every weight gradient,
        the bias sums and the weight-rate term; each side job waits for the chain tensor it consumes."""
        net, Ls = self.net, self.layers
        main = torch.cuda.current_stream()
        self.overlap = self.allow_overlap and a["e"].shape[0] <= 64
        self._bias_jobs = []
        self._bias_cover = []     # bias gradients some launch's own final pass writes (no channel-sum job needed)
        if self._wg is None:
            # (wide decoder: 125 MiB of slabs at batch 16 -- a workspace that had to grow mid-step would be reallocated)
            self._wg = ops.WgradBatch(self.dev, nbytes=(256 if self.wide else 128) << 20, ctx=self.ctx)    # partial sums now, ONE reduction launch for all ten
        loss = torch.empty(4, device=self.dev)   # [main, head0, head1, unused]
        nbits = torch.empty(7, device=self.dev)
        # the one-block final passes of the focal terms, the bias sums and the weight rate feed nothing inside the
        # step: queue them and run all three in one launch at the end (single-stream schedule only)
        defer = want_w and not self.overlap
        ctx = self.ctx if defer else None
        if defer:
            self.ctx.begin()
        fused_loss = self.heads3 and a["e"].shape[0] <= 32 and _NAIVE_OFF()
        if not fused_loss:
            dl2, dl0, dl1 = ops.focal_loss_multi([(a["p2"], gt, dist, 0.9, 1.0), (a["p0"], gt8, None, 0.85, 0.0),
                                                  (a["p1"], gt16, None, 0.85, 0.0)], loss, ctx=ctx)
        def step_metrics():
            # logging counts of NVFPCC.py:174-179, 190-221 (tp / ap / tn / an of the main output and of both heads at 0.5,
            # sse / denom at 0.6): one partial-sum launch, the final pass rides in the finals launch; nvf_step_tail turns
            # them into the per-step ratios behind the all-reduce
            if self.epoch_acc is not None and want_w:
                ops.metrics3([a["p2"], a["p0"], a["p1"]], [gt, gt8, gt16], [dist, None, None], 0.5, 0.6,
                             out=self.step_counts, ctx=ctx)
        heads_deferred = a.get("p2") is None
        if not heads_deferred:
            step_metrics()
        ev_t1 = ev_t0 = None
        if self.heads3:
            hl = [Ls["conv0_cls"], Ls["conv1_cls"], Ls["conv2_cls"]]
            if heads_deferred:
                # the heads' forward, the three focal terms, their logit gradients and the heads' backward-data: ONE launch
                # (forward() left p0 / p1 / p2 to this call)
                assert fused_loss and defer
                (a["p0"], a["p1"], a["p2"]), (dl0, dl1, dl2), (t0, t1, g5) = ops.heads3_fwd_loss_bwd_data(
                    [a["y1"], a["y3"], a["y5"]], [L.w_fwd for L in hl], [L.b_eff for L in hl], [gt8, gt16, gt],
                    [None, None, dist], [0.85, 0.85, 0.9], [0.0, 0.0, 1.0], [1, 2, 0], loss, [L.w_bwd for L in hl],
                    [None, None, a["y5"]], self.ctx,
                    bias_outs=[L.gb for L in hl] if (want_w and _HEAD_BIAS_IN_LOSS) else None)
                head_bias_done = want_w and _HEAD_BIAS_IN_LOSS
                step_metrics()
            elif fused_loss:    # the three focal terms, their logit gradients and the heads' backward-data: one launch
                (dl0, dl1, dl2), (t0, t1, g5) = ops.heads3_loss_bwd_data(
                    [a["p0"], a["p1"], a["p2"]], [gt8, gt16, gt], [None, None, dist], [0.85, 0.85, 0.9],
                    [0.0, 0.0, 1.0], [1, 2, 0], loss, [L.w_bwd for L in hl], [L.cin for L in hl], [None, None, a["y5"]],
                    ctx=ctx, bias_outs=[L.gb for L in hl] if (want_w and _HEAD_BIAS_IN_LOSS) else None)
                head_bias_done = want_w and _HEAD_BIAS_IN_LOSS    # the heads' bias gradients: partials of that launch
            else:
                t0, t1, g5 = ops.heads3_bwd_data([dl0, dl1, dl2], [L.w_bwd for L in hl], [L.cin for L in hl],
                                                 [None, None, a["y5"]])
            heads_job = None
            if want_w:
                heads_job = ([dl0, dl1, dl2], [a["y1"], a["y3"], a["y5"]], [L.gk for L in hl])
                if not (self.narrow and _NAIVE_OFF() and _HEADS_IN_TRUNK5):   # else: workgroups of the five-gradient launch
                    self._wg.add_heads3(*heads_job)
                    heads_job = None
                if fused_loss and head_bias_done:
                    self._bias_cover += [L.gb for L in hl]
                else:
                    self._bias_jobs += [(dl2, hl[2].gb), (dl1, hl[1].gb), (dl0, hl[0].gb)]
        else:
            self._fork()
            with self._on_side():
                t1 = self._dx_conv(Ls["conv1_cls"], dl1, a["y3"])
                ev_t1 = torch.cuda.Event() if self.overlap else None
                if ev_t1 is not None:
                    ev_t1.record()
                t0 = self._dx_conv(Ls["conv0_cls"], dl0, a["y1"])
                ev_t0 = torch.cuda.Event() if self.overlap else None
                if ev_t0 is not None:
                    ev_t0.record()
                if want_w:
                    self._wgrad_conv(Ls["conv2_cls"], dl2, a["y5"])
                    self._wgrad_conv(Ls["conv1_cls"], dl1, a["y3"])
                    self._wgrad_conv(Ls["conv0_cls"], dl0, a["y1"])

        def side_wgrad(fn, L, g, x, **kw):
            if not want_w:
                return
            self._fork()
            with self._on_side():
                fn(L, g, x, **kw)

        # wide decoder: the Winograd backward-data kernels leave the channel sums of what they write (the bias gradients
        # of up2 / up1) as slabs for the reduction launch
        wbias = want_w and self.wide and _WINO16_BIAS and Ls["conv2"].wp_w is not None

        if not self.heads3:
            g5 = self._dx_conv(Ls["conv2_cls"], dl2, a["y5"], mask=a["y5"])
        wg3 = want_w and self.narrow and _NAIVE_OFF()      # conv2 / up2 / conv1 weight gradients in one launch
        if not wg3:
            side_wgrad(self._wgrad_conv, Ls["conv2"], g5, a["y4"])
        if wg3:     # up2's bias gradient = the channel sums of g4: left by the kernel that writes g4
            g4, up2_bias_done = self._dx_conv(Ls["conv2"], g5, a["y4"], mask=a["y4"], bias_out=Ls["up2"].gb)
        elif wbias:
            g4, done = self._dx_conv(Ls["conv2"], g5, a["y4"], mask=a["y4"], bias_out=Ls["up2"].gb)
            side_wgrad(self._wgrad_convT, Ls["up2"], g4, a["y3"], bias=not done)
        else:
            g4 = self._dx_conv(Ls["conv2"], g5, a["y4"], mask=a["y4"])
            side_wgrad(self._wgrad_convT, Ls["up2"], g4, a["y3"])
        if ev_t1 is not None:
            main.wait_event(ev_t1)
        g3 = self._dx_convT(Ls["up2"], g4, a["y3"], mask=a["y3"], addend=t1)
        if wg3:     # launched after the latent tail has been queued (below): the tail rides in that launch; the bias
            # gradients of conv2 and conv1 (channel sums of g5, g3) come out of it too
            if not up2_bias_done:
                self._bias_jobs += [(g4, Ls["up2"].gb)]
        else:
            side_wgrad(self._wgrad_conv, Ls["conv1"], g3, a["y2"])
        if wg3:
            g2, up1_bias_done = self._dx_conv(Ls["conv1"], g3, a["y2"], mask=a["y2"], bias_out=Ls["up1"].gb)
        elif wbias:
            g2, done = self._dx_conv(Ls["conv1"], g3, a["y2"], mask=a["y2"], bias_out=Ls["up1"].gb)
            side_wgrad(self._wgrad_convT, Ls["up1"], g2, a["y1"], bias=not done)
        else:
            g2 = self._dx_conv(Ls["conv1"], g3, a["y2"], mask=a["y2"])
            side_wgrad(self._wgrad_convT, Ls["up1"], g2, a["y1"])
        if ev_t0 is not None:
            main.wait_event(ev_t0)
        g1 = self._dx_convT(Ls["up1"], g2, a["y1"], mask=a["y1"], addend=t0)
        stem_wg0 = want_w and self.fused_stem and defer and not wg3   # conv0's weight gradient rides in the stem's backward
        if wg3:                                      # up1 and conv0 weight gradients: with the other three, below
            self._bias_jobs += ([] if up1_bias_done else [(g2, Ls["up1"].gb)]) + [(g1, Ls["conv0"].gb)]
        elif stem_wg0:
            self._bias_jobs.append((g1, Ls["conv0"].gb))
        else:
            side_wgrad(self._wgrad_convT, Ls["conv0"], g1, a["h0"])
        ig = net.reconstructor.activation
        gview = (lambda n: self._g(n)) if want_w else (lambda n: None)
        gamma_view = None if not want_w else self._g("reconstructor.activation.gamma").view(ig.gamma.shape)
        self._stem_gdn_in_finals = bool(self.fused_stem and defer and want_w)
        tail = _TAIL and defer and not want_emb and a["e"].shape[1] <= 8 and _NAIVE_OFF()
        stem_coop = (_STEM_IN_TRUNK5 and tail and wg3 and self.fused_stem and want_w and a["e"].shape[0] <= 32
                     and heads_job is not None)
        if stem_coop:
            # no launch here: queued in the context, it runs inside add_trunk5's launch below (with the latent tail that
            # consumes dx0); up0's bias gradient comes from per-block channel sums the stage leaves, not from da0
            da0, dx0 = ops.stem_bwd_queue(g1, a["x0"], a["a0"], Ls["conv0"].w_bwd, Ls["up0"].w_bwd, ig.beta, ig.gamma,
                                          gview("reconstructor.activation.beta"), gamma_view, Ls["up0"].gk,
                                          Ls["up0"].gb, self._wg, self.ctx)
        elif self.fused_stem and defer:        # its final launch is shared with the slab reduction / the final passes
            da0, dx0 = ops.stem_bwd_partial(g1, a["x0"], a["a0"], Ls["conv0"].w_bwd, Ls["up0"].w_bwd, ig.beta,
                                            ig.gamma, gview("reconstructor.activation.beta"), gamma_view,
                                            Ls["up0"].gk, self._wg, ctx=ctx, h0=a["h0"] if stem_wg0 else None,
                                            dw_conv0=Ls["conv0"].gk if stem_wg0 else None)
            self._bias_jobs.append((da0, Ls["up0"].gb))
        elif self.fused_stem:
            da0, dx0 = ops.stem_bwd(g1, a["x0"], a["a0"], Ls["conv0"].w_bwd, Ls["up0"].w_bwd, ig.beta, ig.gamma,
                                    gview("reconstructor.activation.beta"), gamma_view,
                                    Ls["up0"].gk if want_w else None)
            if want_w:
                self._bias_jobs.append((da0, Ls["up0"].gb))
        else:
            dh0 = self._dx_convT(Ls["conv0"], g1, a["h0"])
            da0, _, _ = ops.gdn_bwd(a["a0"], ig.beta, ig.gamma, dh0, True, gview("reconstructor.activation.beta"),
                                    gamma_view)
            side_wgrad(self._wgrad_convT, Ls["up0"], da0, a["x0"])
            dx0 = self._dx_convT(Ls["up0"], da0, a["x0"])
        # latent rate (+ the decoder's gradient through the straight-through round)
        ec = net.entropy_coder
        sd = self._step_dev
        g_lat = self.lmbda * self.w1 / n_pts if self._g_lat_dev is None else 1.0
        g2m = net.latent_gen.gdn_2
        if tail:
            # three dependent launches on [B, ch, 2^3] tensors -> one workgroup of the next weight-gradient launch
            # (or of the slab reduction); dlat / dh / dx0 stay referenced until that launch has been enqueued
            dlat, dh = torch.empty_like(a["lat"]), torch.empty_like(a["h"])
            ops.latent_tail_queue(self.ctx, a["lat"], ec.sigma.reshape(-1), ec.mu.reshape(-1), mode, block_ids, dx0, dlat,
                                  gview("entropy_coder.sigma"), gview("entropy_coder.mu"), self._g_lat_dev, g_lat,
                                  self.seed, 0 if sd is not None else self.noise_step, sd, a["h"], g2m.beta, g2m.gamma,
                                  dh, gview("latent_gen.gdn_2.beta"),
                                  self._g("latent_gen.gdn_2.gamma").view(g2m.gamma.shape), a["e"], Ls["latent"].gk,
                                  Ls["latent"].gb)
            de = None
        else:
            _, _, dlat, _, _ = ops.latent_rate(a["lat"], ec.sigma.reshape(-1), ec.mu.reshape(-1), mode,
                                               block_ids=block_ids, want_grad=True, g_host=g_lat,
                                               g_dev=self._g_lat_dev, seed=self.seed,
                                               step=0 if sd is not None else self.noise_step, step_dev=sd,
                                               dx_addend=dx0, dsigma_out=gview("entropy_coder.sigma"),
                                               dmu_out=gview("entropy_coder.mu"))
            dh, _, _ = ops.gdn_bwd(a["h"], g2m.beta, g2m.gamma, dlat, False, gview("latent_gen.gdn_2.beta"),
                                   None if not want_w else self._g("latent_gen.gdn_2.gamma").view(g2m.gamma.shape))
            side_wgrad(self._wgrad_conv, Ls["latent"], dh, a["e"])
            de = self._dx_conv(Ls["latent"], dh, a["e"]) if want_emb else None
        if wg3:
            # conv2 / up2 / conv1 weight gradients, the longest launch of the step, go last of the big kernels: the
            # queued latent tail (a 30 us chain of three dependent stages in ONE workgroup) runs as its first workgroup
            # and is hidden behind them instead of being the critical path of the slab reduction; up1's and conv0's
            # gradients (small VALU kernels) fill the slots that the short matrix-core workgroups leave
            self._wg.add_trunk5([g5, a["y3"], g3, a["y1"], a["h0"]], [a["y4"], g4, a["y2"], g2, g1],
                                [Ls["conv2"].gk, Ls["up2"].gk, Ls["conv1"].gk, Ls["up1"].gk, Ls["conv0"].gk],
                                bias_outs=(Ls["conv2"].gb, Ls["conv1"].gb), heads=heads_job if self.heads3 else None,
                                sums=(([t for t, _ in self._bias_jobs], [o for _, o in self._bias_jobs])
                                      if (self.heads3 and heads_job is not None and defer and self._rate_ready
                                          and want_w and _SUMS_IN_TRUNK5) else None),
                                coef=((fuse["coef_dev"], self._coef_live)
                                      if (fuse is not None and fuse.get("coef_dev") is not None) else None))
        # weight rate: bits of the 7 quantised kernels and, for the decoder update, their gradients (added to the
        # weight gradients, so it follows the wgrads on the side stream); every bias gradient in one reduction
        lm = net.reconstructor.likelihood_model
        g_net = self.lmbda * self.w2 / self.n_points_total
        gs, gm = self._g("reconstructor.likelihood_model.sigma"), self._g("reconstructor.likelihood_model.mu")
        kernels = [Ls[n].mod.kernel for n in TRUNK]
        rate_head, self._rate_ready = self._rate_ready and want_w and defer, False
        fused = None
        self._fork()
        with self._on_side():
            if want_w:     # slab reduction of every weight gradient + all bias sums
                addends = None
                if rate_head:
                    # the weight-rate gradient was computed by the step head: the reduction adds it while it writes the
                    # gradients (every trunk kernel must come out of that launch; otherwise the stand-alone pass)
                    job, addends = self._rate_job()[:2]
                    live = {j[1] for j in self._wg.jobs if j[2] > 0}
                    if not all(Ls[n].gk.data_ptr() in live for n in TRUNK) or len(self._wg.jobs) > 16:
                        rate_head, addends = False, None
                if fuse is not None and rate_head:
                    fused = self._fused_tail(fuse, loss, a["lbits"], nbits, gs, gm)
                sums_done = wg3 and getattr(self._wg, "sums_done", False)    # the partial bias sums rode in add_trunk5
                one_launch = fused is not None and sums_done and rate_head
                if one_launch:      # nothing the final passes read is written by the slab reduction: ONE launch for both
                    ops.weight_rate_final(job, nbits, gs, gm, ctx=ctx)
                    adam = fused[1]
                    if fuse.get("coef_dev") is not None:     # the copy add_trunk5 staged: not the step buffer's words
                        adam.coef_dev = self._coef_live.data_ptr()
                    self._wg.finish_and_flush_tail(addends, adam, fused[0], fused[2])
                elif sums_done and rate_head and fused is None:
                    # data parallelism (the all-reduce and the step tail follow): the same pairing without the optimiser
                    one_launch = True
                    ops.weight_rate_final(job, nbits, gs, gm, ctx=ctx)
                    self._wg.finish_and_flush(addends)
                else:
                    if sums_done:
                        self._wg.finish_with_sums([], [], addends=addends, adam=None if fused is None else fused[1])
                    else:
                        self._wg.finish_with_sums([t for t, _ in self._bias_jobs], [o for _, o in self._bias_jobs],
                                                  addends=addends, adam=None if fused is None else fused[1])
                    if rate_head:
                        ops.weight_rate_final(job, nbits, gs, gm, ctx=ctx)
                    else:
                        ops.weight_rate_batch(kernels, [Ls[n].gk for n in TRUNK], lm.sigma, lm.mu, nbits, gs, gm,
                                              g_host=g_net * self.rate_grad_scale, ctx=ctx)
            else:
                one_launch = False
                ops.weight_rate_batch(kernels, None, lm.sigma, lm.mu, nbits)
            if defer and not one_launch:
                if fused is not None:
                    self.ctx.flush_tail(fused[0], fused[2])
                else:
                    self.ctx.flush()
        self.tail_done = fused is not None
        if self.overlap:
            main.wait_stream(self.side)      # join: nothing below (Adam, frees) may pass the side work
        self.last = {"loss_terms": loss, "latent_bits": a["lbits"], "net_bits": nbits, "n_pts": n_pts}
        return de

    def _fused_tail(self, tail, loss, lbits, nbits, gs, gm):
        """(NvfStepTail, NvfAdamFuse, uncovered ranges) for a step whose optimiser rides in the slab reduction (weight
        gradients) and in the finals launch (everything those final passes write + the ranges): single-GPU steps only.
        ``tail``: dict(coef_dev= / coef_host=, inv_npts_dev= / inv_npts_host=, sched=)."""
        stats = self.epoch_acc is not None
        t = ops.step_tail_args(self.flat_p, self.flat_g, self.flat_m, self.flat_v, tail.get("coef_dev"),
                               tail.get("coef_host", (0.0, 0.0)), loss_terms=loss if stats else None,
                               lbits=lbits if stats else None, nbits=nbits if stats else None,
                               inv_npts_dev=tail.get("inv_npts_dev"), inv_npts_host=tail.get("inv_npts_host", 1.0),
                               nbits_scale=1.0 / self.n_points_total, counts=self.step_counts if stats else None,
                               acc=self.epoch_acc, done=self._tail_done, sched=tail.get("sched"))
        base, esz = self.flat_g.data_ptr(), 4
        cover = []
        for j in self._wg.jobs:
            if j[2] > 0:
                cover.append(((j[1] - base) // esz, (j[1] - base) // esz + j[3]))
        for o in [o for _, o in self._bias_jobs] + self._bias_cover:
            cover.append(((o.data_ptr() - base) // esz, (o.data_ptr() - base) // esz + o.numel()))
        for g1 in (gs, gm):
            cover.append(((g1.data_ptr() - base) // esz, (g1.data_ptr() - base) // esz + g1.numel()))
        if self._stem_gdn_in_finals:
            for name in ("reconstructor.activation.beta", "reconstructor.activation.gamma"):
                off, cnt = self.slices[name]
                cover.append((off, off + cnt))
        n = self.flat_g.numel()
        key = tuple(sorted(c for c in cover if 0 <= c[0] < n))
        ranges = self._tail_ranges.get(key)
        if ranges is None:
            ranges, pos = [], 0
            for lo, hi in key:
                if lo < pos:
                    raise RuntimeError("fused tail: two launches write the same gradient elements")
                if lo > pos:
                    ranges.append((pos, lo))
                pos = hi
            if pos < n:
                ranges.append((pos, n))
            if len(ranges) > 16:
                raise RuntimeError("fused tail: more than 16 uncovered gradient ranges")
            self._tail_ranges[key] = ranges
        return t, ops.adam_fuse_args(t), ranges

    # ------------------------------------------------------------------ steps
    def _batch(self, idx_dev):
        """(gt, dist, gt16, gt8, emb) rows of the mini-batch: one fused gather."""
        return ops.gather_rows_multi([self.gt, self.dist, self.gt16, self.gt8, self.emb], idx_dev)

    def enable_epoch_stats(self):
        """Device accumulators behind NVFPCC.py train's per-epoch log line (the reference syncs ~14 .item()s per step,
        NVFPCC.py:190-221): epoch_acc[16] = the three focal terms, b_latent, b_net, non-finite objective terms,
        non-finite gradient entries, steps, then the per-step ratios Pacc, Nacc, S1Pacc, S1Nacc, S2Pacc, S2Nacc summed
        over the steps, sse, denom (nvf_step_tail)."""
        if self.epoch_acc is None:
            self.epoch_acc = torch.zeros(16, device=self.dev)

    def read_epoch_stats(self, reset=True, reduce=None, world=1):
        """epoch_acc as a float64 numpy array [16]: ONE host sync per epoch.  ``reduce``: the data-parallel SUM
        all-reduce, applied to the 16 floats first; entries that every rank holds identically (b_net, the step count and
        the ratios / sse / denom, which nvf_step_tail derives from all-reduced counts) are divided by ``world`` after it.
        Raises on the reference's NaN checks (NVFPCC.py:199-212: 'Problem in loss' / 'Problem with grad') instead of
        opening an IPython shell; a non-finite gradient entry never reaches its parameter (nvf_step_tail skips it)."""
        acc = self.epoch_acc.clone()
        if reduce is not None:
            reduce(acc)
        acc = acc.double().cpu().numpy()
        if reduce is not None and world > 1:
            acc[4] /= world
            acc[7:16] /= world
        if reset:
            self.epoch_acc.zero_()
        if acc[5] > 0:
            raise ValueError("Problem in loss: %d non-finite objective terms this epoch" % int(acc[5]))
        if acc[6] > 0:
            raise ValueError("Problem with grad: %d non-finite gradient entries this epoch" % int(acc[6]))
        return acc

    def train_log_fields(self, acc, nsteps):
        """The 16 numbers of the reference's TRAIN line after 'seconds]' (NVFPCC.py:261-281), from read_epoch_stats:
        Loss, PosiPenal, PosiGain, Pacc, Nacc, S1 Loss, S2 Loss, S1Pacc, S1Nacc, S2Pacc, S2Nacc, bpp, b_latent, b_net,
        MSE1, PSNR1 -- means over the epoch's mini-batches, MSE1 = sum sse / sum denom (0 / 0 = nan, as there)."""
        cnt = float(nsteps)
        ls, bl, bn = acc[0:3] / cnt, acc[3] / cnt, acc[4] / cnt
        with np.errstate(divide="ignore", invalid="ignore"):
            mse1 = np.float64(acc[14]) / np.float64(acc[15])
            psnr1 = 20 * np.log10(1023 / np.sqrt(mse1 / 3))
        loss = ls.sum() + self.lmbda * (bl * self.w1 + bn * self.w2)
        return [loss, 0.0, 0.0, acc[8] / cnt, acc[9] / cnt, ls[1], ls[2], acc[10] / cnt, acc[11] / cnt,
                acc[12] / cnt, acc[13] / cnt, bl + bn, bl, bn, mse1, psnr1]

    def _tail(self, n_pts):
        """All-reduce hook (data parallelism), then Adam (+ the epoch sums): nvf_step_tail with host coefficients."""
        if self.grad_hook is not None:
            self.grad_hook(self.flat_gx)
        self.opt_step += 1
        t = self.last
        stats = self.epoch_acc is not None
        ops.step_tail(self.flat_p, self.flat_g, self.flat_m, self.flat_v, None,
                      ops.adam_coefficients(self.lr, self.opt_step),
                      loss_terms=t["loss_terms"] if stats else None, lbits=t["latent_bits"] if stats else None,
                      nbits=t["net_bits"] if stats else None, inv_npts_host=1.0 / n_pts,
                      nbits_scale=1.0 / self.n_points_total, counts=self.step_counts if stats else None,
                      acc=self.epoch_acc, done=self._tail_done)

    def _idle_backward(self):
        """A rank whose share of a short last mini-batch is empty (NVFPCC.py:149 with 917 mod 16 = 5 blocks on 8
        GPUs): no block terms, but its 1/W share of the replicated weight-rate gradient (and of d/d sigma, d/d mu of
        the likelihood model) still goes into the all-reduce, so the summed gradient is the single-GPU one."""
        net, Ls = self.net, self.layers
        self.flat_gx.zero_()
        lm = net.reconstructor.likelihood_model
        nbits = torch.empty(7, device=self.dev)
        g_net = self.lmbda * self.w2 / self.n_points_total
        ops.weight_rate_batch([Ls[n].mod.kernel for n in TRUNK], [Ls[n].gk for n in TRUNK], lm.sigma, lm.mu, nbits,
                              self._g("reconstructor.likelihood_model.sigma"),
                              self._g("reconstructor.likelihood_model.mu"), g_host=g_net * self.rate_grad_scale)
        self.last = {"loss_terms": torch.zeros(4, device=self.dev), "latent_bits": torch.zeros(1, device=self.dev),
                     "net_bits": nbits, "n_pts": 1.0}

    def train_step(self, idx_host, q, idx_dev=None, update=True, n_pts=None):
        """One mini-batch decoder update (NVFPCC.py:149-223, minus logging).  Under data parallelism
        ``idx_host`` is this rank's share of the global mini-batch (possibly empty) and ``n_pts`` the occupied-voxel
        count of the WHOLE mini-batch (host-known: every rank derives the same epoch order)."""
        idx_host = np.asarray(idx_host, np.int64)
        self.noise_step += 1       # a fresh draw per train-mode forward whatever q is (latent noise, network.py:4516)
        a = None
        if idx_host.shape[0] == 0:
            self._idle_backward()
            n_pts = 1.0
        else:
            if idx_dev is None:
                idx_dev = torch.from_numpy(idx_host).to(self.dev)
            if n_pts is None:
                n_pts = float(self.counts[idx_host].sum())
            gt, dist, gt16, gt8, e = self.batch_and_prepare(idx_dev, q, with_rate=not self.allow_overlap,
                                                            stem_mode="train")
            a = self.forward(e, "train", idx_dev, defer_heads=True)
            tail = None
            if update and self.grad_hook is None and not self.allow_overlap:
                # single GPU: the optimiser rides in the slab reduction and the finals launch (no all-reduce in between)
                tail = dict(coef_host=ops.adam_coefficients(self.lr, self.opt_step + 1), inv_npts_host=1.0 / n_pts)
            self.backward(a, gt, dist, gt16, gt8, n_pts, "train", idx_dev, want_w=True, want_emb=False, fuse=tail)
            if tail is not None and self.tail_done:
                self.opt_step += 1
                update = False
        if update:
            self._tail(n_pts)
        elif self.grad_hook is not None:
            self.grad_hook(self.flat_gx)
        return a

    def latent_step(self, q, lo=0, hi=None, update=True):
        """Full-batch latent update over blocks [lo, hi) (NVFPCC.py:225-251); no weight gradients."""
        hi = self.N_leaf if hi is None else hi
        self.noise_step += 1
        ids = torch.arange(lo, hi, device=self.dev)
        n_pts = float(self.counts.sum())          # the reference divides by the points of ALL blocks
        self.prepare_weights(q)
        e = self.emb[lo:hi]
        a = self.forward(e, "train", ids)
        de = self.backward(a, self.gt[lo:hi], self.dist[lo:hi], self.gt16[lo:hi], self.gt8[lo:hi], n_pts, "train",
                           ids, want_w=False, want_emb=True)
        if update:
            self.emb_step += 1
            ops.adam_step(self.emb[lo:hi].view(-1), de.view(-1), self.emb_m[lo:hi].view(-1),
                          self.emb_v[lo:hi].view(-1), self.lr_emb, self.emb_step)
        return a, de

    def eval_forward(self, lo=0, hi=None, q=2):
        hi = self.N_leaf if hi is None else hi
        ids = torch.arange(lo, hi, device=self.dev)
        self.prepare_weights(q)
        return self.forward(self.emb[lo:hi], "eval", ids)

    def eval_sums(self, lo=0, hi=None, q=2):
        """Additive sums behind the TEST line (NVFPCC.py:308-364) over blocks [lo, hi): [0:3] the three focal terms,
        [3:21] the 18 metric counts of ops.metrics3 (main output, head 0, head 1), [21] the latent bits.  Every entry is a
        SUM over blocks, so a rank evaluates its contiguous shard and ONE all-reduce of these 22 floats gives the
        full-batch numbers (eval shards contiguously; the reference evaluates on one device)."""
        hi = self.N_leaf if hi is None else hi
        out = torch.zeros(22, device=self.dev)
        if hi <= lo:
            return out
        a = self.eval_forward(lo, hi, q)
        loss = torch.empty(4, device=self.dev)
        gt, dist, gt16, gt8 = self.gt[lo:hi], self.dist[lo:hi], self.gt16[lo:hi], self.gt8[lo:hi]
        ops.focal_loss_multi([(a["p2"], gt, dist, 0.9, 1.0), (a["p0"], gt8, None, 0.85, 0.0),
                              (a["p1"], gt16, None, 0.85, 0.0)], loss)
        c = ops.metrics3([a["p2"], a["p0"], a["p1"]], [gt, gt8, gt16], [dist, None, None], 0.5, 0.6)
        out[0:3] = loss[0:3]
        out[3:21] = c[0:18]
        out[21] = a["lbits"].reshape(-1)[0]
        return out

    def weight_bits(self):
        """Sum of the 7 quantised kernels' rate terms (identical on every rank): host float, one sync."""
        nbits = torch.empty(7, device=self.dev)
        lm = self.net.reconstructor.likelihood_model
        ops.weight_rate_batch([self.layers[n].mod.kernel for n in TRUNK], None, lm.sigma, lm.mu, nbits)
        return float(nbits.sum().item())

    def loss_value(self):
        """Host scalar of the last step's objective (one sync; logging only)."""
        t = self.last
        terms = t["loss_terms"].cpu().numpy()
        b_latent = t["latent_bits"].item() / t["n_pts"]
        b_net = t["net_bits"].sum().item() / self.n_points_total
        return float(terms[0] + terms[1] + terms[2] + self.lmbda * (b_latent * self.w1 + b_net * self.w2))


class GraphedTrainStep:
    """The whole mini-batch step -- step head, forward, losses, backward, [all-reduce], Adam + epoch sums -- captured
    once into a HIP graph and replayed: removes ~90 host-side launches per step (the reference pays ~1 500 aten
    dispatches).  Everything that changes from step to step comes from device memory: block ids, the noise-step
    counter, the rate coefficient lambda*w1/n_pts (and 1/n_pts for the log line) and Adam's two step-dependent
    coefficients (lr / (1 - b1^t), sqrt(1 - b2^t)) are one ROW of a schedule the host uploads once per epoch
    (``load_schedule``); the last kernel of step s (nvf_step_tail) copies row s + 1 over the buffer the kernels read, so
    a replay costs the host one graph launch and the GPU no host-to-device copy.

    Data parallelism: ``collective`` = "graph" captures the all-reduce of the gradient buffer as a node of the graph
    (the hand-over between the compute stream and RCCL's stream is then a graph edge instead of two event waits per
    step); "host" ends the graph after the backward pass and launches the all-reduce hook and the optimiser node from
    the host.  The default is "host" (dist.attach leaves it there; NVF_GRAPH_COLLECTIVE=graph or collective="graph" opts in,
    falling back to "host" in-process if the capture fails)."""

    CAP = 4096       # rows of the device-resident schedule (longer schedules are loaded in pieces)
    # steps per replay of the unrolled graphs, largest first (a run of n loaded steps is replayed greedily: 57 = 3 x 16 + 8
    # + 1 is five graph launches; 20 = 16 + 4 two)
    UNROLL = tuple(sorted({max(int(v), 1) for v in os.environ.get("NVF_GRAPH_UNROLL", "16,8,4,2").split(",") if v.strip()},
                          reverse=True))

    def __init__(self, eng, batch, q, ring=2, collective=None, unroll=None):
        self.eng, self.batch, self.q = eng, batch, q
        un = self.UNROLL if unroll is None else ((unroll,) if isinstance(unroll, int) else tuple(unroll))
        self.unrolls = tuple(u for u in sorted({max(int(v), 1) for v in un}, reverse=True) if u > 1)
        self.unroll = self.unrolls[0] if self.unrolls else 1
        dev = eng.dev
        if collective is None:
            collective = getattr(eng, "collective_mode", None) or os.environ.get("NVF_GRAPH_COLLECTIVE", "host")
        self.collective = collective if eng.grad_hook is not None else "none"
        # the buffer the step's kernels read: [idx (B x i64) | noise step (u64) | lambda*w1/n_pts, 1/n_pts (2 x f32) |
        # Adam coefficients (2 x f32)]; sched = [cursor | unused | CAP + 1 rows of the same layout]
        # -- one allocation [step buffer | cursor | unused | rows], so that loading a schedule is ONE host-to-device copy
        nw = self.nw = batch + 3
        self.sched = torch.zeros(nw + 2 + (self.CAP + 1) * nw, dtype=torch.int64, device=dev)
        self.buf = self.sched[:nw]
        self.cursor, self.rows = self.sched[nw:nw + 1], self.sched[nw + 2:]
        # staging ring: the host may load the next schedule while the copy of the previous one has not executed yet
        self.pins = [torch.zeros(self.sched.numel(), dtype=torch.int64).pin_memory() for _ in range(max(int(ring), 1))]
        self.pin_np = [p.numpy() for p in self.pins]                    # the same memory, for the NumPy row fill
        self.pin_events = [None] * len(self.pins)
        self.pin_gen = [0] * len(self.pins)          # generation of each slot's contents (stale handles are refused)
        self.loads = 0
        self.pending = []            # (n_pts) of the loaded steps not replayed yet
        self.idx = self.buf[:batch]
        self.step = self.buf[batch:batch + 1]
        rate = self.buf[batch + 1:batch + 2].view(torch.float32)
        self.g_lat, self.inv_npts = rate[0:1], rate[1:2]
        self.coef = self.buf[batch + 2:batch + 3].view(torch.float32)
        eng._step_dev, eng._g_lat_dev = self.step, self.g_lat
        first = torch.zeros(nw, dtype=torch.int64)
        first[:batch] = torch.arange(batch) % eng.N_leaf
        first[batch + 1:batch + 2].view(torch.float32)[:] = 1.0
        self.buf.copy_(first)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        acc = eng.epoch_acc
        try:
            with torch.cuda.stream(side):      # warm-up: allocates the grow-only workspaces outside the graph; the
                for _ in range(2):             # tail (no workspace) is left out, so no state is touched
                    self._body(tail=False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if self.collective == "graph":
                try:
                    self._capture(True)
                except Exception as e:       # noqa: BLE001 -- a failed capture of the collective: host launch instead
                    import warnings
                    warnings.warn(f"all-reduce could not be captured into the step graph ({e}); launching it from "
                                  f"the host behind the graph instead")
                    torch.cuda.synchronize()
                    self.collective = eng.collective_mode = "host"
                    self._capture(False)
            else:
                self._capture(self.collective != "host")
        finally:
            eng._step_dev, eng._g_lat_dev = None, None

    def _capture(self, tail):
        """graph: one step.  graphs_u[u]: u steps back to back -- every step's last kernel hands the step buffer
        over to the next schedule row, so the bodies are identical; a graph launch costs ~9 us of idle GPU between two
        replays (measured: 461 us period against 452 us of kernels), which an unrolled graph pays once per u steps.
        Only when the optimiser is inside the graph (not with a host-launched all-reduce)."""
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body(tail=tail)
        self.eng._graphs_captured += 1
        self.out1, self.last1 = self.out, self.last          # each graph writes tensors of its own
        self.graphs_u = {}
        for u in (self.unrolls if tail else ()):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(u):
                    self._body(tail=tail)
            self.graphs_u[u] = (g, self.out, self.last)       # ... an unrolled one: those of its last step
        self.graph_u = self.graphs_u[self.unroll][0] if self.graphs_u else None

    def _body(self, tail):
        eng = self.eng
        gt, dist, gt16, gt8, e = eng.batch_and_prepare(self.idx, self.q, with_rate=not eng.allow_overlap,
                                                       stem_mode="train")
        a = eng.forward(e, "train", self.idx, defer_heads=True)
        spec = None
        if tail and eng.grad_hook is None and not eng.allow_overlap:      # single GPU: no launch of its own for the tail
            spec = dict(coef_dev=self.coef, inv_npts_dev=self.inv_npts, sched=(self.buf, self.rows, self.cursor, self.nw))
        eng.backward(a, gt, dist, gt16, gt8, 1.0, "train", self.idx, want_w=True, want_emb=False, fuse=spec)
        self.out = a
        self.last = dict(eng.last)
        if tail and not (spec is not None and eng.tail_done):
            self._tail()

    def _tail(self):
        eng = self.eng
        if eng.grad_hook is not None:
            eng.grad_hook(eng.flat_gx)
        t = self.last
        stats = eng.epoch_acc is not None
        ops.step_tail(eng.flat_p, eng.flat_g, eng.flat_m, eng.flat_v, self.coef,
                      loss_terms=t["loss_terms"] if stats else None, lbits=t["latent_bits"] if stats else None,
                      nbits=t["net_bits"] if stats else None, inv_npts_dev=self.inv_npts,
                      nbits_scale=1.0 / eng.n_points_total, counts=eng.step_counts if stats else None,
                      acc=eng.epoch_acc, done=eng._tail_done, sched=(self.buf, self.rows, self.cursor, self.nw))

    def stage_schedule(self, steps):
        """Host half of load_schedule: validates `steps` and fills the rows (block ids, noise steps, rate and Adam
        coefficients continuing from the engine's counters) into a pinned staging buffer; nothing reaches the device and no
        engine state changes.  Returns a handle for load_schedule -- valid while the engine's step counters and learning rate
        stay what they were (a training loop can prepare the next epoch's schedule while the GPU still runs this one)."""
        eng, B, nw = self.eng, self.batch, self.nw
        # the arrays form is recognised by its first member being a 2-D ndarray (a tuple of two (ids, n_pts) pairs is a
        # two-step list, not the arrays form)
        def is2d(x):          # ndarray / tensor [n, B], or a sequence of 1-D rows ((ids, n_pts) has a scalar / None second member)
            if isinstance(x, np.ndarray) or torch.is_tensor(x):
                return x.ndim == 2
            return (isinstance(x, (list, tuple)) and len(x) > 0 and
                    all(isinstance(r, (list, tuple, np.ndarray)) and np.ndim(r) == 1 for r in x))
        arrays = len(steps) == 2 and is2d(steps[0])
        if arrays and torch.is_tensor(steps[0]):
            steps = (steps[0].cpu().numpy(), steps[1].cpu().numpy() if torch.is_tensor(steps[1]) else steps[1])
        if arrays:
            ids_all, npts = np.asarray(steps[0], np.int64), np.asarray(steps[1], np.float64)
        else:
            ids_all = np.stack([np.asarray(ids, np.int64).reshape(-1) for ids, _ in steps]) if len(steps) else np.zeros((0, B), np.int64)
            npts = np.array([float(eng.counts[ids_all[k]].sum()) if p is None else float(p)
                             for k, (_, p) in enumerate(steps)], np.float64)
        n = ids_all.shape[0]
        if not (0 < n <= self.CAP):
            raise ValueError("load_schedule: 1..%d steps" % self.CAP)
        if ids_all.shape != (n, B) or npts.shape != (n,):
            raise ValueError("load_schedule: ids must be [n, %d] and n_pts [n]; got %s, %s" % (B, ids_all.shape, npts.shape))
        slot = self.loads % len(self.pins)
        self.loads += 1
        if self.pin_events[slot] is not None:
            self.pin_events[slot].synchronize()
        host = self.pin_np[slot]
        # the rows are filled as ONE NumPy array (per-element writes into a torch tensor cost ~10 us each: milliseconds per
        # epoch of host time that a short timed region would see)
        rows = host[nw + 2:nw + 2 + (n + 1) * nw].reshape(n + 1, nw)      # written in place (pinned memory)
        f32 = rows.view(np.float32)               # [n + 1, 2 nw]
        rows[:n, :B] = ids_all
        rows[:n, B] = eng.noise_step + 1 + np.arange(n)
        f32[:n, 2 * (B + 1)] = (eng.lmbda * eng.w1 / npts).astype(np.float32)
        f32[:n, 2 * (B + 1) + 1] = (1.0 / npts).astype(np.float32)
        f32[:n, 2 * (B + 2):2 * (B + 2) + 2] = ops.adam_coefficients_n(eng.lr, eng.opt_step + 1, n)
        rows[n] = rows[n - 1]                     # what the last step's tail copies (never used)
        host[:nw] = rows[0]                       # the step buffer starts as row 0 ...
        host[nw], host[nw + 1] = 1, 0             # ... and the cursor at 1: the tail of the first step fetches row 1
        self.pin_gen[slot] += 1                   # a handle owns its slot only until the slot is staged again
        m = nw + 2 + (n + 1) * nw
        return {"_staged": True, "slot": slot, "gen": self.pin_gen[slot], "n": n, "npts": npts.tolist(), "words": m,
                "src": self.pins[slot][:m], "dst": self.sched[:m],          # (sliced here, not inside a timed region)
                "state": (eng.noise_step, eng.opt_step, eng.lr, eng.lmbda, eng.w1)}

    def load_schedule(self, steps):
        """steps: [(block ids of this rank [batch], n_pts of the whole mini-batch or None)] in replay order (at most
        CAP) -- or the same as two arrays (ids [n, batch] int64 ndarray, n_pts [n] ndarray) -- or a handle from
        stage_schedule.  One host-to-device copy for all of them; Adam / noise counters continue from the engine's.
        Nothing is touched before the input has been validated."""
        eng = self.eng
        if self.pending:
            raise ValueError("load_schedule: only after every loaded step has been replayed")
        h = steps if (isinstance(steps, dict) and steps.get("_staged")) else self.stage_schedule(steps)
        if h["state"] != (eng.noise_step, eng.opt_step, eng.lr, eng.lmbda, eng.w1):
            raise ValueError("load_schedule: the staged schedule was made for other step counters / coefficients")
        slot, m = h["slot"], h["words"]
        if h.get("gen") != self.pin_gen[slot]:
            raise ValueError("load_schedule: stale handle -- its staging slot has been filled again since (ring=%d)"
                             % len(self.pins))
        self.pending.extend(h["npts"])
        h["dst"].copy_(h["src"], non_blocking=True)
        ev = self.pin_events[slot]
        if ev is None:
            ev = self.pin_events[slot] = torch.cuda.Event()
        ev.record()

    def replay(self):
        """Run the next loaded step."""
        eng = self.eng
        n_pts = self.pending.pop(0)
        eng.noise_step += 1
        eng.opt_step += 1
        self.graph.replay()
        self.out, self.last = self.out1, self.last1
        eng.last = dict(self.last)
        eng.last["n_pts"] = n_pts
        if self.collective == "host":
            self._tail()
        return self.out

    def replay_all(self):
        """Run every loaded step: the run of n steps is cut greedily into the largest unrolled graphs that fit (57 = 16 + 16
        + 16 + 8 + 1) and replayed SMALLEST FIRST: launching a graph costs the host time that grows with its node count, and
        only the first launch of a run is exposed (the later ones are issued while the GPU is busy) -- so the first one
        should be the cheapest (a 20-step timed region: 4 + 16 instead of 16 + 4)."""
        eng = self.eng
        n, sizes = len(self.pending), []
        for U in self.unrolls:
            if U in self.graphs_u:
                while n >= U:
                    sizes.append(U)
                    n -= U
        sizes += [1] * n
        for U in reversed(sizes):
            if U == 1:
                self.replay()
                continue
            g, out, last = self.graphs_u[U]
            n_pts = self.pending[U - 1]
            del self.pending[:U]
            eng.noise_step += U
            eng.opt_step += U
            g.replay()
            self.out, self.last = out, last
            eng.last = dict(self.last)
            eng.last["n_pts"] = n_pts
        return self.out

    def prime(self, rounds=1):
        """Replay every captured graph `rounds` times (real training steps on block ids 0..): the first launch of a graph
        pays a one-off upload that a measurement should not hold.  For benchmarks; training does not need it.  Returns the
        number of optimiser steps run."""
        B, N = self.batch, self.eng.N_leaf
        sizes = [u for u in self.unrolls if u in self.graphs_u] + [1]
        n = 0
        for _ in range(max(int(rounds), 1)):
            for u in sizes:
                self.load_schedule([((np.arange(B) + k * B) % N, None) for k in range(u)])
                self.replay_all()
                n += u
        torch.cuda.synchronize()
        return n

    def __call__(self, idx_host, n_pts=None):
        """One step with its own one-row schedule (tests; the training loop loads an epoch at a time)."""
        self.load_schedule([(idx_host, n_pts)])
        return self.replay()


class EpochDriver:
    """The mini-batch phase of one epoch of NVFPCC.py train (NVFPCC.py:149-223) on this rank: full-size mini-batches
    replay the captured graph of their (share size, q), anything else (the short last batch, an empty share) takes the
    host-launched step.  No host synchronisation: the log line's sums stay in device accumulators
    (TrainEngine.read_epoch_stats, once per epoch)."""

    def __init__(self, eng, batch, rank=0, world=1, use_graph=True):
        from . import dist as nd
        self.eng, self.batch, self.rank, self.world, self.use_graph = eng, int(batch), rank, world, use_graph
        self.nd = nd
        self.graphs = {}
        eng.enable_epoch_stats()

    def run(self, order, q):
        eng, B = self.eng, self.batch
        n = len(order)
        nsteps = (n + B - 1) // B
        order = np.asarray(order, np.int64)
        nfull = n // B
        share = len(range(self.rank, B, self.world))     # this rank's blocks of a full mini-batch
        s = 0
        if self.use_graph and nfull > 0 and share > 0:
            # every full-size mini-batch in one go: the rows of the schedule as arrays (a Python loop over the steps costs
            # the host ~8 us each, with the GPU idle behind the epoch's read-back)
            whole = order[:nfull * B].reshape(nfull, B)
            ids_all, npts = whole[:, self.rank::self.world], eng.counts[whole].sum(axis=1).astype(np.float64)
            key = (share, q)
            g = self.graphs.get(key)
            if g is None:
                g = self.graphs[key] = GraphedTrainStep(eng, share, q)
            while s < nfull:
                e = min(nfull, s + g.CAP)
                g.load_schedule((ids_all[s:e], npts[s:e]))
                g.replay_all()
                s = e
        while s < nsteps:                                # the short last batch, empty shares, or no graph at all
            ids, whole = self.nd.shard_minibatch(order, s, B, self.rank, self.world)
            n_pts = float(eng.counts[whole].sum())
            if self.use_graph and self.world == 1 and s == nfull and len(ids) > 0 and _GRAPH_LAST:
                # one GPU: the short last mini-batch (917 mod 16 = 5 blocks) replays a single-step graph of its own size
                # instead of ~90 host launches (0.25 ms against 0.6 per epoch); same kernels, same bits
                key = (len(ids), q)
                g = self.graphs.get(key)
                if g is None:
                    g = self.graphs[key] = GraphedTrainStep(eng, len(ids), q, unroll=1)
                g.load_schedule((np.asarray(ids, np.int64)[None], np.array([n_pts])))
                g.replay_all()
            else:
                eng.train_step(ids, q, n_pts=n_pts)
            s += 1
        return nsteps
