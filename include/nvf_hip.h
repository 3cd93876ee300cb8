/*
 * nvf_hip.h -- C ABI of libnvf_hip.so: the gfx950 (MI355X) kernels under NVFPCC's
 * per-block neural-volumetric-field hot path.
 *
 * The reference has NO native boundary on this path: NVFPCC.py calls the Python
 * operator classes of utils/network.py, gdn_3d.py and utils/loss.py, which call
 * torch aten ops.  This header is the boundary this build introduces underneath
 * those classes; every entry point names the reference call site it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every tensor is fp32, contiguous, NCDHW, resident in device memory;
 *   - the caller owns every buffer (inputs, outputs, workspace); the library
 *     never allocates, frees or keeps a pointer after returning, and holds NO state of its own between
 *     calls: work that one call hands to a later one (the deferred final passes, the queued latent
 *     tail) travels in a caller-owned NvfStepCtx, so any number of engines, streams and threads can
 *     use the library at once, each with its own context;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call
 *     synchronises; outputs are overwritten unless the argument says accumulate;
 *   - return 0 on success, a positive hipError_t from the launch, or a negative
 *     NVF_E* code for a rejected argument; no C++ exception crosses the ABI;
 *   - summation order inside a kernel depends only on the output element, never
 *     on the batch size or grid, so results are batch- and rank-count-invariant.
 */
#ifndef NVF_HIP_H
#define NVF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVF_OK 0
#define NVF_EINVAL (-1)      /* bad shape / null pointer / unsupported combination */
#define NVF_EWORKSPACE (-2)  /* workspace too small */

#define NVF_ACT_NONE 0
#define NVF_ACT_RELU 1
#define NVF_ACT_SIGMOID 2

int nvf_version(void);

/* ---- step context ---------------------------------------------------------------
 * Host-side state of ONE training step in flight: the queue of deferred final passes (nvf_finals_*) and the
 * queued latent tail (nvf_latent_tail_queue).  Opaque; the caller allocates nvf_step_ctx_bytes() bytes of host
 * memory (8-byte aligned), calls nvf_step_ctx_init once, and passes the pointer to every entry point that takes
 * an `NvfStepCtx* ctx`.  ctx == NULL everywhere means "defer nothing": every final pass is launched by the call
 * that produces its partial sums.  A context must not be used from two threads at the same time; different
 * contexts are independent (tests/test_gpu_engine.py::test_two_step_contexts_do_not_interfere). */
typedef struct NvfStepCtx NvfStepCtx;
size_t nvf_step_ctx_bytes(void);
int nvf_step_ctx_init(NvfStepCtx* ctx);
/* on != 0: launches given this context keep the direct summation order (no Winograd form of a weight gradient): the
 * arithmetic that follows the reference's training trajectory to 1e-7 (torch's conv backward, NVFPCC.py:164-172);
 * 0 (default): the reduced-multiplication forms, statistically equivalent training. */
int nvf_step_ctx_set_direct(NvfStepCtx* ctx, int on);
/* Forms of the merged weight-gradient launches (nvf_wgrad_mfma3_partial / nvf_wgrad_trunk5_*) when the context does not
 * ask for the direct ones: conv2_zsplit in 0..8 (0 = default 1): z work items of conv2's Winograd gradient; conv1_wino
 * != 0: conv1's gradient in the Winograd form as well.  Each choice is another summation order (agreement to fp32
 * rounding): it belongs to the caller's context, the library reads no environment variable for it.  NVF_EINVAL outside
 * the ranges. */
int nvf_step_ctx_set_wgrad_forms(NvfStepCtx* ctx, int conv2_zsplit, int conv1_wino);

/* Adam fused into the launch that produces a gradient: an element g_base[i] written by that launch is followed by the
 * nvf_step_tail update of p_base[i], m_base[i], v_base[i] (same coefficients, same arithmetic, same non-finite rule;
 * bad_count, if non-NULL, is nvf_step_tail's acc + 6).  Applies to outputs inside [g_base, g_base + n) only. */
typedef struct NvfAdamFuse {
  const float* g_base;
  float* p_base;
  float* m_base;
  float* v_base;
  int64_t n;
  const float* coef_dev;
  float coef0_host, coef1_host, beta1, beta2, eps, reserved;
  float* bad_count;
} NvfAdamFuse;

/* ---- weight packing -------------------------------------------------------
 * Re-lays an effective kernel into the two layouts the direct-conv kernels read
 * from scalar registers: w_fwd[ci][k][co] and w_bwd[co][k'][ci].
 * conv  : w is [Cout][Cin][K^3] (F.conv3d, network.py:687,741); w_bwd taps are flipped.
 * convT : w is [Cin][Cout][K^3] (F.conv_transpose3d, network.py:621); w_bwd keeps tap order.
 * Either output may be NULL. */
int nvf_pack_conv_weight(const float* w, int cout, int cin, int k, float* w_fwd, float* w_bwd, void* stream);
int nvf_pack_convT_weight(const float* w, int cin, int cout, int k, float* w_fwd, float* w_bwd, void* stream);

/* ---- effective parameters (network.py:611-620, 677-686, 735-740) ------------
 * w_eff = f_q(kernel) + kernel_init, b_eff = b + b_init for one layer.
 * q = 1: kernel + (U-.5)/16 with U from `u` if non-NULL else Philox(seed, stream_id);
 * q = 2: round(16 k)/16 (half-to-even); any other q: raw kernel. */
int nvf_effective_params(const float* kernel, const float* kernel_init, const float* u, float* w_eff, int n,
                         const float* b, const float* b_init, float* b_eff, int nb, int q, uint64_t seed,
                         uint64_t stream_id, void* stream);

/* All layers of a decoder in one launch.  `table_dev` is a device array of `nlayers` records
 *   { const float *kernel, *kernel_init, *b, *b_init; float *w_fwd, *w_bwd, *b_eff;
 *     int32 dim0, dim1, k3, kind (0 conv / 1 convT), quantised, layer_id, nbias, pad; }
 * (nvf_layer_desc_size() bytes each).  Writes w_eff = f_q(kernel) + kernel_init directly in the packed
 * layouts and b_eff = b + b_init.  Weight noise (q = 1): Philox(seed, ((step + *step_dev) << 8) | layer_id). */
size_t nvf_layer_desc_size(void);
int nvf_prepare_weights(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step,
                        const uint64_t* step_dev, void* stream);

/* ---- gather convolution (F.conv3d fwd; bwd-data of conv3d and conv_transpose3d) --
 * y[b,co,o] = act(bias[co] + sum_{ci,k} x[b,ci, stride*o - pad + k] * w[ci][k][co])
 *             (+ addend[b,co,o]) (* (mask[b,co,o] > 0))
 * w is a packed w_fwd (forward) or w_bwd (backward-data; then cin/cout are swapped
 * by the caller and pad = K-1-p for conv, pad = p with stride 2 for convT).
 * bias, addend, mask may be NULL.  variant: 0 = tuned LDS-tiled kernel (falls back to 1 for shapes without a
 * tiled instantiation), 1 = one-thread-per-output kernel (same fmaf order: bit-identical), >= 2 = tuning alternates.
 * Replaces F.conv3d at network.py:687,741 and the
 * autograd backward of network.py:621,687. */
int nvf_conv3d_gather(const float* x, const float* w, const float* bias, float* y, const float* addend,
                      const float* mask, int batch, int cin, int cout, int k, int stride, int pad, int din,
                      int hin, int win, int dout, int hout, int wout, int act, int variant, void* stream);

/* ---- transposed convolution k5 s2 forward (F.conv_transpose3d, network.py:621) ---
 * y[b,co,o] = act(bias[co] + sum_{ci,k : o+pad-k = 2i} x[b,ci,i] * w[ci][k][co]),
 * dout = 2*din + 3 (pad 0) or 2*din (pad 2, output_padding 1).  w is w_fwd. */
int nvf_convT3d_k5s2_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int cin, int cout,
                         int pad, int din, int hin, int win, int dout, int hout, int wout, int act, int variant,
                         void* stream);

/* ---- matrix-core (v_mfma_f32_16x16x4_f32, exact fp32) form of the 4^3 convolutions with 8 output channels
 * (conv1 / conv2 of chanstr 8,16,8,8: F.conv3d network.py:687 and its autograd backward-data).  Same contract
 * as nvf_conv3d_gather with k = 4, stride 1, but the weights are pre-packed MFMA A-fragments:
 *   nvf_pack_mfma_k4(gather_w [cin][64][8] (= w_fwd, or w_bwd for the backward-data pass), cin, 8, pair_axis, wp)
 *   pair_axis 0: rows pair outputs along x (forward, pad 0); 2: along z (backward-data, pad 3)
 * wp holds nvf_pack_mfma_k4_floats(cin, pair_axis) floats (layout [ci group][ky][kx][kz][lane]).  Per-output accumulation order is fixed
 * (input-channel group, ky, kx, kz), so results do not depend on batch size or tiling; they differ from
 * nvf_conv3d_gather's order by fp32 rounding only.  NVF_EINVAL = no instantiation for this shape. */
size_t nvf_pack_mfma_k4_floats(int cin, int pair_axis);
int nvf_pack_mfma_k4(const float* gather_w, int cin, int cout, int pair_axis, float* wp, void* stream);
/* up to 8 packings (8 output channels each) in one launch */
int nvf_pack_mfma_k4_multi(const float* const* gather_ws, float* const* wps, const int* cins, const int* pair_axes,
                           int n, void* stream);
int nvf_conv3d_k4_mfma(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                       const float* mask, int batch, int cin, int cout, int pad, int pair_axis, int din, int hin,
                       int win, int dout, int hout, int wout, int act, int variant, void* stream);
/* ... with bias_part (backward-data through a ReLU mask: bias NULL, act NVF_ACT_NONE, addend NULL, mask given): the
 * launch also leaves, per (workgroup, wave), the 8 channel sums of the outputs it stored -- *bias_nparts (<= 2048) slabs
 * of 8 floats whose sum (a jtotal = 8 job of nvf_wgrad_reduce_multi) is the bias gradient of the layer below. */
int nvf_conv3d_k4_mfma_bias(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                            const float* mask, int batch, int cin, int cout, int pad, int pair_axis, int din, int hin,
                            int win, int dout, int hout, int wout, int act, int variant, float* bias_part,
                            int* bias_nparts, void* stream);

/* ---- reduced-multiplication backward-data of the valid 4^3 convolutions with 8 -> 8 channels (conv2 of chanstr
 * 8,16,8,8: the autograd backward of F.conv3d, network.py:687, NVFPCC.py:197): Winograd F(2x2, 4x4) over (y, x), direct
 * over z on the matrix cores (conv_wino.hip) -- 2.56 x fewer multiplications than nvf_conv3d_k4_mfma's gather form.
 * BACKWARD PASSES ONLY: fp32 error 1e-6 of max |dx| against 6e-7 for the direct form, but not a fixed fmaf chain per
 * output, so the forward (bit-exact batch invariance) never uses it.
 * dy [batch, 8, din^3]; dx, mask [batch, 8, (din + 3)^3]: dx = mask > 0 ? conv_full(dy, w) : 0 (the ReLU of the layer
 * below).  wp = nvf_pack_mfma_all kind 40 (c0 = c1 = 8) of the layer's gather-form backward weights w_bwd,
 * nvf_pack_wino_k4_floats() floats.  ppc = pairs of output planes per work unit (0: default).  bias_part (optional):
 * *bias_nparts slabs of 8 floats, the channel sums of dx per work unit (the bias gradient of the layer below, a
 * jtotal = 8 job of nvf_wgrad_reduce_multi).  NVF_EINVAL = no instantiation (din 32: conv2, 16: conv1).
 * Two kernels compute this, with the same bits: conv_wino1.hip (one accumulator set per wave, two waves per SIMD: the
 * default, ppc = 0, or bit 16 of ppc set with the low byte = pairs per work unit) and conv_wino.hip (two sets, every plane
 * walked once: an explicit even ppc in the low byte). */
size_t nvf_pack_wino_k4_floats(void);
int nvf_conv3d_k4_wino_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int din, int ppc,
                           float* bias_part, int* bias_nparts, void* stream);
/* The FORWARD of the same layers in the same form, y = relu(conv3d(x, w) + bias), for TRAINING steps only (NVFPCC.py:160,
 * 234): x [batch, 8, din^3] (din 35: conv2, 19: conv1), y [batch, 8, (din - 3)^3], wp = kind 40 of the layer's w_fwd.  Its
 * outputs differ from nvf_conv3d_k4_mfma's by fp32 rounding (1e-6 of max |y|) and are not a fixed fmaf chain per output:
 * eval / encode / decode (NVFPCC.py:316, 516, 628), whose occupancy must be batch-invariant bit for bit, never use it. */
int nvf_conv3d_k4_wino_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int din, int ppc,
                           void* stream);

/* The same form for the WIDE decoder's 4^3 layers (16 -> 16 channels; conv16_wino.hip): rows = the 16 output channels,
 * two output planes in flight, the five input planes of a pair walked once per pair.  Training steps only, as above.
 * wp = nvf_pack_mfma_all kind 41 (c0 = c1 = 16) of w_bwd (backward-data) / w_fwd (forward), nvf_pack_wino16_k4_floats()
 * floats.  bwd: dy [batch, 16, din^3] (din 32 / 16), dx, mask [batch, 16, (din + 3)^3]; fwd: x [batch, 16, din^3]
 * (din 35 / 19), y [batch, 16, (din - 3)^3].  ppc = pairs of output planes per work unit (0: default).  bias_part (optional,
 * bwd): *bias_nparts slabs of 16 floats, the channel sums of dx per work unit (a jtotal = 16 job of nvf_wgrad_reduce_multi*).
 * conv2's backward-data without bias_part runs, by default (ppc 0; or bit 16 of ppc with the low byte = output planes per
 * work unit), in conv16_wino1.hip: one output plane in flight per wave, two waves per SIMD -- the same bits. */
size_t nvf_pack_wino16_k4_floats(void);
int nvf_conv3d_k4_wino16_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int din, int ppc,
                             float* bias_part, int* bias_nparts, void* stream);
int nvf_conv3d_k4_wino16_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int din, int ppc,
                             void* stream);

/* ... and their WEIGHT gradient (wgrad16_wino.hip: Winograd F(4x4, 2x2) over (y, x), rows = the 16 dY channels, columns =
 * the 16 X channels, two z taps per pass): partial sums only -- *nslab <= max_slabs slabs of 16384 floats, layout
 * [co][ci][kz][ky][kx], for a jtotal = 16384 job of nvf_wgrad_reduce_multi*.  dy [batch, 16, w^3] (w 32 / 16), x [batch, 16,
 * (w + 3)^3]; zsplit: z steps of an item split over this many work items (0: default; <= 8). */
int nvf_wgrad16_k4_wino_partial(const float* dy, const float* x, float* slabs, int batch, int w, int zsplit, int max_slabs,
                                int* nslab, void* stream);

/* ... and conv2's WEIGHT gradient in the corresponding form (wgrad_wino.h: Winograd F(4x4, 2x2) over (y, x) -- the taps
 * are the output, 2 x 2 tiles of dy the filter -- direct over z on the matrix cores, every MFMA lane useful): the weight
 * half of the autograd backward of F.conv3d, network.py:687.  dy [batch, 8, 32^3], x [batch, 8, 35^3] -> dw [8][8][4][4][4]
 * (and db [8] = channel sums of dy when db is not NULL).  One launch of per-workgroup slabs + the fixed-order reduction
 * (deterministic; fp32 error 4e-6 of max |dw| against 1e-6 for the direct form).  workspace: 512 * (4096 + 8) floats.
 * zsplit: the z steps of a (block, tile group) are split over this many work items (1..8).  In the training step the same
 * body runs as job 0 of nvf_wgrad_trunk5_* (NVF_WGRAD_WINO=0 selects the direct form there). */
int nvf_wgrad_k4_wino(const float* dy, const float* x, float* dw, float* db, void* workspace, size_t workspace_bytes,
                      int batch, int zsplit, void* stream);

/* ---- matrix-core form of the transposed convolutions k5 s2 with 8 output channels and padding 0 (up1, up2 of
 * chanstr 8,16,8,8; F.conv_transpose3d network.py:621).  Same contract as nvf_convT3d_k5s2_fwd; the weights are
 * MFMA A-fragments: nvf_pack_convT_mfma(w_fwd [cin][125][8], cin, 8, wp), nvf_pack_convT_mfma_floats(cin) floats.
 * Fixed per-output accumulation order (input-channel group, jy, jx, jz): independent of batch and tiling.
 * NVF_EINVAL = no instantiation for this shape. */
size_t nvf_pack_convT_mfma_floats(int cin);
int nvf_pack_convT_mfma(const float* w_fwd, int cin, int cout, float* wp, void* stream);
int nvf_convT3d_k5s2_mfma(const float* x, const float* wp, const float* bias, float* y, int batch, int cin, int cout,
                          int din, int act, int variant, void* stream);

/* ---- matrix-core backward-data of those transposed convolutions (a stride-2 gather convolution, autograd
 * backward of network.py:621): dx[b,ci,i] = (sum_{co,k} g[b,co,2i+k] w[ci][co][k] (+ addend)) (* (mask > 0)),
 * din = 2 dout + 3, cog = 8 (up2) or 16 (up1) output channels, cig channels of g.  Weights:
 * nvf_pack_s2k5_mfma(w_bwd [cig][125][cog], cig, cog, wp), nvf_pack_s2k5_mfma_floats(cig, cog) floats.
 * Fixed per-output accumulation order (channel group, ky, kx window, kz).  NVF_EINVAL = no instantiation. */
size_t nvf_pack_s2k5_mfma_floats(int cig, int cog);
int nvf_pack_s2k5_mfma(const float* gather_w, int cig, int cog, float* wp, void* stream);
int nvf_conv3d_s2k5_mfma(const float* g, const float* wp, float* dx, const float* addend, const float* mask,
                         int batch, int cig, int cog, int din, int dout, int variant, void* stream);

/* ---- matrix-core gather convolutions for 16 / 32 OUTPUT channels (the wide decoder, chanstr 16,32,16,16:
 * F.conv3d network.py:687 forward and backward-data, and the backward-data of the stride-2 transposed convolutions
 * network.py:621).  Rows of the MFMA tile are the output channels (no pairing), K = four input channels.  Same
 * contract as nvf_conv3d_gather; the weights are A fragments of the packed gather weight:
 *   nvf_pack_g16_mfma(gather_w [cin][k^3][cout] (= w_fwd, or w_bwd for a backward-data pass), cin, cout, k, wp),
 *   nvf_pack_g16_mfma_floats(cin, cout, k) floats, layout [ceil(cout/16)][cin/4][tap][lane]; cout = 8 is accepted too
 *   (rows 8..15 of the tile are zero weights and are not stored).
 * Fixed per-output accumulation order (channel group, kz, ky, kx): independent of batch and tiling.
 * NVF_EINVAL = no instantiation for this shape (the caller then uses nvf_conv3d_gather). */
size_t nvf_pack_g16_mfma_floats(int cin, int cout, int k);
int nvf_pack_g16_mfma(const float* gather_w, int cin, int cout, int k, float* wp, void* stream);
int nvf_conv3d_g16_mfma(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                        const float* mask, int batch, int cin, int cout, int k, int stride, int pad, int din, int hin,
                        int win, int dout, int hout, int wout, int act, int variant, void* stream);

/* ---- matrix-core form of the transposed convolutions k5 s2 with 16 / 32 output channels (all four of chanstr
 * 16,32,16,16: up1 / up2 with padding 0, conv0 / up0 with padding 2 and output_padding 1; F.conv_transpose3d
 * network.py:621): the output channels are the MFMA rows, the eight sub-pixel parity classes separate accumulators
 * fed by one B fragment.  Same contract as nvf_convT3d_k5s2_fwd; weights: nvf_pack_convT16_mfma(w_fwd
 * [cin][125][cout], cin, cout, wp), nvf_pack_convT16_mfma_floats(cin, cout) floats, layout [cout/16][cin/4][125][lane].
 * Fixed per-output accumulation order (input-channel group, jy, jx, jz).  NVF_EINVAL = no instantiation. */
size_t nvf_pack_convT16_mfma_floats(int cin, int cout);
int nvf_pack_convT16_mfma(const float* w_fwd, int cin, int cout, float* wp, void* stream);
int nvf_convT3d_k5s2_mfma16(const float* x, const float* wp, const float* bias, float* y, int batch, int cin, int cout,
                            int pad, int din, int act, int variant, void* stream);

/* every MFMA weight packing of a step in one launch (<= 16 jobs): kind 0 / 2 = nvf_pack_mfma_k4 with that pair
 * axis (c0 = cin), 10 = nvf_pack_convT_mfma (c0 = cin), 11 = nvf_pack_convT16_mfma (c0 = cin, c1 = cout),
 * 20 = nvf_pack_s2k5_mfma (c0 = cig, c1 = cog),
 * 30 / 31 = nvf_pack_g16_mfma with k = 4 / 5 (c0 = cin, c1 = cout), 40 / 41 = the Winograd packings of
 * nvf_conv3d_k4_wino_* (c0 = c1 = 8) / nvf_conv3d_k4_wino16_* (c0 = c1 = 16) */
int nvf_pack_mfma_all(const float* const* srcs, float* const* dsts, const int* kinds, const int* c0s, const int* c1s,
                      int n, void* stream);

/* ---- the three classifier heads of the narrow decoder in one launch each (conv0_cls on [16, 8^3], conv1_cls on
 * [8, 16^3], conv2_cls on [8, 32^3], in that order; IConv3d/QConv3d C -> 1, k 3, padding 1, network.py:4761-4768).
 * Same kernels, same results as three nvf_conv3d_gather / nvf_wgrad calls; the two small heads, latency-bound on a
 * few CUs, run beside the big one.  NVF_EINVAL for any other (channels, size) triple. */
int nvf_heads3_fwd(const float* const* xs, const float* const* w_fwds, const float* const* biases, float* const* ps,
                   const int* cs, const int* ss, int batch, int act, void* stream);
int nvf_heads3_bwd_data(const float* const* dlogits, const float* const* w_bwds, float* const* dxs,
                        const float* const* masks, const int* cs, const int* ss, int batch, void* stream);

/* nvf_focal_loss_multi (chain_sigmoid) of the three heads and nvf_heads3_bwd_data in ONE launch (NVFPCC.py:166-184
 * + the heads' backward-data): dls[h] = d term_h / d logit_h is computed while the tiles are staged, written for
 * the weight gradient, and loss[slots[h]] = term_h (final pass deferred while ctx has an open nvf_finals_begin).  dists[h] may be
 * NULL.  batch <= 32 (one loss partial per workgroup); otherwise NVF_EINVAL. */
int nvf_heads3_loss_bwd_data(const float* const* ps, const float* const* gts, const float* const* dists,
                             const float* alphas, const float* betas, const int* slots, float* loss,
                             float* const* dls, const float* const* wbs, float* const* dxs,
                             const float* const* masks, const int* cs, const int* ss, int batch, void* workspace,
                             size_t workspace_bytes, NvfStepCtx* ctx, void* stream);
/* ... and, with bias_outs (three pointers; NULL = the call above), the heads' bias gradients (autograd of the bias add,
 * utils/network.py:741): bias_outs[h][0] = sum of dls[h], from one partial per workgroup of the values it writes anyway;
 * the final pass is deferred like the loss terms' (or launched here). */
int nvf_heads3_loss_bwd_data_bias(const float* const* ps, const float* const* gts, const float* const* dists,
                                  const float* alphas, const float* betas, const int* slots, float* loss,
                                  float* const* dls, const float* const* wbs, float* const* dxs,
                                  const float* const* masks, const int* cs, const int* ss, int batch,
                                  float* const* bias_outs, void* workspace, size_t workspace_bytes, NvfStepCtx* ctx,
                                  void* stream);
/* nvf_heads3_fwd (ps[h] = act(conv(xs[h], ws[h]) + biases[h]): network.py:4761-4768 in mode 'train') and
 * nvf_heads3_loss_bwd_data_bias on those ps (NVFPCC.py:166-184 and the heads' autograd backward-data) in ONE launch: the
 * forward workgroups hand p to the loss workgroups of their block through device-scope stores and arrival counters.  Same
 * bits as the two calls.  flags: 6 * batch * 64 32-bit words (one 256-byte line per counter), zero before the first call; a call leaves them zero (one buffer per
 * stream).  batch <= 32; NVF_EINVAL for other head shapes than nvf_heads3_fwd's. */
int nvf_heads3_fwd_loss_bwd_data(const float* const* xs, const float* const* ws, const float* const* biases,
                                 float* const* ps, int act, const float* const* gts, const float* const* dists,
                                 const float* alphas, const float* betas, const int* slots, float* loss,
                                 float* const* dls, const float* const* wbs, float* const* dxs,
                                 const float* const* masks, const int* cs, const int* ss, int batch,
                                 float* const* bias_outs, void* workspace, size_t workspace_bytes, uint32_t* flags,
                                 NvfStepCtx* ctx, void* stream);
/* partial sums only: slabs[h] receives nslabs[h] (<= max_slabs) slabs of cs[h] * 27 floats, to be added by
 * nvf_wgrad_reduce_multi */
int nvf_heads3_wgrad_partial(const float* const* dlogits, const float* const* xs, float* const* slabs, const int* cs,
                             const int* ss, int batch, int max_slabs, int* nslabs, void* stream);

/* ---- fused stem for chanstr (c0, c1) = (8, 16) or (16, 32), ch <= 8 (network.py:4759-4760; gdn_3d.py:137-159) ----
 * forward : a0 = up0(x0) (convT k5 s2 p2 op1), h0 = IGDN(a0), y1 = ReLU(conv0(h0)); all three are outputs.
 * backward: from g1 = dL/d(conv0 pre-activation): da0 (= dL/d a0, after the IGDN backward) and dx0; when
 *           dbeta_hat, dgamma_hat and dw_up0 are all non-NULL also the IGDN parameter gradients and up0's weight
 *           gradient [ch][c0][5][5][5] (overwritten).  Intermediates in LDS; the workspace (always required,
 *           nvf_stem_bwd_workspace_for bytes) also holds conv0's backward-data partials, batch x c1/2 x c0 x 64 floats.
 * Weights are the packed layouts: *_w_fwd = [cin][125][cout], *_w_bwd = [cout][125][cin]. */
int nvf_stem_fwd(const float* x0, const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                 const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0, float* h0,
                 float* y1, int batch, int ch, int c0, int c1, void* stream);

/* nvf_latent_fwd and nvf_stem_fwd in ONE launch (the stem's workgroups compute their block's rounded latents themselves;
 * one more workgroup produces h, lat, x_rounded and the latent rate of the whole batch); results of the two calls, bit
 * for bit.  utils/network.py:4592-4612, 4514-4539, 4759-4760. */
int nvf_stem_latent_fwd(const float* e, const float* lat_w_fwd, const float* lat_bias, const float* lat_beta_hat,
                        const float* lat_gamma_hat, const int64_t* block_ids, const float* sigma, const float* mu,
                        float* h, float* lat, float* x_rounded, float* bits, int mode, uint64_t seed, uint64_t step,
                        const uint64_t* step_dev, const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                        const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0, float* h0,
                        float* y1, int batch, int ch, int c0, int c1, void* stream);
size_t nvf_stem_bwd_workspace(int batch, int ch);                           /* (c0, c1) = (8, 16) */
size_t nvf_stem_bwd_workspace_for(int batch, int ch, int c0, int c1);       /* 0: no kernel for (c0, c1) */
int nvf_stem_bwd(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                 const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0, float* dx0,
                 float* dbeta_hat, float* dgamma_hat, float* dw_up0, void* workspace, size_t workspace_bytes,
                 int batch, int ch, int c0, int c1, void* stream);

/* nvf_stem_bwd minus its final launch (training step): up0's weight-gradient slabs (*dw_slabs: *nslabs slabs of
 * ch * c0 * 125 floats inside `workspace`) are left to the caller's slab reduction (nvf_wgrad_reduce_multi*), the
 * IGDN parameter gradients to the deferred final passes (nvf_finals_begin; launched at once otherwise).  Same sums,
 * same order as nvf_stem_bwd.  h0 and dw_conv0_slabs (both or neither): the launch that computes conv0's backward-data
 * also leaves conv0's weight gradient (bwd-weight of network.py:621 for conv0) as `batch` slabs of c0 * c1 * 125 floats
 * ([ci][co][k], one per block, inside `workspace`) for the same reduction. */
int nvf_stem_bwd_partial(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                         const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0, float* dx0,
                         float* dbeta_hat, float* dgamma_hat, float** dw_slabs, int* nslabs, void* workspace,
                         size_t workspace_bytes, int batch, int ch, int c0, int c1, const float* h0,
                         float** dw_conv0_slabs, NvfStepCtx* ctx, void* stream);

/* The same work with NO launch of its own (narrow decoder, c0 = 8, c1 = 16, batch <= 32): queued in `ctx`, it runs as the
 * first workgroups of the next five-job weight-gradient launch given that context (nvf_wgrad_trunk5_*), which must also
 * carry a queued latent tail (nvf_latent_tail_queue: the tail is the only consumer of dx0 inside that launch; its
 * dx_addend argument is ignored there).  The stem's backward depends on g1 alone, exactly like conv0's weight gradient
 * in that launch (autograd backward of network.py:4759-4760, gdn_3d.py:137-159).  Needs an open finals queue
 * (nvf_finals_begin).  Outputs as nvf_stem_bwd_partial (same sums, same order, same bits), plus *bias_slabs: `batch`
 * slabs of c0 floats inside `workspace` whose sum is up0's bias gradient (a jtotal = c0 job of nvf_wgrad_reduce_multi*).
 * da0 / dx0 exist once that launch has run.  flags: (batch + 1) * 64 uint32 words of device memory (one 256-byte line per arrival counter), zero before the first use
 * (the launch leaves them zero).  NVF_EINVAL: another shape, no open queue, or a stem backward already queued.
 * nvf_latent_tail_cancel also drops a queued stem backward. */
int nvf_stem_bwd_queue(NvfStepCtx* ctx, const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                       const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0, float* dx0,
                       float* dbeta_hat, float* dgamma_hat, float** dw_slabs, int* nslabs, float** bias_slabs,
                       void* workspace, size_t workspace_bytes, uint32_t* flags, int batch, int ch, int c0, int c1,
                       void* stream);
int nvf_stem_bwd_pending(const NvfStepCtx* ctx);

/* ---- weight gradient (autograd backward of network.py:621,687,741) --------------
 * dw[a][b][k] (+)= sum_{n,i} p[n,a,i] * q[n,b, stride*i - pad + k]      (out_mode 0)
 * dw[b][a][K^3-1-k] (+)= same sum                                        (out_mode 1)
 * conv  : p = dY, q = X, stride 1          -> dw[cout][cin][k]
 * convT : p = X,  q = dY, stride 2         -> dw[cin][cout][k]
 * head  : p = X,  q = dlogit, out_mode 1   -> dw[1][cin][k]
 * Two launches: per-workgroup partial sums into `workspace`, then a fixed-order
 * reduction (deterministic; no float atomics).  nvf_wgrad_workspace() gives the
 * byte size for a batch. */
size_t nvf_wgrad_workspace(int batch, int a, int b, int k, int dp, int hp, int wp);
int nvf_wgrad(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes, int batch, int a,
              int b, int k, int stride, int pad, int dp, int hp, int wp, int dq, int hq, int wq, int out_mode,
              int accumulate, int variant, void* stream);
/* The same gradient in two separate steps, so that a whole backward pass needs ONE reduction launch:
 * nvf_wgrad_partial launches only the partial sums (slabs stay in `workspace`, which the caller must not reuse
 * until the reduction; *nslab = number of slabs, 0 when dw was written directly), nvf_wgrad_reduce_multi adds the
 * slabs of up to 16 gradients in the same fixed order as nvf_wgrad (results identical bit for bit). */
int nvf_wgrad_partial(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes, int batch,
                      int a, int b, int k, int stride, int pad, int dp, int hp, int wp, int dq, int hq, int wq,
                      int out_mode, int variant, int* nslab, void* stream);
int nvf_wgrad_reduce_multi(const float* const* slabs, float* const* dws, const int* nslabs, const int* jtotals,
                           int n, void* stream);

/* the three matrix-core weight gradients of the narrow trunk in one launch (partial sums only): job 0 = conv2
 * (p = dY [B,8,32^3], q = X [B,8,35^3]), job 1 = up2 (p = X [B,8,16^3], q = dY [B,8,35^3]), job 2 = conv1
 * (p = dY [B,8,16^3], q = X [B,8,19^3]); slabs[j] holds 512 slabs of 4096 / 8000 / 4096 floats, nslabs[j] = number
 * written.  Same kernels and results as three nvf_wgrad_partial calls; two workgroups share a CU. */
int nvf_wgrad_mfma3_partial(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                            int* nslabs, NvfStepCtx* ctx, void* stream);

/* up1's and conv0's weight gradients of the narrow trunk in one launch (partial sums): job 0 = up1 (p = X [B,16,8^3],
 * q = dY [B,8,19^3]), job 1 = conv0 (p = X [B,8,4^3], q = dY [B,16,8^3]); slabs[j]: up to 512 slabs of 16000 floats */
int nvf_wgrad_up1_conv0_partial(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                                int* nslabs, void* stream);

/* the five weight gradients of the narrow trunk above the stem in ONE launch (partial sums): jobs 0-2 as
 * nvf_wgrad_mfma3_partial (conv2, up2, conv1), jobs 3-4 as nvf_wgrad_up1_conv0_partial (up1, conv0): the two small VALU
 * jobs fill the slots the short matrix-core workgroups leave while conv2's are still running.  slabs[0..2]: 512 slabs of
 * 4096 / 8000 / 4096 floats, slabs[3..4]: up to 512 slabs of 16000 floats; nslabs[5].  Results identical to the two
 * separate launches.  Like nvf_wgrad_mfma3_partial it carries a queued latent tail as its first workgroup. */
int nvf_wgrad_trunk5_partial(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                             int* nslabs, NvfStepCtx* ctx, void* stream);
/* ... and, with bias_slabs[0] / bias_slabs[2] non-NULL (entry 1 is ignored), the per-workgroup channel sums of conv2's
 * / conv1's dY: nslabs[j] slabs of 8 floats each, whose sum (a jtotal = 8 job of nvf_wgrad_reduce_multi) is that
 * layer's bias gradient -- the kernel holds every dY tile in registers, and its tiles partition dY. */
int nvf_wgrad_trunk5_partial_bias(const float* const* ps, const float* const* qs, float* const* slabs,
                                  float* const* bias_slabs, int batch, int* nslabs, NvfStepCtx* ctx, void* stream);
/* ... and the weight gradients of the narrow decoder's three classifier heads (the contract of
 * nvf_heads3_wgrad_partial: three entries each, at most head_max_slabs slabs of cs[h] * 27 floats) as further workgroups
 * of the same launch: they depend on nothing it produces and run in the slots its other jobs leave. */
int nvf_wgrad_trunk5_heads_partial(const float* const* ps, const float* const* qs, float* const* slabs,
                                   float* const* bias_slabs, const float* const* head_dls, const float* const* head_xs,
                                   float* const* head_slabs, int head_max_slabs, int batch, int* nslabs,
                                   int* head_nslabs, NvfStepCtx* ctx, void* stream);
/* ... and the FIRST pass of nvf_multi_channel_sum over sum_xs (bias gradients sum_outs that no other kernel leaves
 * behind) as further workgroups of the launch; its final pass is queued in ctx (an open nvf_finals_begin) or launched
 * here.  sum_workspace: nvf_multi_channel_sum_workspace(total channels) bytes, untouched until the flush.  With the
 * partial sums made here, no final pass of the step reads what the slab reduction writes: nvf_wgrad_reduce_finals_tail.
 * coef_src / coef_live (both or neither): the launch copies two floats (the optimiser's step coefficients) from
 * coef_src to coef_live for that call. */
int nvf_wgrad_trunk5_heads_sums_partial(const float* const* ps, const float* const* qs, float* const* slabs,
                                        float* const* bias_slabs, const float* const* head_dls,
                                        const float* const* head_xs, float* const* head_slabs, int head_max_slabs,
                                        const float* const* sum_xs, float* const* sum_outs, const int* sum_channels,
                                        const int* sum_spatials, int sum_n, void* sum_workspace,
                                        size_t sum_workspace_bytes, const float* coef_src, float* coef_live,
                                        int batch, int* nslabs, int* head_nslabs,
                                        NvfStepCtx* ctx, void* stream);

/* per-channel sum over batch and space: out[c] (+)= sum x[b,c,:]  (bias gradients);
 * two launches through a caller-owned workspace of nvf_channel_sum_workspace(c) bytes */
size_t nvf_channel_sum_workspace(int c);
int nvf_channel_sum(const float* x, float* out, void* workspace, size_t workspace_bytes, int batch, int c,
                    int spatial, int accumulate, void* stream);

/* several bias gradients at once (<= 12 tensors of the same batch): outs[i][c] = sum xs[i][b,c,:]; two launches */
size_t nvf_multi_channel_sum_workspace(int total_channels);
int nvf_multi_channel_sum(const float* const* xs, float* const* outs, const int* channels, const int* spatials,
                          int ntensors, int batch, void* workspace, size_t workspace_bytes, NvfStepCtx* ctx,
                          void* stream);

/* nvf_wgrad_reduce_multi and the partial pass of nvf_multi_channel_sum in one launch (independent work), then the
 * bias sums' final pass; same results as the two separate calls */
int nvf_wgrad_reduce_multi_and_sums(const float* const* slabs, float* const* dws, const int* nslabs,
                                    const int* jtotals, int n, const float* const* xs, float* const* outs,
                                    const int* channels, const int* spatials, int ntensors, int batch,
                                    void* workspace, size_t workspace_bytes, NvfStepCtx* ctx, void* stream);
/* ... with per-gradient addends (addends[i], or addends itself, may be NULL: dws[i][j] = sum of slabs + addends[i][j] --
 * the weight-rate gradient of nvf_step_head's rate job) and, optionally, the optimiser applied to every element it
 * writes (adam NULL: none). */
int nvf_wgrad_reduce_multi_and_sums_fused(const float* const* slabs, float* const* dws, const int* nslabs,
                                          const int* jtotals, int n, const float* const* addends,
                                          const NvfAdamFuse* adam, const float* const* xs, float* const* outs,
                                          const int* channels, const int* spatials, int ntensors, int batch,
                                          void* workspace, size_t workspace_bytes, NvfStepCtx* ctx, void* stream);

/* The latent tail of a training step (backward of NVFPCC.py:186-196's latent generator on [batch, c <= 8, spatial]
 * tensors): gradient of the latent rate (+ dx_addend) -> GDN backward -> 1x1x1 weight and bias gradients, i.e.
 * nvf_latent_rate (want_grad) + nvf_gdn_bwd + nvf_wgrad + the bias sum.  Queued in `ctx`, it runs as ONE workgroup of
 * the next nvf_wgrad_mfma3_partial / nvf_wgrad_trunk5_partial or nvf_wgrad_reduce_multi_and_sums call that is given the
 * same ctx, whichever comes first (every input must already be enqueued on that call's stream), instead of three
 * dependent launches.  NVF_EINVAL: ctx is not an initialised context, or another tail is pending in it. */
int nvf_latent_tail_queue(NvfStepCtx* ctx, const float* lat, const int64_t* block_ids, const float* sigma, const float* mu,
                          const float* dx_addend, float* dlat, float* dsigma, float* dmu, const float* g_dev,
                          float g_host, int mode, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                          const float* h, const float* beta_hat, const float* gamma_hat, float* dh, float* dbeta_hat,
                          float* dgamma_hat, const float* e, float* dw, float* db, int batch, int c, int spatial);
int nvf_latent_tail_pending(const NvfStepCtx* ctx);
void nvf_latent_tail_cancel(NvfStepCtx* ctx);

/* ---- GDN / IGDN (gdn_3d.py:72-95, 137-159; LowerBound gdn_3d.py:13-29) ----------
 * beta = max(beta_hat, beta_bound)^2 - 2^-36, gamma = max(gamma_hat, 2^-18)^2 - 2^-36;
 * norm[c] = sqrt(beta[c] + sum_j gamma[c][j] x[j]^2); y = x / norm (GDN) or x * norm (IGDN).
 * bwd overwrites dx, dbeta_hat[c], dgamma_hat[c][j] (LowerBound pass-through rule applied to
 * the summed gradient); workspace: nvf_gdn_bwd_workspace(c) bytes; c <= 32. */
int nvf_gdn_fwd(const float* x, const float* beta_hat, const float* gamma_hat, float* y, int batch, int c,
                int spatial, int inverse, void* stream);
size_t nvf_gdn_bwd_workspace(int c);
int nvf_gdn_bwd(const float* x, const float* beta_hat, const float* gamma_hat, const float* dy, float* dx,
                float* dbeta_hat, float* dgamma_hat, void* workspace, size_t workspace_bytes, int batch, int c,
                int spatial, int inverse, void* stream);

/* ---- latent generator + quantiser, forward, in one launch (SingleLayerLatentGen network.py:4592-4612 followed by
 * QuantGaussianLikelihood network.py:4514-4539): h = conv1x1(e) + bias, lat = GDN(h), x_rounded = round(lat),
 * bits[0] as nvf_latent_rate.  Same arithmetic and order as nvf_conv3d_gather(k = 1) + nvf_gdn_fwd +
 * nvf_latent_rate (bit-identical outputs); c <= 8; w_fwd is the packed [ci][1][co] layout. */
int nvf_latent_fwd(const float* e, const float* w_fwd, const float* bias, const float* beta_hat,
                   const float* gamma_hat, const int64_t* block_ids, const float* sigma, const float* mu, float* h,
                   float* lat, float* x_rounded, float* bits, int batch, int c, int spatial, int mode, uint64_t seed,
                   uint64_t step, const uint64_t* step_dev, void* stream);

/* ---- latent quantisation + Gaussian rate (network.py:4514-4539, 145-161) --------
 * x_rounded = round(x) (half-to-even); v = x + (U-.5) (mode 0 = train) or x_rounded (mode 1 = eval);
 * bits[0] = sum -log2(max(Phi((v-mu+.5)/|s|) - Phi((v-mu-.5)/|s|), 1e-8)).
 * U comes from `u` if non-NULL, else Philox keyed by (seed, block_ids[b], step + *step_dev): one stream per
 * leaf block, so the noise does not depend on how blocks are sharded over GPUs; block_ids NULL
 * means 0..B-1.  Gradients (each optional, overwritten) are already multiplied by the upstream
 * gradient g = g_host * (g_dev ? *g_dev : 1): dx = dx_addend + g dbits/dx (identity through the
 * round and the noise; dx_addend, optional, is the decoder's gradient w.r.t. x_rounded), dsigma[c], dmu[c].  The LowerBound at 1e-8 passes when like >= 1e-8 or the incoming
 * gradient is negative (network.py:66-72). */
int nvf_latent_rate(const float* x, const float* u, const int64_t* block_ids, const float* sigma, const float* mu,
                    float* x_rounded, float* bits, float* dx, const float* dx_addend, float* dsigma, float* dmu,
                    const float* g_dev, float g_host, int batch, int c, int spatial, int mode, uint64_t seed,
                    uint64_t step, const uint64_t* step_dev, void* stream);

/* ---- weight rate (network.py:4777-4778, 301-305): one quantised kernel ---------
 * bits[0] = sum -log2(max(Phi((w-mu+1/32)/|s|) - Phi((w-mu-1/32)/|s|), 1e-8)), w = round(16 k)/16.
 * dk[i], dsigma[0], dmu[0] (optional) get g * d bits / d(.), overwritten or accumulated. */
int nvf_weight_rate(const float* kernel, int n, const float* sigma, const float* mu, float* bits, float* dk,
                    float* dsigma, float* dmu, const float* g_dev, float g_host, int accumulate, void* stream);

/* all (<= 8) quantised kernels in two launches: bits[l] per layer; dks[l][i] += g dbits/dk (dks or entries may be
 * NULL); dsigma[0], dmu[0] overwritten with the sum over layers (network.py:4777-4778). */
size_t nvf_weight_rate_batch_workspace(void);
int nvf_weight_rate_batch(const float* const* kernels, float* const* dks, const int* ns, int nlayers,
                          const float* sigma, const float* mu, float* bits, float* dsigma, float* dmu,
                          const float* g_dev, float g_host, void* workspace, size_t workspace_bytes, NvfStepCtx* ctx,
                          void* stream);

/* The same rate term split so that its partial pass can ride in the step head (it depends on the parameters only, not
 * on the mini-batch): nvf_step_head given a job computes the per-workgroup partial sums into job->part
 * (nvf_weight_rate_batch_workspace() bytes) and WRITES g * dbits/dk to job->dk[l] (not accumulated: the caller hands
 * those buffers to nvf_wgrad_reduce_multi_and_sums as addends); nvf_weight_rate_batch_final then adds the partials in
 * the usual fixed order (queued in ctx between nvf_finals_begin / nvf_finals_flush, like nvf_weight_rate_batch's). */
typedef struct NvfRateJob {
  const float* kernel[8];
  float* dk[8];            /* entries may be NULL */
  int32_t n[8];
  int32_t nlayers, reserved;
  const float* sigma;
  const float* mu;
  float* part;
  float g, reserved2;
} NvfRateJob;
int nvf_weight_rate_batch_final(const NvfRateJob* job, float* bits, float* dsigma, float* dmu, NvfStepCtx* ctx,
                                void* stream);


/* ---- deferred final passes ------------------------------------------------------
 * nvf_focal_loss_multi, nvf_multi_channel_sum / nvf_wgrad_reduce_multi_and_sums and nvf_weight_rate_batch end with a
 * tiny launch that adds per-workgroup partial sums in a fixed order.  Between nvf_finals_begin(ctx) and
 * nvf_finals_flush(ctx, stream) the final passes of calls given that ctx (at most one of each kind; further ones are
 * launched as usual) are queued in the context and flush runs them in ONE launch on `stream`: their outputs (the loss
 * terms, the bias gradients, the weight-rate bits and d/dsigma, d/dmu) exist only after the flush.  Same device code:
 * identical results.  Used by the training step, where nothing reads those outputs before the optimiser
 * (NVFPCC.py:161-223).  nvf_finals_begin returns NVF_EINVAL when a queue is already open on ctx. */
int nvf_finals_begin(NvfStepCtx* ctx);
int nvf_finals_flush(NvfStepCtx* ctx, void* stream);
void nvf_finals_cancel(NvfStepCtx* ctx);   /* drop the queue without launching (error paths) */

/* workspace (bytes) for the two-stage reductions of nvf_focal_loss / nvf_metrics */
size_t nvf_reduce_workspace(void);

/* ---- losses (utils/loss.py:61-72, 94-111) fused with their gradient ------------
 * loss[0] (+)= sum -a_t (1-p_t)^2 w ln(p_t),  p_t = max(p or 1-p, 1e-9), a_t = alpha or 1-alpha,
 * w = 1 (dist NULL: get_focal_dense) or dist + gt*beta (get_surf_focal_dense).
 * dp (optional) receives g * d loss / d p with g = g_host * (g_dev ? *g_dev : 1); with chain_sigmoid
 * it is further multiplied by p (1 - p), i.e. it is the gradient w.r.t. the head's logit. */
int nvf_focal_loss(const float* p, const float* gt, const float* dist, float alpha, float beta, float* loss,
                   float* dp, const float* g_dev, float g_host, void* workspace, size_t workspace_bytes, int64_t n,
                   int accumulate, int chain_sigmoid, void* stream);

/* up to 3 focal terms (the objective's main output + two heads, NVFPCC.py:166-184) in one launch pair;
 * loss[t] overwritten; dps[t] (optional) = d loss_t / d p_t, times p(1-p) with chain_sigmoid. */
int nvf_focal_loss_multi(const float* const* ps, const float* const* gts, const float* const* dists,
                         float* const* dps, const float* alphas, const float* betas, const int64_t* ns, int nterm,
                         float* loss, int chain_sigmoid, void* workspace, size_t workspace_bytes, NvfStepCtx* ctx,
                         void* stream);

/* metrics (utils/loss.py:74-84, 113-121): out[0..5] (+)= tp, ap, tn, an at thh_acc; sse, denom at thh_sse */
/* with ctx (an open nvf_finals_begin) the final pass joins the deferred ones: out exists after nvf_finals_flush, and
 * `workspace` must then be a buffer no other deferred pass uses */
int nvf_metrics(const float* p, const float* gt, const float* dist, float thh_acc, float thh_sse, float* out,
                void* workspace, size_t workspace_bytes, int64_t n, int accumulate, NvfStepCtx* ctx, void* stream);

/* the same six sums for up to three (prediction, ground truth[, dist]) pairs in one launch: out[6 t + k], overwritten
 * (the main output and the two coarse heads of the TRAIN / TEST log lines, NVFPCC.py:174-179, 214-221); dists or any
 * dists[t] may be NULL (sse = 0).  workspace: nvf_metrics_workspace() bytes (also enough for nvf_metrics). */
size_t nvf_metrics_workspace(void);
int nvf_metrics3(const float* const* ps, const float* const* gts, const float* const* dists, const int64_t* ns,
                 int nterm, float thh_acc, float thh_sse, float* out, void* workspace, size_t workspace_bytes,
                 NvfStepCtx* ctx, void* stream);

/* dlogit = dp * p * (1 - p)   (sigmoid backward of network.py:4761,4764,4768) */
int nvf_sigmoid_bwd(const float* dp, const float* p, float* dlogit, int64_t n, void* stream);

/* out = dy where y > 0 else 0   (ReLU backward of network.py:4760-4766) */
int nvf_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream);

/* get_se (utils/loss.py:123-128): out[b,0,:] = ((p > thh) * dist)^2, out[b,1,:] = p; out is [B,2,spatial] */
int nvf_squared_error_map(const float* p, const float* dist, float thh, float* out, int batch, int spatial,
                          void* stream);

/* 2x2x2 max pooling, stride 2 (MultiscaleProcessor, NVFPCC.py:76-88) */
int nvf_maxpool2(const float* x, float* y, int batch_channels, int d, int h, int w, void* stream);

/* ---- optimiser (torch.optim.Adam defaults, NVFPCC.py:116,124) -------------------
 * one fused update over a flat buffer; step counts from 1. */
int nvf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, void* stream);

/* The tail of a training step with every per-step scalar in device memory (so it can be a node of a replayed HIP
 * graph, behind the data-parallel all-reduce).  One launch does
 *   - nvf_adam_step over n floats with coef_dev[0] = lr / (1 - beta1^t), coef_dev[1] = sqrt(1 - beta2^t) (coef_dev
 *     NULL: the host values coef0_host, coef1_host) -- nvf_adam_coefficients() gives exactly the two floats
 *     nvf_adam_step would use, so the updates are identical.  An element whose gradient is NaN / +-inf keeps its
 *     parameter and moments and is counted in acc[6] (the reference checks before opt.step(), NVFPCC.py:199-212);
 *   - when acc is non-NULL, the epoch's running sums behind the TRAIN log line (NVFPCC.py:190-221, 256-281, without
 *     the per-step .item() syncs): acc[0..2] += loss_terms[0..2]; acc[3] += lbits[0] * (inv_npts_dev ? *inv_npts_dev :
 *     inv_npts_host) (b_latent, :161); acc[4] += sum(nbits[0..nnb)) * nbits_scale (b_net, :162); acc[5] += number of
 *     non-finite terms among those; acc[7] += 1; and, when counts (the 18 floats of nvf_metrics3) is non-NULL, the
 *     per-step ratios the reference averages: acc[8 + 2t] += tp_t / ap_t, acc[9 + 2t] += tn_t / an_t for t = main
 *     output, head 0, head 1 (get_acc_dense, loss.py:74-84), acc[14] += sse, acc[15] += denom (get_sse1);
 *   - when sched_rows is non-NULL, the hand-over to the next step: row sched_cursor[0] of the caller's schedule
 *     (sched_words int64 words per row: whatever the step's kernels read from sched_buf -- block ids, noise step, rate
 *     coefficient, Adam coefficients) is copied over sched_buf and the cursor advances, so a replayed graph needs no
 *     host-to-device copy per step.
 * done: one uint32 of device memory, zero before the first call (the last workgroup to arrive does the scalars and
 * resets it); required when acc or sched_rows is given. */
typedef struct NvfStepTail {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  const float* coef_dev;
  float coef0_host, coef1_host, beta1, beta2, eps;
  int32_t nnb;
  const float* loss_terms;
  const float* lbits;
  const float* nbits;
  const float* inv_npts_dev;
  float inv_npts_host, nbits_scale;
  const float* counts;
  float* acc;
  int64_t* sched_buf;
  const int64_t* sched_rows;
  uint64_t* sched_cursor;
  uint32_t* done;
  int32_t sched_words, reserved;
} NvfStepTail;
int nvf_step_tail(const NvfStepTail* args, void* stream);

/* nvf_finals_flush and the tail of the training step in the same launch (single-GPU steps: nothing sits between the
 * gradients and the optimiser): every gradient element a queued final pass writes inside [tail->g, tail->g + tail->n)
 * -- bias sums, d/dsigma and d/dmu of the weight rate, the stem's GDN parameter gradients -- gets tail's Adam update on
 * the spot; the elements of the nranges half-open index ranges ranges[2 r], ranges[2 r + 1] (gradients that earlier
 * launches wrote directly, e.g. the latent tail's) get it from a workgroup of their own; the workgroup that ran the
 * loss / rate / metrics passes adds the epoch statistics; the last workgroup to arrive hands the step buffer over to
 * the next schedule row.  Together with nvf_wgrad_reduce_multi_and_sums_fused (the weight gradients) this covers what
 * nvf_step_tail does; the caller makes the ranges the exact complement. */
int nvf_finals_flush_tail(NvfStepCtx* ctx, const NvfStepTail* tail, const int64_t* ranges, int nranges, void* stream);
/* nvf_wgrad_reduce_multi_and_sums_fused (slab reduction with addends and the fused optimiser; no channel sums) and
 * nvf_finals_flush_tail in ONE launch -- for steps whose queued final passes read nothing that reduction writes.
 * Only the final passes' workgroups wait for one another before the schedule hand-over, so the reduction must read
 * nothing from the step buffer: adam->coef_dev may not point into tail->sched_buf (NVF_EINVAL) -- use the copy that
 * nvf_wgrad_trunk5_heads_sums_partial(coef_src, coef_live) staged earlier in the step, or host coefficients. */
int nvf_wgrad_reduce_finals_tail(const float* const* slabs, float* const* dws, const int* nslabs, const int* jtotals,
                                 int n, const float* const* addends, const NvfAdamFuse* adam, NvfStepCtx* ctx,
                                 const NvfStepTail* tail, const int64_t* ranges, int nranges, void* stream);
/* The same pairing without the optimiser, for steps whose gradients are not final yet (data parallelism: the all-reduce
 * and nvf_step_tail follow): nvf_wgrad_reduce_multi with addends + nvf_finals_flush in ONE launch.  Same condition: no
 * queued final pass may read what the reduction writes. */
int nvf_wgrad_reduce_finals(const float* const* slabs, float* const* dws, const int* nslabs, const int* jtotals, int n,
                            const float* const* addends, NvfStepCtx* ctx, void* stream);
int nvf_adam_coefficients(float lr, float beta1, float beta2, int step, float* coef_host);
/* ... of n consecutive steps step0, step0 + 1, ...: coef_host[2 k], coef_host[2 k + 1] (the rows of a schedule) */
int nvf_adam_coefficients_n(float lr, float beta1, float beta2, int step0, int n, float* coef_host);

/* rows: dst[r,:] = src[idx[r],:]  (emb[indices], NVFPCC.py:158) and its transpose
 * dst[idx[r],:] += src[r,:] (indices unique within a call) */
int nvf_gather_rows(const float* src, const int64_t* idx, float* dst, int rows, int width, void* stream);
int nvf_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int rows, int width, void* stream);

/* up to 6 gathers sharing one index vector: dsts[t][r,:] = srcs[t][idx[r],:] */
int nvf_gather_rows_multi(const float* const* srcs, float* const* dsts, const int* widths, int n,
                          const int64_t* idx, int rows, void* stream);

/* The head of a training step (NVFPCC.py:149-160) in ONE launch: nvf_prepare_weights, nvf_pack_mfma_all and
 * nvf_gather_rows_multi.  Pack job j packs layout pack_bwd[j] (0: w_fwd, 1: w_bwd) of layer-table row pack_layers[j]
 * (kinds / c0s / c1s as in nvf_pack_mfma_all; npack may be 0); a packed element recomputes its effective weight from
 * the raw kernel, so nothing in the launch waits for anything else.  Same results as the three calls, bit for bit. */
int nvf_step_head(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                  float* const* pack_dsts, const int* pack_kinds, const int* pack_c0s, const int* pack_c1s,
                  const int* pack_layers, const int* pack_bwd, int npack, const float* const* srcs,
                  float* const* dsts, const int* widths, int n, const int64_t* idx, int rows,
                  const NvfRateJob* rate /* may be NULL: see nvf_weight_rate_batch_final */, void* stream);

/* nvf_step_head AND nvf_stem_latent_fwd (latent generator + quantiser + up0 / IGDN / conv0 of the mini-batch) in ONE
 * launch, (c0, c1) = (8, 16) or (16, 32), ch <= 8, rows <= 32: the stem's workgroups derive their effective weights
 * from the raw parameters of layer-table rows lat_row / up0_row / conv0_row themselves (network.py:611-620, 735-740:
 * the arithmetic of nvf_prepare_weights, element by element) and fetch their latents as emb[idx[b]], so nothing in the
 * launch waits for anything else.  Same results as the two calls, bit for bit.  block ids of the latent noise = idx. */
typedef struct NvfStemHead {
  const float* emb;            /* latent table [N, ch, 2, 2, 2] */
  const float* lat_beta_hat;   /* GDN of the latent generator */
  const float* lat_gamma_hat;
  const float* sigma;          /* entropy coder [ch] */
  const float* mu;
  const float* beta_hat;       /* IGDN of the decoder */
  const float* gamma_hat;
  float* h;                    /* outputs of nvf_stem_latent_fwd */
  float* lat;
  float* x_rounded;
  float* bits;
  float* a0;
  float* h0;
  float* y1;
  int32_t lat_row, up0_row, conv0_row;
  int32_t mode, ch, c0, c1, reserved;
} NvfStemHead;
int nvf_step_head_stem(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step,
                       const uint64_t* step_dev, float* const* pack_dsts, const int* pack_kinds, const int* pack_c0s,
                       const int* pack_c1s, const int* pack_layers, const int* pack_bwd, int npack,
                       const float* const* srcs, float* const* dsts, const int* widths, int n, const int64_t* idx,
                       int rows, const NvfRateJob* rate /* may be NULL */, const NvfStemHead* stem, void* stream);

/* U[0,1) floats, Philox4x32-10 keyed by (seed, stream_id), counter = element index */
int nvf_uniform(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);

/* ---- occupancy thresholding + compaction (NVFPCC.py:520,532-535,631-634) --------
 * For each block b, voxel (z,y,x) with p > thh: coords (origin[b] + (z,y,x)) appended in
 * raster order (b, z, y, x) -- the order torch.nonzero gives.  counts[b] = points of
 * block b.  Two-phase: call with coords == NULL to get counts, exclusive-scan them on
 * the host or device into offsets, then call again with coords/offsets. */
int nvf_threshold_count(const float* p, float thh, int32_t* counts, int batch, int voxels, void* stream);
int nvf_threshold_compact(const float* p, float thh, const int32_t* offsets, const int32_t* origins,
                          int32_t* coords, int batch, int dim, void* stream);

/* ---- pre-processing: exact squared distance of every voxel of every 32^3 leaf block to the nearest input
 * point (util_get_grids.py:19-46; gt_grid = (d2 == 0), dist = sqrt(d2)).  pts int32 [P,3] sorted by block,
 * blk_off [N+1] their ranges, origins int32 [N,3], (nb_off, nb_idx) a CSR list of candidate blocks per block
 * (the block itself first, then occupied blocks within +-2 block steps).  d2out int32 [N,32,32,32], axes (x,y,z). */
int nvf_nearest_dist2(const int32_t* pts, const int32_t* blk_off, const int32_t* origins, const int32_t* nb_off,
                      const int32_t* nb_idx, int32_t* d2out, int nblocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NVF_HIP_H */
