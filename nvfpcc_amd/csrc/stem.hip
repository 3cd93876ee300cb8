// Fused "stem" of the NVF decoder for chanstr (c0, c1) = (8, 16) -- the narrow decoder -- and (16, 32) -- the wide
// one (latent channels ch <= 8); kernels are templates on <C0, C1>, C0 * 64 threads per workgroup:
//
//   forward :  x0 [ch,2^3] --up0 (convT k5 s2 p2)--> a0 [C0,4^3] --IGDN--> h0 --conv0 (convT k5 s2 p2)+ReLU--> y1 [C1,8^3]
//   backward:  g1 = dL/d(conv0 pre-activation) --> dh0 --> IGDN backward (da0, d beta, d gamma) --> dx0,
//              plus up0's weight gradient
//
// These layers are < 1 % of the step's FLOPs but, as separate launches over a
// batch of 16 blocks, each is a latency chain on a handful of CUs.  Here the reference's operator sequence
// (utils/network.py:4759-4760, gdn_3d.py:137-159) runs with every intermediate in LDS: the forward is one launch
// of (block, conv0 channel group) workgroups, the backward two (conv0's backward-data over (block, channel pair)
// workgroups, then one workgroup per block for IGDN / up0).  Weights are copied to LDS with coalesced vector
// loads before use.  Accumulation orders equal those of the per-layer kernels (conv_direct.hip), so the forward
// is bit-identical to the unfused path.
#include "nvf_common.h"
#include "step_ctx.h"
#include "latent_tail.h"
#include "stem_bwd.h"
#include "stem_fwd.h"

namespace {
constexpr int MAXCH = 8;

}  // namespace

template <int C0, int C1, int COG, bool LATENT>
__global__ __launch_bounds__(C0 * 64) void stem_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w0,
                                                           const float* __restrict__ b0,
                                                           const float* __restrict__ beta_hat,
                                                           const float* __restrict__ gamma_hat,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           float* __restrict__ a0, float* __restrict__ h0,
                                                           float* __restrict__ y1, int ch, StemLatent L) {
  __shared__ __attribute__((aligned(16))) float lds[StemFwdLds<C0, C1, COG>::FLOATS];
  const StemPreparedW wp{w0, b0, w1, b1, L.w, L.bw, L.e, ch};
  stem_fwd_body<C0, C1, COG, LATENT, false>(x0, wp, beta_hat, gamma_hat, a0, h0, y1, ch, L, blockIdx.x, blockIdx.y, lds);
}

#ifndef NVF_STEM_COG
#define NVF_STEM_COG 4      // narrow decoder: conv0 output channels per workgroup (tuning: 2, 1)
#endif

template <int C0, int C1, bool LATENT>
static int launch_stem_fwd(const float* x0, const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                           const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0, float* h0,
                           float* y1, int batch, int ch, const StemLatent& L, void* stream) {
  constexpr int COG = C0 == 16 ? 2 : NVF_STEM_COG, PARTS = C1 / (COG * (C0 / 8));   // (wide: 51 us with four channels per group, 40 with two, 48 with one)
  stem_fwd_kernel<C0, C1, COG, LATENT><<<dim3(batch, PARTS + (LATENT ? 1 : 0)), C0 * 64, 0, nvf_stream(stream)>>>(
      x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b, a0, h0, y1, ch, L);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
static bool stem_shape(int c0, int c1) { return (c0 == 8 && c1 == 16) || (c0 == 16 && c1 == 32); }

extern "C" int nvf_stem_fwd(const float* x0, const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                            const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0,
                            float* h0, float* y1, int batch, int ch, int c0, int c1, void* stream) {
  if (!x0 || !up0_w_fwd || !up0_b || !beta_hat || !gamma_hat || !conv0_w_fwd || !conv0_b || !a0 || !h0 || !y1)
    return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || !stem_shape(c0, c1)) return NVF_EINVAL;
  if (c0 == 8)
    return launch_stem_fwd<8, 16, false>(x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b, a0, h0, y1,
                                         batch, ch, StemLatent{}, stream);
  return launch_stem_fwd<16, 32, false>(x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b, a0, h0, y1,
                                        batch, ch, StemLatent{}, stream);
}

// nvf_latent_fwd (e -> h, lat, x_rounded, bits) and nvf_stem_fwd (x_rounded -> a0, h0, y1) in ONE launch; same results
// as the two calls, bit for bit.  Latent tensors are [batch, ch <= 8, 2^3].
extern "C" int nvf_stem_latent_fwd(const float* e, const float* lat_w_fwd, const float* lat_bias,
                                   const float* lat_beta_hat, const float* lat_gamma_hat, const int64_t* block_ids,
                                   const float* sigma, const float* mu, float* h, float* lat, float* x_rounded,
                                   float* bits, int mode, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                                   const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                                   const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0,
                                   float* h0, float* y1, int batch, int ch, int c0, int c1, void* stream) {
  if (!e || !lat_w_fwd || !lat_bias || !lat_beta_hat || !lat_gamma_hat || !sigma || !mu || !h || !lat || !x_rounded ||
      !bits || !up0_w_fwd || !up0_b || !beta_hat || !gamma_hat || !conv0_w_fwd || !conv0_b || !a0 || !h0 || !y1)
    return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || !stem_shape(c0, c1) || (mode != 0 && mode != 1)) return NVF_EINVAL;
  StemLatent L{e, lat_w_fwd, lat_bias, lat_beta_hat, lat_gamma_hat, block_ids, sigma, mu, h, lat, x_rounded, bits,
               step_dev, seed, step, mode, batch};
  if (c0 == 8)
    return launch_stem_fwd<8, 16, true>(nullptr, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b, a0, h0,
                                        y1, batch, ch, L, stream);
  return launch_stem_fwd<16, 32, true>(nullptr, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd, conv0_b, a0, h0, y1,
                                       batch, ch, L, stream);
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
static const int kStemMaxSlabs = 256;
constexpr int stem_ncol(int c0) { return c0 + c0 * c0; }          // IGDN parameter partials
constexpr int kStemWMax = MAXCH * 125;                            // up0 weight-gradient slab: kStemWMax * C0 floats

// conv0 backward-data, split over (block, output-channel pair): part[b][cp][ci][i] = sum over the pair's two co and
// all 125 taps of g1[co, 2 i - 2 + k] w1[ci][co][k].  C0 / 2 waves: wave = input-channel pair, lane = position i.
// The pair's gradients (zero-padded) and weights sit in LDS; each lane runs the fmaf chain (cc, kz, ky, kx ascending),
// the partials are added in ascending pair order by stem_bwd_kernel.
// WG0: the workgroup also leaves its two output channels' share of conv0's weight gradient for this block,
//   slab0[b][ci][co][k] = sum_i h0[b, ci, i] g1[b, co, 2 i - 2 + k]   (i ascending; the caller's slab reduction adds the blocks),
// from the gradient tile it already holds -- a separate launch over 4^3 inputs was 18 us of latency for 0.13 GFLOP.
template <int C0, int C1, bool WG0>
__global__ __launch_bounds__(C0 * 32) void stem_bwd_dh_kernel(const float* __restrict__ g1,
                                                              const float* __restrict__ w1b /* [co][125][ci] */,
                                                              float* __restrict__ part, const float* __restrict__ h0,
                                                              float* __restrict__ slab0) {
  __shared__ __attribute__((aligned(16))) float lds[StemDhLds<C0, C1>::FLOATS];
  stem_bwd_dh_body<C0, C1, WG0, false>(g1, w1b, part, h0, slab0, blockIdx.x, blockIdx.y, lds, StemCoop{});
}

template <int C0, int C1>
__global__ __launch_bounds__(C0 * 64) void stem_bwd_kernel(const float* __restrict__ part, const float* __restrict__ x0,
                                                           const float* __restrict__ a0,
                                                           const float* __restrict__ w0b /* [co][125][ch] */,
                                                           const float* __restrict__ beta_hat,
                                                           const float* __restrict__ gamma_hat, float* __restrict__ da0,
                                                           float* __restrict__ dx0, float* __restrict__ slab_gdn,
                                                           float* __restrict__ slab_w, int batch, int ch, int want_w) {
  __shared__ __attribute__((aligned(16))) float lds[StemBwdLds<C0>::FLOATS];
  stem_bwd_body<C0, C1, C0 * 64, false>(part, x0, a0, w0b, beta_hat, gamma_hat, da0, dx0, slab_gdn, slab_w, batch, ch,
                                        want_w, blockIdx.x, gridDim.x, lds, StemCoop{});
}

// Both finals of the stem backward in one launch: the last workgroup turns the IGDN slabs into parameter gradients
// (fixed-order sum, re-parametrisation chain rule, LowerBound rule), the others add up0's weight-gradient slabs.
__global__ __launch_bounds__(256) void stem_finals(StemGdnFinal f, const float* __restrict__ slab_w,
                                                   float* __restrict__ dw, int nslab, int jtotal) {
  if (blockIdx.x == gridDim.x - 1) {
    stem_gdn_final_body(f, threadIdx.x, 256);
    return;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= jtotal) return;
  float s = 0.f;
  for (int g = 0; g < nslab; ++g) s += slab_w[(size_t)g * jtotal + j];
  dw[j] = s;
}

// workspace of nvf_stem_bwd / nvf_stem_bwd_partial: IGDN slabs, up0 weight-gradient slabs, conv0 backward-data partials
static size_t stem_ws_floats(int batch, int ch, int c0, int c1) {
  const size_t nb = batch > 0 ? batch : 0;
  return (size_t)kStemMaxSlabs * (stem_ncol(c0) + (size_t)ch * c0 * 125) + nb * (c1 / 2) * c0 * 64 +
         nb * c0 * c1 * 125 +                           // + conv0's weight-gradient slabs (one per block)
         nb * c0;                                       // + up0's bias-gradient slabs (nvf_stem_bwd_queue)
}
extern "C" size_t nvf_stem_bwd_workspace(int batch, int ch) { return stem_ws_floats(batch, ch, 8, 16) * sizeof(float); }
extern "C" size_t nvf_stem_bwd_workspace_for(int batch, int ch, int c0, int c1) {
  return stem_shape(c0, c1) ? stem_ws_floats(batch, ch, c0, c1) * sizeof(float) : 0;
}

template <int C0, int C1>
static int launch_stem_bwd(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                           const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0, float* dx0,
                           void* workspace, int batch, int ch, int want_w, float** slab_gdn_out, float** slab_w_out,
                           int* nslab_out, hipStream_t s, const float* h0 = nullptr, float** slab0_out = nullptr) {
  const int nslab = batch < kStemMaxSlabs ? batch : kStemMaxSlabs;
  float* slab_gdn = (float*)workspace;
  float* slab_w = slab_gdn + (size_t)kStemMaxSlabs * stem_ncol(C0);
  float* part = slab_w + (size_t)kStemMaxSlabs * ch * C0 * 125;
  float* slab0 = part + (size_t)batch * (C1 / 2) * C0 * 64;
  if (h0 && slab0_out) {
    stem_bwd_dh_kernel<C0, C1, true><<<dim3(batch, C1 / 2), C0 * 32, 0, s>>>(g1, conv0_w_bwd, part, h0, slab0);
    *slab0_out = slab0;
  } else
    stem_bwd_dh_kernel<C0, C1, false><<<dim3(batch, C1 / 2), C0 * 32, 0, s>>>(g1, conv0_w_bwd, part, nullptr, nullptr);
  stem_bwd_kernel<C0, C1><<<nslab, C0 * 64, 0, s>>>(part, x0, a0, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, slab_gdn,
                                                    slab_w, batch, ch, want_w);
  *slab_gdn_out = slab_gdn; *slab_w_out = slab_w; *nslab_out = nslab;
  return NVF_OK;
}

extern "C" int nvf_stem_bwd(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                            const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0,
                            float* dx0, float* dbeta_hat, float* dgamma_hat, float* dw_up0, void* workspace,
                            size_t workspace_bytes, int batch, int ch, int c0, int c1, void* stream) {
  if (!g1 || !x0 || !a0 || !conv0_w_bwd || !up0_w_bwd || !beta_hat || !gamma_hat || !da0 || !dx0) return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || !stem_shape(c0, c1)) return NVF_EINVAL;
  const int want_w = dbeta_hat && dgamma_hat && dw_up0;
  if (!workspace || workspace_bytes < stem_ws_floats(batch, ch, c0, c1) * sizeof(float)) return NVF_EWORKSPACE;
  hipStream_t s = nvf_stream(stream);
  float *slab_gdn, *slab_w;
  int nslab;
  if (c0 == 8)
    launch_stem_bwd<8, 16>(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, workspace, batch, ch,
                           want_w, &slab_gdn, &slab_w, &nslab, s);
  else
    launch_stem_bwd<16, 32>(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, workspace, batch, ch,
                            want_w, &slab_gdn, &slab_w, &nslab, s);
  if (want_w) {
    const int jtotal = ch * c0 * 125;
    StemGdnFinal f{slab_gdn, beta_hat, gamma_hat, dbeta_hat, dgamma_hat, nslab, c0};
    stem_finals<<<(jtotal + 255) / 256 + 1, 256, 0, s>>>(f, slab_w, dw_up0, nslab, jtotal);
  }
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// nvf_stem_bwd without its final launch: up0's weight-gradient slabs are left to the caller's slab reduction
// (*dw_slabs = nslabs slabs of ch * c0 * 125 floats inside `workspace`, to be added into dw_up0 by
// nvf_wgrad_reduce_multi*), the IGDN parameter gradients go to the deferred final passes (nvf_finals_begin; launched
// at once when nothing is being deferred).  Same sums in the same order as nvf_stem_bwd.  h0 / dw_conv0_slabs (both or
// neither): also leave conv0's weight gradient as `batch` slabs of c0 * c1 * 125 floats ([ci][co][k], one per block)
// inside `workspace` for the same reduction.
extern "C" int nvf_stem_bwd_partial(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                                    const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0,
                                    float* dx0, float* dbeta_hat, float* dgamma_hat, float** dw_slabs, int* nslabs,
                                    void* workspace, size_t workspace_bytes, int batch, int ch, int c0, int c1,
                                    const float* h0, float** dw_conv0_slabs, NvfStepCtx* ctx, void* stream) {
  if (!g1 || !x0 || !a0 || !conv0_w_bwd || !up0_w_bwd || !beta_hat || !gamma_hat || !da0 || !dx0 || !dbeta_hat ||
      !dgamma_hat || !dw_slabs || !nslabs || (h0 != nullptr) != (dw_conv0_slabs != nullptr))
    return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || !stem_shape(c0, c1)) return NVF_EINVAL;
  if (!workspace || workspace_bytes < stem_ws_floats(batch, ch, c0, c1) * sizeof(float)) return NVF_EWORKSPACE;
  hipStream_t s = nvf_stream(stream);
  float *slab_gdn, *slab_w;
  int nslab;
  if (c0 == 8)
    launch_stem_bwd<8, 16>(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, workspace, batch, ch, 1,
                           &slab_gdn, &slab_w, &nslab, s, h0, dw_conv0_slabs);
  else
    launch_stem_bwd<16, 32>(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, workspace, batch, ch, 1,
                            &slab_gdn, &slab_w, &nslab, s, h0, dw_conv0_slabs);
  NVF_LAUNCH_CHECK();
  *dw_slabs = slab_w;
  *nslabs = nslab;
  StemGdnFinal f{slab_gdn, beta_hat, gamma_hat, dbeta_hat, dgamma_hat, nslab, c0};
  return nvf_finals_run_stem_gdn(ctx, f, stream);
}

// The stem's backward as the FIRST workgroups of the next five-gradient launch given this context
// (nvf_wgrad_trunk5_heads_sums_partial and its siblings with five jobs) instead of two launches of its own: stem_bwd.h.
// Needs a queued latent tail in the same context by the time of that launch (the tail consumes dx0 inside it) and an open
// finals queue (nvf_finals_begin: the IGDN parameter gradients are a deferred final pass, as in nvf_stem_bwd_partial).
// Narrow decoder only (c0 = 8, c1 = 16), batch <= kStemCoopMaxBatch.  Outputs as nvf_stem_bwd_partial, plus
// *bias_slabs = `batch` slabs of c0 channel sums of da0 (up0's bias gradient: a jtotal = c0 job of the slab reduction).
// da0 / dx0 exist after that launch.  flags: (batch + 1) * 64 uint32 words (one 256-byte line per counter), zero before the first use; the launch leaves them zero.
extern "C" int nvf_stem_bwd_queue(NvfStepCtx* ctx, const float* g1, const float* x0, const float* a0,
                                  const float* conv0_w_bwd, const float* up0_w_bwd, const float* beta_hat,
                                  const float* gamma_hat, float* da0, float* dx0, float* dbeta_hat, float* dgamma_hat,
                                  float** dw_slabs, int* nslabs, float** bias_slabs, void* workspace,
                                  size_t workspace_bytes, uint32_t* flags, int batch, int ch, int c0, int c1,
                                  void* stream) {
  if (!g1 || !x0 || !a0 || !conv0_w_bwd || !up0_w_bwd || !beta_hat || !gamma_hat || !da0 || !dx0 || !dbeta_hat ||
      !dgamma_hat || !dw_slabs || !nslabs || !bias_slabs || !flags)
    return NVF_EINVAL;
  if (!nvf_ctx_ok(ctx) || !ctx->deferring || ctx->stem_pending) return NVF_EINVAL;
  if (batch <= 0 || batch > kStemCoopMaxBatch || ch <= 0 || ch > MAXCH || c0 != 8 || c1 != 16) return NVF_EINVAL;
  if (!workspace || workspace_bytes < stem_ws_floats(batch, ch, c0, c1) * sizeof(float)) return NVF_EWORKSPACE;
  float* slab_gdn = (float*)workspace;
  float* slab_w = slab_gdn + (size_t)kStemMaxSlabs * stem_ncol(c0);
  float* part = slab_w + (size_t)kStemMaxSlabs * ch * c0 * 125;
  float* slab0 = part + (size_t)batch * (c1 / 2) * c0 * 64;
  float* bias = slab0 + (size_t)batch * c0 * c1 * 125;
  StemBwdJob j{};
  j.g1 = g1; j.w1b = conv0_w_bwd; j.x0 = x0; j.a0 = a0; j.w0b = up0_w_bwd; j.beta_hat = beta_hat; j.gamma_hat = gamma_hat;
  j.part = part; j.da0 = da0; j.dx0 = dx0; j.slab_gdn = slab_gdn; j.slab_w = slab_w;
  j.coop.dh_done = flags; j.coop.stem_done = flags + batch * kStemFlagStride; j.coop.bias_slab = bias;
  j.batch = batch; j.ch = ch; j.nwg = batch;
  ctx->stem = j;
  ctx->stem_pending = 1;
  *dw_slabs = slab_w; *nslabs = batch; *bias_slabs = bias;
  StemGdnFinal f{slab_gdn, beta_hat, gamma_hat, dbeta_hat, dgamma_hat, batch, c0};
  return nvf_finals_run_stem_gdn(ctx, f, stream);
}

extern "C" int nvf_stem_bwd_pending(const NvfStepCtx* ctx) { return nvf_ctx_ok(ctx) && ctx->stem_pending ? 1 : 0; }
