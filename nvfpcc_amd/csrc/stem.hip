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

namespace {
constexpr int MAXCH = 8;

__device__ __forceinline__ float st_beta(float bh) {
  float m = fmaxf(bh, NVF_BETA_BOUND);
  return m * m - NVF_PEDESTAL;
}
__device__ __forceinline__ float st_gamma(float gh) {
  float m = fmaxf(gh, NVF_GAMMA_BOUND);
  return m * m - NVF_PEDESTAL;
}
// dst[e] = src[index(e)] for e < n with U loads of a thread in flight before the first store (a plain copy loop is a
// chain of load -> wait -> store round trips: 16 of them for the wide stem's 64 KB of weights)
template <int NT, int U, class Index>
__device__ __forceinline__ void stem_copy(float* dst, const float* __restrict__ src, int n, int tid, Index index) {
#pragma unroll 1
  for (int e0 = tid; e0 < n; e0 += NT * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      v[u] = e < n ? src[index(e)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      if (e < n) dst[e] = v[u];
    }
  }
}
}  // namespace

// conv0 for one output parity class (EZ,EY,EX): lane = cell, COG output channels in registers, taps unrolled so
// the LDS reads of a whole input channel are in flight together.  Weights come from the LDS copy s_w[ci*125+tap][COG].
template <int C0, int C1, int EZ, int EY, int EX, int COG>
__device__ __forceinline__ void stem_conv0_class(const float* s_h, const float* s_w, const float* __restrict__ b1,
                                                 float* __restrict__ y1, int b, int co0, int v) {
  const int mz = (v >> 4) + 1, my = ((v >> 2) & 3) + 1, mx = (v & 3) + 1;   // cells 1..4 (pad 2)
  float acc[COG];
#pragma unroll
  for (int co = 0; co < COG; ++co) acc[co] = 0.f;
#pragma unroll 2
  for (int ci = 0; ci < C0; ++ci) {
#pragma unroll
    for (int jz = 0; jz < 3 - EZ; ++jz)
#pragma unroll
      for (int jy = 0; jy < 3 - EY; ++jy)
#pragma unroll
        for (int jx = 0; jx < 3 - EX; ++jx) {
          const float hv = s_h[ci * 216 + ((mz - jz + 1) * 6 + (my - jy + 1)) * 6 + (mx - jx + 1)];
          const float* wr = s_w + (ci * 125 + ((EZ + 2 * jz) * 5 + (EY + 2 * jy)) * 5 + EX + 2 * jx) * COG;
#pragma unroll
          for (int co = 0; co < COG; ++co) acc[co] = fmaf(hv, wr[co], acc[co]);
        }
  }
  const int qz = 2 * mz + EZ - 2, qy = 2 * my + EY - 2, qx = 2 * mx + EX - 2;   // in [0, 8)
#pragma unroll
  for (int co = 0; co < COG; ++co)
    y1[((size_t)b * C1 + co0 + co) * 512 + (qz * 8 + qy) * 8 + qx] = fmaxf(acc[co] + b1[co0 + co], 0.f);
}

// grid = (batch, C1 / COG): every workgroup recomputes the (tiny) up0 + IGDN of its block and produces COG of
// conv0's 16 output channels, so a batch of 16 blocks runs on 64 CUs instead of 16.  All weights are copied to
// LDS with coalesced vector loads first: scalar loads in the tap loops were a chain of cache misses.
// The latent generator + quantiser (nvf_latent_fwd) for the launch that also runs the stem: the stem's workgroups
// compute the 8 ch rounded latents of their own block themselves (latent_x_rounded: the same arithmetic), so they do
// not wait for the one workgroup (blockIdx = (0, C1 / COG)) that produces h, lat, x_rounded and the rate for the
// whole batch.
struct StemLatent {
  const float* e;
  const float* w;          // latent generator's w_fwd [ci][co], bias
  const float* bw;
  const float* beta_hat;   // its GDN
  const float* gamma_hat;
  const int64_t* block_ids;
  const float* sigma;
  const float* mu;
  float* h_out;
  float* lat_out;
  float* x_rounded;
  float* bits;
  const uint64_t* step_dev;
  uint64_t seed, step;
  int32_t mode, batch;
};

template <int C0, int C1, int COG, bool LATENT>
__global__ __launch_bounds__(C0 * 64) void stem_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w0,
                                                           const float* __restrict__ b0,
                                                           const float* __restrict__ beta_hat,
                                                           const float* __restrict__ gamma_hat,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           float* __restrict__ a0, float* __restrict__ h0,
                                                           float* __restrict__ y1, int ch, StemLatent L) {
  constexpr int NT = C0 * 64, NCG = C0 / 8;      // NCG groups of eight waves (one per parity class) in the conv0 phase,
  constexpr int PARTS = C1 / (COG * NCG);        //  each with its own COG output channels
  __shared__ float s_x[MAXCH * 8];
  __shared__ float s_a[C0 * 64];
  __shared__ __attribute__((aligned(16))) float s_w0[MAXCH * 125 * C0];
  if (LATENT && blockIdx.y == PARTS) {
    if (blockIdx.x == 0)
      latent_fwd_body(L.e, L.w, L.bw, L.beta_hat, L.gamma_hat, L.block_ids, L.sigma, L.mu, L.h_out, L.lat_out,
                      L.x_rounded, L.bits, L.batch, ch, 8, L.mode, L.seed, L.step, L.step_dev, s_a, s_w0,
                      MAXCH * 125 * C0);
    return;
  }
  __shared__ float s_h[C0 * 216];     // h0 with a one-voxel zero halo: [c][6][6][6], index i + 1
  __shared__ __attribute__((aligned(16))) float s_w1[NCG * C0 * 125 * COG];
  __shared__ float s_beta[C0], s_gamma[C0 * C0];
  __shared__ float s_lat[2 * MAXCH * MAXCH + 2 * MAXCH];   // latent generator: w [ci][co], gamma_hat, bias, beta_hat
  const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  float ev[MAXCH];                                           // this thread's latent element: its ch inputs, fetched together
  if (LATENT) {
    // (parameters through LDS and the inputs up front: as loads inside the fmaf chains they were ch^2 dependent round trips)
    if (tid < ch * ch) { s_lat[tid] = L.w[tid]; s_lat[MAXCH * MAXCH + tid] = L.gamma_hat[tid]; }
    if (tid >= 64 && tid < 64 + ch) {
      s_lat[2 * MAXCH * MAXCH + tid - 64] = L.bw[tid - 64];
      s_lat[2 * MAXCH * MAXCH + MAXCH + tid - 64] = L.beta_hat[tid - 64];
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) ev[i] = (tid < ch * 8 && i < ch) ? L.e[((size_t)b * ch + i) * 8 + (tid & 7)] : 0.f;
  } else if (tid < ch * 8) {
    s_x[tid] = x0[(size_t)b * ch * 8 + tid];
  }
  if (tid >= 64 && tid < 64 + C0) s_beta[tid - 64] = st_beta(beta_hat[tid - 64]);
  if (tid >= 128 && tid < 128 + C0 * C0) s_gamma[tid - 128] = st_gamma(gamma_hat[tid - 128]);
  for (int e = tid; e < C0 * 216; e += NT) s_h[e] = 0.f;
  stem_copy<NT, 16>(s_w0, w0, ch * 125 * C0, tid, [](int e) { return e; });
  stem_copy<NT, 16>(s_w1, w1, NCG * C0 * 125 * COG, tid, [&](int e) {
    const int cg = e / (C0 * 125 * COG), r = e - cg * (C0 * 125 * COG);
    return (r / COG) * C1 + (part * NCG + cg) * COG + r % COG;
  });
  if (LATENT) {
    __syncthreads();
    if (tid < ch * 8)
      s_x[tid] = latent_x_rounded_from(ev, s_lat, s_lat + 2 * MAXCH * MAXCH, s_lat + 2 * MAXCH * MAXCH + MAXCH,
                                       s_lat + MAXCH * MAXCH, tid >> 3, ch);
  }
  __syncthreads();
  {  // up0: a0[co, o] = b0 + sum_ci sum_{k : o + 2 - k = 2 i} x0[ci, i] w0[ci][k][co]
    // per axis the valid taps are k = o (input i = 1) and k = o + 2 (i = 0, if o <= 2): ascending k, the order of
    // the per-layer kernel, without walking the 125 taps.  Lanes run over the output CHANNEL here (the weight row of a
    // tap is C0 consecutive words): with lanes over positions every lane read another tap's row at a stride of C0
    // words -- 2 (C0 = 16) or 4 banks for the whole wave, 11 us of this launch for the wide decoder.
    const int co = tid % C0, vo = tid / C0, oz = vo >> 4, oy = (vo >> 2) & 3, ox = vo & 3;
    float acc = 0.f;
    for (int ci = 0; ci < ch; ++ci)
#pragma unroll
      for (int az = 0; az < 2; ++az) {
        const int kz = oz + 2 * az;
        if (kz > 4) continue;
#pragma unroll
        for (int ay = 0; ay < 2; ++ay) {
          const int ky = oy + 2 * ay;
          if (ky > 4) continue;
#pragma unroll
          for (int ax = 0; ax < 2; ++ax) {
            const int kx = ox + 2 * ax;
            if (kx > 4) continue;
            acc = fmaf(s_x[ci * 8 + (1 - az) * 4 + (1 - ay) * 2 + (1 - ax)],
                       s_w0[(ci * 125 + (kz * 5 + ky) * 5 + kx) * C0 + co], acc);
          }
        }
      }
    const float val = acc + b0[co];
    s_a[co * 64 + vo] = val;
    if (part == 0) a0[(size_t)b * C0 * 64 + co * 64 + vo] = val;
  }
  const int c = tid >> 6, v = tid & 63, oz = v >> 4, oy = (v >> 2) & 3, ox = v & 3;
  __syncthreads();
  {  // IGDN: h0 = a0 * sqrt(beta_c + sum_j gamma_cj a0_j^2)
    float nrm = s_beta[c];
#pragma unroll
    for (int j = 0; j < C0; ++j) {
      const float xj = s_a[j * 64 + v];
      nrm = fmaf(s_gamma[c * C0 + j], xj * xj, nrm);
    }
    const float hv = s_a[tid] * sqrtf(nrm);
    if (part == 0) h0[(size_t)b * C0 * 64 + tid] = hv;
    s_h[c * 216 + ((oz + 1) * 6 + (oy + 1)) * 6 + ox + 1] = hv;
  }
  __syncthreads();
  // conv0: one wave per output parity class (and channel group), one lane per cell
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wv >> 3, co0 = (part * NCG + cg) * COG;
  const float* sw = s_w1 + cg * (C0 * 125 * COG);
  switch (wv & 7) {
    case 0: stem_conv0_class<C0, C1, 0, 0, 0, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 1: stem_conv0_class<C0, C1, 0, 0, 1, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 2: stem_conv0_class<C0, C1, 0, 1, 0, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 3: stem_conv0_class<C0, C1, 0, 1, 1, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 4: stem_conv0_class<C0, C1, 1, 0, 0, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 5: stem_conv0_class<C0, C1, 1, 0, 1, COG>(s_h, sw, b1, y1, b, co0, v); break;
    case 6: stem_conv0_class<C0, C1, 1, 1, 0, COG>(s_h, sw, b1, y1, b, co0, v); break;
    default: stem_conv0_class<C0, C1, 1, 1, 1, COG>(s_h, sw, b1, y1, b, co0, v); break;
  }
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
// da0 / dx0 exist after that launch.  flags: batch + 1 uint32 words, zero before the first use; the launch leaves them zero.
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
  j.coop.dh_done = flags; j.coop.stem_done = flags + batch; j.coop.bias_slab = bias;
  j.batch = batch; j.ch = ch; j.nwg = batch;
  ctx->stem = j;
  ctx->stem_pending = 1;
  *dw_slabs = slab_w; *nslabs = batch; *bias_slabs = bias;
  StemGdnFinal f{slab_gdn, beta_hat, gamma_hat, dbeta_hat, dgamma_hat, batch, c0};
  return nvf_finals_run_stem_gdn(ctx, f, stream);
}

extern "C" int nvf_stem_bwd_pending(const NvfStepCtx* ctx) { return nvf_ctx_ok(ctx) && ctx->stem_pending ? 1 : 0; }
