// Pieces of the latent path's backward pass shared by pointwise.hip (the stand-alone launches) and wgrad.hip (the
// "latent tail": rate gradient -> GDN backward -> 1x1x1 weight gradient run by ONE workgroup inside the launch that
// adds up the weight-gradient slabs, instead of three dependent ~5-10 us launches on [B, c <= 8, 2^3] tensors).
#pragma once
#include "nvf_common.h"
#include "finals.h"

// ---- GDN re-parametrisation (gdn_3d.py:40-60) ------------------------------------------------------------------
#define NVF_PEDESTAL 1.4551915228366852e-11f  /* 2^-36 */

__device__ __forceinline__ float gdn_beta(float bh) {
  float m = fmaxf(bh, NVF_BETA_BOUND);
  return m * m - NVF_PEDESTAL;
}
__device__ __forceinline__ float gdn_gamma(float gh) {
  float m = fmaxf(gh, NVF_GAMMA_BOUND);
  return m * m - NVF_PEDESTAL;
}

// ---- Gaussian rate helpers (network.py:145-161) ----------------------------------------------------------------
__device__ __forceinline__ float std_cdf(float z) { return 0.5f * (1.f + erff(z / 1.41421356237309515f)); }
__device__ __forceinline__ float std_pdf(float z) { return 0.3989422804014327f * expf(-0.5f * z * z); }

struct RateTerm {
  float bits, dv, dmu, dsig;  // value and derivatives w.r.t. v, mu, |sigma|
};

// gsign: sign of the gradient arriving at `bits` (LowerBound passes when like >= 1e-8 OR the
// incoming gradient on `like` is negative; that gradient is gsign * (-1/(like ln2))).
__device__ __forceinline__ RateTerm rate_term(float v, float mu, float sabs, float half, float gsign) {
  const float inv_ln2 = 1.4426950408889634f;
  float up = (v - mu + half) / sabs, lo = (v - mu - half) / sabs;
  float like = std_cdf(up) - std_cdf(lo);
  float cl = fmaxf(like, 1e-8f);
  RateTerm r;
  r.bits = -1.f * logf(cl) / 0.6931471805599453f;
  float dbits_dlike = -inv_ln2 / cl;
  bool pass = (like >= 1e-8f) || (gsign * dbits_dlike < 0.f);
  if (!pass) dbits_dlike = 0.f;
  float pu = std_pdf(up), pl = std_pdf(lo);
  r.dv = dbits_dlike * (pu - pl) / sabs;
  r.dmu = -r.dv;
  r.dsig = dbits_dlike * (-(pu * up - pl * lo) / sabs);
  return r;
}

// latent quantisation + rate: one workgroup, channel-major loops (fixed order).  The arithmetic is that of a
// 1024-thread workgroup whatever blockDim.x is (a divisor of 1024, whole waves): real wave w plays the virtual waves
// w, w + nw, ..., each virtual wave's sum goes to red[], thread 0 adds them in wave order.  red: 48 floats.
constexpr int kRateVT = 1024;
__device__ __forceinline__ void latent_rate_body(const float* __restrict__ x, const float* __restrict__ u,
                                                 const int64_t* __restrict__ block_ids, const float* __restrict__ sigma,
                                                 const float* __restrict__ mu, float* __restrict__ x_rounded,
                                                 float* __restrict__ bits, float* __restrict__ dx,
                                                 const float* __restrict__ dx_addend, float* __restrict__ dsigma,
                                                 float* __restrict__ dmu, const float* __restrict__ g_dev, float g_host,
                                                 int batch, int c, int spatial, int mode, uint64_t seed, uint64_t step_in,
                                                 const uint64_t* __restrict__ step_dev, float* red,
                                                 float* scratch = nullptr, int scratch_floats = 0) {
  const uint64_t step = step_in + (step_dev ? step_dev[0] : 0ull);
  const float g = g_host * (g_dev ? g_dev[0] : 1.f);
  const float gsign = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int NVW = kRateVT / 64;
  const long nel = (long)batch * spatial;
  if (scratch && nel <= kRateVT && 3 * c * (nel + NVW) <= scratch_floats) {
    // Small tensors (a mini-batch of 2^3 latents): every (channel, element) term at once, then the sums in the order
    // of the channel-major loops below (element e of a channel belongs to virtual wave e / 64, lane e % 64).
    float* tb_ = scratch;                     // [3][c][nel] bits, dsigma, dmu terms
    float* rd_ = scratch + 3 * c * nel;       // [3][c][NVW] virtual-wave sums
    const long cn = (long)c * nel;
    for (long p = threadIdx.x; p < cn; p += blockDim.x) {
      const int ch = (int)(p / nel);
      const long e = p - (long)ch * nel;
      const long b = e / spatial;
      const int s = (int)(e % spatial);
      const long idx = (b * c + ch) * spatial + s;
      const float sabs = fabsf(sigma[ch]), m = mu[ch];
      float xv = x[idx];
      float xr = rintf(xv);
      if (x_rounded) x_rounded[idx] = xr;
      float v = xr;
      if (mode == 0) {
        float uu;
        if (u) {
          uu = u[idx];
        } else {
          uint64_t blk = block_ids ? (uint64_t)block_ids[b] : (uint64_t)b;
          uu = nvf_uniform01(seed, (blk << 20) ^ step * 0x9E3779B97F4A7C15ull, (uint64_t)(ch * spatial + s));
        }
        v = xv + (uu - 0.5f);
      }
      RateTerm r = rate_term(v, m, sabs, 0.5f, gsign);
      tb_[p] = r.bits;
      tb_[cn + p] = r.dsig;
      tb_[2 * cn + p] = r.dmu;
      if (dx) dx[idx] = (dx_addend ? dx_addend[idx] : 0.f) + g * r.dv;
    }
    __syncthreads();
    for (int q = wave; q < c * NVW; q += nw) {          // (channel, virtual wave) pairs
      const int ch = q / NVW, vw = q - ch * NVW;
      const long e = vw * 64 + lane;
      const bool in = e < nel;
      float sb = 0.f, ss = 0.f, sm_ = 0.f;
      if (in) {
        sb += tb_[ch * nel + e];
        ss += tb_[cn + ch * nel + e];
        sm_ += tb_[2 * cn + ch * nel + e];
      }
      sb = nvf_wave_sum(sb);
      ss = nvf_wave_sum(ss);
      sm_ = nvf_wave_sum(sm_);
      if (lane == 0) { rd_[q] = sb; rd_[c * NVW + q] = ss; rd_[2 * c * NVW + q] = sm_; }
    }
    __syncthreads();
    if ((int)threadIdx.x < c) {
      const int ch = threadIdx.x;
      float tb = 0.f, tsg = 0.f, tm = 0.f;
      for (int i = 0; i < NVW; ++i) tb += rd_[ch * NVW + i];
      for (int i = 0; i < NVW; ++i) tsg += rd_[c * NVW + ch * NVW + i];
      for (int i = 0; i < NVW; ++i) tm += rd_[2 * c * NVW + ch * NVW + i];
      const float sraw = sigma[ch];
      float sgn = sraw > 0.f ? 1.f : (sraw < 0.f ? -1.f : 0.f);
      if (dsigma) dsigma[ch] = g * tsg * sgn;
      if (dmu) dmu[ch] = g * tm;
      rd_[ch * NVW] = tb;
    }
    __syncthreads();
    if (threadIdx.x == 0 && bits) {
      float total = 0.f;
      for (int ch = 0; ch < c; ++ch) total += rd_[ch * NVW];
      bits[0] = total;
    }
    __syncthreads();
    return;
  }
  float total_bits = 0.f;
  for (int ch = 0; ch < c; ++ch) {
    const float sraw = sigma[ch], sabs = fabsf(sraw), m = mu[ch];
    for (int vw = wave; vw < NVW; vw += nw) {
      float sb = 0.f, ss = 0.f, sm_ = 0.f;
      for (long e = vw * 64 + lane; e < (long)batch * spatial; e += kRateVT) {
        long b = e / spatial;
        int s = (int)(e % spatial);
        long idx = (b * c + ch) * spatial + s;
        float xv = x[idx];
        float xr = rintf(xv);
        if (x_rounded) x_rounded[idx] = xr;
        float v = xr;
        if (mode == 0) {
          float uu;
          if (u) {
            uu = u[idx];
          } else {
            uint64_t blk = block_ids ? (uint64_t)block_ids[b] : (uint64_t)b;
            uu = nvf_uniform01(seed, (blk << 20) ^ step * 0x9E3779B97F4A7C15ull, (uint64_t)(ch * spatial + s));
          }
          v = xv + (uu - 0.5f);
        }
        RateTerm r = rate_term(v, m, sabs, 0.5f, gsign);
        sb += r.bits;
        ss += r.dsig;
        sm_ += r.dmu;
        if (dx) dx[idx] = (dx_addend ? dx_addend[idx] : 0.f) + g * r.dv;
      }
      sb = nvf_wave_sum(sb);
      ss = nvf_wave_sum(ss);
      sm_ = nvf_wave_sum(sm_);
      if (lane == 0) { red[vw] = sb; red[NVW + vw] = ss; red[2 * NVW + vw] = sm_; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float tb = 0.f, tsg = 0.f, tm = 0.f;
      for (int i = 0; i < NVW; ++i) tb += red[i];
      for (int i = 0; i < NVW; ++i) tsg += red[NVW + i];
      for (int i = 0; i < NVW; ++i) tm += red[2 * NVW + i];
      total_bits += tb;
      float sgn = sraw > 0.f ? 1.f : (sraw < 0.f ? -1.f : 0.f);
      if (dsigma) dsigma[ch] = g * tsg * sgn;
      if (dmu) dmu[ch] = g * tm;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && bits) bits[0] = total_bits;
}

// ---- latent generator + quantiser, forward (nvf_latent_fwd; also one workgroup of nvf_stem_latent_fwd) --------
// e -> 1x1x1 conv (+ bias) -> GDN -> round / noise + rate for ALL blocks by one workgroup; the arithmetic of
// nvf_conv3d_gather (k = 1) + nvf_gdn_fwd + nvf_latent_rate, and that of a 1024-thread workgroup whatever blockDim.x
// is (see latent_rate_body).  s_par: 144 + 16 floats of LDS.
__device__ __forceinline__ void latent_fwd_body(const float* __restrict__ e, const float* __restrict__ w,
                                                const float* __restrict__ bw, const float* __restrict__ beta_hat,
                                                const float* __restrict__ gamma_hat,
                                                const int64_t* __restrict__ block_ids, const float* __restrict__ sigma,
                                                const float* __restrict__ mu, float* __restrict__ h_out,
                                                float* __restrict__ lat_out, float* __restrict__ x_rounded,
                                                float* __restrict__ bits, int batch, int c, int spatial, int mode,
                                                uint64_t seed, uint64_t step_in, const uint64_t* __restrict__ step_dev,
                                                float* s_par, float* scratch = nullptr, int scratch_floats = 0) {
  float* s_w = s_par;            // [64]  w_fwd layout [ci][co]
  float* s_gamma = s_par + 64;   // [64]
  float* s_b = s_par + 128;      // [8]
  float* s_beta = s_par + 136;   // [8]
  float* red = s_par + 144;      // [16]
  if ((int)threadIdx.x < c * c) {
    s_w[threadIdx.x] = w[threadIdx.x];
    s_gamma[threadIdx.x] = gdn_gamma(gamma_hat[threadIdx.x]);
  }
  if ((int)threadIdx.x < c) {
    s_b[threadIdx.x] = bw[threadIdx.x];
    s_beta[threadIdx.x] = gdn_beta(beta_hat[threadIdx.x]);
  }
  __syncthreads();
  const uint64_t step = step_in + (step_dev ? step_dev[0] : 0ull);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int NVW = kRateVT / 64;
  const long nel = (long)batch * spatial;
  if (scratch && nel <= kRateVT && c * (nel + NVW) <= scratch_floats) {
    // a mini-batch: every (channel, element) at once, then the rate sums in the order of the loops below
    float* tb_ = scratch;                 // [c][nel] rate terms
    float* rd_ = scratch + c * nel;       // [c][NVW] virtual-wave sums
    const long cn = (long)c * nel;
    for (long p = threadIdx.x; p < cn; p += blockDim.x) {
      const int ch = (int)(p / nel);
      const long el = p - (long)ch * nel;
      const long b = el / spatial;
      const int sp = (int)(el % spatial);
      const float sabs = fabsf(sigma[ch]), m = mu[ch];
      float h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < c) {
          float acc = 0.f;
          for (int i = 0; i < c; ++i) acc = fmaf(e[(b * c + i) * spatial + sp], s_w[i * c + j], acc);
          h[j] = acc + s_b[j];
        }
      }
      float nrm = s_beta[ch], hc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < c) {
          nrm = fmaf(s_gamma[ch * c + j], h[j] * h[j], nrm);
          if (j == ch) hc = h[j];
        }
      }
      const float xv = hc / sqrtf(nrm);
      const long idx = (b * c + ch) * spatial + sp;
      h_out[idx] = hc;
      lat_out[idx] = xv;
      const float xr = rintf(xv);
      x_rounded[idx] = xr;
      float v = xr;
      if (mode == 0) {
        const uint64_t blk = block_ids ? (uint64_t)block_ids[b] : (uint64_t)b;
        const float uu = nvf_uniform01(seed, (blk << 20) ^ step * 0x9E3779B97F4A7C15ull, (uint64_t)(ch * spatial + sp));
        v = xv + (uu - 0.5f);
      }
      tb_[p] = rate_term(v, m, sabs, 0.5f, 0.f).bits;
    }
    __syncthreads();
    for (int q = wave; q < c * NVW; q += nw) {
      const int ch = q / NVW, vw = q - ch * NVW;
      const long el = vw * 64 + lane;
      float sb = 0.f;
      if (el < nel) sb += tb_[ch * nel + el];
      sb = nvf_wave_sum(sb);
      if (lane == 0) rd_[q] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float total = 0.f;
      for (int ch = 0; ch < c; ++ch) {
        float tb = 0.f;
        for (int i = 0; i < NVW; ++i) tb += rd_[ch * NVW + i];
        total += tb;
      }
      bits[0] = total;
    }
    return;
  }
  float total_bits = 0.f;
  for (int ch = 0; ch < c; ++ch) {
    const float sabs = fabsf(sigma[ch]), m = mu[ch];
    for (int vw = wave; vw < NVW; vw += nw) {
      float sb = 0.f;
      for (long el = vw * 64 + lane; el < (long)batch * spatial; el += kRateVT) {
        const long b = el / spatial;
        const int sp = (int)(el % spatial);
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < c) {
            float acc = 0.f;
            for (int i = 0; i < c; ++i) acc = fmaf(e[(b * c + i) * spatial + sp], s_w[i * c + j], acc);
            h[j] = acc + s_b[j];
          }
        }
        float nrm = s_beta[ch], hc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < c) {
            nrm = fmaf(s_gamma[ch * c + j], h[j] * h[j], nrm);
            if (j == ch) hc = h[j];
          }
        }
        const float xv = hc / sqrtf(nrm);
        const long idx = (b * c + ch) * spatial + sp;
        h_out[idx] = hc;
        lat_out[idx] = xv;
        const float xr = rintf(xv);
        x_rounded[idx] = xr;
        float v = xr;
        if (mode == 0) {
          const uint64_t blk = block_ids ? (uint64_t)block_ids[b] : (uint64_t)b;
          const float uu = nvf_uniform01(seed, (blk << 20) ^ step * 0x9E3779B97F4A7C15ull, (uint64_t)(ch * spatial + sp));
          v = xv + (uu - 0.5f);
        }
        sb += rate_term(v, m, sabs, 0.5f, 0.f).bits;
      }
      sb = nvf_wave_sum(sb);
      if (lane == 0) red[vw] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float tb = 0.f;
      for (int i = 0; i < NVW; ++i) tb += red[i];
      total_bits += tb;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) bits[0] = total_bits;
}

// rounded latent of ONE element (block b, channel ch, voxel sp) -- what latent_fwd_body writes to x_rounded, for a
// consumer that wants it without waiting for that workgroup (the fused stem); parameters straight from global memory
__device__ __forceinline__ float latent_x_rounded(const float* __restrict__ e, const float* __restrict__ w,
                                                  const float* __restrict__ bw, const float* __restrict__ beta_hat,
                                                  const float* __restrict__ gamma_hat, long b, int ch, int sp, int c,
                                                  int spatial) {
  float h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < c) {
      float acc = 0.f;
      for (int i = 0; i < c; ++i) acc = fmaf(e[(b * c + i) * spatial + sp], w[i * c + j], acc);
      h[j] = acc + bw[j];
    }
  }
  float nrm = gdn_beta(beta_hat[ch]), hc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < c) {
      nrm = fmaf(gdn_gamma(gamma_hat[ch * c + j]), h[j] * h[j], nrm);
      if (j == ch) hc = h[j];
    }
  }
  return rintf(hc / sqrtf(nrm));
}

// the same element from its c inputs already in registers and the parameters anywhere (LDS): the arithmetic of
// latent_x_rounded / latent_fwd_body (fmaf over i ascending, then the GDN sum over j ascending)
__device__ __forceinline__ float latent_x_rounded_from(const float (&ev)[8], const float* w, const float* bw,
                                                       const float* beta_hat, const float* gamma_hat, int ch, int c) {
  float h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < c) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < c) acc = fmaf(ev[i], w[i * c + j], acc);
      h[j] = acc + bw[j];
    }
  }
  float nrm = gdn_beta(beta_hat[ch]), hc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < c) {
      nrm = fmaf(gdn_gamma(gamma_hat[ch * c + j]), h[j] * h[j], nrm);
      if (j == ch) hc = h[j];
    }
  }
  return rintf(hc / sqrtf(nrm));
}

// ---- the latent tail ---------------------------------------------------------------------------------------------
struct LatentTail {
  // rate of the latents (+ the decoder's gradient arriving at them): d lat
  const float* lat;
  const int64_t* block_ids;
  const float* sigma;
  const float* mu;
  const float* dx_addend;
  float* dlat;
  float* dsigma;
  float* dmu;
  const float* g_dev;
  const uint64_t* step_dev;
  uint64_t seed, step;
  // GDN of the latent generator (forward direction): d h, d beta_hat, d gamma_hat
  const float* h;
  const float* beta_hat;
  const float* gamma_hat;
  float* dh;
  float* dbeta_hat;
  float* dgamma_hat;
  // 1x1x1 convolution of the latent generator: d kernel [c][c], d bias [c]
  const float* e;
  float* dw;
  float* db;
  float g_host;
  int32_t batch, c, spatial, mode;
};

constexpr int kTailMaxC = 8, kTailGdnT = 128;
constexpr int kTailGdnLds = 3 * kTailMaxC * (kTailGdnT + 1) + kTailMaxC * kTailMaxC + kTailMaxC;
constexpr int kTailRateLds = 3 * kTailMaxC * (128 + kRateVT / 64);      // a mini-batch of 16 blocks at c = 8
constexpr int kTailLds = (kTailGdnLds > kTailRateLds ? kTailGdnLds : kTailRateLds) + 48;   // floats of LDS the tail needs

// GDN backward of a tensor small enough for one workgroup: the arithmetic of gdn_bwd_kernel with gridDim.x == 1.
// blockDim.x / kTailGdnT threads share a voxel (thread (voxel, cg) owns the channels cg, cg + NG, ...); every thread
// takes part in the barriers.
__device__ __forceinline__ void gdn_bwd_one_workgroup(const float* __restrict__ x, const float* __restrict__ beta_hat,
                                                      const float* __restrict__ gamma_hat, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int batch, int c, int spatial, int inverse,
                                                      float* __restrict__ dbeta_hat, float* __restrict__ dgamma_hat,
                                                      float* sm) {
  constexpr int T = kTailGdnT, LD = T + 1;
  float* ts = sm;
  float* xs = sm + c * LD;
  float* ns = xs + c * LD;                   // norms (kept out of registers: the host kernel wants few of them)
  float* gam = ns + c * LD;
  float* bet = gam + c * c;
  const int tid = threadIdx.x;
  const int ng = (int)blockDim.x / T > 0 ? (int)blockDim.x / T : 1;
  const int vt = tid % T, cg = tid / T;
  const bool worker = cg < ng && (int)blockDim.x >= T;
  const int ncol = c + c * c;
  const long nvox = (long)batch * spatial;
  for (int i = tid; i < c * c; i += blockDim.x) gam[i] = gdn_gamma(gamma_hat[i]);
  for (int i = tid; i < c; i += blockDim.x) bet[i] = gdn_beta(beta_hat[i]);
  float own = 0.f;                           // ncol <= 72 < T: at most one parameter column per thread
  for (long base = 0; base < nvox; base += T) {
    const long v = base + vt;
    const bool live = worker && v < nvox;
    const long b = live ? v / spatial : 0;
    const int s = live ? (int)(v % spatial) : 0;
    const float* xb = x + b * c * spatial + s;
    const float* gb = dy + b * c * spatial + s;
    if (worker) {
#pragma unroll 1
      for (int ch = cg; ch < c; ch += ng) {
        const float xv = live ? xb[(long)ch * spatial] : 0.f;
        xs[ch * LD + vt] = xv * xv;
      }
    }
    __syncthreads();
    if (worker) {
#pragma unroll 1
      for (int ch = cg; ch < c; ch += ng) {
        float t = 0.f;
        if (live) {
          float acc = bet[ch];
#pragma unroll 1
          for (int j = 0; j < c; ++j) acc = fmaf(gam[ch * c + j], xs[j * LD + vt], acc);
          const float nrm = sqrtf(acc);
          const float xv = xb[(long)ch * spatial], g = gb[(long)ch * spatial];
          t = inverse ? g * xv / nrm : -g * xv / (nrm * nrm * nrm);
          ns[ch * LD + vt] = nrm;
        }
        ts[ch * LD + vt] = t;
      }
    }
    __syncthreads();
    if (live) {
#pragma unroll 1
      for (int i = cg; i < c; i += ng) {
        const float nrm = ns[i * LD + vt];
        float mix = 0.f;
#pragma unroll 1
        for (int ch = 0; ch < c; ++ch) mix = fmaf(ts[ch * LD + vt], gam[ch * c + i], mix);
        float xv = xb[(long)i * spatial], g = gb[(long)i * spatial];
        dx[(b * c + i) * spatial + s] = (inverse ? g * nrm : g / nrm) + xv * mix;
      }
    }
    if (tid < ncol) {
      const int p = tid;
      float sum = 0.f;
      if (p < c) {
#pragma unroll 4
        for (int k = 0; k < T; ++k) sum += ts[p * LD + k];
      } else {
        int ch = (p - c) / c, j = (p - c) % c;
#pragma unroll 4
        for (int k = 0; k < T; ++k) sum = fmaf(ts[ch * LD + k], xs[j * LD + k], sum);
      }
      own += 0.5f * sum;
    }
    __syncthreads();
  }
  if (tid < ncol) {
    const int p = tid;
    const float sv = 0.f + own;
    if (p < c) {
      const float h = beta_hat[p];
      const float g = sv * 2.f * fmaxf(h, NVF_BETA_BOUND);
      dbeta_hat[p] = (h >= NVF_BETA_BOUND || g < 0.f) ? g : 0.f;
    } else {
      const float h = gamma_hat[p - c];
      const float g = sv * 2.f * fmaxf(h, NVF_GAMMA_BOUND);
      dgamma_hat[p - c] = (h >= NVF_GAMMA_BOUND || g < 0.f) ? g : 0.f;
    }
  }
}

// One workgroup (>= 192 threads, whole waves), lds = kTailLds floats.  Stage results travel through global memory
// (they are outputs anyway); a WORKGROUP-scope fence + barrier separates the stages: producer and consumer waves sit on
// one CU, so the stores only have to be complete -- a device-scope __threadfence() writes the XCD's L2 back (the L2s of
// the eight XCDs are not coherent with each other), and with the host kernel's stores in flight each of the two cost
// about 20 us (measured with s_memrealtime stamps: rate 23, GDN 20, weight gradient 8 us in the wide step).
// dx_addend: the decoder's gradient at the rounded latents (t.dx_addend, or the carrier launch's own copy of it)
__device__ __forceinline__ void latent_tail_body(const LatentTail& t, float* lds, const float* dx_addend) {
  float* red = lds + kTailLds - 48;
  latent_rate_body(t.lat, nullptr, t.block_ids, t.sigma, t.mu, nullptr, nullptr, t.dlat, dx_addend, t.dsigma, t.dmu,
                   t.g_dev, t.g_host, t.batch, t.c, t.spatial, t.mode, t.seed, t.step, t.step_dev, red, lds,
                   kTailLds - 48);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  gdn_bwd_one_workgroup(t.h, t.beta_hat, t.gamma_hat, t.dlat, t.dh, t.batch, t.c, t.spatial, 0, t.dbeta_hat,
                        t.dgamma_hat, lds);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  // d kernel[a][b] = sum dh[n, a, pos] e[n, b, pos] (one wave per output, lanes 64 apart, fixed-order wave sum: the
  // arithmetic of wgrad_naive), then d bias[a] = sum dh[n, a, pos]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
  const long total = (long)t.batch * t.spatial;
  for (int j = wave; j < t.c * t.c + t.c; j += nwave) {
    const bool bias = j >= t.c * t.c;
    const int a = bias ? j - t.c * t.c : j / t.c, b = bias ? 0 : j % t.c;
    float acc = 0.f;
    for (long el = lane; el < total; el += 64) {
      const long n = el / t.spatial, r = el - n * t.spatial;
      const float pv = t.dh[(n * t.c + a) * t.spatial + r];
      acc = bias ? acc + pv : fmaf(pv, t.e[(n * t.c + b) * t.spatial + r], acc);
    }
    acc = nvf_wave_sum(acc);
    if (lane == 0) (bias ? t.db : t.dw)[bias ? a : j] = acc;
  }
}
