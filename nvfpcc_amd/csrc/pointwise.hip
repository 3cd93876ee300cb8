// Element-wise, normalisation, rate, loss, optimiser and compaction kernels of the NVF
// hot path on gfx950.  All of them are HBM- or latency-bound byte movers; the rules that
// matter are coalesced (16 B / lane where the layout allows) accesses, one pass over the
// data with the gradient fused into the forward where the caller wants it, and fixed-order
// two-stage reductions instead of float atomics so results are reproducible.
#include "nvf_common.h"
#include "step_ctx.h"
#include "pack_mfma.h"
#include "latent_tail.h"
#include "stem_fwd.h"

#define NVF_GRID(n, bs) ((unsigned)(((n) + (bs)-1) / (bs)))

extern "C" int nvf_version(void) { return 100; }

// ---------------------------------------------------------------------------
// generic fixed-order finaliser: out[j] (+)= sum_g part[g*ncol + j]
// ---------------------------------------------------------------------------
__global__ void finalize_partials(const float* __restrict__ part, float* __restrict__ out, int nrow, int ncol,
                                  int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncol) return;
  float s = 0.f;
  for (int g = 0; g < nrow; ++g) s += part[(size_t)g * ncol + j];
  out[j] = accumulate ? out[j] + s : s;
}

// ---------------------------------------------------------------------------
// effective parameters (network.py:611-620, 677-686, 735-740)
// ---------------------------------------------------------------------------
__global__ void effective_params_kernel(const float* __restrict__ kernel, const float* __restrict__ kernel_init,
                                        const float* __restrict__ u, float* __restrict__ w_eff, int n,
                                        const float* __restrict__ b, const float* __restrict__ b_init,
                                        float* __restrict__ b_eff, int nb, int q, uint64_t seed, uint64_t stream_id) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float k = kernel[i];
    if (q == 1) {
      float uu = u ? u[i] : nvf_uniform01(seed, stream_id, (uint64_t)i);
      k = k + (uu - 0.5f) * 0.0625f;
    } else if (q == 2) {
      k = rintf(k * 16.f) / 16.f;
    }
    w_eff[i] = k + kernel_init[i];
  } else if (i < n + nb) {
    int j = i - n;
    b_eff[j] = b[j] + b_init[j];
  }
}

extern "C" int nvf_effective_params(const float* kernel, const float* kernel_init, const float* u, float* w_eff, int n,
                                    const float* b, const float* b_init, float* b_eff, int nb, int q, uint64_t seed,
                                    uint64_t stream_id, void* stream) {
  if (!kernel || !kernel_init || !w_eff || n <= 0 || nb < 0) return NVF_EINVAL;
  if (nb > 0 && (!b || !b_init || !b_eff)) return NVF_EINVAL;
  effective_params_kernel<<<NVF_GRID(n + nb, 256), 256, 0, nvf_stream(stream)>>>(kernel, kernel_init, u, w_eff, n, b,
                                                                                  b_init, b_eff, nb, q, seed,
                                                                                  stream_id);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// All layers of the decoder in ONE launch: w_eff = f_q(kernel) + kernel_init written straight into
// the two scalar-load layouts (w_fwd[ci][k][co], w_bwd[co][k'][ci]) and b_eff = b + b_init.
struct NvfLayerDesc {
  const float* kernel;
  const float* kernel_init;
  const float* b;
  const float* b_init;
  float* w_fwd;
  float* w_bwd;
  float* b_eff;
  int32_t dim0, dim1, k3;   // kernel is [dim0][dim1][k3]
  int32_t kind;             // 0: conv [cout][cin][k] (w_bwd taps flipped); 1: convT [cin][cout][k]
  int32_t quantised;        // 1: Q-layer (q applies); 0: I-layer (raw kernel)
  int32_t layer_id;         // weight-noise stream id
  int32_t nbias, pad_;
};

// w_eff of element i of a layer's kernel: f_q(kernel) + kernel_init (q = 1: uniform noise of one quantisation step,
// q = 2: rounding to 1/16)
__device__ __forceinline__ float effective_weight(const NvfLayerDesc& d, int i, int qq, uint64_t seed, uint64_t sid) {
  float k = d.kernel[i];
  if (qq == 1) k = k + (nvf_uniform01(seed, sid, (uint64_t)i) - 0.5f) * 0.0625f;
  else if (qq == 2) k = rintf(k * 16.f) / 16.f;
  return k + d.kernel_init[i];
}

__device__ __forceinline__ void prepare_weights_body(const NvfLayerDesc* __restrict__ table, int q, uint64_t seed,
                                                     uint64_t step, const uint64_t* __restrict__ step_dev, int layer,
                                                     int bx, int nbx) {
  const NvfLayerDesc d = table[layer];
  const int n = d.dim0 * d.dim1 * d.k3;
  const int qq = d.quantised ? q : 0;
  const uint64_t st = step + (step_dev ? step_dev[0] : 0ull);
  const uint64_t sid = (st << 8) | (uint64_t)d.layer_id;
  for (int i = bx * blockDim.x + threadIdx.x; i < n + d.nbias; i += nbx * blockDim.x) {
    if (i < n) {
      const float w = effective_weight(d, i, qq, seed, sid);
      const int t = i % d.k3, i1 = (i / d.k3) % d.dim1, i0 = i / (d.k3 * d.dim1);
      if (d.kind == 0) {  // i0 = co, i1 = ci
        if (d.w_fwd) d.w_fwd[(i1 * d.k3 + t) * d.dim0 + i0] = w;
        if (d.w_bwd) d.w_bwd[(i0 * d.k3 + (d.k3 - 1 - t)) * d.dim1 + i1] = w;
      } else {            // i0 = ci, i1 = co
        if (d.w_fwd) d.w_fwd[(i0 * d.k3 + t) * d.dim1 + i1] = w;
        if (d.w_bwd) d.w_bwd[(i1 * d.k3 + t) * d.dim0 + i0] = w;
      }
    } else {
      const int j = i - n;
      d.b_eff[j] = d.b[j] + d.b_init[j];
    }
  }
}

__global__ void prepare_weights_kernel(const NvfLayerDesc* __restrict__ table, int q, uint64_t seed, uint64_t step,
                                       const uint64_t* __restrict__ step_dev) {
  prepare_weights_body(table, q, seed, step, step_dev, blockIdx.y, blockIdx.x, gridDim.x);
}

extern "C" size_t nvf_layer_desc_size(void) { return sizeof(NvfLayerDesc); }

extern "C" int nvf_prepare_weights(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step,
                                   const uint64_t* step_dev, void* stream) {
  if (!table_dev || nlayers <= 0) return NVF_EINVAL;
  prepare_weights_kernel<<<dim3(16, nlayers), 256, 0, nvf_stream(stream)>>>((const NvfLayerDesc*)table_dev, q, seed,
                                                                           step, step_dev);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// GDN / IGDN (gdn_3d.py:72-95, 137-159)
// ---------------------------------------------------------------------------
__global__ void gdn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ beta_hat,
                               const float* __restrict__ gamma_hat, float* __restrict__ y, int batch, int c,
                               int spatial, int inverse) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (b, ch, s)
  long total = (long)batch * c * spatial;
  if (idx >= total) return;
  int s = idx % spatial;
  int ch = (idx / spatial) % c;
  long b = idx / ((long)spatial * c);
  const float* xb = x + b * c * spatial + s;
  float acc = gdn_beta(beta_hat[ch]);
  for (int j = 0; j < c; ++j) {
    float xj = xb[(long)j * spatial];
    acc = fmaf(gdn_gamma(gamma_hat[ch * c + j]), xj * xj, acc);
  }
  float nrm = sqrtf(acc);
  float xv = xb[(long)ch * spatial];
  y[idx] = inverse ? xv * nrm : xv / nrm;
}

extern "C" int nvf_gdn_fwd(const float* x, const float* beta_hat, const float* gamma_hat, float* y, int batch, int c,
                           int spatial, int inverse, void* stream) {
  if (!x || !beta_hat || !gamma_hat || !y || batch <= 0 || c <= 0 || spatial <= 0) return NVF_EINVAL;
  long total = (long)batch * c * spatial;
  gdn_fwd_kernel<<<NVF_GRID(total, 256), 256, 0, nvf_stream(stream)>>>(x, beta_hat, gamma_hat, y, batch, c, spatial,
                                                                       inverse);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// backward: one thread per voxel (b, s).  t_c = dy_c x_c / n_c (IGDN) or -dy_c x_c / n_c^3 (GDN);
// dx_i = dy_i n_i^{+-1} + x_i sum_c t_c gamma_ci ; dbeta_c = sum t_c / 2 ; dgamma_cj = sum t_c x_j^2 / 2.
// Per-workgroup partial parameter sums go to a slab; gdn_bwd_final adds the slabs in order and
// applies the re-parametrisation chain rule with the LowerBound pass-through rule (gdn_3d.py:24-29).
static const int kGdnMaxC = 32;
static const int kGdnThreads = 128;
static const int kGdnMaxSlabs = 256;

// A workgroup walks tiles of kGdnThreads voxels; CG threads share a voxel, thread (voxel, cg) owning the channels
// cg, cg + CG, ... of it (the per-voxel channel loops are the serial part of this kernel: on the [B, 32, 4^3] tensor
// of the wide decoder one thread per voxel meant 8 workgroups walking 5000 dependent iterations each).  The
// re-parametrised beta / gamma sit in LDS.  The arithmetic and its order do not depend on CG.
template <int CG>
__global__ __launch_bounds__(kGdnThreads * CG) void gdn_bwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ beta_hat,
                                                                   const float* __restrict__ gamma_hat,
                                                                   const float* __restrict__ dy, float* __restrict__ dx,
                                                                   float* __restrict__ slabs, int batch, int c,
                                                                   int spatial, int inverse, int vox_per_wg,
                                                                   float* __restrict__ dbeta_hat,
                                                                   float* __restrict__ dgamma_hat) {
  extern __shared__ float sm[];             // ts[c][T+1], xs[c][T+1], gam[c][c], bet[c]
  constexpr int T = kGdnThreads, LD = T + 1, NT = T * CG, MAXK = kGdnMaxC / CG;
  float* ts = sm;
  float* xs = sm + c * LD;
  float* gam = xs + c * LD;
  float* bet = gam + c * c;
  const int tid = threadIdx.x, vt = tid % T, cg = tid / T;
  const int ncol = c + c * c;
  const long nvox = (long)batch * spatial;
  const long v_lo = (long)blockIdx.x * vox_per_wg;
  long v_hi = v_lo + vox_per_wg;
  if (v_hi > nvox) v_hi = nvox;
  for (int i = tid; i < c * c; i += NT) gam[i] = gdn_gamma(gamma_hat[i]);
  for (int i = tid; i < c; i += NT) bet[i] = gdn_beta(beta_hat[i]);
  // running parameter partials of the columns this thread owns
  float own[(kGdnMaxC + kGdnMaxC * kGdnMaxC + NT - 1) / NT];
#pragma unroll
  for (int i = 0; i < (int)(sizeof(own) / sizeof(float)); ++i) own[i] = 0.f;

  for (long base = v_lo; base < v_hi; base += T) {
    const long v = base + vt;
    const bool live = v < v_hi;
    const long b = live ? v / spatial : 0;
    const int s = live ? (int)(v % spatial) : 0;
    const float* xb = x + b * c * spatial + s;
    const float* gb = dy + b * c * spatial + s;
    float xr[MAXK], gr[MAXK], nr[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int ch = cg + CG * k;
      if (ch < c) {
        xr[k] = live ? xb[(long)ch * spatial] : 0.f;
        gr[k] = live ? gb[(long)ch * spatial] : 0.f;
        xs[ch * LD + vt] = xr[k] * xr[k];
      }
    }
    __syncthreads();      // (also: gam / bet are complete)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int ch = cg + CG * k;
      if (ch < c) {
        float t = 0.f;
        nr[k] = 1.f;
        if (live) {
          float acc = bet[ch];
          for (int j = 0; j < c; ++j) acc = fmaf(gam[ch * c + j], xs[j * LD + vt], acc);
          const float nrm = sqrtf(acc);
          const float xv = xr[k], g = gr[k];
          t = inverse ? g * xv / nrm : -g * xv / (nrm * nrm * nrm);
          nr[k] = nrm;
        }
        ts[ch * LD + vt] = t;
      }
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int i = cg + CG * k;
        if (i < c) {
          const float nrm = nr[k];
          float mix = 0.f;
          for (int ch = 0; ch < c; ++ch) mix = fmaf(ts[ch * LD + vt], gam[ch * c + i], mix);
          float xv = xr[k], g = gr[k];
          dx[(b * c + i) * spatial + s] = (inverse ? g * nrm : g / nrm) + xv * mix;
        }
      }
    }
    // parameter partials: column p < c is dbeta_p, column c + ch*c + j is dgamma_{ch,j}
    int slot = 0;
    for (int p = tid; p < ncol; p += NT, ++slot) {
      float sum = 0.f;
      if (p < c) {
        for (int k = 0; k < T; ++k) sum += ts[p * LD + k];
      } else {
        int ch = (p - c) / c, j = (p - c) % c;
        for (int k = 0; k < T; ++k) sum = fmaf(ts[ch * LD + k], xs[j * LD + k], sum);
      }
      own[slot] += 0.5f * sum;
    }
    __syncthreads();
  }
  int slot = 0;
  if (gridDim.x == 1) {
    // a single workgroup holds the complete sums: finish here (what gdn_bwd_final does for one slab: 0 + s = s)
    for (int p = tid; p < ncol; p += NT, ++slot) {
      const float sv = 0.f + own[slot];
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
    return;
  }
  for (int p = tid; p < ncol; p += NT, ++slot) slabs[(size_t)blockIdx.x * ncol + p] = own[slot];
}

__global__ void gdn_bwd_final(const float* __restrict__ slabs, const float* __restrict__ beta_hat,
                              const float* __restrict__ gamma_hat, float* __restrict__ dbeta_hat,
                              float* __restrict__ dgamma_hat, int nslab, int c) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int ncol = c + c * c;
  if (p >= ncol) return;
  float s = 0.f;
  for (int g = 0; g < nslab; ++g) s += slabs[(size_t)g * ncol + p];
  if (p < c) {
    float h = beta_hat[p];
    float g = s * 2.f * fmaxf(h, NVF_BETA_BOUND);   // d/d(clamped) of clamped^2 - pedestal
    dbeta_hat[p] = (h >= NVF_BETA_BOUND || g < 0.f) ? g : 0.f;
  } else {
    float h = gamma_hat[p - c];
    float g = s * 2.f * fmaxf(h, NVF_GAMMA_BOUND);
    dgamma_hat[p - c] = (h >= NVF_GAMMA_BOUND || g < 0.f) ? g : 0.f;
  }
}

extern "C" size_t nvf_gdn_bwd_workspace(int c) { return (size_t)kGdnMaxSlabs * (c + c * c) * sizeof(float); }

extern "C" int nvf_gdn_bwd(const float* x, const float* beta_hat, const float* gamma_hat, const float* dy, float* dx,
                           float* dbeta_hat, float* dgamma_hat, void* workspace, size_t workspace_bytes, int batch,
                           int c, int spatial, int inverse, void* stream) {
  if (!x || !beta_hat || !gamma_hat || !dy || !dx || !dbeta_hat || !dgamma_hat || !workspace) return NVF_EINVAL;
  if (batch <= 0 || c <= 0 || c > kGdnMaxC || spatial <= 0) return NVF_EINVAL;
  if (workspace_bytes < nvf_gdn_bwd_workspace(c)) return NVF_EWORKSPACE;
  long nvox = (long)batch * spatial;
  long per = (nvox + kGdnMaxSlabs - 1) / kGdnMaxSlabs;
  per = (per + kGdnThreads - 1) / kGdnThreads * kGdnThreads;
  int nslab = (int)((nvox + per - 1) / per);
  size_t lds = ((size_t)2 * c * (kGdnThreads + 1) + c * c + c) * sizeof(float);
  hipStream_t s = nvf_stream(stream);
  // few voxels and many channels (the decoder's IGDN sits on 4^3 grids): spread the channels of a voxel over 8 / 4 threads
  const bool few = nslab < 128;
  if (few && c % 8 == 0)
    gdn_bwd_kernel<8><<<nslab, kGdnThreads * 8, lds, s>>>(x, beta_hat, gamma_hat, dy, dx, (float*)workspace, batch, c,
                                                         spatial, inverse, (int)per, dbeta_hat, dgamma_hat);
  else if (few && c % 4 == 0)
    gdn_bwd_kernel<4><<<nslab, kGdnThreads * 4, lds, s>>>(x, beta_hat, gamma_hat, dy, dx, (float*)workspace, batch, c,
                                                         spatial, inverse, (int)per, dbeta_hat, dgamma_hat);
  else
    gdn_bwd_kernel<1><<<nslab, kGdnThreads, lds, s>>>(x, beta_hat, gamma_hat, dy, dx, (float*)workspace, batch, c,
                                                      spatial, inverse, (int)per, dbeta_hat, dgamma_hat);
  if (nslab > 1)      // one workgroup (the latent GDN of a mini-batch) finishes the parameter gradients itself
    gdn_bwd_final<<<NVF_GRID(c + c * c, 64), 64, 0, s>>>((const float*)workspace, beta_hat, gamma_hat, dbeta_hat,
                                                         dgamma_hat, nslab, c);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// Gaussian rate helpers (network.py:145-161)
// ---------------------------------------------------------------------------
// latent quantisation + rate: one workgroup, channel-major loops (fixed order, C + 1 block reductions)
__global__ __launch_bounds__(1024) void latent_rate_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                           const int64_t* __restrict__ block_ids,
                                                           const float* __restrict__ sigma,
                                                           const float* __restrict__ mu, float* __restrict__ x_rounded,
                                                           float* __restrict__ bits, float* __restrict__ dx,
                                                           const float* __restrict__ dx_addend,
                                                           float* __restrict__ dsigma, float* __restrict__ dmu,
                                                           const float* __restrict__ g_dev, float g_host, int batch,
                                                           int c, int spatial, int mode, uint64_t seed, uint64_t step_in,
                                                           const uint64_t* __restrict__ step_dev) {
  __shared__ float red[48];
  __shared__ float scratch[kTailRateLds];
  latent_rate_body(x, u, block_ids, sigma, mu, x_rounded, bits, dx, dx_addend, dsigma, dmu, g_dev, g_host, batch, c,
                   spatial, mode, seed, step_in, step_dev, red, scratch, kTailRateLds);
}

extern "C" int nvf_latent_rate(const float* x, const float* u, const int64_t* block_ids, const float* sigma,
                               const float* mu, float* x_rounded, float* bits, float* dx, const float* dx_addend,
                               float* dsigma, float* dmu, const float* g_dev, float g_host, int batch, int c,
                               int spatial, int mode,
                               uint64_t seed, uint64_t step, const uint64_t* step_dev, void* stream) {
  if (!x || !sigma || !mu || batch <= 0 || c <= 0 || spatial <= 0) return NVF_EINVAL;
  if (mode != 0 && mode != 1) return NVF_EINVAL;
  latent_rate_kernel<<<1, 1024, 0, nvf_stream(stream)>>>(x, u, block_ids, sigma, mu, x_rounded, bits, dx, dx_addend,
                                                         dsigma, dmu, g_dev, g_host, batch, c, spatial, mode, seed,
                                                         step, step_dev);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// The latent generator and the latent quantiser in one launch (forward only): e -> 1x1x1 conv (+ bias) -> GDN ->
// round / noise + rate.  Same arithmetic, in the same order, as nvf_conv3d_gather (k = 1) + nvf_gdn_fwd +
// nvf_latent_rate: h and lat are written too (the backward pass reads them).  c <= 8.
__global__ __launch_bounds__(1024) void latent_fwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                          const float* __restrict__ bw,
                                                          const float* __restrict__ beta_hat,
                                                          const float* __restrict__ gamma_hat,
                                                          const int64_t* __restrict__ block_ids,
                                                          const float* __restrict__ sigma, const float* __restrict__ mu,
                                                          float* __restrict__ h_out, float* __restrict__ lat_out,
                                                          float* __restrict__ x_rounded, float* __restrict__ bits,
                                                          int batch, int c, int spatial, int mode, uint64_t seed,
                                                          uint64_t step_in, const uint64_t* __restrict__ step_dev) {
  __shared__ float s_par[160];
  __shared__ float scratch[kTailRateLds / 3];
  latent_fwd_body(e, w, bw, beta_hat, gamma_hat, block_ids, sigma, mu, h_out, lat_out, x_rounded, bits, batch, c, spatial,
                  mode, seed, step_in, step_dev, s_par, scratch, kTailRateLds / 3);
}

extern "C" int nvf_latent_fwd(const float* e, const float* w_fwd, const float* bias, const float* beta_hat,
                              const float* gamma_hat, const int64_t* block_ids, const float* sigma, const float* mu,
                              float* h, float* lat, float* x_rounded, float* bits, int batch, int c, int spatial,
                              int mode, uint64_t seed, uint64_t step, const uint64_t* step_dev, void* stream) {
  if (!e || !w_fwd || !bias || !beta_hat || !gamma_hat || !sigma || !mu || !h || !lat || !x_rounded || !bits)
    return NVF_EINVAL;
  if (batch <= 0 || c <= 0 || c > 8 || spatial <= 0 || (mode != 0 && mode != 1)) return NVF_EINVAL;
  latent_fwd_kernel<<<1, 1024, 0, nvf_stream(stream)>>>(e, w_fwd, bias, beta_hat, gamma_hat, block_ids, sigma, mu, h,
                                                        lat, x_rounded, bits, batch, c, spatial, mode, seed, step,
                                                        step_dev);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// weight rate of one quantised kernel (network.py:4777-4778, 301-305)
__global__ __launch_bounds__(1024) void weight_rate_kernel(const float* __restrict__ kernel, int n,
                                                           const float* __restrict__ sigma,
                                                           const float* __restrict__ mu, float* __restrict__ bits,
                                                           float* __restrict__ dk, float* __restrict__ dsigma,
                                                           float* __restrict__ dmu, const float* __restrict__ g_dev,
                                                           float g_host, int accumulate) {
  __shared__ float red[16];
  const float g = g_host * (g_dev ? g_dev[0] : 1.f);
  const float gsign = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
  const float sraw = sigma[0], sabs = fabsf(sraw), m = mu[0];
  float sb = 0.f, ss = 0.f, sm_ = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = rintf(kernel[i] * 16.f) / 16.f;
    RateTerm r = rate_term(v, m, sabs, 0.03125f, gsign);
    sb += r.bits;
    ss += r.dsig;
    sm_ += r.dmu;
    if (dk) dk[i] = accumulate ? dk[i] + g * r.dv : g * r.dv;
  }
  float tb = nvf_block_sum(sb, red);
  float tsg = nvf_block_sum(ss, red);
  float tm = nvf_block_sum(sm_, red);
  if (threadIdx.x == 0) {
    if (bits) bits[0] = tb;
    float sgn = sraw > 0.f ? 1.f : (sraw < 0.f ? -1.f : 0.f);
    if (dsigma) dsigma[0] = accumulate ? dsigma[0] + g * tsg * sgn : g * tsg * sgn;
    if (dmu) dmu[0] = accumulate ? dmu[0] + g * tm : g * tm;
  }
}

extern "C" int nvf_weight_rate(const float* kernel, int n, const float* sigma, const float* mu, float* bits, float* dk,
                               float* dsigma, float* dmu, const float* g_dev, float g_host, int accumulate,
                               void* stream) {
  if (!kernel || !sigma || !mu || n <= 0) return NVF_EINVAL;
  weight_rate_kernel<<<1, 1024, 0, nvf_stream(stream)>>>(kernel, n, sigma, mu, bits, dk, dsigma, dmu, g_dev, g_host,
                                                         accumulate);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// all quantised kernels of the decoder in two launches: workgroups own fixed chunks of one layer each and
// write (bits, dsigma, dmu) partials; one workgroup then adds the partials in chunk order (reproducible).
// dk is ADDED to the weight gradients already in place.

// partial sums of workgroup `wg` (its fixed chunk of one layer): part[3 wg ..] = bits, d/dsigma, d/dmu; dk (if given)
// is ADDED to (accumulate) or overwritten with g * dbits/dk
__device__ __forceinline__ void weight_rate_partial_body(const WeightRateBatch& b, const float* __restrict__ sigma,
                                                         const float* __restrict__ mu, float* __restrict__ part, float g,
                                                         int wg, int accumulate, float* red, int nthreads = 0) {
  // nthreads (0 = blockDim.x): the threads that walk the chunk -- a carrier launch with larger workgroups keeps the
  // 256-thread arithmetic (the others add zeros to the block sums: the same bits)
  if (nthreads <= 0) nthreads = blockDim.x;
  const float gsign = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
  const float sabs = fabsf(sigma[0]), m = mu[0];
  int l = 0;
  while (l + 1 < b.nlayers && wg >= b.first_wg[l + 1]) ++l;
  const int lo = (wg - b.first_wg[l]) * b.chunk;
  const int hi = min(lo + b.chunk, b.n[l]);
  const float* k = b.kernel[l];
  float* dk = b.dk[l];
  float sb = 0.f, ss = 0.f, sm_ = 0.f;
  for (int i = lo + threadIdx.x; i < hi && (int)threadIdx.x < nthreads; i += nthreads) {
    float v = rintf(k[i] * 16.f) / 16.f;
    RateTerm r = rate_term(v, m, sabs, 0.03125f, gsign);
    sb += r.bits;
    ss += r.dsig;
    sm_ += r.dmu;
    if (dk) dk[i] = accumulate ? dk[i] + g * r.dv : g * r.dv;
  }
  float tb = nvf_block_sum(sb, red);
  float tsg = nvf_block_sum(ss, red);
  float tm = nvf_block_sum(sm_, red);
  if (threadIdx.x == 0) {
    part[3 * wg] = tb;
    part[3 * wg + 1] = tsg;
    part[3 * wg + 2] = tm;
  }
}

__global__ __launch_bounds__(256) void weight_rate_batch_kernel(WeightRateBatch b, const float* __restrict__ sigma,
                                                                const float* __restrict__ mu,
                                                                float* __restrict__ part,
                                                                const float* __restrict__ g_dev, float g_host) {
  __shared__ float red[16];
  weight_rate_partial_body(b, sigma, mu, part, g_host * (g_dev ? g_dev[0] : 1.f), blockIdx.x, 1, red);
}

// workgroup layout of the batch: fixed chunks, <= ~256 + nlayers workgroups (the final pass walks them in this order)
static int weight_rate_batch_desc(const float* const* kernels, float* const* dks, const int* ns, int nlayers,
                                  WeightRateBatch& b) {
  if (!kernels || !ns || nlayers <= 0 || nlayers > 8) return NVF_EINVAL;
  long total = 0;
  for (int i = 0; i < nlayers; ++i) {
    if (!kernels[i] || ns[i] <= 0) return NVF_EINVAL;
    b.kernel[i] = kernels[i];
    b.dk[i] = dks ? dks[i] : nullptr;
    b.n[i] = ns[i];
    total += ns[i];
  }
  b.nlayers = nlayers;
  long chunk = (total + 255) / 256;
  if (chunk < 256) chunk = 256;              // at least one element per thread
  b.chunk = (int)chunk;
  int wg = 0;
  for (int i = 0; i < nlayers; ++i) {
    b.first_wg[i] = wg;
    wg += (int)((ns[i] + chunk - 1) / chunk);
  }
  b.first_wg[nlayers] = wg;
  return wg > 512 ? NVF_EINVAL : NVF_OK;
}

__global__ void weight_rate_batch_final(WeightRateBatch b, const float* __restrict__ part,
                                        const float* __restrict__ sigma, float* __restrict__ bits,
                                        float* __restrict__ dsigma, float* __restrict__ dmu,
                                        const float* __restrict__ g_dev, float g_host) {
  weight_rate_batch_final_body(b, part, sigma, bits, dsigma, dmu, g_dev, g_host, threadIdx.x);
}

extern "C" int nvf_weight_rate_batch_final(const NvfRateJob* job, float* bits, float* dsigma, float* dmu,
                                           NvfStepCtx* ctx, void* stream) {
  if (!job || !bits || !job->sigma || !job->mu || !job->part) return NVF_EINVAL;
  WeightRateBatch b{};
  const int rc = weight_rate_batch_desc(job->kernel, job->dk, job->n, job->nlayers, b);
  if (rc != NVF_OK) return rc;
  if (!nvf_finals_push_rate(ctx, b, job->part, job->sigma, bits, dsigma, dmu, nullptr, job->g))
    weight_rate_batch_final<<<1, 64, 0, nvf_stream(stream)>>>(b, job->part, job->sigma, bits, dsigma, dmu, nullptr, job->g);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" size_t nvf_weight_rate_batch_workspace(void) { return (size_t)3 * 512 * sizeof(float); }

extern "C" int nvf_weight_rate_batch(const float* const* kernels, float* const* dks, const int* ns, int nlayers,
                                     const float* sigma, const float* mu, float* bits, float* dsigma, float* dmu,
                                     const float* g_dev, float g_host, void* workspace, size_t workspace_bytes,
                                     NvfStepCtx* ctx, void* stream) {
  if (!kernels || !ns || nlayers <= 0 || nlayers > 8 || !sigma || !mu || !bits || !workspace) return NVF_EINVAL;
  if (workspace_bytes < nvf_weight_rate_batch_workspace()) return NVF_EWORKSPACE;
  WeightRateBatch b{};
  const int rc = weight_rate_batch_desc(kernels, dks, ns, nlayers, b);
  if (rc != NVF_OK) return rc;
  const int wg = b.first_wg[nlayers];
  hipStream_t s = nvf_stream(stream);
  weight_rate_batch_kernel<<<wg, 256, 0, s>>>(b, sigma, mu, (float*)workspace, g_dev, g_host);
  if (!nvf_finals_push_rate(ctx, b, (const float*)workspace, sigma, bits, dsigma, dmu, g_dev, g_host))
    weight_rate_batch_final<<<1, 64, 0, s>>>(b, (const float*)workspace, sigma, bits, dsigma, dmu, g_dev, g_host);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// focal losses fused with their gradient (utils/loss.py:61-72, 94-111)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ p, const float* __restrict__ gt,
                                                    const float* __restrict__ dist, float alpha, float beta,
                                                    float* __restrict__ part, float* __restrict__ dp,
                                                    const float* __restrict__ g_dev, float g_host, long n,
                                                    int chain_sigmoid) {
  __shared__ float red[16];
  const float g = g_host * (g_dev ? g_dev[0] : 1.f);
  const float a1 = alpha, a0 = 1.f - alpha;  // fp32 "-alpha + 1" as the reference evaluates it
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o;
    s += focal_elem(p[i], gt[i], dist ? dist[i] : 0.f, dist != nullptr, a1, a0, beta, chain_sigmoid, o, g);
    if (dp) dp[i] = o;
  }
  float t = nvf_block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

extern "C" size_t nvf_reduce_workspace(void) { return (size_t)kLossMaxWG * 8 * sizeof(float); }

extern "C" int nvf_focal_loss(const float* p, const float* gt, const float* dist, float alpha, float beta, float* loss,
                              float* dp, const float* g_dev, float g_host, void* workspace, size_t workspace_bytes,
                              int64_t n, int accumulate, int chain_sigmoid, void* stream) {
  if (!p || !gt || !loss || !workspace || n <= 0) return NVF_EINVAL;
  if (workspace_bytes < nvf_reduce_workspace()) return NVF_EWORKSPACE;
  int nwg = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (nwg > kLossMaxWG) nwg = kLossMaxWG;
  hipStream_t s = nvf_stream(stream);
  focal_kernel<<<nwg, 256, 0, s>>>(p, gt, dist, alpha, beta, (float*)workspace, dp, g_dev, g_host, (long)n,
                                   chain_sigmoid);
  finalize_partials<<<1, 64, 0, s>>>((const float*)workspace, loss, nwg, 1, accumulate);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// the three focal terms of the objective (main output + two heads, NVFPCC.py:166-184) in one launch pair

// every thread takes float4 groups (all loads of a group in flight together); a thread's terms are added in index
// order, the block sum is the fixed-order nvf_block_sum
__global__ __launch_bounds__(256) void focal_multi_kernel(FocalMulti m, float* __restrict__ part, int chain_sigmoid) {
  __shared__ float red[16];
  const int t = blockIdx.y;
  if ((int)blockIdx.x >= m.nwg[t]) return;
  const float* p = m.p[t];
  const float* gt = m.gt[t];
  const float* dist = m.dist[t];
  float* dp = m.dp[t];
  const float a1 = m.alpha[t], a0 = 1.f - m.alpha[t], beta = m.beta[t];
  float s = 0.f;
  const long n4 = m.n[t] >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)m.nwg[t] * blockDim.x) {
    const float4 pv = ((const float4*)p)[i], gv = ((const float4*)gt)[i];
    const float4 dv = dist ? ((const float4*)dist)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    s += focal_elem(pv.x, gv.x, dv.x, dist != nullptr, a1, a0, beta, chain_sigmoid, o.x);
    s += focal_elem(pv.y, gv.y, dv.y, dist != nullptr, a1, a0, beta, chain_sigmoid, o.y);
    s += focal_elem(pv.z, gv.z, dv.z, dist != nullptr, a1, a0, beta, chain_sigmoid, o.z);
    s += focal_elem(pv.w, gv.w, dv.w, dist != nullptr, a1, a0, beta, chain_sigmoid, o.w);
    if (dp) ((float4*)dp)[i] = o;
  }
  if (blockIdx.x == 0)                                         // tail (n not a multiple of four)
    for (long i = 4 * n4 + threadIdx.x; i < m.n[t]; i += blockDim.x) {
      float o;
      s += focal_elem(p[i], gt[i], dist ? dist[i] : 0.f, dist != nullptr, a1, a0, beta, chain_sigmoid, o);
      if (dp) dp[i] = o;
    }
  float tot = nvf_block_sum(s, red);
  if (threadIdx.x == 0) part[t * kLossMaxWG + blockIdx.x] = tot;
}

__global__ void focal_multi_final(FocalMulti m, const float* __restrict__ part, float* __restrict__ loss, int nterm) {
  focal_multi_final_body(m, part, loss, nterm, threadIdx.x);
}

extern "C" int nvf_focal_loss_multi(const float* const* ps, const float* const* gts, const float* const* dists,
                                    float* const* dps, const float* alphas, const float* betas, const int64_t* ns,
                                    int nterm, float* loss, int chain_sigmoid, void* workspace,
                                    size_t workspace_bytes, NvfStepCtx* ctx, void* stream) {
  if (!ps || !gts || !alphas || !ns || !loss || !workspace || nterm <= 0 || nterm > 3) return NVF_EINVAL;
  if (workspace_bytes < nvf_reduce_workspace()) return NVF_EWORKSPACE;
  FocalMulti m{};
  int maxwg = 1;
  for (int t = 0; t < nterm; ++t) {
    if (!ps[t] || !gts[t] || ns[t] <= 0) return NVF_EINVAL;
    m.p[t] = ps[t]; m.gt[t] = gts[t]; m.dist[t] = dists ? dists[t] : nullptr; m.dp[t] = dps ? dps[t] : nullptr;
    m.alpha[t] = alphas[t]; m.beta[t] = betas ? betas[t] : 0.f; m.n[t] = (long)ns[t];
    if ((((uintptr_t)ps[t] | (uintptr_t)gts[t] | (uintptr_t)m.dist[t] | (uintptr_t)m.dp[t]) & 15) != 0)
      return NVF_EINVAL;                                        // float4 access
    int nwg = (int)((ns[t] + 256 * 4 - 1) / (256 * 4));        // one float4 group per thread up to kLossMaxWG groups
    if (nwg > kLossMaxWG) nwg = kLossMaxWG;
    m.nwg[t] = nwg;
    if (nwg > maxwg) maxwg = nwg;
  }
  hipStream_t s = nvf_stream(stream);
  focal_multi_kernel<<<dim3(maxwg, nterm), 256, 0, s>>>(m, (float*)workspace, chain_sigmoid);
  if (!nvf_finals_push_focal(ctx, m, (const float*)workspace, loss, nterm))
    focal_multi_final<<<1, 64 * nterm, 0, s>>>(m, (const float*)workspace, loss, nterm);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// metrics (utils/loss.py:74-84, 113-121): tp, ap, tn, an at thh_acc; sse, denom at thh_sse -- of up to three
// (prediction, ground truth) pairs in one launch (blockIdx.y = term): the main output and the two coarse heads, whose
// accuracies the reference's log line prints beside it (NVFPCC.py:174-179, 261-281).  dist may be NULL (sse = 0).
struct MetricsMulti {
  const float* p[3];
  const float* gt[3];
  const float* dist[3];
  long n[3];
  int nwg[3];
};

__global__ __launch_bounds__(256) void metrics_kernel(MetricsMulti m, float thh_acc, float thh_sse,
                                                      float* __restrict__ part) {
  __shared__ float red[16];
  const int t = blockIdx.y;
  if ((int)blockIdx.x >= m.nwg[t]) return;
  const float* __restrict__ p = m.p[t];
  const float* __restrict__ gt = m.gt[t];
  const float* __restrict__ dist = m.dist[t];
  const long n = m.n[t];
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)m.nwg[t] * blockDim.x) {
    float pv = p[i];
    bool occ = gt[i] != 0.f;
    v[0] += (pv > thh_acc && occ) ? 1.f : 0.f;
    v[1] += occ ? 1.f : 0.f;
    v[2] += (pv <= thh_acc && !occ) ? 1.f : 0.f;
    v[3] += occ ? 0.f : 1.f;
    if (pv > thh_sse) {
      float dv = dist ? dist[i] : 0.f;
      v[4] += dv * dv;
      v[5] += 1.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float tot = nvf_block_sum(v[k], red);
    if (threadIdx.x == 0) part[((size_t)t * kLossMaxWG + blockIdx.x) * 6 + k] = tot;
  }
}

__global__ void metrics_final_kernel(const float* __restrict__ part, float* __restrict__ out, int n0, int n1, int n2,
                                     int nterm, int accumulate) {
  const int32_t nwg[3] = {n0, n1, n2};
  metrics_final_body(part, out, nwg, nterm, accumulate, threadIdx.x);
}

extern "C" size_t nvf_metrics_workspace(void) { return (size_t)3 * kLossMaxWG * 6 * sizeof(float); }

static int metrics_launch(const float* const* ps, const float* const* gts, const float* const* dists,
                          const int64_t* ns, int nterm, float thh_acc, float thh_sse, float* out, void* workspace,
                          size_t workspace_bytes, int accumulate, NvfStepCtx* ctx, void* stream) {
  if (!ps || !gts || !ns || !out || !workspace || nterm < 1 || nterm > 3) return NVF_EINVAL;
  if (workspace_bytes < nvf_metrics_workspace()) return NVF_EWORKSPACE;
  MetricsMulti m{};
  int most = 0;
  for (int t = 0; t < nterm; ++t) {
    if (!ps[t] || !gts[t] || ns[t] <= 0) return NVF_EINVAL;
    m.p[t] = ps[t]; m.gt[t] = gts[t]; m.dist[t] = dists ? dists[t] : nullptr; m.n[t] = (long)ns[t];
    int nwg = (int)((ns[t] + 256 * 8 - 1) / (256 * 8));
    m.nwg[t] = nwg > kLossMaxWG ? kLossMaxWG : nwg;
    if (m.nwg[t] > most) most = m.nwg[t];
  }
  hipStream_t s = nvf_stream(stream);
  metrics_kernel<<<dim3(most, nterm), 256, 0, s>>>(m, thh_acc, thh_sse, (float*)workspace);
  if (!nvf_finals_push_metrics(ctx, (const float*)workspace, out, m.nwg, nterm, accumulate))
    metrics_final_kernel<<<1, 64, 0, s>>>((const float*)workspace, out, m.nwg[0], m.nwg[1], m.nwg[2], nterm,
                                          accumulate);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_metrics(const float* p, const float* gt, const float* dist, float thh_acc, float thh_sse, float* out,
                           void* workspace, size_t workspace_bytes, int64_t n, int accumulate, NvfStepCtx* ctx,
                           void* stream) {
  return metrics_launch(&p, &gt, &dist, &n, 1, thh_acc, thh_sse, out, workspace, workspace_bytes, accumulate, ctx,
                        stream);
}

// out[6 t + k]: the six sums of term t (t < nterm <= 3), overwritten
extern "C" int nvf_metrics3(const float* const* ps, const float* const* gts, const float* const* dists,
                            const int64_t* ns, int nterm, float thh_acc, float thh_sse, float* out, void* workspace,
                            size_t workspace_bytes, NvfStepCtx* ctx, void* stream) {
  return metrics_launch(ps, gts, dists, ns, nterm, thh_acc, thh_sse, out, workspace, workspace_bytes, 0, ctx, stream);
}

// ---------------------------------------------------------------------------
// small element-wise kernels
// ---------------------------------------------------------------------------
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p,
                                   float* __restrict__ dlogit, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float pv = p[i];
    dlogit[i] = dp[i] * ((1.f - pv) * pv);
  }
}

extern "C" int nvf_sigmoid_bwd(const float* dp, const float* p, float* dlogit, int64_t n, void* stream) {
  if (!dp || !p || !dlogit || n <= 0) return NVF_EINVAL;
  sigmoid_bwd_kernel<<<NVF_GRID(n, 256), 256, 0, nvf_stream(stream)>>>(dp, p, dlogit, (long)n);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int nvf_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream) {
  if (!dy || !y || !out || n <= 0) return NVF_EINVAL;
  relu_bwd_kernel<<<NVF_GRID(n, 256), 256, 0, nvf_stream(stream)>>>(dy, y, out, (long)n);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// get_se (utils/loss.py:123-128): out[b,0,:] = ((p > thh) * dist)^2, out[b,1,:] = p
__global__ void se_kernel(const float* __restrict__ p, const float* __restrict__ dist, float thh,
                          float* __restrict__ out, long n, int spatial) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long b = i / spatial;
  int s = (int)(i % spatial);
  float pv = p[i];
  float d = pv > thh ? dist[i] : 0.f;
  out[(2 * b) * spatial + s] = d * d;
  out[(2 * b + 1) * spatial + s] = pv;
}

extern "C" int nvf_squared_error_map(const float* p, const float* dist, float thh, float* out, int batch, int spatial,
                                     void* stream) {
  if (!p || !dist || !out || batch <= 0 || spatial <= 0) return NVF_EINVAL;
  long n = (long)batch * spatial;
  se_kernel<<<NVF_GRID(n, 256), 256, 0, nvf_stream(stream)>>>(p, dist, thh, out, n, spatial);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int d, int h, int w) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int ow = w / 2, oh = h / 2, od = d / 2;
  int ox = i % ow;
  long r = i / ow;
  int oy = r % oh;
  r /= oh;
  int oz = r % od;
  long bc = r / od;
  const float* xb = x + bc * (long)d * h * w;
  float m = -INFINITY;
  for (int dz = 0; dz < 2; ++dz)
    for (int dy = 0; dy < 2; ++dy)
      for (int dxx = 0; dxx < 2; ++dxx)
        m = fmaxf(m, xb[((long)(2 * oz + dz) * h + (2 * oy + dy)) * w + 2 * ox + dxx]);
  y[i] = m;
}

extern "C" int nvf_maxpool2(const float* x, float* y, int batch_channels, int d, int h, int w, void* stream) {
  if (!x || !y || batch_channels <= 0 || d < 2 || h < 2 || w < 2) return NVF_EINVAL;
  long total = (long)batch_channels * (d / 2) * (h / 2) * (w / 2);
  maxpool2_kernel<<<NVF_GRID(total, 256), 256, 0, nvf_stream(stream)>>>(x, y, total, d, h, w);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// Adam, torch defaults (A.7 of SURVEY.md): p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float step_size, float b1, float b2, float eps,
                            float bc2_sqrt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  float mi = m[i] * b1 + gi * (1.f - b1);
  float vi = v[i] * b2 + (gi * gi) * (1.f - b2);
  m[i] = mi;
  v[i] = vi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

extern "C" int nvf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return NVF_EINVAL;
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  adam_kernel<<<NVF_GRID(n, 256), 256, 0, nvf_stream(stream)>>>(p, g, m, v, (long)n, (float)(lr / bc1), beta1, beta2,
                                                                eps, (float)sqrt(bc2));
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// The tail of a training step with every per-step scalar in DEVICE memory, so that it can sit inside a replayed HIP graph
// (and behind the data-parallel all-reduce): Adam with coef[0] = lr / (1 - b1^t), coef[1] = sqrt(1 - b2^t) staged by the
// host (the floats nvf_adam_step computes: identical updates); the epoch's running sums behind NVFPCC.py's log line
// (:190-221, 256-281) and the counters behind its NaN checks (:199-212), read once per epoch instead of synchronising
// every step; and the hand-over to the NEXT step: the last workgroup to finish copies the next row of the caller's
// step schedule (block ids, noise step, rate coefficients, Adam coefficients) over the buffer the step's kernels read,
// so a replayed graph needs no host-to-device copy per step.  An element whose gradient is not finite keeps its
// parameter and moments (it is counted): the parameters at the raise are finite.
__global__ void step_tail_kernel(NvfStepTail a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (i < a.n) {
    const float step_size = a.coef_dev ? a.coef_dev[0] : a.coef0_host, bc2_sqrt = a.coef_dev ? a.coef_dev[1] : a.coef1_host;
    const float gi = a.g[i];
    bad = !(fabsf(gi) <= 3.402823466e38f);          // NaN or +-inf
    if (!bad) {
      const float mi = a.m[i] * a.beta1 + gi * (1.f - a.beta1);
      const float vi = a.v[i] * a.beta2 + (gi * gi) * (1.f - a.beta2);
      a.m[i] = mi;
      a.v[i] = vi;
      const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
      a.p[i] = a.p[i] - step_size * (mi / denom);
    }
  }
  if (a.acc) {
    const unsigned long long any = __ballot(bad);
    if (any && (threadIdx.x & 63) == 0) atomicAdd(a.acc + 6, (float)__popcll(any));   // integer-valued: order-free
  }
  if (!a.acc && !a.sched_rows) return;
  // the last workgroup to arrive does the per-step scalars: every other workgroup has read its coefficients by then
  __shared__ int last;
  bool is_last = gridDim.x == 1;
  if (gridDim.x > 1) {
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.done, 1u) == gridDim.x - 1;
    __syncthreads();
    is_last = last != 0;
  }
  if (!is_last) return;
  if (a.acc && threadIdx.x == 0) {
    int bad_terms = 0;
    for (int t = 0; t < 3; ++t) {
      const float x = a.loss_terms[t];
      a.acc[t] += x;
      bad_terms += !(fabsf(x) <= 3.402823466e38f);
    }
    const float bl = a.lbits[0] * (a.inv_npts_dev ? a.inv_npts_dev[0] : a.inv_npts_host);
    float nb = 0.f;
    for (int l = 0; l < a.nnb; ++l) nb += a.nbits[l];
    nb *= a.nbits_scale;
    a.acc[3] += bl;
    a.acc[4] += nb;
    bad_terms += !(fabsf(bl) <= 3.402823466e38f) + !(fabsf(nb) <= 3.402823466e38f);
    a.acc[5] += (float)bad_terms;
    a.acc[7] += 1.f;
    if (a.counts) {      // per-step ratios, as get_acc_dense returns them (0/0 = NaN included), summed over the epoch
      for (int t = 0; t < 3; ++t) {
        a.acc[8 + 2 * t] += a.counts[6 * t] / a.counts[6 * t + 1];
        a.acc[9 + 2 * t] += a.counts[6 * t + 2] / a.counts[6 * t + 3];
      }
      a.acc[14] += a.counts[4];
      a.acc[15] += a.counts[5];
    }
  }
  if (a.sched_rows) {
    // thread 0's statistics read 1/n_pts from the step buffer (word batch + 1); the copy below overwrites that word from
    // another wave once the per-rank batch exceeds 62: order them (is_last is uniform over the workgroup)
    // ... and the cursor is read ONCE, by thread 0, and broadcast: thread 0 advances it as soon as its own words are
    // copied, so a wave that fetched it late would copy part of the NEXT row (mixed block ids / Adam coefficients once a
    // row spans more than one wave: per-rank batches of 62 and more)
    __shared__ unsigned long long cur_s;
    if (threadIdx.x == 0) cur_s = a.sched_cursor[0];
    __syncthreads();
    const unsigned long long c = cur_s;
    const int64_t* row = a.sched_rows + c * (unsigned long long)a.sched_words;
    for (int w = threadIdx.x; w < a.sched_words; w += blockDim.x) a.sched_buf[w] = row[w];
    if (threadIdx.x == 0) a.sched_cursor[0] = c + 1;
  }
  if (threadIdx.x == 0 && gridDim.x > 1) a.done[0] = 0u;
}

extern "C" int nvf_step_tail(const NvfStepTail* args, void* stream) {
  if (!args) return NVF_EINVAL;
  const NvfStepTail a = *args;
  if (!a.p || !a.g || !a.m || !a.v || a.n <= 0) return NVF_EINVAL;
  if (a.acc && (!a.loss_terms || !a.lbits || !a.nbits || a.nnb <= 0 || a.nnb > 16)) return NVF_EINVAL;
  if (a.sched_rows && (!a.sched_buf || !a.sched_cursor || a.sched_words <= 0)) return NVF_EINVAL;
  if ((a.acc || a.sched_rows) && !a.done) return NVF_EINVAL;
  step_tail_kernel<<<NVF_GRID(a.n, 256), 256, 0, nvf_stream(stream)>>>(a);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// the two floats nvf_step_tail reads from coef_dev for optimiser step `step` (>= 1): what nvf_adam_step passes
extern "C" int nvf_adam_coefficients(float lr, float beta1, float beta2, int step, float* coef_host) {
  if (!coef_host || step < 1) return NVF_EINVAL;
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  coef_host[0] = (float)(lr / bc1);
  coef_host[1] = (float)sqrt(bc2);
  return NVF_OK;
}

// ... for n consecutive steps step0, step0 + 1, ...: coef_host[2 k], coef_host[2 k + 1] (a schedule's rows in one call)
extern "C" int nvf_adam_coefficients_n(float lr, float beta1, float beta2, int step0, int n, float* coef_host) {
  if (!coef_host || step0 < 1 || n < 0) return NVF_EINVAL;
  for (int k = 0; k < n; ++k) {
    const int rc = nvf_adam_coefficients(lr, beta1, beta2, step0 + k, coef_host + 2 * k);
    if (rc != NVF_OK) return rc;
  }
  return NVF_OK;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                   float* __restrict__ dst, int rows, int width) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * width) return;
  int r = i / width, c = i % width;
  dst[i] = src[idx[r] * width + c];
}

__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                        float* __restrict__ dst, int rows, int width) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * width) return;
  int r = i / width, c = i % width;
  dst[idx[r] * width + c] += src[i];
}

extern "C" int nvf_gather_rows(const float* src, const int64_t* idx, float* dst, int rows, int width, void* stream) {
  if (!src || !idx || !dst || rows <= 0 || width <= 0) return NVF_EINVAL;
  gather_rows_kernel<<<NVF_GRID((long)rows * width, 256), 256, 0, nvf_stream(stream)>>>(src, idx, dst, rows, width);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int rows, int width,
                                    void* stream) {
  if (!src || !idx || !dst || rows <= 0 || width <= 0) return NVF_EINVAL;
  scatter_add_rows_kernel<<<NVF_GRID((long)rows * width, 256), 256, 0, nvf_stream(stream)>>>(src, idx, dst, rows,
                                                                                            width);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// up to 6 row gathers that share one index vector (the mini-batch's grids, pyramid and latents) in one launch
struct GatherMulti {
  const float* src[6];
  float* dst[6];
  int32_t width[6];
  int32_t n;
};

__device__ __forceinline__ void gather_rows_multi_body(const GatherMulti& g, const int64_t* __restrict__ idx, int rows,
                                                       int t, int bx, int nbx) {
  if (t >= g.n) return;
  const int width = g.width[t];
  const long total = (long)rows * width;
  const float* src = g.src[t];
  float* dst = g.dst[t];
  if ((width & 3) == 0) {
    const long total4 = total >> 2;
    const int w4 = width >> 2;
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < total4; i += (long)nbx * blockDim.x) {
      const long r = i / w4;
      const int c = (int)(i - r * w4);
      ((float4*)dst)[i] = ((const float4*)src)[idx[r] * w4 + c];
    }
  } else {
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < total; i += (long)nbx * blockDim.x) {
      const long r = i / width;
      dst[i] = src[idx[r] * width + (i - r * width)];
    }
  }
}

__global__ void gather_rows_multi_kernel(GatherMulti g, const int64_t* __restrict__ idx, int rows) {
  gather_rows_multi_body(g, idx, rows, blockIdx.y, blockIdx.x, gridDim.x);
}

static int gather_multi_desc(const float* const* srcs, float* const* dsts, const int* widths, int n, int rows,
                             GatherMulti& g, long& wg) {
  if (!srcs || !dsts || !widths || n <= 0 || n > 6 || rows <= 0) return NVF_EINVAL;
  long biggest = 0;
  for (int t = 0; t < n; ++t) {
    if (!srcs[t] || !dsts[t] || widths[t] <= 0) return NVF_EINVAL;
    g.src[t] = srcs[t]; g.dst[t] = dsts[t]; g.width[t] = widths[t];
    if ((long)rows * widths[t] > biggest) biggest = (long)rows * widths[t];
  }
  g.n = n;
  wg = (biggest / 4 + 255) / 256;
  if (wg > 1024) wg = 1024;
  if (wg < 1) wg = 1;
  return NVF_OK;
}

extern "C" int nvf_gather_rows_multi(const float* const* srcs, float* const* dsts, const int* widths, int n,
                                     const int64_t* idx, int rows, void* stream) {
  if (!idx) return NVF_EINVAL;
  GatherMulti g{};
  long wg = 0;
  const int rc = gather_multi_desc(srcs, dsts, widths, n, rows, g, wg);
  if (rc != NVF_OK) return rc;
  gather_rows_multi_kernel<<<dim3((unsigned)wg, n), 256, 0, nvf_stream(stream)>>>(g, idx, rows);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// The head of a training step in ONE launch: the effective weights of every layer (nvf_prepare_weights), the gather
// of the mini-batch rows (nvf_gather_rows_multi) and the matrix-core weight packings (nvf_pack_mfma_all).  The
// packings do not wait for the prepared layouts: a packed element recomputes its effective weight from the raw
// kernel (same arithmetic, same noise index), so the three parts are independent.  Workgroups [0, 16 nlayers)
// prepare, the next 16 npack pack, the rest gather.  Results are those of the three calls, bit for bit.
__device__ __forceinline__ int layout_to_kernel_index(const NvfLayerDesc& d, int bwd, int si) {
  int i0, i1, t;
  if (d.kind == 0) {                    // conv [cout = i0][cin = i1][k]
    if (!bwd) { i0 = si % d.dim0; t = (si / d.dim0) % d.k3; i1 = si / (d.dim0 * d.k3); }
    else { i1 = si % d.dim1; t = d.k3 - 1 - (si / d.dim1) % d.k3; i0 = si / (d.dim1 * d.k3); }
  } else {                              // convT [cin = i0][cout = i1][k]
    if (!bwd) { i1 = si % d.dim1; t = (si / d.dim1) % d.k3; i0 = si / (d.dim1 * d.k3); }
    else { i0 = si % d.dim0; t = (si / d.dim0) % d.k3; i1 = si / (d.dim0 * d.k3); }
  }
  return (i0 * d.dim1 + i1) * d.k3 + t;
}

struct RateInHead {
  WeightRateBatch b;
  const float* sigma;
  const float* mu;
  float* part;
  float g;
  int32_t nwg;
};

__global__ void step_head_kernel(const NvfLayerDesc* __restrict__ table, int nlayers, int q, uint64_t seed,
                                 uint64_t step, const uint64_t* __restrict__ step_dev, PackJobs pk, GatherMulti g,
                                 const int64_t* __restrict__ idx, int rows, int gwg, int wpl, RateInHead rate) {
  int bid = blockIdx.x;        // wpl workgroups per layer / pack job
  if (bid < rate.nwg) {        // the weight-rate term's partial sums: parameters only, nothing of the mini-batch
    __shared__ float red[16];
    weight_rate_partial_body(rate.b, rate.sigma, rate.mu, rate.part, rate.g, bid, 0, red);
    return;
  }
  bid -= rate.nwg;
  if (bid < wpl * nlayers) { prepare_weights_body(table, q, seed, step, step_dev, bid / wpl, bid % wpl, wpl); return; }
  bid -= wpl * nlayers;
  if (bid < wpl * pk.n) {
    const int job = bid / wpl;
    const NvfLayerDesc d = table[pk.layer[job]];
    const int qq = d.quantised ? q : 0, bwd = pk.bwd[job];
    const uint64_t st = step + (step_dev ? step_dev[0] : 0ull);
    const uint64_t sid = (st << 8) | (uint64_t)d.layer_id;
    pack_mfma_body(pk, job, bid % wpl, wpl, [&](int, int si) {
      return effective_weight(d, layout_to_kernel_index(d, bwd, si), qq, seed, sid);
    });
    return;
  }
  bid -= wpl * pk.n;
  gather_rows_multi_body(g, idx, rows, bid / gwg, bid % gwg, gwg);
}

// pack_*: as nvf_pack_mfma_all without sources -- pack job j packs layout pack_bwd[j] (0 w_fwd, 1 w_bwd) of table row
// pack_layers[j]; npack may be 0.
extern "C" int nvf_step_head(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step,
                             const uint64_t* step_dev, float* const* pack_dsts, const int* pack_kinds,
                             const int* pack_c0s, const int* pack_c1s, const int* pack_layers, const int* pack_bwd,
                             int npack, const float* const* srcs, float* const* dsts, const int* widths, int n,
                             const int64_t* idx, int rows, const NvfRateJob* rate_job, void* stream) {
  if (!table_dev || nlayers <= 0 || !idx || npack < 0) return NVF_EINVAL;
  PackJobs pk{};
  if (npack > 0) {
    if (!pack_layers || !pack_bwd) return NVF_EINVAL;
    const int rc = pack_jobs_desc(nullptr, pack_dsts, pack_kinds, pack_c0s, pack_c1s, npack, pk);
    if (rc != NVF_OK) return rc;
    for (int j = 0; j < npack; ++j) {
      if (pack_layers[j] < 0 || pack_layers[j] >= nlayers) return NVF_EINVAL;
      pk.layer[j] = pack_layers[j]; pk.bwd[j] = pack_bwd[j] ? 1 : 0;
    }
  }
  GatherMulti g{};
  long wg = 0;
  const int rc = gather_multi_desc(srcs, dsts, widths, n, rows, g, wg);
  if (rc != NVF_OK) return rc;
  // workgroups per layer and pack job: 64 (narrow: 10.0 us with 16, 8.0 with 32, 7.2 with 64); 128 for the wide decoder's
  // kernels (up to 64 000 weights each: 30.6 us with 16, 13.7 with 64, 12.1 with 128)
  int wpl = 64;
  for (int j = 0; j < npack; ++j)
    if (pack_c0s && pack_c1s && pack_c0s[j] * pack_c1s[j] >= 256) wpl = 128;
  RateInHead rate{};
  if (rate_job) {
    if (!rate_job->sigma || !rate_job->mu || !rate_job->part) return NVF_EINVAL;
    const int rrc = weight_rate_batch_desc(rate_job->kernel, rate_job->dk, rate_job->n, rate_job->nlayers, rate.b);
    if (rrc != NVF_OK) return rrc;
    rate.sigma = rate_job->sigma; rate.mu = rate_job->mu; rate.part = rate_job->part; rate.g = rate_job->g;
    rate.nwg = rate.b.first_wg[rate.b.nlayers];
  }
  step_head_kernel<<<rate.nwg + wpl * nlayers + wpl * npack + (unsigned)(wg * n), 256, 0, nvf_stream(stream)>>>(
      (const NvfLayerDesc*)table_dev, nlayers, q, seed, step, step_dev, pk, g, idx, rows, (int)wg, wpl, rate);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---- the step head AND the stem's forward (with the latent generator + quantiser) in one launch --------------------------
// The stem's workgroups derive their weights from the raw parameters (effective_weight: the arithmetic of the weight
// preparation, element by element) and fetch their latent rows through idx, so they depend on nothing the rest of the
// launch writes: two latency-bound launches (9.4 + 11.5 us at batch 16) become one.  Workgroups have C0 * 64 threads (the
// stem's); the head's jobs are blockDim-agnostic, the weight-rate partial pass keeps its 256-thread arithmetic.
struct StemRawW {
  NvfLayerDesc lat, up0l, conv0l;
  const float* emb;
  const int64_t* idx;
  uint64_t seed, st;
  int32_t ch, q;
  __device__ __forceinline__ float weight(const NvfLayerDesc& d, int e) const {
    return effective_weight(d, layout_to_kernel_index(d, 0, e), d.quantised ? q : 0, seed, (st << 8) | (uint64_t)d.layer_id);
  }
  // Four consecutive RAW elements i0 .. i0 + 3 (i0 a multiple of 4) per call: two 16-byte loads and ONE Philox block --
  // effective_weight's arithmetic per element (nvf_uniform01(i) = word i & 3 of block i >> 2).  A transposed conv's raw
  // layout is [ci][co][k]: walking it in order is coalesced; the forward layout is a scatter into LDS.
  __device__ __forceinline__ void weight4(const NvfLayerDesc& d, int i0, float (&w)[4]) const {
    const nvf_f4u k4 = *(const nvf_f4u*)(d.kernel + i0), n4 = *(const nvf_f4u*)(d.kernel_init + i0);
    const float kk[4] = {k4.a, k4.b, k4.c, k4.d}, nn[4] = {n4.a, n4.b, n4.c, n4.d};
    const int qq = d.quantised ? q : 0;
    uint32_t r[4] = {0, 0, 0, 0};
    if (qq == 1) nvf_philox(seed, (st << 8) | (uint64_t)d.layer_id, (uint64_t)(i0 >> 2), r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float k = kk[j];
      if (qq == 1) k = k + ((float)(r[j] >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.0625f;
      else if (qq == 2) k = rintf(k * 16.f) / 16.f;
      w[j] = k + nn[j];
    }
  }
  template <int NT, int C0>
  __device__ __forceinline__ void fill_up0(float* s_w0, int nch, int tid) const {
    // raw [ci][co = C0][125]: all of it, in order
    for (int i0 = 4 * tid; i0 < nch * C0 * 125; i0 += 4 * NT) {
      float w[4];
      weight4(up0l, i0, w);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j, k = i % 125, cc = i / 125, co = cc % C0, ci = cc / C0;
        s_w0[(ci * 125 + k) * C0 + co] = w[j];
      }
    }
  }
  template <int NT, int C0, int C1, int COG>
  __device__ __forceinline__ void fill_conv0(float* s_w1, int part, int tid) const {
    // raw [ci = C0][co = C1][125]: per input channel the run of this part's NCG * COG output channels (a multiple of 4
    // elements long and starting at one: 125 * COG * k with COG = 4)
    constexpr int NCG = C0 / 8, RUN = NCG * COG * 125;
    static_assert(RUN % 4 == 0 && (C1 * 125) % 4 == 0, "16-byte pieces: runs start and end on multiples of four elements");
    for (int t = tid; t < C0 * (RUN / 4); t += NT) {
      const int ci = t / (RUN / 4), r0 = 4 * (t - ci * (RUN / 4));
      float w[4];
      weight4(conv0l, (ci * C1 + part * NCG * COG) * 125 + r0, w);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + j, k = r % 125, cl = r / 125, cg = cl / COG, c = cl % COG;
        s_w1[cg * (C0 * 125 * COG) + (ci * 125 + k) * COG + c] = w[j];
      }
    }
  }
  __device__ __forceinline__ float up0_b(int co) const { return up0l.b[co] + up0l.b_init[co]; }
  __device__ __forceinline__ float conv0_b(int co) const { return conv0l.b[co] + conv0l.b_init[co]; }
  __device__ __forceinline__ float lat_w(int i) const { return weight(lat, i); }
  __device__ __forceinline__ float lat_b(int j) const { return lat.b[j] + lat.b_init[j]; }
  __device__ __forceinline__ float e(int b, int i, int sp) const { return emb[((size_t)idx[b] * ch + i) * 8 + sp]; }
};

// (launch bounds ask for four waves per SIMD = two workgroups per CU: the stem's body wants 131 registers, three more than
// let a second 8-wave workgroup in -- the head's ~2000 short workgroups then queued behind 256 slots instead of 512, 65 of
// them held by the stem: 20.0 us for the launch against 15.2 with 128 registers and three spilled values)
template <int C0, int C1, int COG>
__global__ __launch_bounds__(C0 * 64, 4) void step_head_stem_kernel(const NvfLayerDesc* __restrict__ table, int nlayers, int q,
                                                                 uint64_t seed, uint64_t step,
                                                                 const uint64_t* __restrict__ step_dev, PackJobs pk,
                                                                 GatherMulti g, const int64_t* __restrict__ idx, int rows,
                                                                 int gwg, int wpl, RateInHead rate, NvfStemHead sj) {
  __shared__ __attribute__((aligned(16))) float lds[StemFwdLds<C0, C1, COG>::FLOATS];
  constexpr int PARTS = C1 / (COG * (C0 / 8));
  int bid = blockIdx.x;
  const int nstem = 1 + rows * PARTS;        // the latent workgroup (the longest chain) first, then (block, part)
#ifndef NVF_HS_SKIP
#define NVF_HS_SKIP 0                        // tuning builds: 1 = the stem's workgroups do nothing, 2 = the head's (results meaningless)
#endif
  if ((NVF_HS_SKIP & 1) && bid < nstem) return;
  if ((NVF_HS_SKIP & 2) && bid >= nstem) return;
  if (bid < nstem) {
    const uint64_t st = step + (step_dev ? step_dev[0] : 0ull);
    const StemRawW wp{table[sj.lat_row], table[sj.up0_row], table[sj.conv0_row], sj.emb, idx, seed, st, sj.ch, q};
    const StemLatent L{nullptr, nullptr, nullptr, sj.lat_beta_hat, sj.lat_gamma_hat, idx, sj.sigma, sj.mu, sj.h, sj.lat,
                       sj.x_rounded, sj.bits, step_dev, seed, step, sj.mode, rows};
    const int b = bid == 0 ? 0 : (bid - 1) / PARTS, part = bid == 0 ? PARTS : (bid - 1) % PARTS;
    stem_fwd_body<C0, C1, COG, true, true>(nullptr, wp, sj.beta_hat, sj.gamma_hat, sj.a0, sj.h0, sj.y1, sj.ch, L, b, part,
                                           lds);
    return;
  }
  bid -= nstem;
  if (bid < rate.nwg) {        // the weight-rate term's partial sums: parameters only, nothing of the mini-batch
    weight_rate_partial_body(rate.b, rate.sigma, rate.mu, rate.part, rate.g, bid, 0, lds, 256);
    return;
  }
  bid -= rate.nwg;
  if (bid < wpl * nlayers) { prepare_weights_body(table, q, seed, step, step_dev, bid / wpl, bid % wpl, wpl); return; }
  bid -= wpl * nlayers;
  if (bid < wpl * pk.n) {
    const int job = bid / wpl;
    const NvfLayerDesc d = table[pk.layer[job]];
    const int qq = d.quantised ? q : 0, bwd = pk.bwd[job];
    const uint64_t st = step + (step_dev ? step_dev[0] : 0ull);
    const uint64_t sid = (st << 8) | (uint64_t)d.layer_id;
    pack_mfma_body(pk, job, bid % wpl, wpl, [&](int, int si) {
      return effective_weight(d, layout_to_kernel_index(d, bwd, si), qq, seed, sid);
    });
    return;
  }
  bid -= wpl * pk.n;
  gather_rows_multi_body(g, idx, rows, bid / gwg, bid % gwg, gwg);
}

extern "C" int nvf_step_head_stem(const void* table_dev, int nlayers, int q, uint64_t seed, uint64_t step,
                                  const uint64_t* step_dev, float* const* pack_dsts, const int* pack_kinds,
                                  const int* pack_c0s, const int* pack_c1s, const int* pack_layers, const int* pack_bwd,
                                  int npack, const float* const* srcs, float* const* dsts, const int* widths, int n,
                                  const int64_t* idx, int rows, const NvfRateJob* rate_job, const NvfStemHead* stem,
                                  void* stream) {
  if (!table_dev || nlayers <= 0 || !idx || npack < 0 || !stem) return NVF_EINVAL;
  const NvfStemHead sj = *stem;
  const bool narrow = sj.c0 == 8 && sj.c1 == 16, wide = sj.c0 == 16 && sj.c1 == 32;
  if (!(narrow || wide) || sj.ch <= 0 || sj.ch > kStemFwdMaxCh || rows <= 0 || rows > 32) return NVF_EINVAL;
  if (sj.lat_row < 0 || sj.lat_row >= nlayers || sj.up0_row < 0 || sj.up0_row >= nlayers || sj.conv0_row < 0 ||
      sj.conv0_row >= nlayers || (sj.mode != 0 && sj.mode != 1))
    return NVF_EINVAL;
  if (!sj.emb || !sj.lat_beta_hat || !sj.lat_gamma_hat || !sj.sigma || !sj.mu || !sj.beta_hat || !sj.gamma_hat || !sj.h ||
      !sj.lat || !sj.x_rounded || !sj.bits || !sj.a0 || !sj.h0 || !sj.y1)
    return NVF_EINVAL;
  PackJobs pk{};
  if (npack > 0) {
    if (!pack_layers || !pack_bwd) return NVF_EINVAL;
    const int rc = pack_jobs_desc(nullptr, pack_dsts, pack_kinds, pack_c0s, pack_c1s, npack, pk);
    if (rc != NVF_OK) return rc;
    for (int j = 0; j < npack; ++j) {
      if (pack_layers[j] < 0 || pack_layers[j] >= nlayers) return NVF_EINVAL;
      pk.layer[j] = pack_layers[j]; pk.bwd[j] = pack_bwd[j] ? 1 : 0;
    }
  }
  GatherMulti g{};
  long wg = 0;
  const int rc = gather_multi_desc(srcs, dsts, widths, n, rows, g, wg);
  if (rc != NVF_OK) return rc;
  // 512-thread (wide decoder: 1024-thread) workgroups: a half (quarter) as many as nvf_step_head's for the same threads
  const int tscale = (narrow ? 2 : 4) * nvf_tune_int("NVF_HS_GSCALE", 1);
  wg = (wg + tscale - 1) / tscale;
  // (nvf_step_head: 64 workgroups of 256 threads per layer; 128 for the wide decoder's kernels.  Tuning builds: NVF_HS_WPL)
  const int wpl = nvf_tune_int("NVF_HS_WPL", 32) > 0 ? nvf_tune_int("NVF_HS_WPL", 32) : 32;
  RateInHead rate{};
  if (rate_job) {
    if (!rate_job->sigma || !rate_job->mu || !rate_job->part) return NVF_EINVAL;
    const int rrc = weight_rate_batch_desc(rate_job->kernel, rate_job->dk, rate_job->n, rate_job->nlayers, rate.b);
    if (rrc != NVF_OK) return rrc;
    rate.sigma = rate_job->sigma; rate.mu = rate_job->mu; rate.part = rate_job->part; rate.g = rate_job->g;
    rate.nwg = rate.b.first_wg[rate.b.nlayers];
  }
  const int parts = narrow ? 16 / 4 : 32 / (2 * 2);          // conv0 channel parts per block (COG 4; wide: 2 x 2 groups)
  const unsigned grid = 1 + rows * parts + rate.nwg + wpl * nlayers + wpl * npack + (unsigned)(wg * n);
  if (narrow)
    step_head_stem_kernel<8, 16, 4><<<grid, 512, 0, nvf_stream(stream)>>>(
        (const NvfLayerDesc*)table_dev, nlayers, q, seed, step, step_dev, pk, g, idx, rows, (int)wg, wpl, rate, sj);
  else
    step_head_stem_kernel<16, 32, 2><<<grid, 1024, 0, nvf_stream(stream)>>>(
        (const NvfLayerDesc*)table_dev, nlayers, q, seed, step, step_dev, pk, g, idx, rows, (int)wg, wpl, rate, sj);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void uniform_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t stream_id) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = nvf_uniform01(seed, stream_id, (uint64_t)i);
}

extern "C" int nvf_uniform(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  if (!out || n <= 0) return NVF_EINVAL;
  uniform_kernel<<<NVF_GRID(n, 256), 256, 0, nvf_stream(stream)>>>(out, (long)n, seed, stream_id);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// occupancy thresholding + compaction (NVFPCC.py:520, 532-535, 631-634)
// one workgroup per block; raster order kept (ballot prefix inside a wave, wave offsets
// through LDS, chunks walked in order)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void threshold_count_kernel(const float* __restrict__ p, float thh,
                                                               int32_t* __restrict__ counts, int voxels) {
  __shared__ float red[16];
  const float* pb = p + (size_t)blockIdx.x * voxels;
  float c = 0.f;
  for (int i = threadIdx.x; i < voxels; i += blockDim.x) c += pb[i] > thh ? 1.f : 0.f;
  float t = nvf_block_sum(c, red);  // exact: counts <= 2^24
  if (threadIdx.x == 0) counts[blockIdx.x] = (int32_t)t;
}

extern "C" int nvf_threshold_count(const float* p, float thh, int32_t* counts, int batch, int voxels, void* stream) {
  if (!p || !counts || batch <= 0 || voxels <= 0 || voxels > (1 << 24)) return NVF_EINVAL;
  threshold_count_kernel<<<batch, 1024, 0, nvf_stream(stream)>>>(p, thh, counts, voxels);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ __launch_bounds__(1024) void threshold_compact_kernel(const float* __restrict__ p, float thh,
                                                                 const int32_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ origins,
                                                                 int32_t* __restrict__ coords, int dim) {
  __shared__ int wave_cnt[16];
  __shared__ int running;
  const int b = blockIdx.x, voxels = dim * dim * dim;
  const float* pb = p + (size_t)b * voxels;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int oz = origins ? origins[3 * b] : 0, oy = origins ? origins[3 * b + 1] : 0, ox = origins ? origins[3 * b + 2] : 0;
  if (threadIdx.x == 0) running = offsets[b];
  __syncthreads();
  for (int base = 0; base < voxels; base += blockDim.x) {
    const int i = base + threadIdx.x;
    const bool hit = i < voxels && pb[i] > thh;
    const unsigned long long mask = __ballot(hit);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(mask);
    __syncthreads();
    int off = running;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    if (hit) {
      const int z = i / (dim * dim), y = (i / dim) % dim, x = i % dim;
      int32_t* o = coords + (size_t)(off + before) * 3;
      o[0] = oz + z;
      o[1] = oy + y;
      o[2] = ox + x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < nw; ++k) t += wave_cnt[k];
      running += t;
    }
    __syncthreads();
  }
}

extern "C" int nvf_threshold_compact(const float* p, float thh, const int32_t* offsets, const int32_t* origins,
                                     int32_t* coords, int batch, int dim, void* stream) {
  if (!p || !offsets || !coords || batch <= 0 || dim <= 0 || dim > 256) return NVF_EINVAL;
  threshold_compact_kernel<<<batch, 1024, 0, nvf_stream(stream)>>>(p, thh, offsets, origins, coords, dim);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
