// Fused "stem" of the NVF decoder for chanstr c0 = 8, c1 = 16 (latent channels ch <= 8):
//
//   forward :  x0 [ch,2^3] --up0 (convT k5 s2 p2)--> a0 [8,4^3] --IGDN--> h0 --conv0 (convT k5 s2 p2)+ReLU--> y1 [16,8^3]
//   backward:  g1 = dL/d(conv0 pre-activation) --> dh0 --> IGDN backward (da0, d beta, d gamma) --> dx0,
//              plus up0's weight gradient
//
// These layers are < 1 % of the step's FLOPs but, as separate launches over a
// batch of 16 blocks, each is a latency chain on a handful of CUs.  Here one workgroup owns one block, keeps
// every intermediate in LDS, and walks the reference's operator sequence (utils/network.py:4759-4760,
// gdn_3d.py:137-159) in one launch.  Accumulation orders equal those of the per-layer kernels
// (conv_direct.hip), so the forward is bit-identical to the unfused path.
#include "nvf_common.h"

#define NVF_PEDESTAL 1.4551915228366852e-11f
#define NVF_BETA_BOUND 1.0000072759311445e-03f
#define NVF_GAMMA_BOUND 3.814697265625e-06f

namespace {
constexpr int C0 = 8, C1 = 16, MAXCH = 8;

__device__ __forceinline__ float st_beta(float bh) {
  float m = fmaxf(bh, NVF_BETA_BOUND);
  return m * m - NVF_PEDESTAL;
}
__device__ __forceinline__ float st_gamma(float gh) {
  float m = fmaxf(gh, NVF_GAMMA_BOUND);
  return m * m - NVF_PEDESTAL;
}
}  // namespace

__global__ __launch_bounds__(512) void stem_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w0,
                                                       const float* __restrict__ b0,
                                                       const float* __restrict__ beta_hat,
                                                       const float* __restrict__ gamma_hat,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       float* __restrict__ a0, float* __restrict__ h0,
                                                       float* __restrict__ y1, int ch) {
  __shared__ float s_x[MAXCH * 8];
  __shared__ float s_a[C0 * 64];
  __shared__ float s_h[C0 * 216];     // h0 with a one-voxel zero halo: [c][6][6][6], index i + 1
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < ch * 8) s_x[tid] = x0[(size_t)b * ch * 8 + tid];
  for (int e = tid; e < C0 * 216; e += 512) s_h[e] = 0.f;
  __syncthreads();
  const int c = tid >> 6, v = tid & 63, oz = v >> 4, oy = (v >> 2) & 3, ox = v & 3;
  {  // up0: a0[co = c, o] = b0 + sum_ci sum_{k : o + 2 - k = 2 i} x0[ci, i] w0[ci][k][co]
    float acc = 0.f;
    for (int ci = 0; ci < ch; ++ci)
      for (int kz = 0; kz < 5; ++kz) {
        const int uz = oz + 2 - kz;
        if (uz < 0 || (uz & 1) || (uz >> 1) >= 2) continue;
        for (int ky = 0; ky < 5; ++ky) {
          const int uy = oy + 2 - ky;
          if (uy < 0 || (uy & 1) || (uy >> 1) >= 2) continue;
          for (int kx = 0; kx < 5; ++kx) {
            const int ux = ox + 2 - kx;
            if (ux < 0 || (ux & 1) || (ux >> 1) >= 2) continue;
            acc = fmaf(s_x[ci * 8 + (uz >> 1) * 4 + (uy >> 1) * 2 + (ux >> 1)],
                       w0[(ci * 125 + (kz * 5 + ky) * 5 + kx) * C0 + c], acc);
          }
        }
      }
    const float val = acc + b0[c];
    s_a[tid] = val;
    a0[(size_t)b * C0 * 64 + tid] = val;
  }
  __syncthreads();
  {  // IGDN: h0 = a0 * sqrt(beta_c + sum_j gamma_cj a0_j^2)
    float nrm = st_beta(beta_hat[c]);
    for (int j = 0; j < C0; ++j) {
      const float xj = s_a[j * 64 + v];
      nrm = fmaf(st_gamma(gamma_hat[c * C0 + j]), xj * xj, nrm);
    }
    const float hv = s_a[tid] * sqrtf(nrm);
    h0[(size_t)b * C0 * 64 + tid] = hv;
    s_h[c * 216 + ((oz + 1) * 6 + (oy + 1)) * 6 + ox + 1] = hv;
  }
  __syncthreads();
  {  // conv0: one wave per output parity class, one lane per cell, all 16 output channels in registers
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ez = wv >> 2, ey = (wv >> 1) & 1, ex = wv & 1;
    const int mz = (v >> 4) + 1, my = ((v >> 2) & 3) + 1, mx = (v & 3) + 1;   // cells 1..4 (pad 2)
    float acc[C1];
#pragma unroll
    for (int co = 0; co < C1; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < C0; ++ci)
      for (int jz = 0; jz < 3 - ez; ++jz)
        for (int jy = 0; jy < 3 - ey; ++jy)
          for (int jx = 0; jx < 3 - ex; ++jx) {
            const float hv = s_h[ci * 216 + ((mz - jz + 1) * 6 + (my - jy + 1)) * 6 + (mx - jx + 1)];
            const float* wr = w1 + (size_t)(ci * 125 + ((ez + 2 * jz) * 5 + (ey + 2 * jy)) * 5 + ex + 2 * jx) * C1;
#pragma unroll
            for (int co = 0; co < C1; ++co) acc[co] = fmaf(hv, wr[co], acc[co]);
          }
    const int qz = 2 * mz + ez - 2, qy = 2 * my + ey - 2, qx = 2 * mx + ex - 2;   // in [0, 8)
#pragma unroll
    for (int co = 0; co < C1; ++co)
      y1[((size_t)b * C1 + co) * 512 + (qz * 8 + qy) * 8 + qx] = fmaxf(acc[co] + b1[co], 0.f);
  }
}

extern "C" int nvf_stem_fwd(const float* x0, const float* up0_w_fwd, const float* up0_b, const float* beta_hat,
                            const float* gamma_hat, const float* conv0_w_fwd, const float* conv0_b, float* a0,
                            float* h0, float* y1, int batch, int ch, int c0, int c1, void* stream) {
  if (!x0 || !up0_w_fwd || !up0_b || !beta_hat || !gamma_hat || !conv0_w_fwd || !conv0_b || !a0 || !h0 || !y1)
    return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || c0 != C0 || c1 != C1) return NVF_EINVAL;
  stem_fwd_kernel<<<batch, 512, 0, nvf_stream(stream)>>>(x0, up0_w_fwd, up0_b, beta_hat, gamma_hat, conv0_w_fwd,
                                                         conv0_b, a0, h0, y1, ch);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
static const int kStemMaxSlabs = 256;
static const int kStemNcol = C0 + C0 * C0;          // IGDN parameter partials
static const int kStemWMax = MAXCH * C0 * 125;      // up0 weight-gradient slab

__global__ __launch_bounds__(512) void stem_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ x0,
                                                       const float* __restrict__ a0,
                                                       const float* __restrict__ w1b /* [co16][125][ci8] */,
                                                       const float* __restrict__ w0b /* [co8][125][ch] */,
                                                       const float* __restrict__ beta_hat,
                                                       const float* __restrict__ gamma_hat, float* __restrict__ da0,
                                                       float* __restrict__ dx0, float* __restrict__ slab_gdn,
                                                       float* __restrict__ slab_w, int batch, int ch, int want_w) {
  __shared__ float s_g[C1 * 1331];      // g1 with a two-voxel zero halo: [co][11][11][11], index q + 2
  __shared__ float s_part[8][512];      // per-wave partial dh0
  __shared__ float s_dh[512], s_a[512], s_n[512], s_t[512];
  __shared__ float s_da[C0 * 343];      // da0 with a two-voxel halo: [co][7][7][7], index q + 2
  __shared__ float s_x[MAXCH * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tid >> 6, v = lane, iz = v >> 4, iy = (v >> 2) & 3, ix = v & 3;
  for (int e = tid; e < C1 * 1331; e += 512) s_g[e] = 0.f;
  for (int e = tid; e < C0 * 343; e += 512) s_da[e] = 0.f;
  float own_gdn = 0.f;                  // thread p < 72 owns IGDN partial p
  float own_w[(kStemWMax + 511) / 512]; // up0 weight-gradient outputs j = tid + 512 r
#pragma unroll
  for (int r = 0; r < (kStemWMax + 511) / 512; ++r) own_w[r] = 0.f;
  const int jtotal = ch * C0 * 125;
  __syncthreads();

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    // ---- stage g1[b] (interior of the padded tile), a0[b], x0[b]
    for (int e = tid; e < C1 * 512; e += 512) {
      const int co = e >> 9, q = e & 511;
      s_g[co * 1331 + (((q >> 6) + 2) * 11 + ((q >> 3) & 7) + 2) * 11 + (q & 7) + 2] = g1[(size_t)b * C1 * 512 + e];
    }
    s_a[tid] = a0[(size_t)b * C0 * 64 + tid];
    if (tid < ch * 8) s_x[tid] = x0[(size_t)b * ch * 8 + tid];
    __syncthreads();
    // ---- conv0 backward-data: dh0[ci, i] = sum_co sum_k g1[co, 2 i - 2 + k] w1[ci][co][k]; wave wv takes two co
    {
      float acc[C0];
#pragma unroll
      for (int ci = 0; ci < C0; ++ci) acc[ci] = 0.f;
      for (int cc = 0; cc < 2; ++cc) {
        const int co = 2 * wv + cc;
        const float* gp = s_g + co * 1331 + ((2 * iz) * 11 + 2 * iy) * 11 + 2 * ix;   // q + 2 = 2 i + k
#pragma unroll 1
        for (int kz = 0; kz < 5; ++kz)
#pragma unroll 1
          for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
              const float gv = gp[(kz * 11 + ky) * 11 + kx];
              const float* wr = w1b + (size_t)(co * 125 + (kz * 5 + ky) * 5 + kx) * C0;    // wave-uniform
#pragma unroll
              for (int ci = 0; ci < C0; ++ci) acc[ci] = fmaf(gv, wr[ci], acc[ci]);
            }
      }
#pragma unroll
      for (int ci = 0; ci < C0; ++ci) s_part[wv][ci * 64 + lane] = acc[ci];
    }
    __syncthreads();
    {
      float dh = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) dh += s_part[w][tid];
      s_dh[tid] = dh;
      // IGDN forward quantities of this voxel/channel: n_c, t_c = dh_c a_c / n_c
      float nrm = st_beta(beta_hat[c]);
      for (int j = 0; j < C0; ++j) {
        const float xj = s_a[j * 64 + v];
        nrm = fmaf(st_gamma(gamma_hat[c * C0 + j]), xj * xj, nrm);
      }
      nrm = sqrtf(nrm);
      s_n[tid] = nrm;
      s_t[tid] = dh * s_a[tid] / nrm;
    }
    __syncthreads();
    {  // da0_i = dh_i n_i + a_i sum_c t_c gamma_ci  (i = this thread's channel)
      float mix = 0.f;
      for (int cc = 0; cc < C0; ++cc) mix = fmaf(s_t[cc * 64 + v], st_gamma(gamma_hat[cc * C0 + c]), mix);
      const float d = s_dh[tid] * s_n[tid] + s_a[tid] * mix;
      da0[(size_t)b * C0 * 64 + tid] = d;
      s_da[c * 343 + ((iz + 2) * 7 + iy + 2) * 7 + ix + 2] = d;
      if (want_w && tid < kStemNcol) {   // parameter partials: p < 8: d beta_p ; else d gamma_{cc,j}
        float sum = 0.f;
        if (tid < C0) {
          for (int k = 0; k < 64; ++k) sum += s_t[tid * 64 + k];
        } else {
          const int cc = (tid - C0) / C0, j = (tid - C0) % C0;
          for (int k = 0; k < 64; ++k) {
            const float xj = s_a[j * 64 + k];
            sum = fmaf(s_t[cc * 64 + k], xj * xj, sum);
          }
        }
        own_gdn += 0.5f * sum;
      }
    }
    __syncthreads();
    // ---- up0 backward-data: dx0[ci, i] = sum_co sum_k da0[co, 2 i - 2 + k] w0[ci][co][k]; 8 lanes (co) per output
    if (tid < ch * 64) {
      const int out = tid >> 3, co = tid & 7, ci = out >> 3, i = out & 7;
      const int jz = i >> 2, jy = (i >> 1) & 1, jx = i & 1;
      const float* dp = s_da + co * 343 + ((2 * jz) * 7 + 2 * jy) * 7 + 2 * jx;
      float acc = 0.f;
#pragma unroll 1
      for (int kz = 0; kz < 5; ++kz)
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx)
            acc = fmaf(dp[(kz * 7 + ky) * 7 + kx], w0b[(size_t)(co * 125 + (kz * 5 + ky) * 5 + kx) * ch + ci], acc);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (co == 0) dx0[(size_t)b * ch * 8 + out] = acc;
    }
    // ---- up0 weight gradient: dW0[ci][co][k] += sum_i x0[ci, i] da0[co, 2 i - 2 + k]
    if (want_w) {
#pragma unroll
      for (int r = 0; r < (kStemWMax + 511) / 512; ++r) {
        const int j = tid + 512 * r;
        if (j < jtotal) {
          const int k = j % 125, co = (j / 125) % C0, ci = j / (125 * C0);
          const int kz = k / 25, ky = (k / 5) % 5, kx = k % 5;
          const float* dp = s_da + co * 343 + (kz * 7 + ky) * 7 + kx;
          float acc = own_w[r];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc = fmaf(s_x[ci * 8 + i], dp[((2 * (i >> 2)) * 7 + 2 * ((i >> 1) & 1)) * 7 + 2 * (i & 1)], acc);
          own_w[r] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (want_w) {
    if (tid < kStemNcol) slab_gdn[(size_t)blockIdx.x * kStemNcol + tid] = own_gdn;
#pragma unroll
    for (int r = 0; r < (kStemWMax + 511) / 512; ++r) {
      const int j = tid + 512 * r;
      if (j < jtotal) slab_w[(size_t)blockIdx.x * jtotal + j] = own_w[r];
    }
  }
}

// IGDN parameter gradients from the slabs: fixed-order sum, re-parametrisation chain rule, LowerBound rule
__global__ void stem_gdn_final(const float* __restrict__ slabs, const float* __restrict__ beta_hat,
                               const float* __restrict__ gamma_hat, float* __restrict__ dbeta_hat,
                               float* __restrict__ dgamma_hat, int nslab) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= kStemNcol) return;
  float s = 0.f;
  for (int g = 0; g < nslab; ++g) s += slabs[(size_t)g * kStemNcol + p];
  if (p < C0) {
    const float h = beta_hat[p];
    const float g = s * 2.f * fmaxf(h, NVF_BETA_BOUND);
    dbeta_hat[p] = (h >= NVF_BETA_BOUND || g < 0.f) ? g : 0.f;
  } else {
    const float h = gamma_hat[p - C0];
    const float g = s * 2.f * fmaxf(h, NVF_GAMMA_BOUND);
    dgamma_hat[p - C0] = (h >= NVF_GAMMA_BOUND || g < 0.f) ? g : 0.f;
  }
}

__global__ void stem_w_final(const float* __restrict__ slabs, float* __restrict__ dw, int nslab, int jtotal) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= jtotal) return;
  float s = 0.f;
  for (int g = 0; g < nslab; ++g) s += slabs[(size_t)g * jtotal + j];
  dw[j] = s;
}

extern "C" size_t nvf_stem_bwd_workspace(int ch) {
  return (size_t)kStemMaxSlabs * (kStemNcol + (size_t)ch * C0 * 125) * sizeof(float);
}

extern "C" int nvf_stem_bwd(const float* g1, const float* x0, const float* a0, const float* conv0_w_bwd,
                            const float* up0_w_bwd, const float* beta_hat, const float* gamma_hat, float* da0,
                            float* dx0, float* dbeta_hat, float* dgamma_hat, float* dw_up0, void* workspace,
                            size_t workspace_bytes, int batch, int ch, int c0, int c1, void* stream) {
  if (!g1 || !x0 || !a0 || !conv0_w_bwd || !up0_w_bwd || !beta_hat || !gamma_hat || !da0 || !dx0) return NVF_EINVAL;
  if (batch <= 0 || ch <= 0 || ch > MAXCH || c0 != C0 || c1 != C1) return NVF_EINVAL;
  const int want_w = dbeta_hat && dgamma_hat && dw_up0;
  if (want_w && (!workspace || workspace_bytes < nvf_stem_bwd_workspace(ch))) return NVF_EWORKSPACE;
  const int nslab = batch < kStemMaxSlabs ? batch : kStemMaxSlabs;
  float* slab_gdn = (float*)workspace;
  float* slab_w = slab_gdn ? slab_gdn + (size_t)kStemMaxSlabs * kStemNcol : nullptr;
  hipStream_t s = nvf_stream(stream);
  stem_bwd_kernel<<<nslab, 512, 0, s>>>(g1, x0, a0, conv0_w_bwd, up0_w_bwd, beta_hat, gamma_hat, da0, dx0, slab_gdn,
                                        slab_w, batch, ch, want_w);
  if (want_w) {
    stem_gdn_final<<<(kStemNcol + 63) / 64, 64, 0, s>>>(slab_gdn, beta_hat, gamma_hat, dbeta_hat, dgamma_hat, nslab);
    const int jtotal = ch * C0 * 125;
    stem_w_final<<<(jtotal + 255) / 256, 256, 0, s>>>(slab_w, dw_up0, nslab, jtotal);
  }
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
