// Device bodies of the stem's backward (stem.hip, top comment) -- shared by the stand-alone kernels of stem.hip and by the
// five-gradient launch (wgrad.hip), which carries them as its FIRST workgroups: the stem's backward needs nothing but
// g1 (like conv0's weight gradient in that launch) and feeds nothing but the latent tail, so as two launches of its own
// it was 16 us of latency chains on the critical path in front of a 69 us launch it does not feed.
//
// Inside one launch the hand-over between workgroups (conv0's backward-data partials -> per-block IGDN / up0 stage -> latent
// tail) cannot use a launch boundary.  The L2s of the eight XCDs are not coherent for ordinary accesses, and an agent-scope
// release fence writes a whole XCD's L2 back (measured ~20 us with the launch's other stores in flight), so instead
//   * every word that crosses workgroups is written and read with DEVICE-SCOPE accesses (relaxed agent-scope atomics: sc1
//     stores go through to memory, sc1 loads never hit a stale line) -- 36 KB per step in all;
//   * a producer waits for its stores to be acknowledged (s_waitcnt vmcnt(0)), its workgroup meets at a barrier, and one
//     thread bumps an agent-scope arrival counter; the consumer's thread 0 polls it, then a barrier releases its workgroup.
// Forward progress: producers have the LOWEST workgroup ids of the launch and wait for nothing; every consumer has a higher
// id than its producers, and the host only uses this form while all of them fit in the first dispatch round (ids < 512,
// two 256-thread workgroups per CU), so a consumer never occupies a slot a producer still needs.
#pragma once
#include "nvf_common.h"

constexpr int kStemMaxCh = 8;
constexpr int kStemFlagStride = 64;      // words between two arrival counters: one 256-byte line each (a shared line made the
                                         // producers' adds queue behind the consumers' polls: heads.hip measured 52 vs 30 us)
constexpr int kStemCoopMaxBatch = 32;    // 9 workgroups per block + the tail inside the first dispatch round (< 512)

__device__ __forceinline__ void nvf_store_dev(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float nvf_load_dev(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread of the workgroup calls it after its last device-scope store
__device__ __forceinline__ void nvf_coop_signal(unsigned* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread of the workgroup calls it before its first device-scope load; the only consumer resets the counter
__device__ __forceinline__ void nvf_coop_wait(unsigned* counter, unsigned target) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next step's launch
  }
  __syncthreads();
}

// What the cooperative (in-launch) form adds to the stand-alone kernels' arguments
struct StemCoop {
  unsigned* dh_done;     // [batch] arrivals of the (block, channel pair) workgroups, zero between launches
  unsigned* stem_done;   // [1] arrivals of the per-block workgroups
  float* bias_slab;      // [batch][C0] channel sums of da0 (up0's bias gradient: one slab per block for the slab reduction)
};

namespace stem_detail {
__device__ __forceinline__ float beta_of(float bh) {
  float m = fmaxf(bh, NVF_BETA_BOUND);
  return m * m - NVF_PEDESTAL;
}
__device__ __forceinline__ float gamma_of(float gh) {
  float m = fmaxf(gh, NVF_GAMMA_BOUND);
  return m * m - NVF_PEDESTAL;
}
// dst[e] = src[e] for e < n with U loads of a thread in flight before the first store
template <int NT, int U>
__device__ __forceinline__ void copy(float* dst, const float* __restrict__ src, int n, int tid) {
#pragma unroll 1
  for (int e0 = tid; e0 < n; e0 += NT * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      v[u] = e < n ? src[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      if (e < n) dst[e] = v[u];
    }
  }
}
// dst[e] = gen(e) for e < n, U values of a thread in flight before the first store
template <int NT, int U, class Gen>
__device__ __forceinline__ void copy_from(float* dst, int n, int tid, Gen gen) {
#pragma unroll 1
  for (int e0 = tid; e0 < n; e0 += NT * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      v[u] = e < n ? gen(e) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * NT;
      if (e < n) dst[e] = v[u];
    }
  }
}
}  // namespace stem_detail

constexpr int stem_ncol_of(int c0) { return c0 + c0 * c0; }          // IGDN parameter partials per slab

// ---- conv0 backward-data, split over (block b, output-channel pair cp): part[b][cp][ci][i] = sum over the pair's two co
// and all 125 taps of g1[co, 2 i - 2 + k] w1[ci][co][k].  C0 / 2 waves (C0 * 32 threads): wave = input-channel pair,
// lane = position i.  The pair's gradients (zero-padded) and weights sit in LDS; each lane runs the fmaf chain (cc, kz, ky,
// kx ascending), the partials are added in ascending pair order by stem_bwd_body.
// WG0: the workgroup also leaves its two output channels' share of conv0's weight gradient for this block,
//   slab0[b][ci][co][k] = sum_i h0[b, ci, i] g1[b, co, 2 i - 2 + k]   (i ascending; the caller's slab reduction adds the blocks).
// COOP: the partials leave with device-scope stores and the workgroup signals coop.dh_done[b].
template <int C0, int C1>
struct StemDhLds {
  static constexpr int G = 2 * 1331, W = 2 * 125 * C0, FLOATS = G + W;
};

template <int C0, int C1, bool WG0, bool COOP>
__device__ __forceinline__ void stem_bwd_dh_body(const float* __restrict__ g1, const float* __restrict__ w1b /* [co][125][ci] */,
                                                 float* __restrict__ part, const float* __restrict__ h0,
                                                 float* __restrict__ slab0, int b, int cp, float* lds, const StemCoop& coop) {
  constexpr int NT = C0 * 32;
  float* s_g = lds;                                   // [cc][11][11][11], index q + 2
  float* s_w = lds + StemDhLds<C0, C1>::G;            // [cc][k][ci]   (G is even: 8-byte aligned for the float2 reads)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < 2 * 1331; e += NT) s_g[e] = 0.f;
  stem_detail::copy<NT, 8>(s_w, w1b + (size_t)cp * 2 * 125 * C0, 2 * 125 * C0, tid);
  __syncthreads();
  for (int e = tid; e < 2 * 512; e += NT) {
    const int cc = e >> 9, q = e & 511;
    s_g[cc * 1331 + (((q >> 6) + 2) * 11 + ((q >> 3) & 7) + 2) * 11 + (q & 7) + 2] =
        g1[((size_t)b * C1 + 2 * cp) * 512 + e];
  }
  __syncthreads();
  const int iz = lane >> 4, iy = (lane >> 2) & 3, ix = lane & 3;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 1
  for (int cc = 0; cc < 2; ++cc) {
    const float* gp = s_g + cc * 1331 + ((2 * iz) * 11 + 2 * iy) * 11 + 2 * ix;   // q + 2 = 2 i + k
    const float* wp = s_w + cc * 125 * C0 + 2 * wv;
#pragma unroll 1
    for (int kz = 0; kz < 5; ++kz)
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float gv = gp[(kz * 11 + ky) * 11 + kx];
          const float2 w = *(const float2*)(wp + ((kz * 5 + ky) * 5 + kx) * C0);
          acc0 = fmaf(gv, w.x, acc0);
          acc1 = fmaf(gv, w.y, acc1);
        }
  }
  float* o = part + (((size_t)b * (C1 / 2) + cp) * C0 + 2 * wv) * 64 + lane;
  if (COOP) {
    nvf_store_dev(o, acc0);
    nvf_store_dev(o + 64, acc1);
  } else {
    o[0] = acc0;
    o[64] = acc1;
  }
  if (WG0) {
    __syncthreads();                                   // the weights are no longer read: their LDS holds h0[b] now
    float* s_h = s_w;
    for (int e = tid; e < C0 * 64; e += NT) s_h[e] = h0[(size_t)b * C0 * 64 + e];
    __syncthreads();
    // thread = (input channel, kz, ky): ten sums (two output channels x five kx) share every h0 read
    for (int jj = tid; jj < C0 * 25; jj += NT) {
      const int ci = jj / 25, r = jj % 25, kz = r / 5, ky = r % 5;
      const float* gp = s_g + (kz * 11 + ky) * 11;
      const float* hp = s_h + ci * 64;
      float a[2][5];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) a[cc][kx] = 0.f;
#pragma unroll 4
      for (int i = 0; i < 64; ++i) {
        const float hv = hp[i];
        const int base = ((2 * (i >> 4)) * 11 + 2 * ((i >> 2) & 3)) * 11 + 2 * (i & 3);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) a[cc][kx] = fmaf(hv, gp[cc * 1331 + base + kx], a[cc][kx]);
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
          slab0[((size_t)b * C0 + ci) * C1 * 125 + (2 * cp + cc) * 125 + r * 5 + kx] = a[cc][kx];
    }
  }
  if (COOP) nvf_coop_signal(coop.dh_done + b * kStemFlagStride);
}

// ---- per-block stage: dh0 = sum of the channel-pair partials -> IGDN backward (da0, slabs of d beta / d gamma) -> up0
// backward-data (dx0) and up0's weight-gradient slab.  Workgroup `wg` of `nwg` takes blocks wg, wg + nwg, ...
// NT threads: C0 * 64 in the stand-alone kernel (one (channel, voxel) element per thread); the cooperative form runs with
// the carrier launch's 256 threads, every thread then walks its elements t = tid, tid + NT, ... -- per element the same
// arithmetic in the same order, so the two forms give the same bits.
template <int C0>
struct StemBwdLds {
  static constexpr int LS = 65;        // row stride of the [channel][64] tiles (one column of every row: 64 banks, not one)
  static constexpr int DH = 0, A = DH + C0 * 64, N = A + C0 * LS, T = N + C0 * 64, DA = T + C0 * LS,
                       X = DA + C0 * 343, W0 = X + kStemMaxCh * 8, BET = W0 + C0 * 125 * kStemMaxCh, GAM = BET + C0,
                       FLOATS = GAM + C0 * C0;
};

template <int C0, int C1, int NT, bool COOP>
__device__ __forceinline__ void stem_bwd_body(const float* __restrict__ part, const float* __restrict__ x0,
                                              const float* __restrict__ a0,
                                              const float* __restrict__ w0b /* [co][125][ch] */,
                                              const float* __restrict__ beta_hat, const float* __restrict__ gamma_hat,
                                              float* __restrict__ da0, float* __restrict__ dx0,
                                              float* __restrict__ slab_gdn, float* __restrict__ slab_w, int batch, int ch,
                                              int want_w, int wg, int nwg, float* lds, const StemCoop& coop) {
  using L = StemBwdLds<C0>;
  constexpr int NE = C0 * 64, EPT = (NE + NT - 1) / NT;                   // elements, elements per thread
  constexpr int NCOL = stem_ncol_of(C0), NPAIR = (C0 * 125 + NT - 1) / NT;   // (co, k) pairs per thread
  constexpr int LS = L::LS, MAXCH = kStemMaxCh;
  static_assert(NE % NT == 0 && NCOL <= NT, "whole passes over the elements");
  float *s_dh = lds + L::DH, *s_a = lds + L::A, *s_n = lds + L::N, *s_t = lds + L::T;
  float* s_da = lds + L::DA;            // da0 with a two-voxel halo: [co][7][7][7], index q + 2
  float *s_x = lds + L::X, *s_w0 = lds + L::W0, *s_bet = lds + L::BET, *s_gam = lds + L::GAM;
  const int tid = threadIdx.x;
  for (int e = tid; e < C0 * 343; e += NT) s_da[e] = 0.f;
  stem_detail::copy<NT, 16>(s_w0, w0b, C0 * 125 * ch, tid);
  for (int e = tid; e < C0 * C0; e += NT) s_gam[e] = stem_detail::gamma_of(gamma_hat[e]);
  if (tid < C0) s_bet[tid] = stem_detail::beta_of(beta_hat[tid]);
  float own_gdn = 0.f;                  // thread p < NCOL owns IGDN partial p
  float own_w[NPAIR][MAXCH];            // up0 weight gradient: thread owns pairs p = tid + NT r = (co, k), all ch inputs
#pragma unroll
  for (int r = 0; r < NPAIR; ++r)
#pragma unroll
    for (int ci = 0; ci < MAXCH; ++ci) own_w[r][ci] = 0.f;
  const int jtotal = ch * C0 * 125;
  __syncthreads();

  for (int b = wg; b < batch; b += nwg) {
    if (COOP) nvf_coop_wait(coop.dh_done + b * kStemFlagStride, C1 / 2);
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int t = tid + r * NT, c = t >> 6, v = t & 63;
      s_a[c * LS + v] = a0[(size_t)b * NE + t];
      // dh0 = the channel-pair partials of conv0's backward-data, added in ascending order
      float dh = 0.f;
#pragma unroll
      for (int w = 0; w < C1 / 2; ++w) {
        const float* pp = part + ((size_t)b * (C1 / 2) + w) * NE + t;
        dh += COOP ? nvf_load_dev(pp) : *pp;
      }
      s_dh[t] = dh;
    }
    if (tid < ch * 8) s_x[tid] = x0[(size_t)b * ch * 8 + tid];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int t = tid + r * NT, c = t >> 6, v = t & 63;
      const float dh = s_dh[t];
      // IGDN forward quantities of this voxel/channel: n_c, t_c = dh_c a_c / n_c
      float nrm = s_bet[c];
#pragma unroll
      for (int j = 0; j < C0; ++j) {
        const float xj = s_a[j * LS + v];
        nrm = fmaf(s_gam[c * C0 + j], xj * xj, nrm);
      }
      nrm = sqrtf(nrm);
      s_n[t] = nrm;
      s_t[c * LS + v] = dh * s_a[c * LS + v] / nrm;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {  // da0_i = dh_i n_i + a_i sum_c t_c gamma_ci  (i = this element's channel)
      const int t = tid + r * NT, c = t >> 6, v = t & 63, iz = v >> 4, iy = (v >> 2) & 3, ix = v & 3;
      float mix = 0.f;
#pragma unroll
      for (int cc = 0; cc < C0; ++cc) mix = fmaf(s_t[cc * LS + v], s_gam[cc * C0 + c], mix);
      const float d = s_dh[t] * s_n[t] + s_a[c * LS + v] * mix;
      da0[(size_t)b * NE + t] = d;
      s_da[c * 343 + ((iz + 2) * 7 + iy + 2) * 7 + ix + 2] = d;
      if (COOP) {                       // up0's bias gradient: this block's channel sum (a wave holds one channel)
        const float s = nvf_wave_sum(d);
        if (v == 0) coop.bias_slab[(size_t)b * C0 + c] = s;
      }
    }
    if (want_w && tid < NCOL) {   // parameter partials: p < C0: d beta_p ; else d gamma_{cc,j}
      float sum = 0.f;
      if (tid < C0) {
#pragma unroll 8
        for (int k = 0; k < 64; ++k) sum += s_t[tid * LS + k];
      } else {
        const int cc = (tid - C0) / C0, j = (tid - C0) % C0;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
          const float xj = s_a[j * LS + k];
          sum = fmaf(s_t[cc * LS + k], xj * xj, sum);
        }
      }
      own_gdn += 0.5f * sum;
    }
    __syncthreads();
    // ---- up0 backward-data: dx0[ci, i] = sum_co sum_k da0[co, 2 i - 2 + k] w0[ci][co][k]; C0 lanes (co) per output
#pragma unroll 1
    for (int r = 0; r < EPT; ++r) {
      const int t = tid + r * NT;
      if (t < ch * 8 * C0) {            // (uniform over each group of C0 lanes: the shuffles below stay inside it)
        const int out = t / C0, co = t % C0, ci = out >> 3, i = out & 7;
        const int jz = i >> 2, jy = (i >> 1) & 1, jx = i & 1;
        const float* dp = s_da + co * 343 + ((2 * jz) * 7 + 2 * jy) * 7 + 2 * jx;
        const float* wp = s_w0 + co * 125 * ch + ci;
        float acc = 0.f;
#pragma unroll 1
        for (int kz = 0; kz < 5; ++kz)
#pragma unroll
          for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
              acc = fmaf(dp[(kz * 7 + ky) * 7 + kx], wp[((kz * 5 + ky) * 5 + kx) * ch], acc);
#pragma unroll
        for (int m = 1; m < C0; m <<= 1) acc += __shfl_xor(acc, m, 64);
        if (co == 0) {
          if (COOP) nvf_store_dev(dx0 + (size_t)b * ch * 8 + out, acc);
          else dx0[(size_t)b * ch * 8 + out] = acc;
        }
      }
    }
    // ---- up0 weight gradient: dW0[ci][co][k] += sum_i x0[ci, i] da0[co, 2 i - 2 + k].  A thread owns (co, k) pairs and
    // all ch input channels: the eight da0 values of a pair are read once for the ch sums
    if (want_w) {
#pragma unroll
      for (int r = 0; r < NPAIR; ++r) {
        const int p = tid + NT * r;
        if (p < C0 * 125) {
          const int kk = p % 125, co = p / 125;
          const int kz = kk / 25, ky = (kk / 5) % 5, kx = kk % 5;
          const float* dp = s_da + co * 343 + (kz * 7 + ky) * 7 + kx;
          float dv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) dv[i] = dp[((2 * (i >> 2)) * 7 + 2 * ((i >> 1) & 1)) * 7 + 2 * (i & 1)];
#pragma unroll
          for (int ci = 0; ci < MAXCH; ++ci)
            if (ci < ch) {
              float acc = own_w[r][ci];
#pragma unroll
              for (int i = 0; i < 8; ++i) acc = fmaf(s_x[ci * 8 + i], dv[i], acc);
              own_w[r][ci] = acc;
            }
        }
      }
    }
    __syncthreads();
  }
  if (COOP) nvf_coop_signal(coop.stem_done);     // dx0 of this workgroup's blocks is out: the latent tail may read it
  if (want_w) {
    if (tid < NCOL) slab_gdn[(size_t)wg * NCOL + tid] = own_gdn;
#pragma unroll
    for (int r = 0; r < NPAIR; ++r) {
      const int p = tid + NT * r;
      if (p < C0 * 125) {
#pragma unroll
        for (int ci = 0; ci < MAXCH; ++ci)
          if (ci < ch) slab_w[(size_t)wg * jtotal + (size_t)ci * C0 * 125 + p] = own_w[r][ci];
      }
    }
  }
}

// The stem's backward queued for the five-gradient launch (nvf_stem_bwd_queue): the arguments of the two bodies above
struct StemBwdJob {
  const float* g1;
  const float* w1b;
  const float* x0;
  const float* a0;
  const float* w0b;
  const float* beta_hat;
  const float* gamma_hat;
  float* part;
  float* da0;
  float* dx0;
  float* slab_gdn;
  float* slab_w;
  StemCoop coop;
  int32_t batch, ch, nwg, pad_;
};
