// Device body of the stem's forward (stem.hip, top comment): x0 --up0--> a0 --IGDN--> h0 --conv0 + ReLU--> y1 with every
// intermediate in LDS, and -- LATENT -- the latent generator + quantiser in front of it.  Shared by the stand-alone kernel of
// stem.hip (weights from the layouts nvf_prepare_weights leaves) and by the step head (pointwise.hip: nvf_step_head_stem),
// where the stem's workgroups derive their weights from the RAW parameters themselves (the arithmetic of the weight
// preparation, element by element: same bits) and so depend on nothing the step head produces: one launch instead of two
// latency chains.  A weight provider WP supplies
//   fill_up0<NT, C0>(s_w0, ch, tid)              : up0's effective weights into LDS, forward layout [ci][k][co]
//   fill_conv0<NT, C0, C1, COG>(s_w1, part, tid) : conv0's, channels [part * NCG * COG ...), layout [cg][ci][k][COG]
//   up0_b(co), conv0_b(co) : effective biases;   lat_w(i) ([ci][co]), lat_b(j) : the latent generator's
//   e(b, i, sp) : latent-table entry of mini-batch row b, channel i, voxel sp
#pragma once
#include "nvf_common.h"
#include "latent_tail.h"
#include "stem_bwd.h"

constexpr int kStemFwdMaxCh = 8;

// conv0 for one output parity class (EZ,EY,EX): lane = cell, COG output channels in registers, taps unrolled so
// the LDS reads of a whole input channel are in flight together.  Weights come from the LDS copy s_w[ci*125+tap][COG].
template <int C0, int C1, int EZ, int EY, int EX, int COG>
__device__ __forceinline__ void stem_conv0_class(const float* s_h, const float* s_w, const float* s_b1,
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
    y1[((size_t)b * C1 + co0 + co) * 512 + (qz * 8 + qy) * 8 + qx] = fmaxf(acc[co] + s_b1[co0 + co], 0.f);
}

// The latent generator + quantiser (nvf_latent_fwd) for the launch that also runs the stem: the stem's workgroups
// compute the 8 ch rounded latents of their own block themselves (latent_x_rounded: the same arithmetic), so they do
// not wait for the one workgroup that produces h, lat, x_rounded and the rate for the whole batch.
struct StemLatent {
  const float* e;
  const float* w;          // latent generator's w_fwd [ci][co], bias (prepared-weights provider only)
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

// the weights as nvf_prepare_weights left them (stand-alone kernel)
struct StemPreparedW {
  const float* w0;
  const float* b0;
  const float* w1;
  const float* b1;
  const float* lw;
  const float* lb;
  const float* ev;         // gathered latent rows [batch][ch][8]
  int32_t ch;
  template <int NT, int C0>
  __device__ __forceinline__ void fill_up0(float* s_w0, int nch, int tid) const {
    const float* w = w0;
    stem_detail::copy_from<NT, 16>(s_w0, nch * 125 * C0, tid, [&](int e) { return w[e]; });
  }
  template <int NT, int C0, int C1, int COG>
  __device__ __forceinline__ void fill_conv0(float* s_w1, int part, int tid) const {
    constexpr int NCG = C0 / 8;
    const float* w = w1;
    stem_detail::copy_from<NT, 16>(s_w1, NCG * C0 * 125 * COG, tid, [&](int e) {
      const int cg = e / (C0 * 125 * COG), r = e - cg * (C0 * 125 * COG);
      return w[(r / COG) * C1 + (part * NCG + cg) * COG + r % COG];
    });
  }
  __device__ __forceinline__ float up0_b(int co) const { return b0[co]; }
  __device__ __forceinline__ float conv0_b(int co) const { return b1[co]; }
  __device__ __forceinline__ float lat_w(int i) const { return lw[i]; }
  __device__ __forceinline__ float lat_b(int j) const { return lb[j]; }
  __device__ __forceinline__ float e(int b, int i, int sp) const { return ev[((size_t)b * ch + i) * 8 + sp]; }
};

template <int C0, int C1, int COG>
struct StemFwdLds {
  static constexpr int NCG = C0 / 8;
  static constexpr int X = 0, A = X + kStemFwdMaxCh * 8, W0 = A + C0 * 64, H = W0 + kStemFwdMaxCh * 125 * C0,
                       W1 = H + C0 * 216, BETA = W1 + NCG * C0 * 125 * COG, GAMMA = BETA + C0, LAT = GAMMA + C0 * C0,
                       B0 = LAT + 2 * kStemFwdMaxCh * kStemFwdMaxCh + 2 * kStemFwdMaxCh, B1 = B0 + C0, FLOATS = B1 + C1;
  // the latent workgroup's private copies (raw-parameter provider): gathered rows, weights, bias -- in the H / W1 region
  static constexpr int LE = H, LW = LE + 32 * kStemFwdMaxCh * 8, LB = LW + kStemFwdMaxCh * kStemFwdMaxCh;
  static_assert(LB + kStemFwdMaxCh <= BETA, "latent copies fit behind the scratch region");
};

// one workgroup of C0 * 64 threads: (block b, conv0 channel part `part`), or -- part == PARTS, b == 0, LATENT -- the
// latent generator + quantiser of the whole mini-batch.  COPY_LATENT: that workgroup first copies the rows / weights the
// provider derives into LDS (the raw-parameter provider), else it reads L.e / L.w / L.bw directly.
template <int C0, int C1, int COG, bool LATENT, bool COPY_LATENT, class WP>
__device__ __forceinline__ void stem_fwd_body(const float* __restrict__ x0, const WP& wp,
                                              const float* __restrict__ beta_hat, const float* __restrict__ gamma_hat,
                                              float* __restrict__ a0, float* __restrict__ h0, float* __restrict__ y1,
                                              int ch, const StemLatent& L, int b, int part, float* lds) {
  using LD = StemFwdLds<C0, C1, COG>;
  constexpr int NT = C0 * 64, NCG = C0 / 8, MAXCH = kStemFwdMaxCh;   // NCG groups of eight waves (one per parity class)
  constexpr int PARTS = C1 / (COG * NCG);                              //  in the conv0 phase, each with its own COG channels
  float *s_x = lds + LD::X, *s_a = lds + LD::A, *s_w0 = lds + LD::W0;
  const int tid = threadIdx.x;
  if (LATENT && part == PARTS) {
    if (b == 0) {
      if (COPY_LATENT) {
        float *s_e = lds + LD::LE, *s_lw = lds + LD::LW, *s_lb = lds + LD::LB;
        for (int e = tid; e < L.batch * ch * 8; e += NT) s_e[e] = wp.e(e / (ch * 8), (e >> 3) % ch, e & 7);
        if (tid < ch * ch) s_lw[tid] = wp.lat_w(tid);
        if (tid >= 64 && tid < 64 + ch) s_lb[tid - 64] = wp.lat_b(tid - 64);
        __syncthreads();
        latent_fwd_body(s_e, s_lw, s_lb, L.beta_hat, L.gamma_hat, L.block_ids, L.sigma, L.mu, L.h_out, L.lat_out,
                        L.x_rounded, L.bits, L.batch, ch, 8, L.mode, L.seed, L.step, L.step_dev, s_a, s_w0,
                        MAXCH * 125 * C0);
      } else {
        latent_fwd_body(L.e, L.w, L.bw, L.beta_hat, L.gamma_hat, L.block_ids, L.sigma, L.mu, L.h_out, L.lat_out,
                        L.x_rounded, L.bits, L.batch, ch, 8, L.mode, L.seed, L.step, L.step_dev, s_a, s_w0,
                        MAXCH * 125 * C0);
      }
    }
    return;
  }
  float* s_h = lds + LD::H;           // h0 with a one-voxel zero halo: [c][6][6][6], index i + 1
  float* s_w1 = lds + LD::W1;
  float *s_beta = lds + LD::BETA, *s_gamma = lds + LD::GAMMA;
  float* s_lat = lds + LD::LAT;       // latent generator: w [ci][co], gamma_hat, bias, beta_hat
  float *s_b0 = lds + LD::B0, *s_b1 = lds + LD::B1;
  float ev[MAXCH];                    // this thread's latent element: its ch inputs, fetched together
  if (LATENT) {
    // (parameters through LDS and the inputs up front: as loads inside the fmaf chains they were ch^2 dependent round trips)
    if (tid < ch * ch) { s_lat[tid] = wp.lat_w(tid); s_lat[MAXCH * MAXCH + tid] = L.gamma_hat[tid]; }
    if (tid >= 64 && tid < 64 + ch) {
      s_lat[2 * MAXCH * MAXCH + tid - 64] = wp.lat_b(tid - 64);
      s_lat[2 * MAXCH * MAXCH + MAXCH + tid - 64] = L.beta_hat[tid - 64];
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) ev[i] = (tid < ch * 8 && i < ch) ? wp.e(b, i, tid & 7) : 0.f;
  } else if (tid < ch * 8) {
    s_x[tid] = x0[(size_t)b * ch * 8 + tid];
  }
  if (tid >= 64 && tid < 64 + C0) s_beta[tid - 64] = stem_detail::beta_of(beta_hat[tid - 64]);
  if (tid >= 128 && tid < 128 + C0 * C0) s_gamma[tid - 128] = stem_detail::gamma_of(gamma_hat[tid - 128]);
  if (tid >= 192 && tid < 192 + C0) s_b0[tid - 192] = wp.up0_b(tid - 192);
  if (tid >= 256 && tid < 256 + C1) s_b1[tid - 256] = wp.conv0_b(tid - 256);
  for (int e = tid; e < C0 * 216; e += NT) s_h[e] = 0.f;
  wp.template fill_up0<NT, C0>(s_w0, ch, tid);
  wp.template fill_conv0<NT, C0, C1, COG>(s_w1, part, tid);
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
    const float val = acc + s_b0[co];
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
    case 0: stem_conv0_class<C0, C1, 0, 0, 0, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 1: stem_conv0_class<C0, C1, 0, 0, 1, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 2: stem_conv0_class<C0, C1, 0, 1, 0, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 3: stem_conv0_class<C0, C1, 0, 1, 1, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 4: stem_conv0_class<C0, C1, 1, 0, 0, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 5: stem_conv0_class<C0, C1, 1, 0, 1, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    case 6: stem_conv0_class<C0, C1, 1, 1, 0, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
    default: stem_conv0_class<C0, C1, 1, 1, 1, COG>(s_h, sw, s_b1, y1, b, co0, v); break;
  }
}
