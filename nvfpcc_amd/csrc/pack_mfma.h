// Device side of nvf_pack_mfma_all, shared with the one-launch head of the training step (pointwise.hip): all MFMA
// A-fragment packings of a step.  `src(job, i)` yields element i of the job's source weight layout.
#pragma once
#include "nvf_common.h"

struct PackJobs {
  const float* src[16];
  float* dst[16];
  int32_t kind[16], c0[16], c1[16], total[16];
  int32_t layer[16], bwd[16];      // step head only: row of the layer table and which layout (0 w_fwd, 1 w_bwd) src is
  int32_t n;
};

template <class Src>
__device__ __forceinline__ void pack_mfma_body(const PackJobs& m, int job, int bx, int nbx, Src src) {
  float* __restrict__ wp = m.dst[job];
  const int kind = m.kind[job], total = m.total[job];
  if (kind == 40) {
    // Winograd (y, x) backward-data of a 4^3 conv (conv_wino.hip): [g][zw][f][lane], lane = (co, s) + 16 k, value
    // U = G w' G^T of the gather-form kernel slice kz = zw - s, G = Cook-Toom F(2, 4) on {0, 1, -1, 2, inf} with the
    // rational factors of B^T folded in (rows / 2, / 2, / 6, / -6, 1).  Every output of slice (ci, kz, co) goes to
    // (zw = kz, s = 0) and (zw = kz + 1, s = 1); the entries without a tap (zw = 0, s = 1 and zw = 4, s = 0) are zeroed
    const float G[5][4] = {{0.5f, 0.f, 0.f, 0.f},
                           {0.5f, 0.5f, 0.5f, 0.5f},
                           {1.f / 6.f, -1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                           {-1.f / 6.f, -2.f / 6.f, -4.f / 6.f, -8.f / 6.f},
                           {0.f, 0.f, 0.f, 1.f}};
    // four lanes per slice: lane ky fetches the four weights of its kernel row (src may draw counter-RNG noise per
    // element: 4 draws per thread, not 16), transforms the row along x, and the quad exchanges rows by shuffles; lane ky
    // then produces the outputs of fy = ky (lane 0 also fy = 4).  Same arithmetic as one thread per (slice, fy).
    const int nthreads = nbx * blockDim.x;                       // a multiple of 64: whole quads
    for (int t = bx * blockDim.x + threadIdx.x; t < 256 * 4; t += nthreads) {        // quads are whole: in or out together
      const int ky = t & 3, sl = t >> 2, cog = sl % 8, kz = (sl / 8) % 4, ci = sl / 32, g = ci / 4, k = ci % 4;
      float w4[4], mine[5];
      for (int kx = 0; kx < 4; ++kx) w4[kx] = src(job, (ci * 64 + (kz * 4 + ky) * 4 + kx) * 8 + cog);
      for (int fx = 0; fx < 5; ++fx) {
        float r = 0.f;
        for (int kx = 0; kx < 4; ++kx) r = fmaf(G[fx][kx], w4[kx], r);
        mine[fx] = r;
      }
      float row[4][5];
      for (int q = 0; q < 4; ++q)
        for (int fx = 0; fx < 5; ++fx) row[q][fx] = __shfl(mine[fx], (int)((threadIdx.x & 63u) & ~3u) + q, 64);
      for (int pass = 0; pass < 2; ++pass) {
        const int fy = pass == 0 ? ky : 4;
        if (pass == 1 && ky != 0) break;
        for (int fx = 0; fx < 5; ++fx) {
          float v = 0.f;
          for (int q = 0; q < 4; ++q) v = fmaf(G[fy][q], row[q][fx], v);
          const int f = fy * 5 + fx;
          wp[((g * 5 + kz) * 25 + f) * 64 + 16 * k + 2 * cog] = v;
          wp[((g * 5 + kz + 1) * 25 + f) * 64 + 16 * k + 2 * cog + 1] = v;
        }
      }
    }
    for (int r0 = bx * blockDim.x + threadIdx.x; r0 < 3200; r0 += nthreads) {
      int r = r0;
      const int cog = r % 8; r /= 8;
      const int k = r % 4; r /= 4;
      const int f = r % 25; r /= 25;
      const int g = r % 2, s1 = r / 2;                      // s1 = 1: (zw 0, s 1); 0: (zw 4, s 0)
      wp[((g * 5 + (s1 ? 0 : 4)) * 25 + f) * 64 + 16 * k + 2 * cog + s1] = 0.f;
    }
    return;
  }
  if (kind == 41) {
    // Winograd (y, x) form of a 4^3 conv with 16 -> 16 channels (conv16_wino.hip): [g][tz][f][lane], lane = co + 16 k, value
    // U = G w G^T of the gather-form kernel slice (ci = 4 g + k, kz = tz, co); G and the quad scheme as kind 40
    const float G[5][4] = {{0.5f, 0.f, 0.f, 0.f},
                           {0.5f, 0.5f, 0.5f, 0.5f},
                           {1.f / 6.f, -1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                           {-1.f / 6.f, -2.f / 6.f, -4.f / 6.f, -8.f / 6.f},
                           {0.f, 0.f, 0.f, 1.f}};
    const int nthreads = nbx * blockDim.x;
    for (int t = bx * blockDim.x + threadIdx.x; t < 1024 * 4; t += nthreads) {
      const int ky = t & 3, sl = t >> 2, cog = sl % 16, kz = (sl / 16) % 4, ci = sl / 64, g = ci / 4, k = ci % 4;
      float w4[4], mine[5];
      for (int kx = 0; kx < 4; ++kx) w4[kx] = src(job, (ci * 64 + (kz * 4 + ky) * 4 + kx) * 16 + cog);
      for (int fx = 0; fx < 5; ++fx) {
        float r = 0.f;
        for (int kx = 0; kx < 4; ++kx) r = fmaf(G[fx][kx], w4[kx], r);
        mine[fx] = r;
      }
      float row[4][5];
      for (int q = 0; q < 4; ++q)
        for (int fx = 0; fx < 5; ++fx) row[q][fx] = __shfl(mine[fx], (int)((threadIdx.x & 63u) & ~3u) + q, 64);
      for (int pass = 0; pass < 2; ++pass) {
        const int fy = pass == 0 ? ky : 4;
        if (pass == 1 && ky != 0) break;
        for (int fx = 0; fx < 5; ++fx) {
          float v = 0.f;
          for (int q = 0; q < 4; ++q) v = fmaf(G[fy][q], row[q][fx], v);
          wp[((g * 4 + kz) * 25 + fy * 5 + fx) * 64 + 16 * k + cog] = v;
        }
      }
    }
    return;
  }
  for (int idx = bx * blockDim.x + threadIdx.x; idx < total; idx += nbx * blockDim.x) {
    int r = idx;
    const int lane = r % 64; r /= 64;
    const int i = lane & 15, k = lane >> 4;
    float v = 0.f;
    if (kind == 0 || kind == 2) {                    // 4^3 conv, 8 output channels: [g][ty][tx][tz][lane]
      const int KEZ = kind == 2 ? 5 : 4, KEX = kind == 0 ? 5 : 4;
      const int tz = r % KEZ; r /= KEZ;
      const int tx = r % KEX; r /= KEX;
      const int ty = r % 4, g = r / 4;
      const int cog = i >> 1, s = i & 1, ci = 4 * g + k;
      const int kz = kind == 2 ? tz - s : tz, kx = kind == 0 ? tx - s : tx;
      if (kz >= 0 && kz < 4 && kx >= 0 && kx < 4) v = src(job, (ci * 64 + (kz * 4 + ty) * 4 + kx) * 8 + cog);
    } else if (kind == 10) {                         // transposed conv forward: [g][75 class/tap fragments][lane]
      int f = r % 75;
      const int g = r / 75;
      int ez = 0, ey = 0;
      if (f >= 63) { ez = 1; ey = 1; f -= 63; }
      else if (f >= 45) { ez = 1; f -= 45; }
      else if (f >= 27) { ey = 1; f -= 27; }
      const int jx = f % 3, jy = (f / 3) % (3 - ey), jz = f / (3 * (3 - ey));
      const int co = i >> 1, ex = i & 1, ci = 4 * g + k;
      const int kz = ez + 2 * jz, ky = ey + 2 * jy, kx = ex + 2 * jx;
      if (kx < 5) v = src(job, (ci * 125 + (kz * 5 + ky) * 5 + kx) * 8 + co);
    } else if (kind == 12) {                         // ... with the kx = 4 taps on rows (co, ey): [g][50 + 15 fragments][lane]
      // (convt_mfma.hip, REPAIR: training steps only.)  Fragments 0..49: the classes (ez, ey) = (0,0) (0,1) (1,0) (1,1),
      // inside a class [jz][jy][jx = 0, 1], rows (co, ex); 50..64: the x-edge taps (jx = 2: kx = 4, even outputs only),
      // ez = 0 then 1, [jz][jy = 0..2], rows (co, ey) -- the odd row parity has no ky = 5 tap (jy = 2: zero)
      int f = r % 65;
      const int g = r / 65, ci = 4 * g + k, co = i >> 1, par = i & 1;
      if (f < 50) {
        int ez = 0, ey = 0;
        if (f >= 42) { ez = 1; ey = 1; f -= 42; }
        else if (f >= 30) { ez = 1; f -= 30; }
        else if (f >= 18) { ey = 1; f -= 18; }
        const int jx = f % 2, jy = (f / 2) % (3 - ey), jz = f / (2 * (3 - ey));
        v = src(job, (ci * 125 + ((ez + 2 * jz) * 5 + ey + 2 * jy) * 5 + par + 2 * jx) * 8 + co);
      } else {
        f -= 50;
        const int ez = f >= 9 ? 1 : 0;
        if (ez) f -= 9;
        const int jy = f % 3, jz = f / 3, ky = par + 2 * jy;
        if (ky < 5) v = src(job, (ci * 125 + ((ez + 2 * jz) * 5 + ky) * 5 + 4) * 8 + co);
      }
    } else if (kind == 11) {                         // transposed conv forward, 16 output channels: [g][125][lane]
      int f = r % 125, cls = 0;
      const int g = r / 125;
      for (;; ++cls) {
        const int n = (3 - (cls >> 2)) * (3 - ((cls >> 1) & 1)) * (3 - (cls & 1));
        if (f < n) break;
        f -= n;
      }
      const int ez = cls >> 2, ey = (cls >> 1) & 1, ex = cls & 1;
      const int jx = f % (3 - ex), jy = (f / (3 - ex)) % (3 - ey), jz = f / ((3 - ex) * (3 - ey));
      const int cout = m.c1[job], g4 = g % (m.c0[job] / 4), cg = g / (m.c0[job] / 4);
      v = src(job, ((4 * g4 + k) * 125 + ((ez + 2 * jz) * 5 + ey + 2 * jy) * 5 + ex + 2 * jx) * cout + cg * 16 + i);
    } else if (kind == 30 || kind == 31) {           // 16-row gather convolution (conv16_mfma.hip): [cog][g][tap][lane]
      const int k3 = kind == 30 ? 64 : 125, cin = m.c0[job], cout = m.c1[job];
      const int tap = r % k3; r /= k3;
      const int g = r % (cin / 4), cg = r / (cin / 4);
      if (cg * 16 + i < cout) v = src(job, ((4 * g + k) * k3 + tap) * cout + cg * 16 + i);
    } else {                                         // stride-2 gather (transposed conv backward-data)
      const int cog = m.c1[job], pair = cog == 8, KEX = pair ? 7 : 5;
      const int tx = r % KEX; r /= KEX;
      const int ky = r % 5; r /= 5;
      const int kz = r % 5, g = r / 5;
      const int ch = 4 * g + k;
      const int co = pair ? i >> 1 : i, kx = pair ? tx - 2 * (i & 1) : tx;
      if (kx >= 0 && kx < 5) v = src(job, (ch * 125 + (kz * 5 + ky) * 5 + kx) * cog + co);
    }
    wp[idx] = v;
  }
}

// kinds: 0 / 2 = nvf_pack_mfma_k4 with that pair axis (c0 = cin); 10 = nvf_pack_convT_mfma (c0 = cin); 12 = its training-step form with the kx = 4 taps on rows (co, ey) (65 fragments per channel group);
// 11 = nvf_pack_convT16_mfma (c0 = cin, c1 = cout); 20 = nvf_pack_s2k5_mfma (c0 = cig, c1 = cog); 30 / 31 = nvf_pack_g16_mfma with k = 4 / 5 (c0 = cin, c1 = cout); 40 = Winograd form of a 4^3 conv (c0 = c1 = 8; conv_wino.hip); 41 = the same with 16 -> 16 channels (conv16_wino.hip).
// Fills m (sources optional: the step head derives them).
static inline int pack_jobs_desc(const float* const* srcs, float* const* dsts, const int* kinds, const int* c0s,
                                 const int* c1s, int n, PackJobs& m) {
  if (!dsts || !kinds || !c0s || !c1s || n <= 0 || n > 16) return NVF_EINVAL;
  for (int j = 0; j < n; ++j) {
    if (!dsts[j] || c0s[j] <= 0 || c0s[j] % 4) return NVF_EINVAL;
    m.src[j] = srcs ? srcs[j] : nullptr; m.dst[j] = dsts[j]; m.kind[j] = kinds[j]; m.c0[j] = c0s[j]; m.c1[j] = c1s[j];
    if (kinds[j] == 0 || kinds[j] == 2) m.total[j] = (c0s[j] / 4) * 4 * (kinds[j] == 0 ? 5 : 4) * (kinds[j] == 2 ? 5 : 4) * 64;
    else if (kinds[j] == 10) m.total[j] = (c0s[j] / 4) * 75 * 64;
    else if (kinds[j] == 12) m.total[j] = (c0s[j] / 4) * 65 * 64;
    else if (kinds[j] == 40 && c0s[j] == 8 && c1s[j] == 8) m.total[j] = 2 * 5 * 25 * 64;
    else if (kinds[j] == 41 && c0s[j] == 16 && c1s[j] == 16) m.total[j] = 4 * 4 * 25 * 64;
    else if (kinds[j] == 11 && c1s[j] > 0 && c1s[j] % 16 == 0) m.total[j] = (c1s[j] / 16) * (c0s[j] / 4) * 125 * 64;
    else if (kinds[j] == 20 && (c1s[j] == 8 || c1s[j] == 16)) m.total[j] = (c0s[j] / 4) * 25 * (c1s[j] == 8 ? 7 : 5) * 64;
    else if ((kinds[j] == 30 || kinds[j] == 31) && c1s[j] > 0 && (c1s[j] % 16 == 0 || c1s[j] == 8))
      m.total[j] = ((c1s[j] + 15) / 16) * (c0s[j] / 4) * (kinds[j] == 30 ? 64 : 125) * 64;
    else return NVF_EINVAL;
  }
  m.n = n;
  return NVF_OK;
}
