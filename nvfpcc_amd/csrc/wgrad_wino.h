// Weight gradient of the valid 4x4x4 convolution with 8 -> 8 channels (conv2: dY 32^3, X 35^3) in a reduced-
// multiplication form: Winograd F(4x4, 2x2) over (y, x) -- the 4 x 4 taps are the OUTPUT, a 2 x 2 tile of dY the
// "filter", a 5 x 5 window of X the input -- direct over z on the matrix cores.  Reference site: the weight half of the
// autograd backward of F.conv3d, utils/network.py:687 (NVFPCC.py:197; 66 % of the reference's CPU step).
//
//   dW[co][ci][kz][ky][kx] = sum_{n,z,y,x} dY[n,co,z,y,x] X[n,ci,z+kz,y+ky,x+kx]
//   per (y, x) tile T (dY rows 2R..2R+1, columns 2C..2C+1; X window 5 x 5 at (2R, 2C)) and plane pair (z, z + kz):
//   dW_T[kz][:, :] = A^T [ (G g_T G^T) * (B^T x_T B) ] A,  summed over tiles in the TRANSFORM domain:
//   M[f][kz][co][ci] = sum_{n,z,T} Gh[f][co][n,z,T] Xh[f][ci][n,z+kz,T],   dW = A^T M A once per workgroup
//   (G = [1 0; 1 1; 1 -1; 1 2; 0 1], B^T the integer form of wino_common.h, its factors folded into A^T): 25 products per
//   tile and plane pair instead of 64 -- and, unlike the direct form, no x padding (32 -> 36).  fp32 error against
//   float64 3.9e-6 of max |dW| (direct 1.0e-6; gradient goldens are held to 2e-4).
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32), every lane useful: K = four tiles; rows (co, a) take Gh of plane z + a,
// columns (ci, b) take Xh of plane z + 1 + 2 b: D[(co,a)][(ci,b)] += ... is M[kz = 1 + 2 b - a], all four taps of kz in one
// 16 x 16 tile; z advances by one per step (each dY plane meets a = 0 and a = 1 once, each X plane b = 0 and b = 1 once).
// The A operand of lane (co, a, tile k) is Gh of ITS tile, the B operand of lane (ci, b, k) Xh of its tile: both transforms
// run in the lane that feeds them (21 + 90 VALU operations per step and 25 MFMAs).  A wave walks z for one group of four
// tiles of a tile row; its windows (X: 5 rows x 11 columns x 8 channels per plane, three planes live; dY: 2 x 8 x 8, two
// planes) sit in a 7 KB LDS ring of its own, the next planes arrive in registers (three 16-byte buffer loads per step) and are
// committed after the step's reads -- no barrier before the epilogue.  Epilogue: A^T M A in registers, then the existing
// cross-wave sum (one padded LDS region per wave, fixed order) and one 4096-float slab per workgroup.
#pragma once
#include "wino_common.h"

template <int W_>
struct WWCfg {
  static constexpr int W = W_, WQ = W_ + 3, TPR = W_ / 2, NGR = TPR / 4, NGRP = TPR * NGR, NSTEP = W_ + 1;
  static constexpr int XRS = 12, XCS = 5 * XRS;              // X window rows of 11 words (+1: 8-byte reads)
  static constexpr int xps_for() { int v = 8 * XCS; while (v % 32 != 16) ++v; return v; }
  static constexpr int XPS = xps_for();                      // plane stride: b = 1 lanes land on banks + 32
  static constexpr int GRS = 8, GCS = 20, GPS = 8 * GCS + 4;
  static constexpr int WLDS = 3 * XPS + 2 * GPS;             // floats of LDS per wave
  static constexpr int LDSF = 4 * WLDS;
  static constexpr int NXE = 8 * 5 * 11, NLX = (NXE + 63) / 64, NGE = 8 * 2 * 8, NLG = NGE / 64;
  static_assert(TPR % 4 == 0 && NGE % 64 == 0, "groups of four tiles");
};

// `lds` must hold max(C::LDSF, 4 * region) floats (region = 4096 + 64: the epilogue's per-wave sums)
template <class C>
__device__ __forceinline__ void wgrad_k4_wino_body(const float* __restrict__ g, const float* __restrict__ x,
                                                   float* __restrict__ slabs, const WgDims& d, int bx, float* lds,
                                                   int region) {
  constexpr int W = C::W, WQ = C::WQ, XRS = C::XRS, XCS = C::XCS, XPS = C::XPS, GRS = C::GRS, GCS = C::GCS, GPS = C::GPS;
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4, ch = i & 7, hs = i >> 3;
  float* xr = lds + wave * C::WLDS;
  float* gr = xr + 3 * XPS;
  for (int e = lane; e < C::WLDS; e += 64) xr[e] = 0.f;
  f32x4_ acc[25];
#pragma unroll
  for (int f = 0; f < 25; ++f) acc[f] = f32x4_{0.f, 0.f, 0.f, 0.f};
  float bs = 0.f;
  // staging descriptors: an X plane slice is 40 (channel, row) rows of three 16-byte pieces (tile groups start on 32-byte
  // boundaries; the twelfth word of a row is never read), a dY slice 16 rows of two: 2 + 1 loads per lane
  typedef unsigned wwu4 __attribute__((ext_vector_type(4)));
  constexpr int kOob = 0x7ffffff0;
  int voffx[2], ldsx[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = lane + 64 * k, r = e / 3, seg = e % 3, c = r / 5, row = r % 5;
    voffx[k] = e < 120 ? ((c * WQ * WQ + row) * WQ + 4 * seg) * 4 : kOob;
    ldsx[k] = e < 120 ? c * XCS + row * XRS + 4 * seg : -1;
  }
  const int voffg = lane < 32 ? (((lane >> 2) * W * W + ((lane >> 1) & 1)) * W + 4 * (lane & 1)) * 4 : kOob;
  const int ldsg = lane < 32 ? (lane >> 2) * GCS + ((lane >> 1) & 1) * GRS + 4 * (lane & 1) : -1;
  const int zsplit = d.tiles_z > 0 ? d.tiles_z : 1;            // z steps of a (block, tile group) shared by this many items
  // XCD-local work: workgroups bx, bx + 8, ... share an XCD (and its L2); they take consecutive item ranges, i.e. the
  // neighbouring tile groups of the same blocks, when the job's workgroup count d.tiles_y is a multiple of 8
  const int nwg = d.tiles_y;
  const int bxl = (nwg > 0 && nwg % 8 == 0) ? (bx & 7) * (nwg >> 3) + (bx >> 3) : bx;
  const int first = bxl * d.items_per_wg, last = min(first + d.items_per_wg, d.items);
#pragma unroll 1
  for (int item = first + wave; item < last; item += 4) {
    const int it = __builtin_amdgcn_readfirstlane(item);
    const int zs = it % zsplit, grp = (it / zsplit) % C::NGRP, n = it / (zsplit * C::NGRP);
    const int tr = grp / C::NGR, tg = grp % C::NGR;
    const int per = (C::NSTEP + zsplit - 1) / zsplit;
    const int z0 = -1 + zs * per, z1 = min(z0 + per, W);       // steps z0 .. z1 - 1 of -1 .. W - 1
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * 8 * WQ * WQ * WQ), 0,
                                                                         8 * WQ * WQ * WQ * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(g + (size_t)n * 8 * W * W * W), 0,
                                                                         8 * W * W * W * 4, 0x00020000);
    wwu4 xv[2], gv;
    auto load_x = [&](int p) {                                  // X plane p (always inside the tensor when called)
      const int so = __builtin_amdgcn_readfirstlane(((p * WQ + 2 * tr) * WQ + 8 * tg) * 4);
#pragma unroll
      for (int k = 0; k < 2; ++k) xv[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx[k], so, 0);
    };
    auto load_g = [&](int p) {                                  // dY plane p; outside [0, W): zeros
      const bool in = p >= 0 && p < W;
      const int so = __builtin_amdgcn_readfirstlane(in ? ((p * W + 2 * tr) * W + 8 * tg) * 4 : 0);
      gv = __builtin_amdgcn_raw_buffer_load_b128(rg, in ? voffg : kOob, so, 0);
    };
    auto commit_x = [&](int slot) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (ldsx[k] >= 0) *(wwu4*)(xr + slot * XPS + ldsx[k]) = xv[k];
    };
    auto commit_g = [&](int slot) {
      if (ldsg >= 0) *(wwu4*)(gr + slot * GPS + ldsg) = gv;
    };
    // prologue: dY planes z0, z0 + 1; X planes z0 + 1 .. z0 + 3 (slot of plane p: p & 1 / p % 3)
    load_g(z0); commit_g(z0 & 1);
    load_g(z0 + 1); commit_g((z0 + 1) & 1);
    int sx = (z0 + 1) % 3;                                      // slot of X plane z + 1 (scalar)
    load_x(z0 + 1); commit_x(sx);
    load_x(z0 + 2); commit_x(sx == 2 ? 0 : sx + 1);
    load_x(z0 + 3); commit_x(sx == 0 ? 2 : sx - 1);
    int sa = (z0 + hs) & 1;                                     // per lane: slot of its dY plane z + a
    int sb = (z0 + 1 + 2 * hs) % 3;                             //           slot of its X plane z + 1 + 2 b
#pragma unroll 1
    for (int z = z0; z < z1; ++z) {
      const bool more = z + 1 < z1;
      if (more) { load_g(z + 2); load_x(z + 4); }
      const float* gp = gr + sa * GPS + ch * GCS + 2 * kq;
      const float* xp = xr + sb * XPS + ch * XCS + 2 * kq;
      const wino_f2 g0 = *(const wino_f2*)gp, g1 = *(const wino_f2*)(gp + GRS);
      // transforms on the packed fp32 pipe (wino_common.h): every vector instruction is SIMD time next to the MFMAs
      float Xh[25], Gh[25];
      {
        wino_f2 a[5], bb[5], ea[5], eb[5];
        float c[5], ec[5];
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          a[dy] = *(const wino_f2*)(xp + dy * XRS);
          bb[dy] = *(const wino_f2*)(xp + dy * XRS + 2);
          c[dy] = xp[dy * XRS + 4];
        }
        wino_bt2(a[0], a[1], a[2], a[3], a[4], ea[0], ea[1], ea[2], ea[3], ea[4]);
        wino_bt2(bb[0], bb[1], bb[2], bb[3], bb[4], eb[0], eb[1], eb[2], eb[3], eb[4]);
        wino_bt(c[0], c[1], c[2], c[3], c[4], ec[0], ec[1], ec[2], ec[3], ec[4]);
#pragma unroll
        for (int fy = 0; fy < 5; ++fy)
          wino_bt_row(ea[fy], eb[fy], ec[fy], Xh[5 * fy], Xh[5 * fy + 1], Xh[5 * fy + 2], Xh[5 * fy + 3], Xh[5 * fy + 4]);
      }
      bs += (g0.x + g0.y) + (g1.x + g1.y);
      {                                                         // Gh = G g G^T, G = [1 0; 1 1; 1 -1; 1 2; 0 1]
        // y pass on the pairs (x0, x1) the LDS reads return, then per row (x, x + y, x - y, x + 2 y, y)
        const wino_f2 cy[5] = {g0, g0 + g1, g0 - g1, wino_fma2(wino_f2{2.f, 2.f}, g1, g0), g1};
#pragma unroll
        for (int fy = 0; fy < 5; ++fy) {
          const wino_f2 pm = wino_fma2(cy[fy].yy, wino_f2{1.f, -1.f}, cy[fy].xx);
          Gh[5 * fy] = cy[fy].x;
          Gh[5 * fy + 1] = pm.x;
          Gh[5 * fy + 2] = pm.y;
          Gh[5 * fy + 3] = fmaf(2.f, cy[fy].y, cy[fy].x);
          Gh[5 * fy + 4] = cy[fy].y;
        }
      }
#pragma unroll
      for (int f = 0; f < 25; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gh[f], Xh[f], acc[f], 0, 0, 0);
      if (more) {                                               // dY plane z is dead -> z + 2; X plane z + 1 -> z + 4
        commit_g(z & 1);
        commit_x(sx);
      }
      sx = sx == 2 ? 0 : sx + 1;
      sa ^= 1;
      sb = sb == 2 ? 0 : sb + 1;
    }
  }
  // ---- epilogue: bias partials, A^T M A, cross-wave sum, one slab per workgroup ----
  __syncthreads();                                              // every wave is done with its ring: the sums overlay it
  if (d.bias_slab) {                                            // lanes 0..7 of a wave: (co = lane, a = 0, k = 0)
    float bsv = bs + __shfl_xor(bs, 16, 64);
    bsv += __shfl_xor(bsv, 32, 64);
    if (lane < 8) lds[wave * 8 + lane] = bsv;
    __syncthreads();
    if (tid < 8) d.bias_slab[(size_t)bx * 8 + tid] = ((lds[tid] + lds[8 + tid]) + lds[16 + tid]) + lds[24 + tid];
    __syncthreads();
  }
  {
    float* reg = lds + wave * region;
    const int ci = ch, bsel = hs;
    // A^T with B^T's factors folded in: rows ky, columns f
    const float AT[4][5] = {{0.5f, 0.5f, 1.f / 6.f, -1.f / 6.f, 0.f},
                            {0.f, 0.5f, -1.f / 6.f, -2.f / 6.f, 0.f},
                            {0.f, 0.5f, 1.f / 6.f, -4.f / 6.f, 0.f},
                            {0.f, 0.5f, -1.f / 6.f, -8.f / 6.f, 1.f}};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * kq + r, co = m & 7, a = m >> 3, kz = 1 + 2 * bsel - a;
      float t[5][4];                                            // t[fy][kx] = sum_fx AT[kx][fx] M[fy][fx]
#pragma unroll
      for (int fy = 0; fy < 5; ++fy)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          float s = 0.f;
#pragma unroll
          for (int fx = 0; fx < 5; ++fx)
            if (AT[kx][fx] != 0.f) s = fmaf(AT[kx][fx], acc[fy * 5 + fx][r], s);
          t[fy][kx] = s;
        }
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          float s = 0.f;
#pragma unroll
          for (int fy = 0; fy < 5; ++fy)
            if (AT[ky][fy] != 0.f) s = fmaf(AT[ky][fy], t[fy][kx], s);
          const int o = (co * 8 + ci) * 64 + kz * 16 + ky * 4 + kx;
          reg[o + (o >> 6)] = s;
        }
    }
  }
  __syncthreads();
  float* slab = slabs + (size_t)bx * 4096;
  for (int o = tid; o < 4096; o += 256) {
    const int p = o + (o >> 6);
    slab[o] = ((lds[p] + lds[region + p]) + lds[2 * region + p]) + lds[3 * region + p];
  }
}
