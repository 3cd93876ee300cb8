// Weight gradients of the one-channel classifier heads (conv0_cls / conv1_cls / conv2_cls: Conv3d(C -> 1, k = 3,
// padding 1), utils/network.py:735-742, 677-687; autograd's bwd-weight of :741 / :687) on the matrix cores:
//
//   dW[c][t] = sum_{b, p} x[b, c, p] * dl[b, p - t + 1]          t = (kz, ky, kx), p = (z, y, x), dl = d loss / d logit
//
// One output channel gives an implicit GEMM nothing to put in its rows -- unless the roles are swapped: the INPUT
// position is the reduction index and the 27 taps are the columns (the scatter form):
//
//   D[c][t] += sum_{k=0..3} A[c][k] * B[k][t]      A[c][k] = x[c, z, y, x0 + k]          rows = C channels (8 or 16)
//                                                   B[k][t] = dl[z - kz + 1, y - ky + 1, x0 + k - kx + 1]   (0 outside)
//
// with K = four consecutive x positions and the taps in two 16-column tiles (27 of 32 columns useful).  Per four
// input positions and C channels that is two v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chains) fed by ONE A read and two
// B reads from LDS; the VALU kernel this replaces spent ~190 instructions per 108 FMAs (heads.hip).  The x tile has no
// halo (every element is read once), the dl tile a one-voxel halo of zeros.
//
// A workgroup (4 waves) walks items = (batch element, TZ planes, TY rows, all S columns); both tiles of the NEXT item are
// in registers (buffer loads: scalar descriptor + scalar offsets + one per-thread voffset) while this item's MFMAs issue.
// Its four waves split an item's K groups; their partial D tiles are added in wave order through LDS at the end and
// leave as ONE slab of C * 27 floats per workgroup (added in a fixed order by nvf_wgrad_reduce_multi): no atomics.
#pragma once
#include "nvf_common.h"

typedef float hw4 __attribute__((ext_vector_type(4)));
typedef unsigned hwu4 __attribute__((ext_vector_type(4)));

constexpr int kHwOob = 0x7ffffff0;      // a voffset beyond every descriptor's range: the load returns 0

template <int C_, int S_, int TZ_, int TY_>
struct HMCfg {
  static constexpr int C = C_, S = S_, TZ = TZ_, TY = TY_;
  static_assert(C == 8 || C == 16, "rows of the MFMA tile");
  static_assert(S % 4 == 0 && (TY * S) % 4 == 0, "float4 staging");
  static constexpr int mod32(int v, int r) { return v + ((r - v % 32) + 32) % 32; }
  // x tile [c][z][y][x]: the 32 lanes of a read group are 16 channels x 2 consecutive words -> channel stride == 2 (mod 32)
  static constexpr int ACS = mod32(TZ * TY * S, 2);
  // dl tile [(TZ + 2)][(TY + 2)][S + 2], word 0 of a row = x -1
  static constexpr int BRS = S + 2, BPS = (TY + 2) * BRS;
  static constexpr int AW = C * ACS, BW = (TZ + 2) * BPS;
  static constexpr int BOFF = (AW + 3) / 4 * 4;
  static constexpr int LDSF = BOFF + BW;
  static constexpr int KG = TZ * TY * (S / 4);          // K groups (four x positions) per item
  static_assert(KG % 4 == 0, "K groups split evenly over four waves");
  static constexpr int SEG4 = TY * S / 4;               // float4s of one (channel, plane) segment of the x tile
  static constexpr int NA4 = C * TZ * SEG4;             // float4s of the x tile
  static constexpr int UA = (NA4 + 255) / 256;
  static constexpr int NB4 = (TZ + 2) * (TY + 2) * (S / 4);   // float4s of the dl tile (rows of S words)
  static constexpr int UB = (NB4 + 255) / 256;
  static_assert(256 % SEG4 == 0 || SEG4 % 256 == 0, "a thread's x float4s differ by whole segments");
  static constexpr int EPI = 4 * 2 * 256;               // epilogue: four waves x two tiles x 256 sums
  static constexpr int SMEM = LDSF > EPI ? LDSF : EPI;
};

template <class H>
__device__ __forceinline__ void head_wgrad_mfma_body(const float* __restrict__ dl, const float* __restrict__ x,
                                                     float* __restrict__ slabs, int items, int items_per_wg, int bx,
                                                     float* lds, int ct = H::C) {
  // ct: channels of the tensor x / rows of a slab (> C when this call handles one group of C channels of a wider head:
  // x and slabs then point at the group's first channel / row)
  constexpr int C = H::C, S = H::S, TZ = H::TZ, TY = H::TY, ACS = H::ACS, BRS = H::BRS, BPS = H::BPS, UA = H::UA,
                UB = H::UB, SEG4 = H::SEG4;
  float* la = lds;
  float* lb = lds + H::BOFF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, k = lane >> 4;
  constexpr int TILES_Y = S / TY, TILES_Z = S / TZ, TILES = TILES_Y * TILES_Z;
  const int first = bx * items_per_wg, last = min(first + items_per_wg, items);
  // ---- per-thread constants of the staging (the same for every item)
  // x tile: float4 i = tid + u * 256 -> (xq, y, z, c); its global word inside the batch element and its LDS word
  int a_voff[UA], a_lds[UA];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int i = tid + u * 256;
    const int xq = i % (S / 4), r = i / (S / 4), yy = r % TY, t2 = r / TY, zz = t2 % TZ, c = t2 / TZ;
    const bool live = i < H::NA4;
    a_voff[u] = live ? (((c * S + zz) * S + yy) * S + 4 * xq) * 4 : kHwOob;
    a_lds[u] = live ? c * ACS + (zz * TY + yy) * S + 4 * xq : -1;
  }
  // dl tile: float4 i -> (xq, row yi of TY + 2, plane zi of TZ + 2)
  int b_rel[UB], b_lds[UB], b_zy[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int i = tid + u * 256;
    const int xq = i % (S / 4), r = i / (S / 4), yi = r % (TY + 2), zi = r / (TY + 2);
    const bool live = i < H::NB4;
    b_rel[u] = (((zi - 1) * S + (yi - 1)) * S + 4 * xq) * 4;      // bytes from the tile's (z0, y0, 0) element
    b_lds[u] = live ? zi * BPS + yi * BRS + 1 + 4 * xq : -1;
    b_zy[u] = live ? (zi << 8) | yi : -1;
  }
  // ---- per-lane constants of the MFMA operands
  // A: row i16 = channel (rows >= C are zero), K index k: word c * ACS + (z * TY + y) * S + x0 + k
  const bool a_row = i16 < C;
  const int a_lane = (a_row ? i16 : 0) * ACS + k;
  // B: column i16 of tile tt = tap t = 16 tt + i16: word (z - kz + 2) * BPS + (y - ky + 2) * BRS + x0 + k - kx + 2
  int b_lane[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = min(16 * tt + i16, 26);                        // columns 27..31 repeat tap 26 (never stored)
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    b_lane[tt] = (2 - kz) * BPS + (2 - ky) * BRS + k - kx + 2;
  }
  hw4 acc[2] = {hw4{0.f, 0.f, 0.f, 0.f}, hw4{0.f, 0.f, 0.f, 0.f}};
  hwu4 av[UA], bv[UB];
  auto load = [&](int item) {
    const int tile = item % TILES, b = item / TILES;
    const int y0 = (tile % TILES_Y) * TY, z0 = (tile / TILES_Y) * TZ;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * ct * S * S * S), 0,
                                                                        C * S * S * S * 4, 0x00020000);
    const int xs = ((z0 * S + y0) * S) * 4;
#pragma unroll
    for (int u = 0; u < UA; ++u) av[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[u], xs, 0);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dl + (size_t)b * S * S * S), 0,
                                                                        S * S * S * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int gz = z0 - 1 + (b_zy[u] >> 8), gy = y0 - 1 + (b_zy[u] & 255);
      const bool ok = b_zy[u] >= 0 && gz >= 0 && gz < S && gy >= 0 && gy < S;
      // the whole byte offset goes into voffset (it can be negative relative to the tile: no scalar part)
      bv[u] = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? xs + b_rel[u] : kHwOob, 0, 0);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < UA; ++u)
      if (a_lds[u] >= 0) {
        float* d = la + a_lds[u];                                 // 8-byte aligned (ACS is even): two float2 stores
        *(float2*)d = make_float2(__uint_as_float(av[u].x), __uint_as_float(av[u].y));
        *(float2*)(d + 2) = make_float2(__uint_as_float(av[u].z), __uint_as_float(av[u].w));
      }
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (b_lds[u] >= 0) {
        float* d = lb + b_lds[u];
        d[0] = __uint_as_float(bv[u].x); d[1] = __uint_as_float(bv[u].y);
        d[2] = __uint_as_float(bv[u].z); d[3] = __uint_as_float(bv[u].w);
      }
  };
  for (int i = tid; i < H::LDSF; i += 256) lds[i] = 0.f;          // the x = -1 / x = S columns of the dl tile stay zero
  if (first < last) load(first);
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    __syncthreads();                                              // zero fill done / the previous item's reads done
    store();
    __syncthreads();
    if (item + 1 < last) load(item + 1);
    // this wave's K groups: g = wave + 4 i -> (x group, row y, plane z) of the tile
    constexpr int PER = H::KG / 4;
    float a_cur, b0_cur, b1_cur;
    auto operands = [&](int g, float& a, float& b0, float& b1) {
      const int xg = g % (S / 4), r = g / (S / 4);                // r = z * TY + y
      const int yy = r % TY, zz = r / TY;
      const float av_ = la[a_lane + r * S + 4 * xg];
      a = a_row ? av_ : 0.f;
      const int bb = zz * BPS + yy * BRS + 4 * xg;
      b0 = lb[b_lane[0] + bb];
      b1 = lb[b_lane[1] + bb];
    };
    operands(wave, a_cur, b0_cur, b1_cur);
#pragma unroll 4
    for (int i = 0; i < PER; ++i) {
      float a_nxt = 0.f, b0_nxt = 0.f, b1_nxt = 0.f;
      if (i + 1 < PER) operands(wave + 4 * (i + 1), a_nxt, b0_nxt, b1_nxt);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b0_cur, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b1_cur, acc[1], 0, 0, 0);
      a_cur = a_nxt; b0_cur = b0_nxt; b1_cur = b1_nxt;
    }
  }
  // ---- the four waves' partial tiles, added in wave order; lane holds rows 4 (lane >> 4) + r, column i16
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[(wave * 2 + tt) * 256 + (4 * k + r) * 16 + i16] = acc[tt][r];
  __syncthreads();
  float* slab = slabs + (size_t)bx * (ct * 27);
  for (int o = tid; o < C * 27; o += 256) {
    const int c = o / 27, t = o % 27, tt = t >> 4, col = t & 15;
    const int w = tt * 256 + c * 16 + col;
    slab[o] = ((lds[w] + lds[512 + w]) + lds[1024 + w]) + lds[1536 + w];
  }
}


// the three heads of one decoder as block ranges of a launch: workgroups [0, n2) the big head, then head 1, then head 0
struct HeadsW3 {
  const float* dl[3];
  const float* x[3];
  float* slabs[3];
  int32_t n[3], items[3], per[3];     // n[h]: workgroups of head h (slabs x channel groups)
};
constexpr int hmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// G0: channel groups of head 0 (the wide decoder's first head has 32 channels = two row tiles of H0::C = 16)
template <class H0, class H1, class H2, int G0 = 1>
__device__ __forceinline__ void heads3_wgrad_mfma_dispatch(const HeadsW3& m, int bid, float* lds) {
  if (bid < m.n[2]) { head_wgrad_mfma_body<H2>(m.dl[2], m.x[2], m.slabs[2], m.items[2], m.per[2], bid, lds); return; }
  bid -= m.n[2];
  if (bid < m.n[1]) { head_wgrad_mfma_body<H1>(m.dl[1], m.x[1], m.slabs[1], m.items[1], m.per[1], bid, lds); return; }
  bid -= m.n[1];
  constexpr int S3 = H0::S * H0::S * H0::S;
  const int per_group = m.n[0] / G0, g = bid / per_group;
  head_wgrad_mfma_body<H0>(m.dl[0], m.x[0] + (size_t)g * H0::C * S3, m.slabs[0] + g * H0::C * 27, m.items[0], m.per[0],
                           bid - g * per_group, lds, H0::C * G0);
}

// geometry of the three jobs (nslabs[h] = slabs of head h; head 0 runs G0 workgroups per slab)
template <class H0, class H1, class H2, int G0 = 1>
static inline int heads3_wgrad_mfma_fill(HeadsW3& m, const float* const* dls, const float* const* xs,
                                         float* const* slabs, int batch, int max_slabs, int* nslabs) {
  const int items[3] = {batch * (H0::S / H0::TZ) * (H0::S / H0::TY), batch * (H1::S / H1::TZ) * (H1::S / H1::TY),
                        batch * (H2::S / H2::TZ) * (H2::S / H2::TY)};
  for (int h = 0; h < 3; ++h) {
    if (!dls[h] || !xs[h] || !slabs[h]) return NVF_EINVAL;
    m.dl[h] = dls[h]; m.x[h] = xs[h]; m.slabs[h] = slabs[h];
    int n = items[h] < max_slabs ? items[h] : max_slabs;
    const int per = (items[h] + n - 1) / n;
    n = (items[h] + per - 1) / per;
    m.n[h] = h == 0 ? n * G0 : n; m.items[h] = items[h]; m.per[h] = per;
    nslabs[h] = n;
  }
  return NVF_OK;
}

// the narrow decoder's heads: conv0_cls [16, 8^3], conv1_cls [8, 16^3], conv2_cls [8, 32^3]
using HeadW0 = HMCfg<16, 8, 4, 8>;
using HeadW1 = HMCfg<8, 16, 4, 8>;
using HeadW2 = HMCfg<8, 32, 2, 8>;
// the wide decoder's: conv0_cls [32, 8^3] (two groups of 16 channels), conv1_cls [16, 16^3], conv2_cls [16, 32^3]
using HeadWw0 = HMCfg<16, 8, 4, 8>;
using HeadWw1 = HMCfg<16, 16, 4, 8>;
using HeadWw2 = HMCfg<16, 32, 2, 8>;
