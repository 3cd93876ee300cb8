// Matrix-core (v_mfma_f32_16x16x4_f32, exact fp32) forward of the stride-2, 5^3 transposed convolutions
// with 8 output channels and no padding (up1: 16 -> 8 channels, 8^3 -> 19^3; up2: 8 -> 8, 16^3 -> 35^3;
// F.conv_transpose3d, utils/network.py:621).
//
// Sub-pixel form: output o = 2c + e per axis (cell c, parity e) reads inputs i = c - j through taps
// k = e + 2j (j = 0..2 for e = 0, j = 0..1 for e = 1): all 125 taps do useful work, no inserted zeros.
// MFMA mapping: rows = (co, ex) -- the two x parities of a cell share their inputs, the odd one has a
// zero weight on jx = 2 (5 of 6 row-taps useful); K = four input channels; columns = 16 consecutive
// cells of the FLATTENED (cy, cx) cell plane.  The LDS image of an input plane has row stride = cells
// per row (= input width + 2), two zero words in front of every row and zero rows around it, so
//   address(cell p, jy, jx) = p - jy * NCELL - jx + const
// is linear in p (a row's overrun lands on the next row's zero words): any 16 consecutive cells are one
// conflict-free ds_read_b32, and 18- or 10-cell rows cost no padding columns.  The (ez, ey) parity
// classes are separate accumulators; one B fragment feeds every (plane parity, row parity) that uses it.
// Per output the accumulation order is fixed: (ci group, jy, jx, jz).
//
// REPAIR (training steps behind the engine's `winograd` switch; pack kind 12): the x-edge taps (jx = 2: kx = 4, which
// only the even outputs have) run on rows (co, ey) instead -- one product serves both row parities of a plane parity,
// 15 fragments per channel group instead of 25 half-empty ones (65 instead of 75 in all).  Their sums are kept in
// accumulators of their own and added to the even outputs in the epilogue, so an even output's sum ends with its
// kx = 4 taps: another order, other bits (evaluation, encode and decode keep the 75-fragment form).
#include "nvf_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef NVF_CT_DBG
#define NVF_CT_DBG 0     // tuning builds: 1 = no MFMAs, 2 = no epilogue stores, 4 = no A-fragment staging (bit mask)
#endif

namespace {

// number of A fragments per channel group: sum over classes (ez,ey) of (3-ez)(3-ey) * 3
constexpr int kAPerGroup = 75;

__host__ __device__ constexpr int a_index(int ez, int ey, int jz, int jy, int jx) {
  // classes in order (0,0) (0,1) (1,0) (1,1); inside a class [jz][jy][jx]
  const int base = ez == 0 ? (ey == 0 ? 0 : 27) : (ey == 0 ? 27 + 18 : 27 + 18 + 18);
  return base + (jz * (3 - ey) + jy) * 3 + jx;
}

constexpr int kAPerGroupR = 65;       // REPAIR: 50 (classes x [jz][jy][jx < 2]) + 15 x-edge fragments (ez; [jz][jy])
__host__ __device__ constexpr int a_index_r(int ez, int ey, int jz, int jy, int jx) {
  const int base = ez == 0 ? (ey == 0 ? 0 : 18) : (ey == 0 ? 30 : 42);
  return base + (jz * (3 - ey) + jy) * 2 + jx;
}
__host__ __device__ constexpr int a_index_edge(int ez, int jz, int jy) { return 50 + (ez ? 9 : 0) + jz * 3 + jy; }

__global__ void pack_convT_mfma_kernel(const float* __restrict__ wf /* [cin][125][8] */, float* __restrict__ wp,
                                       int cin) {
  const int total = (cin / 4) * kAPerGroup * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx % 64, f = (idx / 64) % kAPerGroup, g = idx / (64 * kAPerGroup);
    int ez = 0, ey = 0, r = f;
    if (r >= 27 + 18 + 18) { ez = 1; ey = 1; r -= 63; }
    else if (r >= 27 + 18) { ez = 1; r -= 45; }
    else if (r >= 27) { ey = 1; r -= 27; }
    const int jx = r % 3, jy = (r / 3) % (3 - ey), jz = r / (3 * (3 - ey));
    const int i = lane & 15, co = i >> 1, ex = i & 1, ci = 4 * g + (lane >> 4);
    const int kz = ez + 2 * jz, ky = ey + 2 * jy, kx = ex + 2 * jx;
    wp[idx] = kx < 5 ? wf[(ci * 125 + (kz * 5 + ky) * 5 + kx) * 8 + co] : 0.f;
  }
}

template <int CIN_, int NIN_, int NCT_, int NSPLIT_, int NW_ = 4, bool REPAIR_ = false>
struct TMCfg {
  static constexpr int CIN = CIN_, NIN = NIN_, NCT = NCT_, NSPLIT = NSPLIT_;
  static constexpr bool REPAIR = REPAIR_;
  static constexpr int KA = REPAIR_ ? kAPerGroupR : kAPerGroup;   // A fragments per channel group
  static constexpr int NCELL = NIN + 2;                        // cells per axis; outputs 2 NIN + 3
  static constexpr int NPT = (NCELL * NCELL + 15) / 16;        // column tiles of a cell plane
  static constexpr int NW = NW_, NTH = NW_ * 64, CPW = NW * NCT;  // waves, threads, column tiles per workgroup
  static_assert(CPW * NSPLIT >= NPT, "the splits cover the plane");
  static constexpr int PLANE = (NCELL + 3) * NCELL + 18;       // LDS words per (channel, plane), zero margins
  static constexpr int cs_for(int v) { while (v % 32 != 16) ++v; return v; }
  static constexpr int CS = cs_for(3 * PLANE);                 // channel stride: second channel -> banks 16..31
  static constexpr int NG = CIN / 4;
  static constexpr int XS = CIN * CS;                          // input image
  static constexpr int AS = NG * KA * 64;                      // A fragments
  static_assert((XS + AS) * 4 <= 160 * 1024, "LDS");
};

// Item -> (batch element, cell plane, column split).  A cell plane cz reads the input planes cz - jz, jz = 0..2: the two
// first and two last cell planes have one or two of them outside the input, and those (ez, jz) blocks are skipped
// (cost_of() fifteenths of a full plane: 6, 12 / 9, 3).  Items are ordered by cost, heaviest first, so that whatever hands
// them out -- the dispatcher when the grid covers the items, the stride loop below otherwise -- ends on the short ones.
template <int NIN>
__device__ __forceinline__ void convT_item(int item, int nsplit, int batch, int& b, int& cz, int& split) {
  constexpr int NCELL = NIN + 2;
  split = item % nsplit;
  b = (item / nsplit) % batch;
  const int k = item / (nsplit * batch);                       // rank by cost: NIN - 2 full planes, then 1, NIN, 0, NIN + 1
  cz = k < NIN - 2 ? k + 2 : (k == NIN - 2 ? 1 : (k == NIN - 1 ? NIN : (k == NIN ? 0 : NCELL - 1)));
}

// Persistent workgroups: the A fragments are fetched once per workgroup; a workgroup then walks the items
// w, w + G, w + 2 G ... with the next item's three input planes already
// in registers while this item's MFMAs issue (two workgroups share a CU and fill each other's barriers).
template <class T>
__global__ __launch_bounds__(T::NTH) void convT_k5s2_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ y, int act,
                                                       int items, int batch) {
  constexpr int CIN = T::CIN, NIN = T::NIN, NCELL = T::NCELL, NCT = T::NCT, NPT = T::NPT, PLANE = T::PLANE, CS = T::CS,
                NG = T::NG, NOUT = 2 * NIN + 3, NTH = T::NTH;
  __shared__ __attribute__((aligned(16))) float xs[T::XS];
  __shared__ __attribute__((aligned(16))) float as[T::AS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- stage: every global load (A fragments, the three input planes cz-2 .. cz of every channel of the first
  // item) is issued before anything is waited for; the image is zeroed while they are in flight
  constexpr int NA4 = (T::AS / 4 + NTH - 1) / NTH;             // float4 A loads per thread
  constexpr int ITEMS = CIN * 3 * NIN * NIN / 4;               // float4 input loads (rows are NIN = 8 or 16 floats)
  constexpr int NX4 = (ITEMS + NTH - 1) / NTH;
  float4 xv[NX4];
  auto load_x = [&](int item) {
    int b, cz, split_;
    convT_item<NIN>(item, T::NSPLIT, batch, b, cz, split_);
    const float* xb = x + (size_t)b * CIN * NIN * NIN * NIN;
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * NTH;
      const int xq = i % (NIN / 4), iy = (i / (NIN / 4)) % NIN, pl = (i / (NIN / 4 * NIN)) % 3, c = i / (NIN / 4 * NIN * 3);
      const int zi = cz - 2 + pl;
      const bool ok = i < ITEMS && zi >= 0 && zi < NIN;
      xv[u] = ok ? *(const float4*)(xb + (((size_t)c * NIN + zi) * NIN + iy) * NIN + 4 * xq)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_x = [&]() {                                       // planes outside the input are written as zeros
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * NTH;
      if (i < ITEMS) {
        const int xq = i % (NIN / 4), iy = (i / (NIN / 4)) % NIN, pl = (i / (NIN / 4 * NIN)) % 3, c = i / (NIN / 4 * NIN * 3);
        float* dst = xs + c * CS + pl * PLANE + (iy + 2) * NCELL + 4 * xq + 2;
        dst[0] = xv[u].x; dst[1] = xv[u].y; dst[2] = xv[u].z; dst[3] = xv[u].w;
      }
    }
  };
  {
    // the A fragments go L2 -> LDS by DMA (1 KB per wave instruction, no registers); the first item's planes and the zero
    // fill of the image pass under their latency
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    if (!(NVF_CT_DBG & 4)) {
#pragma unroll
      for (int u = 0; u < NA4; ++u) {
        const int i = tid + u * NTH;
        if (i < T::AS / 4)
          __builtin_amdgcn_global_load_lds((glb_vp)(wp + (size_t)i * 4), (lds_vp)(as + (u * NTH + wave * 64) * 4), 16, 0, 0);
      }
    }
    if ((int)blockIdx.x < items) load_x(blockIdx.x);
    for (int i = tid * 4; i < T::XS; i += NTH * 4) *(float4*)(xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the first __syncthreads of the item loop publishes it)
  }
  const int j = lane & 15, kq = lane >> 4;
  const size_t cstride = (size_t)NOUT * NOUT * NOUT;
  // ---- epilogue of one item: lane holds rows i = 4 kq + r -> co = 2 kq + (r >> 1), ex = r & 1 of cell p
  auto epilogue = [&](int item, const f32x4 (&res)[NCT][2][2], const f32x4 (&edge)[NCT][2]) {
    int b, cz, split;
    convT_item<NIN>(item, T::NSPLIT, batch, b, cz, split);
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int tl = split * T::CPW + c * T::NW + wave;
      const int p = 16 * tl + j;
      if (tl >= NPT || p >= NCELL * NCELL) continue;
      const int cy = p / NCELL, cx = p % NCELL;
#pragma unroll
      for (int ez = 0; ez < 2; ++ez)
#pragma unroll
        for (int ey = 0; ey < 2; ++ey) {
          const int oz = 2 * cz + ez, oy = 2 * cy + ey;
          if (oz >= NOUT || oy >= NOUT) continue;
#pragma unroll
          for (int r = 0; r < 4; r += 2) {        // rows r, r + 1 = the two x parities of a cell: one 8-byte store
            const int co = 2 * kq + (r >> 1), ox = 2 * cx;
            const float bv = bias ? bias[co] : 0.f;
            float* o = y + ((size_t)b * 8 + co) * cstride + ((size_t)oz * NOUT + oy) * NOUT + ox;
            // (REPAIR: the lane's x-edge rows are (co, ey) of the same two channels: row r + ey)
            const float even = T::REPAIR ? res[c][ez][ey][r] + edge[c][ez][r + ey] : res[c][ez][ey][r];
            const float v0 = nvf_act(even + bv, act), v1 = nvf_act(res[c][ez][ey][r + 1] + bv, act);
            if (NVF_CT_DBG & 2) { if (v0 == 12345.f) o[0] = v1; continue; }
            if (ox + 1 < NOUT) *(nvf_f2u*)o = nvf_f2u{v0, v1};
            else if (ox < NOUT) o[0] = v0;
          }
        }
    }
  };
  // Round r of the stride loop hands out items r G .. r G + G - 1 (G workgroups, sorted by cost).  Workgroups w and
  // w + G / 2 share a CU (the dispatcher fills the CUs once, then a second time), so odd rounds run the CUs backwards on
  // their second workgroup: a CU's second-round items are the k-th heaviest and the k-th lightest, and with the items of
  // this kernel (160 full planes + 4 x 48 partial ones at batch 16) every CU ends with the same MFMA count
  const int G = gridDim.x, half = G >> 1, w = blockIdx.x;
  auto item_of = [&](int r) {
    const int k = ((r & 1) && !(G & 1)) ? (w < half ? w : G + half - 1 - w) : w;
    return r * G + k;
  };
  const int rounds = (items + G - 1) / G;
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    const int item = item_of(r);
    if (item >= items) break;                                  // only the last round is partial
    const int split = item % T::NSPLIT;
    __syncthreads();                                           // zero fill done / the previous item's reads done
    store_x();
    __syncthreads();
    if (item_of(r + 1) < items) load_x(item_of(r + 1));
    f32x4 acc[NCT][2][2];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e >> 1][e & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 edge[NCT][2];                                        // REPAIR only (else never touched: no registers)
#pragma unroll
    for (int c = 0; c < NCT; ++c) edge[c][0] = edge[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int colbase[NCT];
    // YM: bit jy set = some cell of one of this wave's column tiles reads an input row that exists (cy - jy in [0, NIN)).
    // The first / last tiles of a cell plane lie in the rows cy = 0, 1 / NIN, NIN + 1, where one or two of the three jy taps
    // read only the zero margin: those MFMAs are skipped (2 of 21 tiles' worth for up2, 1.3 of 7 for up1) -- exact zeros,
    // so the sums keep their bits and their order, like the plane mask below.  Wave-uniform.
    int ym = 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int tl = split * T::CPW + c * T::NW + wave;          // wave-uniform
      // cell p = 16 tl + j reads plane word (cy - jy + 2) NCELL + cx - jx + 2 = p + 2 NCELL + 2 - jy NCELL - jx
      // (a column slot past the last tile of the plane computes on the last tile's data and stores nothing)
      colbase[c] = kq * CS + 16 * min(tl, NPT - 1) + j + 2 * NCELL + 2;
      const int tt = min(tl, NPT - 1), cy_lo = (16 * tt) / NCELL, cy_hi = min(16 * tt + 15, NCELL * NCELL - 1) / NCELL;
#pragma unroll
      for (int jy = 0; jy < 3; ++jy)
        if (cy_hi - jy >= 0 && cy_lo - jy <= NIN - 1) ym |= 1 << jy;
    }
    ym = __builtin_amdgcn_readfirstlane(ym);
    // MASK: bit jz set = input plane cz - jz exists.  A skipped block would have added exact zeros, so the sums are the
    // same bits as the unmasked loop's; the order of the remaining terms is unchanged
    auto mfma_phase = [&](auto maskc) {
      constexpr int MASK = decltype(maskc)::value;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        const float* xg = xs + g * 4 * CS;
        const float* ag = as + g * T::KA * 64 + lane;
#pragma unroll
        for (int jy = 0; jy < 3; ++jy) {
          if (!((ym >> jy) & 1)) continue;                     // (scalar branch around 9 - 27 MFMAs)
#pragma unroll
          for (int jx = 0; jx < 3; ++jx)
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) {
              if (!((MASK >> jz) & 1)) continue;
              if (T::REPAIR && jx == 2) {                      // x-edge taps: rows (co, ey), one product per plane parity
                float ae[2];
#pragma unroll
                for (int ez = 0; ez < 2; ++ez) ae[ez] = jz <= 2 - ez ? ag[a_index_edge(ez, jz, jy) * 64] : 0.f;
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                  const float bv = xg[colbase[c] + (2 - jz) * PLANE - jy * NCELL - jx];
#pragma unroll
                  for (int ez = 0; ez < 2; ++ez)
                    if (jz <= 2 - ez && !(NVF_CT_DBG & 1))
                      edge[c][ez] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[ez], bv, edge[c][ez], 0, 0, 0);
                }
                continue;
              }
              float a[2][2];
#pragma unroll
              for (int ez = 0; ez < 2; ++ez)
#pragma unroll
                for (int ey = 0; ey < 2; ++ey)
                  a[ez][ey] = (jz <= 2 - ez && jy <= 2 - ey)
                                  ? ag[(T::REPAIR ? a_index_r(ez, ey, jz, jy, jx) : a_index(ez, ey, jz, jy, jx)) * 64] : 0.f;
#pragma unroll
              for (int c = 0; c < NCT; ++c) {
                const float bv = xg[colbase[c] + (2 - jz) * PLANE - jy * NCELL - jx];   // input plane cz - jz
#pragma unroll
                for (int ez = 0; ez < 2; ++ez)
#pragma unroll
                  for (int ey = 0; ey < 2; ++ey)
                    if (jz <= 2 - ez && jy <= 2 - ey && !(NVF_CT_DBG & 1))
                      acc[c][ez][ey] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ez][ey], bv, acc[c][ez][ey], 0, 0, 0);
              }
            }
        }
      }
    };
    {
      int b_, cz, split_;
      convT_item<NIN>(item, T::NSPLIT, batch, b_, cz, split_);
      const int mask = (cz < NIN ? 1 : 0) | ((cz >= 1 && cz <= NIN) ? 2 : 0) | (cz >= 2 ? 4 : 0);     // wave-uniform
      switch (mask) {
        case 7: mfma_phase(std::integral_constant<int, 7>{}); break;
        case 3: mfma_phase(std::integral_constant<int, 3>{}); break;
        case 6: mfma_phase(std::integral_constant<int, 6>{}); break;
        case 1: mfma_phase(std::integral_constant<int, 1>{}); break;
        default: mfma_phase(std::integral_constant<int, 4>{}); break;
      }
    }
    epilogue(item, acc, edge);
  }
}

}  // namespace

extern "C" size_t nvf_pack_convT_mfma_floats(int cin) { return (size_t)(cin / 4) * kAPerGroup * 64; }

// w_fwd = the [cin][125][8] packed forward weight of a transposed convolution with 8 output channels
extern "C" int nvf_pack_convT_mfma(const float* w_fwd, int cin, int cout, float* wp, void* stream) {
  if (!w_fwd || !wp || cin <= 0 || cin % 4 || cout != 8) return NVF_EINVAL;
  const int total = (int)nvf_pack_convT_mfma_floats(cin);
  pack_convT_mfma_kernel<<<(total + 255) / 256, 256, 0, nvf_stream(stream)>>>(w_fwd, wp, cin);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// y[b,co,o] = act(bias[co] + sum_{ci,k : o - k = 2 i} x[b,ci,i] w[ci][k][co]), padding 0, dout = 2 din + 3.
// NVF_EINVAL = no instantiation for this shape (the caller then uses nvf_convT3d_k5s2_fwd).
// workgroups per launch: 512 = two resident per CU, each walking items with a stride; NVF_CT_CAP (tuning) raises it so
// that the dispatcher hands the items out one workgroup each
static int convT_cap() {
  const int cap = nvf_tune_int("NVF_CT_CAP", 512);
  return cap < 1 ? 512 : cap;
}

extern "C" int nvf_convT3d_k5s2_mfma(const float* x, const float* wp, const float* bias, float* y, int batch, int cin,
                                     int cout, int din, int act, int variant, void* stream) {
  if (!x || !wp || !y || batch <= 0 || cout != 8) return NVF_EINVAL;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
#define NVF_TM(VAR, CI, NIN, NCT, NSPLIT, ...)                                                         \
  if (rc == 1 && variant == VAR && cin == CI && din == NIN) {                                          \
    using T = TMCfg<CI, NIN, NCT, NSPLIT, ##__VA_ARGS__>;                                              \
    const int items = batch * T::NCELL * NSPLIT;                                                       \
    const int cap = convT_cap();                                                                       \
    convT_k5s2_mfma<T><<<items < cap ? items : cap, T::NTH, 0, s>>>(x, wp, bias, y, act, items, batch); \
    rc = NVF_OK;                                                                                       \
  }
  NVF_TM(0, 8, 16, 2, 3)     // up2: 21 column tiles per cell plane, 8 per workgroup
  NVF_TM(0, 16, 8, 2, 1)     // up1: all 7 column tiles of a cell plane in one workgroup
  NVF_TM(2, 8, 16, 3, 2)
  NVF_TM(3, 8, 16, 6, 1)
  NVF_TM(4, 8, 16, 1, 6)
  NVF_TM(2, 16, 8, 1, 2)
  NVF_TM(5, 16, 8, 1, 1, 8)  // up1: eight waves, one column tile each (two waves per SIMD instead of one on 160 CUs)
  NVF_TM(5, 8, 16, 1, 3, 8)  // up2: eight waves x one tile, three splits
  NVF_TM(6, 8, 16, 2, 2, 8)  // up2: eight waves x two tiles = 16 of 21 column tiles, two splits (11 empty slots of 32)
  NVF_TM(15, 16, 8, 1, 1, 8, true)   // variant 5 with the x-edge taps on rows (co, ey): wp = pack kind 12 (training steps)
  NVF_TM(15, 8, 16, 1, 3, 8, true)
#undef NVF_TM
  if (rc == 1) return NVF_EINVAL;
  NVF_LAUNCH_CHECK();
  return rc;
}
