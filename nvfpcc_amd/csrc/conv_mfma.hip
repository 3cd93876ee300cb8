// MFMA (matrix-core) 4x4x4 convolutions of the NVF decoder trunk on gfx950: conv1 / conv2 forward
// (F.conv3d, utils/network.py:687) and their autograd backward-data passes, for 8 -> 8 channels.
//
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain at the fp32 vector rate, and unlike the VALU it
// reaches that rate from ONE wave per SIMD -- which is all a batch of 16 blocks can put on 256 CUs.
// With Cout = 8 a plain implicit GEMM would leave half of the 16 tile rows empty; instead the rows are
// (co, s) where s in {0,1} picks one of two outputs that are ADJACENT along a "pair axis", and the
// reduction runs over a 5-wide window along that axis (each row uses 4 of the 5 taps, its weights are
// zero on the fifth), so 80 % of every MFMA is useful work:
//
//   D[(co,s)][col] += sum_{k=0..3} A[(co,s)][k] * B[k][col]          k = 4 consecutive input channels
//   pair axis x (forward):        col = 16 cells m of one row, output x = 2m + s, B = in[ci][z+tz][y+ty][2m + tx]
//                                 A = W[co][ci][tz][ty][tx - s],  tx in 0..4
//   pair axis z (backward-data):  col = 4x4 patch of (y, x),     output z = 2q + s, B = in[ci][2q + tz][y+ty][x+tx]
//                                 A = W[co][ci][tz - s][ty][tx],  tz in 0..4
//
// (backward-data pairs along z because its 35- / 19-wide rows do not split into 16-cell tiles, while
// 36 = 9 * 4 and 20 = 5 * 4 patches do.)  A wave owns NC columns of NT tiles stacked along z; one B
// fragment (a ds_read_b32 per lane) feeds every tile/tap pair of the column that touches that input
// plane, so there are ~0.4 LDS reads per MFMA.  The A fragments come pre-packed (nvf_pack_mfma_k4)
// and stay in registers; four input channels of the tile are staged per step by LDS-DMA, with the
// channel stride chosen so the 32 lanes of a ds_read group hit 32 distinct banks.
// Per output the accumulation order is fixed (ci group, ty, tx, tz), independent of batch and tile.
#include "nvf_common.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MDims {
  int din, hin, win, dout, hout, wout, pad, act, tiles_x, tiles_y, tiles_z;
  int dbg;   // tuning runs only (variant >= 100): 1 = no staging, 2 = no MFMAs; results are then meaningless
             // (bit 2, NVF_A_GLOBAL=1: every wave fetches its A fragments from global memory itself; bit 3: every tile
             //  is staged as if it lay inside the tensor -- reads around it, the caller owns that memory)
  float* bias_part;   // EPI 1 only, optional: per (workgroup, wave) the 8 channel sums of the outputs it stored -- the
                      // bias gradient of the layer below (its masked output gradient is what this pass writes)
};

// ---------------------------------------------------------------------------------------------------
// A-fragment packing.  gw is a gather-form weight [cin][64 taps][8] (the w_fwd layout of a conv, or its
// w_bwd layout for the backward-data pass).  Output: wp[g][ty][tx][lane][tz (padded to KEZP)].
// ---------------------------------------------------------------------------------------------------
extern "C" size_t nvf_pack_mfma_k4_floats(int cin, int pair_axis) {
  return (size_t)(cin / 4) * 4 * (pair_axis == 0 ? 5 : 4) * (pair_axis == 2 ? 5 : 4) * 64;   // 5120 per channel group
}

extern "C" int nvf_pack_mfma_k4_multi(const float* const* gather_ws, float* const* wps, const int* cins,
                                      const int* pair_axes, int n, void* stream);

extern "C" int nvf_pack_mfma_k4(const float* gather_w, int cin, int cout, int pair_axis, float* wp, void* stream) {
  if (cout != 8) return NVF_EINVAL;
  return nvf_pack_mfma_k4_multi(&gather_w, &wp, &cin, &pair_axis, 1, stream);     // one job of the multi-packer
}

// several packings in one launch (the step engine packs conv1 / conv2, forward and backward, after every
// weight preparation)
struct PackMulti {
  const float* gw[8];
  float* wp[8];
  int32_t cin[8], pair[8];
};
__global__ void pack_mfma_k4_multi_kernel(PackMulti m) {
  const int job = blockIdx.y;
  const float* __restrict__ gw = m.gw[job];
  float* __restrict__ wp = m.wp[job];
  const int cin = m.cin[job], pair = m.pair[job];
  const int KEZ = pair == 2 ? 5 : 4, KEX = pair == 0 ? 5 : 4;
  const int total = (cin / 4) * 4 * KEX * KEZ * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int lane = r % 64; r /= 64;
    const int tz = r % KEZ; r /= KEZ;
    const int tx = r % KEX; r /= KEX;
    const int ty = r % 4;
    const int g = r / 4;
    const int i = lane & 15, cog = i >> 1, s = i & 1, ci = 4 * g + (lane >> 4);
    const int kz = pair == 2 ? tz - s : tz, ky = ty, kx = pair == 0 ? tx - s : tx;
    float v = 0.f;
    if (kz >= 0 && kz < 4 && kx >= 0 && kx < 4) v = gw[(ci * 64 + (kz * 4 + ky) * 4 + kx) * 8 + cog];
    wp[idx] = v;
  }
}

extern "C" int nvf_pack_mfma_k4_multi(const float* const* gather_ws, float* const* wps, const int* cins,
                                      const int* pair_axes, int n, void* stream) {
  if (!gather_ws || !wps || !cins || !pair_axes || n <= 0 || n > 8) return NVF_EINVAL;
  PackMulti m{};
  for (int i = 0; i < n; ++i) {
    if (!gather_ws[i] || !wps[i] || cins[i] <= 0 || cins[i] % 4 || (pair_axes[i] != 0 && pair_axes[i] != 2))
      return NVF_EINVAL;
    m.gw[i] = gather_ws[i]; m.wp[i] = wps[i]; m.cin[i] = cins[i]; m.pair[i] = pair_axes[i];
  }
  pack_mfma_k4_multi_kernel<<<dim3(8, n), 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------------
template <int CIN_, int PAIR_, int CTY_, int CTX_, int RY_, int RX_, int NWC_, int NWZ_, int NT_>
struct MCv {
  static constexpr int CIN = CIN_, PAIR = PAIR_, CTY = CTY_, CTX = CTX_, RY = RY_, RX = RX_, NT = NT_;
  static constexpr int NWC = NWC_, NWZ = NWZ_, NW = NWC_ * NWZ_;   // waves split the columns (NWC) and z (NWZ)
  static_assert(CTY * CTX == 16, "a column tile has 16 columns");
  static_assert((RY * RX) % NWC == 0, "columns split evenly over the waves");
  static_assert(CIN % 4 == 0, "K runs over groups of four input channels");
  static constexpr int NC = RY * RX / NWC;             // tile columns per wave
  static constexpr int XS = PAIR == 0 ? 2 : 1;         // x step between neighbouring cells
  static constexpr int ZS = PAIR == 2 ? 2 : 1;         // z step between stacked tiles
  static constexpr int KEZ = PAIR == 2 ? 5 : 4, KEY = 4, KEX = PAIR == 0 ? 5 : 4;
  static constexpr int NA = (CIN / 4) * KEY * KEX * KEZ;  // A fragments of the layer = registers per lane
  static constexpr int OZ = NWZ * NT * ZS, OY = RY * CTY, OX = RX * CTX * XS;   // outputs of one workgroup
  static constexpr int IZW = ZS * (NT - 1) + KEZ;      // input planes one wave reads
  static constexpr int IZ = ZS * (NWZ * NT - 1) + KEZ, IY = OY + 3, IXW = OX + 3;
  static constexpr int rs_for(int w) {
    if (PAIR == 0) return (w + 1) & ~1;
    int r = w;
    while (r % 16 != 8) ++r;                           // patch rows land on banks 0, 8, 16, 24 (+ 0..3)
    return r;
  }
  static constexpr int RS = rs_for(IXW);
  static constexpr int PS = IY * RS;
  static constexpr int cs_for(int v) {
    if (PAIR == 0) return v | 1;                       // odd: the second channel of a read group -> odd banks
    while (v % 8 != 4) ++v;                            // == 4 (mod 8): -> banks 4..7 (+ 8 n)
    return v;
  }
  static constexpr int CS = cs_for(IZ * PS);
  static constexpr int BUF = (4 * CS + 3) / 4 * 4;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two tile buffers in LDS");
  static_assert(IY < 256 && IXW < 256, "packed tile coordinates");
  // geometry of column cc (of the workgroup) and lane j: LDS word offset of its first tap, output (y, x) offsets
  static __device__ __forceinline__ int lds_off(int cc, int j) {
    return ((cc / RX) * CTY + j / CTX) * RS + ((cc % RX) * CTX + j % CTX) * XS;
  }
  static __device__ __forceinline__ bool out_yx(int cc, int j, int& oy, int& ox) {
    oy = (cc / RX) * CTY + j / CTX;
    ox = ((cc % RX) * CTX + j % CTX) * XS;
    return true;
  }
};

// Pair axis x with FLATTENED columns, for outputs whose rows are not a multiple of 16 cells (conv2's backward-data:
// 35 wide = 18 cells): the input is the zero-padded gradient, staged with an LDS row stride of exactly 2 * CPR
// words, so that a row's last taps fall on the next row's (zero) left margin and
//     address(cell p = y * CPR + m, ty, tx) = 2 p + ty * RS + tx
// is linear in p: any 16 consecutive cells of the RY-row region are one column tile, no padding columns.
template <int CIN_, int CPR_, int RY_, int NWC_, int NWZ_, int NT_>
struct MCvFlat {
  static constexpr int CIN = CIN_, PAIR = 0, CPR = CPR_, RY = RY_, NT = NT_;
  static constexpr int NWC = NWC_, NWZ = NWZ_, NW = NWC_ * NWZ_;
  static constexpr int NCOLS = (RY * CPR + 15) / 16;
  static_assert(NCOLS % NWC == 0 && CIN % 4 == 0, "columns split evenly over the waves");
  static constexpr int NC = NCOLS / NWC;
  static constexpr int XS = 2, ZS = 1, KEZ = 4, KEY = 4, KEX = 5;
  static constexpr int NA = (CIN / 4) * KEY * KEX * KEZ;
  static constexpr int OZ = NWZ * NT, OY = RY, OX = 2 * CPR;
  // (one extra staged row: the last tap row's overrun must land on staged zeros as well)
  static constexpr int IZW = NT + 3, IZ = NWZ * NT + 3, IY = RY + 4, IXW = 2 * CPR, RS = 2 * CPR;
  static constexpr int PS = IY * RS;
  static constexpr int CS = (IZ * PS + 64) | 1;          // slack: the unused cells of the last column tile read on
  static constexpr int BUF = (4 * CS + 3) / 4 * 4;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two tile buffers in LDS");
  static_assert(IY < 256 && IXW < 256, "packed tile coordinates");
  static __device__ __forceinline__ int lds_off(int cc, int j) { return 2 * (16 * cc + j); }
  static __device__ __forceinline__ bool out_yx(int cc, int j, int& oy, int& ox) {
    const int p = 16 * cc + j;
    oy = p / CPR;
    ox = 2 * (p % CPR);
    return p < RY * CPR;
  }
};

// ---------------------------------------------------------------------------------------------------
// The kernel is persistent and weight-stationary: one workgroup of NW waves (one per SIMD) per CU walks
// its share of the tiles; every lane keeps ALL A fragments in registers for the whole launch (CIN/4 * 80
// VGPRs -- a wave that is alone on its SIMD owns 512), and the input tiles stream through two LDS buffers
// by LDS-DMA, the loads of step s+1 in flight while the MFMAs of step s issue (a step = one tile x one
// group of four input channels; one barrier per step).  Workgroups of one XCD take a contiguous range of
// tiles, so the halo re-reads of neighbouring tiles hit that XCD's L2.
// ---------------------------------------------------------------------------------------------------
template <class C>
struct MStage {
  static constexpr int NE = C::IZ * C::PS;                          // LDS words per channel
  static constexpr int NIT = (NE + C::NW * 64 - 1) / (C::NW * 64);  // DMA instructions per wave per channel
  int rel[NIT];   // element offset from the tile's first input element (same for every tile and channel)
  int pk[NIT];    // xx | yi << 8 | zi << 16, or -1: nothing to stage in this slot

  __device__ __forceinline__ void init(int wave, int lane, const MDims& d) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = (i * C::NW + wave) * 64 + lane;
      const int xx = e % C::RS, t = e / C::RS, yi = t % C::IY, zi = t / C::IY;
      const bool live = e < NE && xx < C::IXW;
      rel[i] = live ? (zi * d.hin + yi) * d.win + xx : 0;
      pk[i] = live ? (xx | (yi << 8) | (zi << 16)) : -1;
    }
  }

  // four channels starting at xg (= batch item, first channel of the group) -> LDS buffer at word offset `buf`
  __device__ __forceinline__ void issue(const float* __restrict__ xg, float* lds, unsigned lds0, int buf, int wave,
                                        int lane, int gz0, int gy0, int gx0, const MDims& d) const {
    const long plane = (long)d.hin * d.win;
    const bool interior = (d.dbg & 8) || (gz0 >= 0 && gz0 + C::IZ <= d.din && gy0 >= 0 && gy0 + C::IY <= d.hin && gx0 >= 0 &&
                          gx0 + C::IXW <= d.win);          // wave-uniform (dbg 8: tuning runs, see MDims)
    const float* x0 = xg + ((long)gz0 * d.hin + gy0) * d.win + gx0;   // may lie outside the tensor; only in-range
#pragma unroll 1                                                      // elements are ever dereferenced
    for (int c = 0; c < 4; ++c) {
      const float* xc = x0 + c * (long)d.din * plane;
      const int cbase = buf + c * C::CS;
      // scalar base (the tile's first element of this channel; wave-uniform) + the lane's constant byte offset: a DMA
      // instruction costs no vector address arithmetic (a per-lane 64-bit pointer was two VALU adds + waits per piece)
      const float* xcs = nvf_uniform_ptr(xc);
      if (interior) {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
          if (pk[i] >= 0) nvf_glds_row(xcs, (unsigned)rel[i] * 4u, lds0 + (unsigned)(cbase + (i * C::NW + wave) * 64) * 4u);
      } else {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int gx = gx0 + (pk[i] & 255), gy = gy0 + ((pk[i] >> 8) & 255), gz = gz0 + (pk[i] >> 16);
          const bool ok = pk[i] >= 0 && gx >= 0 && gx < d.win && gy >= 0 && gy < d.hin && gz >= 0 && gz < d.din;
          if (ok) nvf_glds_row(xcs, (unsigned)rel[i] * 4u, lds0 + (unsigned)(cbase + (i * C::NW + wave) * 64) * 4u);
          else if (pk[i] >= 0) lds[cbase + (i * C::NW + wave) * 64 + lane] = 0.f;
        }
      }
    }
  }
};

template <class C, int G>
__device__ __forceinline__ void mfma_step(const float* ldsb, const int (&colbase)[C::NC], const float (&A)[C::NA],
                                          f32x4 (&acc)[C::NC][C::NT]) {
  constexpr int NC = C::NC, NT = C::NT, RS = C::RS, PS = C::PS, KEZ = C::KEZ, KEY = C::KEY, KEX = C::KEX, ZS = C::ZS;
#pragma unroll
  for (int ty = 0; ty < KEY; ++ty)
#pragma unroll
    for (int tx = 0; tx < KEX; ++tx)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int zi = 0; zi < C::IZW; ++zi) {
          const float bv = ldsb[colbase[c] + zi * PS + ty * RS + tx];
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            const int tz = zi - ZS * q;
            if (tz >= 0 && tz < KEZ)
              acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[((G * KEY + ty) * KEX + tx) * KEZ + tz], bv, acc[c][q],
                                                               0, 0, 0);
          }
        }
}

// Epilogue.  A lane holds, per tile, rows i = 4 kq + r of column j: co = 2 kq + (r >> 1), s = r & 1.
// EPI 0: y = relu(acc + bias)   (forward)        EPI 1: y = mask > 0 ? acc : 0   (backward-data through a ReLU)
// EPI 2: everything nvf_conv3d_gather offers (bias, act, addend, mask), decided at run time.
template <class C, int EPI>
__device__ __forceinline__ void mfma_store(const f32x4 (&acc)[C::NC][C::NT], const float* __restrict__ bias,
                                           float* __restrict__ y, const float* __restrict__ addend,
                                           const float* __restrict__ mask, const MDims& d, int b, int ozw, int oy0,
                                           int ox0, int j, int wc, int kq, float (&bsum)[2]) {
  const size_t cstride = (size_t)d.dout * d.hout * d.wout;
  const size_t base = ((size_t)b * 8 + 2 * kq) * cstride;
  float bv[2] = {0.f, 0.f};
  if (EPI != 1 && bias) { bv[0] = bias[2 * kq]; bv[1] = bias[2 * kq + 1]; }
#pragma unroll
  for (int c = 0; c < C::NC; ++c) {
    const int cc = wc * C::NC + c;
    int oy, ox;
    if (!C::out_yx(cc, j, oy, ox)) continue;
    oy += oy0;
    ox += ox0;
    if (oy >= d.hout || ox >= d.wout) continue;
#pragma unroll
    for (int q = 0; q < C::NT; ++q) {
      const int oz = ozw + C::ZS * q;
      if (oz >= d.dout) continue;
      const size_t o = base + ((size_t)oz * d.hout + oy) * d.wout + ox;
      // element (h, s): channel 2 kq + h, second output of the pair if s; its offset from o:
      const size_t pstep = C::PAIR == 2 ? (size_t)d.hout * d.wout : 1;
      const bool second = C::PAIR == 2 ? oz + 1 < d.dout : ox + 1 < d.wout;
      float v[4];
      if (EPI == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[c][q][r] + bv[r >> 1], 0.f);
      } else if (EPI == 1) {
        float m[4];
        if (C::PAIR == 0 && second) {
          const nvf_f2u m0 = *(const nvf_f2u*)(mask + o), m1 = *(const nvf_f2u*)(mask + o + cstride);
          m[0] = m0.a; m[1] = m0.b; m[2] = m1.a; m[3] = m1.b;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) m[r] = ((r & 1) && !second) ? 0.f : mask[o + (r >> 1) * cstride + (r & 1) * pstep];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = m[r] > 0.f ? acc[c][q][r] : 0.f;
        bsum[0] += second ? v[0] + v[1] : v[0];
        bsum[1] += second ? v[2] + v[3] : v[2];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = nvf_act(acc[c][q][r] + bv[r >> 1], d.act);
          const size_t oo = o + (r >> 1) * cstride + (r & 1) * pstep;
          if ((r & 1) && !second) continue;
          if (addend) v[r] += addend[oo];
          if (mask) v[r] = mask[oo] > 0.f ? v[r] : 0.f;
        }
      }
      if (C::PAIR == 0 && second) {      // the pair is adjacent in x: one 8-byte store at any 4-byte phase (35-wide rows)
        *(nvf_f2u*)(y + o) = nvf_f2u{v[0], v[1]};
        *(nvf_f2u*)(y + o + cstride) = nvf_f2u{v[2], v[3]};
      } else {
        y[o] = v[0];
        y[o + cstride] = v[2];
        if (second) {
          y[o + pstep] = v[1];
          y[o + cstride + pstep] = v[3];
        }
      }
    }
  }
}

// (Measured and dropped, round 3: four-wave configurations with TWO workgroups per CU -- 256 VGPRs each, both LDS buffer
// pairs resident, one's epilogue / tile hand-over under the other's MFMAs -- conv2 forward 52.4 vs 51.0 us, backward-data
// 81.5 vs 79.9: the kernels are bound by the MFMA rate at the clock the chip holds, not by their bubbles.)
template <class C, int EPI>
__global__ __launch_bounds__(C::NW * 64) void conv_k4_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            const float* __restrict__ addend,
                                                            const float* __restrict__ mask, MDims d, int total) {
  constexpr int NC = C::NC, NT = C::NT, RS = C::RS, CS = C::CS, ZS = C::ZS, CIN = C::CIN, NG = CIN / 4, BUF = C::BUF;
  static_assert(NG % 2 == 0, "the two LDS buffers alternate per channel group");
  // the layer's A fragments (NA x 64 floats, the same for every wave) pass through LDS once per workgroup when they fit
  // and eight waves would otherwise pull them through the L1 (conv2: -0.6 / -1.1 us; with four waves, conv1, the extra
  // barrier-to-first-MFMA latency costs more than the L1 traffic saved: +0.7 us)
  constexpr bool A_LDS = C::NW >= 8 && (2 * BUF + C::NA * 64) * 4 <= 160 * 1024;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF + (A_LDS ? C::NA * 64 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tiles of this workgroup: XCD k (workgroups k, k+8, ...) owns tiles [k per, (k+1) per)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int per = (total + 7) >> 3;
  const int t_lo = xcd * per, t_hi = min(t_lo + per, total);
  int t = t_lo + slot;
  if (t >= t_hi) {                     // nothing to do: its share of the bias partials is zero
    if (EPI == 1 && d.bias_part && (lane & 15) == 0)
      for (int h = 0; h < 2; ++h) d.bias_part[((size_t)blockIdx.x * C::NW + wave) * 8 + 2 * (lane >> 4) + h] = 0.f;
    return;
  }
  const int ntile = d.tiles_x * d.tiles_y * d.tiles_z;
  const int j = lane & 15, kq = lane >> 4;
  const int laneB = kq * CS;
  int colbase[NC];
  const int wc = wave % C::NWC, wz = wave / C::NWC;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int cc = wc * NC + c;
    colbase[c] = laneB + wz * NT * ZS * C::PS + C::lds_off(cc, j);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  const size_t vol = (size_t)d.din * d.hin * d.win;
  MStage<C> st;
  st.init(wave, lane, d);
  auto origin = [&](int tt, int& b, int& oz0, int& oy0, int& ox0) {
    const int tile = tt % ntile;
    b = tt / ntile;
    ox0 = (tile % d.tiles_x) * C::OX;
    oy0 = ((tile / d.tiles_x) % d.tiles_y) * C::OY;
    oz0 = (tile / (d.tiles_x * d.tiles_y)) * C::OZ;
  };
  int b, oz0, oy0, ox0;
  origin(t, b, oz0, oy0, ox0);
  const bool a_lds = A_LDS && !(d.dbg & 4);
  if (a_lds) {                     // one copy per workgroup by LDS-DMA, in flight with the first tile's staging
    const float* wps = nvf_uniform_ptr(wp);
#pragma unroll 4
    for (int i = wave; i < C::NA; i += C::NW)
      nvf_glds_row(wps, (unsigned)(i * 64 + lane) * 4u, lds0 + (unsigned)(2 * BUF + i * 64) * 4u);
  }
  st.issue(x + (size_t)b * CIN * vol, lds, lds0, 0, wave, lane, oz0 - d.pad, oy0 - d.pad, ox0 - d.pad, d);
  // every A fragment of the layer, resident in registers for the whole launch.  Only the first channel group's are
  // fetched before the loop: the explicit vmcnt(0) that ends a staging step waits for EVERY outstanding load, so the
  // other groups' fragments are requested right after the first barrier (the first tile's first step is peeled) and
  // arrive under group 0's MFMAs -- all NA up front was 6.5 us of the 52 us launch (s_memrealtime stamps; it is bound
  // by the L1's throughput: 8 waves x 40 KB).  With the LDS copy above, the L1 sees the 40 KB once and every wave takes
  // its registers from LDS after the first barrier (160 conflict-free ds_reads).
  constexpr int NA0 = C::NA / NG;
  float A[C::NA];
  if (!a_lds) {
#pragma unroll
    for (int i = 0; i < NA0; ++i) A[i] = wp[(size_t)i * 64 + lane];
  }

  bool more = false;
  int bn = 0, ozn = 0, oyn = 0, oxn = 0;
  float bsum[2] = {0.f, 0.f};                                // channels 2 kq, 2 kq + 1 of the outputs this lane stored
  auto tile = [&](auto firstc) {
    constexpr bool FIRST = decltype(firstc)::value;
    const int tn = t + wpx;                                  // this workgroup's next tile
    more = tn < t_hi;
    if (more) origin(tn, bn, ozn, oyn, oxn);
    f32x4 acc[NC][NT];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of step (t, g) has landed
      __syncthreads();                                              // ... everyone's; the other buffer is free
      if (d.dbg & 1) {
      } else if (g + 1 < NG)
        st.issue(x + ((size_t)b * CIN + 4 * (g + 1)) * vol, lds, lds0, ((g + 1) & 1) * BUF, wave, lane, oz0 - d.pad,
                 oy0 - d.pad, ox0 - d.pad, d);
      else if (more)
        st.issue(x + (size_t)bn * CIN * vol, lds, lds0, 0, wave, lane, ozn - d.pad, oyn - d.pad, oxn - d.pad, d);
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (FIRST) {                                   // the remaining fragments are requested here
      if (a_lds) {
        const float* la = lds + 2 * BUF + lane;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) A[i] = la[i * 64];
      } else {
#pragma unroll
        for (int i = NA0; i < C::NA; ++i) A[i] = wp[(size_t)i * 64 + lane];
      }
    }
    if (!(d.dbg & 2)) mfma_step<C, 0>(lds, colbase, A, acc);
    if constexpr (NG > 1) { step(std::integral_constant<int, 1>{}); if (!(d.dbg & 2)) mfma_step<C, (NG > 1 ? 1 : 0)>(lds + BUF, colbase, A, acc); }
    if constexpr (NG > 2) { step(std::integral_constant<int, 2>{}); if (!(d.dbg & 2)) mfma_step<C, (NG > 2 ? 2 : 0)>(lds, colbase, A, acc); }
    if constexpr (NG > 3) { step(std::integral_constant<int, 3>{}); if (!(d.dbg & 2)) mfma_step<C, (NG > 3 ? 3 : 0)>(lds + BUF, colbase, A, acc); }
    mfma_store<C, EPI>(acc, bias, y, addend, mask, d, b, oz0 + ZS * wz * NT, oy0, ox0, j, wc, kq, bsum);
  };
  tile(std::true_type{});
#pragma unroll 1
  while (more) {
    t += wpx; b = bn; oz0 = ozn; oy0 = oyn; ox0 = oxn;
    tile(std::false_type{});
  }
  if (EPI == 1 && d.bias_part) {       // the 16 lanes j of a kq row, in a fixed butterfly order; lane j = 0 writes
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = bsum[h];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (j == 0) d.bias_part[((size_t)blockIdx.x * C::NW + wave) * 8 + 2 * kq + h] = v;
    }
  }
}

// (an immutable per-device constant, initialised once and thread-safely: the only process-wide value the library keeps)
static int nvf_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    return cus > 0 ? cus : 256;
  }();
  return n;
}

template <class C>
static int launch_mfma(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                       const float* mask, int batch, MDims d, hipStream_t s, int* bias_nparts = nullptr) {
  d.tiles_x = (d.wout + C::OX - 1) / C::OX;
  d.tiles_y = (d.hout + C::OY - 1) / C::OY;
  d.tiles_z = (d.dout + C::OZ - 1) / C::OZ;
  const int total = d.tiles_x * d.tiles_y * d.tiles_z * batch;
  int grid = (nvf_cu_count() + 7) / 8 * 8;                  // one workgroup per CU, a multiple of the 8 XCDs
  const int need = ((total + 7) / 8) * 8;
  if (grid > need) grid = need;
  if (bias_nparts) *bias_nparts = grid * C::NW;
  if (bias && d.act == NVF_ACT_RELU && !addend && !mask)
    conv_k4_mfma<C, 0><<<grid, C::NW * 64, 0, s>>>(x, wp, bias, y, addend, mask, d, total);
  else if (!bias && d.act == NVF_ACT_NONE && !addend && mask)
    conv_k4_mfma<C, 1><<<grid, C::NW * 64, 0, s>>>(x, wp, bias, y, addend, mask, d, total);
  else
    conv_k4_mfma<C, 2><<<grid, C::NW * 64, 0, s>>>(x, wp, bias, y, addend, mask, d, total);
  return NVF_OK;
}

// y[b,co,o] = act(bias[co] + sum_{ci,k} x[b,ci,o - pad + k] w[ci][k][co]) (+ addend) (masked), k = 4^3, 8 output
// channels; wp = nvf_pack_mfma_k4 of w with pair_axis 0 (pad 0: the forward pass) or 2 (pad 3: backward-data).
// Returns NVF_EINVAL for shapes without an instantiation (the caller then uses nvf_conv3d_gather).
extern "C" int nvf_conv3d_k4_mfma_bias(const float* x, const float* wp, const float* bias, float* y,
                                       const float* addend, const float* mask, int batch, int cin, int cout, int pad,
                                       int pair_axis, int din, int hin, int win, int dout, int hout, int wout, int act,
                                       int variant, float* bias_part, int* bias_nparts, void* stream);

extern "C" int nvf_conv3d_k4_mfma(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                                  const float* mask, int batch, int cin, int cout, int pad, int pair_axis, int din,
                                  int hin, int win, int dout, int hout, int wout, int act, int variant, void* stream) {
  return nvf_conv3d_k4_mfma_bias(x, wp, bias, y, addend, mask, batch, cin, cout, pad, pair_axis, din, hin, win, dout,
                                 hout, wout, act, variant, nullptr, nullptr, stream);
}

// ... with bias_part (backward-data through a ReLU mask only: no bias, act none, mask given): per (workgroup, wave) the
// 8 channel sums of the stored outputs, *bias_nparts slabs of 8 floats (at most 2048) whose sum is the bias gradient of
// the layer whose masked output gradient this pass writes.
extern "C" int nvf_conv3d_k4_mfma_bias(const float* x, const float* wp, const float* bias, float* y,
                                       const float* addend, const float* mask, int batch, int cin, int cout, int pad,
                                       int pair_axis, int din, int hin, int win, int dout, int hout, int wout, int act,
                                       int variant, float* bias_part, int* bias_nparts, void* stream) {
  if (!x || !wp || !y || batch <= 0 || cout != 8) return NVF_EINVAL;
  if (dout != din + 2 * pad - 3 || hout != hin + 2 * pad - 3 || wout != win + 2 * pad - 3) return NVF_EINVAL;
  if (bias_part && (!bias_nparts || bias || act != NVF_ACT_NONE || addend || !mask)) return NVF_EINVAL;
  MDims d{din, hin, win, dout, hout, wout, pad, act, 0, 0, 0, 0, bias_part};
  if (variant >= 100) { d.dbg = variant / 100; variant %= 100; }
  d.dbg |= nvf_tune_int("NVF_A_GLOBAL", 0) ? 4 : 0;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
#define NVF_M(VAR, CI, PA, WLO, WHI, CTY, CTX, RY, RX, NWC, NWZ, NT)                                   \
  if (rc == 1 && variant == VAR && cin == CI && pair_axis == PA && wout >= WLO && wout <= WHI)        \
    rc = launch_mfma<MCv<CI, PA, CTY, CTX, RY, RX, NWC, NWZ, NT>>(x, wp, bias, y, addend, mask, batch, d, s, bias_nparts);
  NVF_M(0, 8, 0, 17, 32, 1, 16, 8, 1, 4, 2, 2)    // conv2 forward: 8 rows x 4 planes x 32 per workgroup, 8 waves
  NVF_M(0, 8, 0, 9, 16, 2, 8, 4, 1, 4, 1, 2)      // conv1 forward: 8 rows x 2 planes x 16
  NVF_M(0, 8, 2, 21, 36, 4, 4, 3, 3, 1, 4, 1)     // conv2 backward-data: 12 x 12 patch x 8 planes, waves along z
  NVF_M(0, 8, 2, 9, 20, 4, 4, 5, 1, 1, 4, 1)      // conv1 backward-data: 20 x 4 strip x 8 planes
  // tuning alternatives
  NVF_M(2, 8, 0, 17, 32, 1, 16, 4, 1, 4, 1, 4)
  NVF_M(3, 8, 0, 17, 32, 1, 16, 8, 1, 4, 1, 2)
  NVF_M(4, 8, 0, 17, 32, 1, 16, 8, 1, 8, 1, 4)
  NVF_M(5, 8, 0, 17, 32, 1, 16, 16, 1, 8, 1, 4)
  NVF_M(6, 8, 0, 17, 32, 1, 16, 8, 1, 4, 1, 4)
  NVF_M(7, 8, 0, 17, 32, 1, 16, 4, 1, 4, 1, 2)    // conv2 forward: 4 rows x 2 planes, four waves, two workgroups per CU
  NVF_M(2, 8, 2, 21, 36, 4, 4, 3, 3, 3, 1, 3)
  NVF_M(3, 8, 2, 21, 36, 4, 4, 2, 2, 4, 1, 3)
  NVF_M(4, 8, 2, 21, 36, 4, 4, 1, 9, 1, 4, 1)
  NVF_M(5, 8, 2, 21, 36, 4, 4, 3, 3, 3, 2, 1)
  NVF_M(2, 8, 0, 9, 16, 2, 8, 4, 1, 4, 1, 4)
  NVF_M(3, 8, 0, 9, 16, 2, 8, 4, 1, 2, 1, 2)
  NVF_M(4, 8, 0, 9, 16, 2, 8, 4, 1, 4, 2, 1)      // conv1 forward: the same 8 x 2 x 16 tile on eight waves (one plane each)
  NVF_M(5, 8, 0, 9, 16, 2, 8, 4, 1, 4, 2, 2)      // 8 rows x 4 planes x 16 on eight waves
  NVF_M(2, 8, 2, 9, 20, 4, 4, 5, 1, 1, 2, 1)
  NVF_M(3, 8, 2, 9, 20, 4, 4, 5, 5, 5, 1, 1)
#undef NVF_M
#define NVF_MF(VAR, CI, WLO, WHI, CPR, RY, NWC, NWZ, NT)                                               \
  if (rc == 1 && variant == VAR && cin == CI && pair_axis == 0 && pad == 3 && wout >= WLO && wout <= WHI) \
    rc = launch_mfma<MCvFlat<CI, CPR, RY, NWC, NWZ, NT>>(x, wp, bias, y, addend, mask, batch, d, s, bias_nparts);
  // backward-data with pair axis x and flattened columns (weights packed with pair_axis 0 from w_bwd)
  NVF_MF(0, 8, 33, 36, 18, 7, 4, 2, 2)     // conv2 backward-data: 7 rows x 4 planes x 36 = 8 column tiles, 8 waves
  NVF_MF(2, 8, 33, 36, 18, 7, 4, 1, 7)
  NVF_MF(3, 8, 33, 36, 18, 7, 8, 1, 4)
  NVF_MF(4, 8, 33, 36, 18, 7, 4, 1, 4)
  NVF_MF(5, 8, 33, 36, 18, 7, 4, 1, 2)     // 7 rows x 2 planes, four waves, two workgroups per CU
  NVF_MF(0, 8, 17, 20, 10, 8, 5, 1, 4)     // conv1 backward-data: 19 wide = 10 cells; 8 rows = 5 column tiles
  NVF_MF(2, 8, 17, 20, 10, 8, 5, 1, 2)
  NVF_MF(3, 8, 17, 20, 10, 19, 4, 2, 1)    // whole 19 x 20 planes (12 column tiles), 2 planes per workgroup: 36.5 us
  NVF_MF(4, 8, 17, 20, 10, 19, 4, 1, 2)    // 31.5 us (the z-pair mapping above: 28.1)
#undef NVF_MF
  if (rc == 1) return NVF_EINVAL;
  NVF_LAUNCH_CHECK();
  return rc;
}
