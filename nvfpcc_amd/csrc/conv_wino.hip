// Backward-data (and the training-step forward) of the 4x4x4 convolutions (conv2: 32^3 <-> 35^3, conv1: 16^3 <-> 19^3;
// 8 -> 8 channels) in a reduced-multiplication form: Winograd F(2x2, 4x4) over (y, x), direct over z with the pair trick of conv_mfma.hip.
// Reference site: F.conv3d in mode 'train' and its autograd backward, utils/network.py:687 (NVFPCC.py:160, 197).  Training
// steps only: the eval / encode / decode forward keeps the direct fixed-order form (bit-exact batch invariance, the 2e-6
// occupancy contract).
//
//   out[ci, z, y, x] = sum_co sum_{tz,ty,tx} g[co, z + tz, y + ty, x + tx] w'[co][tz,ty,tx][ci]        (gather form: g is
//   the zero-padded output gradient, w' = w_bwd, the flipped kernel).  For the 2 x 2 outputs of tile (R, X) on plane z:
//
//   out_tile = A^T [ sum_co sum_tz U[f][tz][co][ci] * V[f][z + tz][co][tile] ] A,     f = (fy, fx) in 5 x 5
//   V = B^T g_tile B  (5 x 5 window at (2R, 2X)),   U = G w'_{tz} G^T  (packed once per step: nvf_pack kind 40)
//
// with the Cook-Toom matrices on {0, 1, -1, 2, inf}, the rational factors moved into G so that B^T is small integers:
// 25 products per 2 x 2 outputs and tz instead of 64 -- 2.56 x fewer multiplications.  fp32 error against fp64:
// 1.0e-6 of max |out| (direct form 6.0e-7; tools/winograd_probe.py, /tmp-probe in DESIGN section 12).
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32): rows (ci, s) = two output planes z = 2q + s of a PAIR q, columns = 16
// tiles, K = four gradient channels; a plane p = 2q + zw (zw = 0..4) feeds pair q with A[(ci,s)][co] = U[f][zw - s][co][ci]
// (zero where zw - s is no tap: 4/5 of every MFMA useful).  The B operand of lane (tile j, co kq) is V[f] of ITS tile
// and channel, so the transformed data never leaves the registers of the lane that computed it: raw 5 x 5 windows are
// read from the wave's own LDS image of the plane (flattened tiles: 16 consecutive tiles of the 18 x 18 tile plane; row
// stride 50 = 18 (mod 32) makes the window reads of 16 tiles x 2 channels conflict-free 8-byte reads), transformed by
// 57 vector instructions per (plane, channel group) (wino_common.h) and multiplied into up to three live pairs (75 MFMAs).  A
// wave walks a chunk of pairs down z with TWO accumulator sets (25 frequencies x 4 registers each; the plane that completes a
// pair is multiplied into its set first, the pair is emitted, and the same set starts the next pair); a finished pair goes
// through the inverse transform in registers and leaves as 8-byte stores behind the ReLU mask.  (conv_wino1.hip is the
// one-set, two-waves-per-SIMD form of the same arithmetic and the default for conv2 and conv1's backward-data.)  Planes are fetched one
// ahead with 16-byte buffer loads (8 rows per instruction) held in registers and committed to LDS after the current
// plane's reads; no barrier after the prologue -- waves share only the A fragments (64 KB of LDS per workgroup).
#include "wino_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wn_u4 __attribute__((ext_vector_type(4)));
typedef unsigned wn_u2 __attribute__((ext_vector_type(2)));

constexpr int kWinoAFloats = 2 * 5 * 25 * 64;     // [g][zw][f][lane]

#ifndef NVF_WINO_DBG
#define NVF_WINO_DBG 0        // tuning builds (tools/ab_build.py .. -DNVF_WINO_DBG=1): WDims::dbg switches phases off.  In the
#endif                        // regular build the switches are compiled out (as run-time tests they cost 150 moves per step)
#define WDBG(bit) (NVF_WINO_DBG && (d.dbg & (bit)))

extern "C" size_t nvf_pack_wino_k4_floats(void) { return (size_t)kWinoAFloats; }

struct WDims {
  int batch, units, ppc;        // work units = (block, z chunk, column group); ppc pairs per chunk
  int dbg;                      // tuning runs only (ppc >> 8): 1 no MFMAs, 2 no emit, 4 no transform, 8 no staging, 16 no A copy, 32 no zero fill; results meaningless
  float* bias_part;             // optional: per unit the 8 channel sums of what it stored
};

// DIN: input extent; PAD: zero padding of the gather (3: backward-data = full correlation of the output gradient, 0: the
// forward pass); output extent DIN + 2 PAD - 3.  The kernel works in PADDED input coordinates p = input index + PAD.
template <int DIN_, int PAD_>
struct WCfg {
  static constexpr int DIN = DIN_, PAD = PAD_, DOUT = DIN_ + 2 * PAD_ - 3, TPR = (DOUT + 1) / 2, NTILE = TPR * TPR;
  static constexpr int NCG = (NTILE + 15) / 16, NPAIR = TPR;
  // tile rows a group of 16 consecutive flattened tiles can touch: 16 | 16 tiles per row: 1; 8: 2; 18: 2; 10: 3
  static constexpr int SPAN = TPR % 16 == 0 ? 1 : (16 % TPR == 0 ? 16 / TPR : (14 + TPR) / TPR + 1);
  static constexpr int NR = 2 * SPAN + 3;                 // raw rows staged per plane and channel
  static constexpr int SEGS = (DIN + 3) / 4, RPI = 64 / SEGS, NROW = 8 * NR, NLD = (NROW + RPI - 1) / RPI;
  static constexpr int rs_for() {
    int r = 2 * TPR + 4 > PAD + 4 * SEGS ? 2 * TPR + 4 : PAD + 4 * SEGS;
    while (r % 32 != TPR % 32) ++r;
    return r;
  }
  static constexpr int RS = rs_for();                     // 2 RS = 2 TPR (mod 64): window address linear in the tile index
  static constexpr int cs_for() { int c = NR * RS; while (c % 64 != 32) ++c; return c; }
  static constexpr int CS = cs_for();                     // the second channel of a 32-lane read group: banks + 32
  static constexpr int BUF = 8 * CS;
  static_assert(RS % 2 == 0 && CS % 2 == 0, "8-byte window reads");
  static_assert((kWinoAFloats + 4 * BUF) * 4 <= 160 * 1024, "LDS");
};

// EPI 1: y = mask > 0 ? acc : 0 (backward-data through the ReLU of the layer below; `mask` = that layer's output)
// EPI 0: y = relu(acc + bias[channel])  (forward; `mask` = the 8 biases)
template <class C, int EPI>
__global__ __launch_bounds__(256) void conv_k4_wino(const float* __restrict__ g, const float* __restrict__ wp,
                                                    float* __restrict__ y, const float* __restrict__ mask, WDims d) {
  constexpr int DIN = C::DIN, PAD = C::PAD, DOUT = C::DOUT, TPR = C::TPR, RS = C::RS, CS = C::CS, NLD = C::NLD;
  __shared__ __attribute__((aligned(16))) float lds[kWinoAFloats + 4 * C::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* raw = lds + kWinoAFloats + wave * C::BUF;
  if (!WDBG(32))
    for (int i = lane; i < C::BUF; i += 64) raw[i] = 0.f;        // the margins stay zero for the whole launch
  // XCD k (workgroups k, k + 8, ...) takes a CONTIGUOUS range of work units: neighbouring column groups and z chunks of a
  // block share input rows / planes, and each XCD has its own L2 (round-robin units made every XCD fetch every block)
  const int per = (int)(gridDim.x >> 3);                         // the grid is a multiple of 8
  const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int unit_ = __builtin_amdgcn_readfirstlane(wg * 4 + wave);
  const int j = lane & 15, kq = lane >> 4;
  const bool idle = unit_ >= d.units;                            // a wave past the last unit: zero partials, no work
  const int unit = idle ? 0 : unit_;
  const int nchunk = (C::NPAIR + d.ppc - 1) / d.ppc;
  const int cg = unit % C::NCG, zc = (unit / C::NCG) % nchunk, b = unit / (C::NCG * nchunk);
  const int q0 = zc * d.ppc, q1 = min(q0 + d.ppc, C::NPAIR);
  const int tl = 16 * cg + j;
  const bool tvalid = tl < C::NTILE;
  const int t = tvalid ? tl : C::NTILE - 1;
  const int R = t / TPR, X = t % TPR, R0 = (16 * cg) / TPR;
  const float* win = raw + 2 * (R - R0) * RS + 2 * X + kq * CS;
  const float* abase = lds + lane;

  // staging descriptors: load k covers rows (k RPI + lane / SEGS) of the (channel, row) list, 16 bytes per lane
  int voff[NLD], ldst[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int ri = k * C::RPI + lane / C::SEGS, seg = lane % C::SEGS;
    const int co = ri / C::NR, row = ri % C::NR, yd = 2 * R0 + row - PAD;
    const bool live = ri < C::NROW && lane < C::RPI * C::SEGS;
    const bool ok = live && yd >= 0 && yd < DIN;
    voff[k] = ok ? ((co * DIN * DIN + yd) * DIN + 4 * seg) * 4 : 0x7ffffff0;      // beyond the descriptor: reads 0
    ldst[k] = live ? co * CS + row * RS + PAD + 4 * seg : -1;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g + (size_t)b * 8 * DIN * DIN * DIN), 0, 8 * DIN * DIN * DIN * 4, 0x00020000);
  wn_u4 st[NLD];
  // (unconditional loads: a plane outside the tensor takes the out-of-range offset in every lane and reads zeros -- a
  // branch here makes the loaded registers a phi, which the compiler resolves with a wait right behind the loads; the
  // plane offset is forced into an SGPR or every load becomes a waterfall loop)
  auto fetch = [&](int p) {
    const int pz = p - PAD;
    if (WDBG(8)) return;
    const bool pin = pz >= 0 && pz < DIN;
    const int so = __builtin_amdgcn_readfirstlane(pin ? pz * DIN * DIN * 4 : 0);
#pragma unroll
    for (int k = 0; k < NLD; ++k) st[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pin ? voff[k] : 0x7ffffff0, so, 0);
  };
  auto commit = [&]() {
    if (WDBG(8)) return;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (ldst[k] < 0) continue;
      float* o = raw + ldst[k];
      if constexpr (PAD & 1) {                                    // odd word: 4 + 8 + 4 bytes
        o[0] = __uint_as_float(st[k].x);
        *(float2*)(o + 1) = float2{__uint_as_float(st[k].y), __uint_as_float(st[k].z)};
        o[3] = __uint_as_float(st[k].w);
      } else {                                                    // (a row's last segment may run one word past the
        *(float2*)o = float2{__uint_as_float(st[k].x), __uint_as_float(st[k].y)};       // row: no window reads it)
        *(float2*)(o + 2) = float2{__uint_as_float(st[k].z), __uint_as_float(st[k].w)};
      }
    }
  };

  // two accumulator sets: pair q lives in set q & 1 (25 frequencies x 4 registers); the matrix cores' accumulators are
  // the AGPR half of the register file, which 3 x 100 would overflow -- so the plane that completes pair s - 2 (tap 4)
  // is multiplied into that set FIRST, the pair is emitted, and the same set then starts pair s with the same V
  f32x4 acc[2][25];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 25; ++f) acc[s][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};

  // V = B^T (5 x 5 window of channel 4 gi + kq) B
  auto transform = [&](auto gi, float (&V)[25]) {
    constexpr int G = decltype(gi)::value;
    const float* p = win + G * 4 * CS;
    if (WDBG(4)) {
#pragma unroll
      for (int f = 0; f < 25; ++f) V[f] = 1.f + f;
      return;
    }
    // y pass on the packed pipe: the window's x pairs (0,1) and (2,3) are the 8-byte LDS reads themselves; then the x
    // pass row by row (wino_common.h)
    wino_f2 a[5], bb[5], ea[5], eb[5];
    float c[5], ec[5];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      a[dy] = *(const wino_f2*)(p + dy * RS);
      bb[dy] = *(const wino_f2*)(p + dy * RS + 2);
      c[dy] = p[dy * RS + 4];
    }
    wino_bt2(a[0], a[1], a[2], a[3], a[4], ea[0], ea[1], ea[2], ea[3], ea[4]);
    wino_bt2(bb[0], bb[1], bb[2], bb[3], bb[4], eb[0], eb[1], eb[2], eb[3], eb[4]);
    wino_bt(c[0], c[1], c[2], c[3], c[4], ec[0], ec[1], ec[2], ec[3], ec[4]);
#pragma unroll
    for (int fy = 0; fy < 5; ++fy)
      wino_bt_row(ea[fy], eb[fy], ec[fy], V[5 * fy], V[5 * fy + 1], V[5 * fy + 2], V[5 * fy + 3], V[5 * fy + 4]);
  };
  auto mfma25 = [&](auto slot, auto zwc, auto gi, const float (&V)[25]) {
    constexpr int S = decltype(slot)::value, ZW = decltype(zwc)::value, G = decltype(gi)::value;
    constexpr bool FIRST = ZW == 0 && G == 0;       // the first block of a pair starts from zero: no clearing pass
    const float* ap = abase + (G * 5 + ZW) * 25 * 64;
    if (WDBG(1)) return;
    __builtin_amdgcn_sched_barrier(0);        // keeps the A reads of other (slot, tap) blocks out of this one: registers
#pragma unroll
    for (int f = 0; f < 25; ++f)
      acc[S][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[f * 64], V[f], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[S][f], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- a finished pair: mask / output addressing through buffer descriptors (out-of-range lanes read 0 and store nothing:
  // no divergent branches).  Lane part of the offset per output row yo; the (channel half, plane) part is scalar.
  const size_t cstride = (size_t)DOUT * DOUT * DOUT;
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(EPI == 1 ? mask + (size_t)b * 8 * cstride : mask), 0, EPI == 1 ? (int)(8 * cstride * 4) : 32, 0x00020000);
  float bias2[2] = {0.f, 0.f};                            // EPI 0: the biases of this lane's two channels
  if constexpr (EPI == 0) { bias2[0] = mask[2 * kq]; bias2[1] = mask[2 * kq + 1]; }
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)b * 8 * cstride), 0,
                                                                         (int)(8 * cstride * 4), 0x00020000);
  constexpr int kOob = 0x7ffffff0;
  const bool full = 2 * X + 1 < DOUT;                     // the tile's second x output exists
  int vo[2], vs64[2], vs32[2];
#pragma unroll
  for (int yo = 0; yo < 2; ++yo) {
    const bool ok = tvalid && 2 * R + yo < DOUT;
    const int o = (int)(((size_t)(2 * kq) * cstride + (size_t)(2 * R + yo) * DOUT + 2 * X) * 4);
    vo[yo] = ok ? o : kOob;
    vs64[yo] = ok && full ? o : kOob;
    vs32[yo] = ok && !full ? o : kOob;
  }
  wn_u2 mk[8];                                            // the ReLU mask of the pair being finished, fetched a plane ahead
  auto mask_fetch = [&](int q) {
    if constexpr (EPI != 1) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool zin = 2 * q + (r & 1) < DOUT;            // wave-uniform
      const int so = __builtin_amdgcn_readfirstlane(
          zin ? (int)(((size_t)(r >> 1) * cstride + (size_t)(2 * q + (r & 1)) * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) mk[2 * r + yo] = __builtin_amdgcn_raw_buffer_load_b64(rs_m, zin ? vo[yo] : kOob, so, 0);
    }
  };
  // A^T M A per row, ReLU mask, stores, channel sums (a masked-out or out-of-range output is 0 and adds nothing)
  auto emit = [&](auto slot, int q) {
    constexpr int S = decltype(slot)::value;
    if (WDBG(2)) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float c[2][5];
#pragma unroll
      for (int fx = 0; fx < 5; ++fx) {
        const float m0 = acc[S][fx][r], m1 = acc[S][5 + fx][r], m2 = acc[S][10 + fx][r], m3 = acc[S][15 + fx][r],
                    m4 = acc[S][20 + fx][r];
        c[0][fx] = (m0 + m1) + (m2 + m3);
        c[1][fx] = (m1 - m2) + fmaf(2.f, m3, m4);
      }
      const bool zin = 2 * q + (r & 1) < DOUT;            // wave-uniform
      const int so = __builtin_amdgcn_readfirstlane(
          zin ? (int)(((size_t)(r >> 1) * cstride + (size_t)(2 * q + (r & 1)) * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        float o0 = (c[yo][0] + c[yo][1]) + (c[yo][2] + c[yo][3]);
        float o1 = (c[yo][1] - c[yo][2]) + fmaf(2.f, c[yo][3], c[yo][4]);
        if constexpr (EPI == 1) {
          const wn_u2 m = mk[2 * r + yo];
          o0 = __uint_as_float(m.x) > 0.f ? o0 : 0.f;
          o1 = (full && __uint_as_float(m.y) > 0.f) ? o1 : 0.f;
        } else {
          o0 = fmaxf(o0 + bias2[r >> 1], 0.f);
          o1 = fmaxf(o1 + bias2[r >> 1], 0.f);
        }
        // (a plane beyond the tensor: its mask was read as zeros, so o0 = o1 = 0; the stores take the out-of-range offset)
        __builtin_amdgcn_raw_buffer_store_b64(wn_u2{__float_as_uint(o0), __float_as_uint(o1)}, rs_y, zin ? vs64[yo] : kOob, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rs_y, zin ? vs32[yo] : kOob, so, 0);
        bsum[r >> 1] += o0 + o1;
      }
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  // one pair step: planes 2s (taps 4 / 0 / 2 of pairs s-2, s, s-1) and 2s + 1 (taps 1 / 3 of pairs s, s-1); SA = set of s
  auto step = [&](auto sa, int s) -> bool {
    constexpr int SA = decltype(sa)::value;
    using A = std::integral_constant<int, SA>;
    using B = std::integral_constant<int, 1 - SA>;
    const bool hA = s < q1, hB = s - 1 >= q0 && s - 1 < q1, hC = s - 2 >= q0;
    const bool last = s == q1 + 1;
    if (hC) mask_fetch(s - 2);
    if (!last) fetch(2 * s + 1);
    const bool pin = 2 * s - PAD >= 0 && 2 * s - PAD < DIN;
    float V0[25], V1[25];
    if (pin) {
      transform(I0{}, V0);
      transform(I1{}, V1);
      if (hC) { mfma25(A{}, I4{}, I0{}, V0); mfma25(A{}, I4{}, I1{}, V1); }
    }
    if (hC) emit(A{}, s - 2);
    if (pin) {
      if (hA) { mfma25(A{}, I0{}, I0{}, V0); mfma25(A{}, I0{}, I1{}, V1); }
      if (hB) { mfma25(B{}, I2{}, I0{}, V0); mfma25(B{}, I2{}, I1{}, V1); }
    }
    if (last) return false;
    commit();
    fetch(2 * s + 2);
    if (2 * s + 1 - PAD >= 0 && 2 * s + 1 - PAD < DIN) {
      transform(I0{}, V0);
      if (hA) mfma25(A{}, I1{}, I0{}, V0);
      if (hB) mfma25(B{}, I3{}, I0{}, V0);
      transform(I1{}, V1);
      if (hA) mfma25(A{}, I1{}, I1{}, V1);
      if (hB) mfma25(B{}, I3{}, I1{}, V1);
    }
    commit();
    return true;
  };
  // prologue: the first plane's loads go out BEFORE the A fragments are copied, so that their latency (HBM on a cold
  // tile) passes under the 64 KB copy instead of after it
  if (!idle) fetch(2 * q0);
  if (!WDBG(16)) {                          // the A fragments, L2 -> LDS by DMA (1 KB per wave instruction, no registers)
    constexpr int NV = kWinoAFloats / 4, NI = (NV + 255) / 256;
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * 256 + tid < NV)
        __builtin_amdgcn_global_load_lds((glb_vp)(wp + (size_t)(i * 256 + tid) * 4), (lds_vp)(lds + (i * 256 + wave * 64) * 4), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (idle) {
    if (d.bias_part && j == 0) { d.bias_part[(size_t)unit_ * 8 + 2 * kq] = 0.f; d.bias_part[(size_t)unit_ * 8 + 2 * kq + 1] = 0.f; }
    return;
  }
  commit();
  int s = q0;                                         // q0 is even (ppc is): the set of pair q is q & 1
#pragma unroll 1
  for (;;) {
    s = __builtin_amdgcn_readfirstlane(s);
    if (!step(I0{}, s++)) break;
    if (!step(I1{}, s++)) break;
  }
  if (d.bias_part) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = bsum[h];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (j == 0) d.bias_part[(size_t)unit * 8 + 2 * kq + h] = v;
    }
  }
}

template <class C, int EPI>
static int launch_wino(const float* x, const float* wp, float* y, const float* aux, int batch, int ppc, float* bias_part,
                       int* bias_nparts, hipStream_t s) {
  const int dbg = ppc >> 8;
  ppc &= 255;
  if (ppc & 1) return NVF_EINVAL;       // chunks start at even pairs (two alternating accumulator sets)
  const int nchunk = (C::NPAIR + ppc - 1) / ppc;
  WDims d{batch, batch * nchunk * C::NCG, ppc, dbg, bias_part};
  const int grid = ((d.units + 3) / 4 + 7) / 8 * 8;      // a multiple of the 8 XCDs (idle workgroups write zero partials)
  if (bias_nparts) *bias_nparts = grid * 4;
  conv_k4_wino<C, EPI><<<grid, 256, 0, s>>>(x, wp, y, aux, d);
  return NVF_OK;
}

// dx[b, ci, :] = relu-mask( sum_co conv_full(dy[b, co], w) ): backward-data of a valid 4^3 convolution with 8 -> 8 channels
// through the ReLU of the layer below.  dy [batch, 8, din^3] (din = 32: conv2, 16: conv1), dx / mask [batch, 8, (din + 3)^3];
// wp = nvf_pack_mfma_all kind 40 of the layer's w_bwd (nvf_pack_wino_k4_floats() floats).  bias_part (optional):
// *bias_nparts slabs of 8 channel sums of dx (the bias gradient of the layer below).  ppc: pairs of output planes per
// work unit (0 = default; even).  NVF_EINVAL for shapes without an instantiation.
// (a caller that wants the two-set kernel of this file for conv2 passes an explicit ppc: no process-wide switch)
static constexpr bool wino1_default() { return true; }

int nvf_wino1_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, float* bias_part,
                  int* bias_nparts, hipStream_t s);
int nvf_wino1_bwd16(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, float* bias_part,
                    int* bias_nparts, hipStream_t s);
int nvf_wino1_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s);
int nvf_wino1_fwd19(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s);

extern "C" int nvf_conv3d_k4_wino_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int din,
                                      int ppc, float* bias_part, int* bias_nparts, void* stream) {
  if (!dy || !wp || !dx || !mask || batch <= 0 || (bias_part && !bias_nparts)) return NVF_EINVAL;
  int rc;
  // conv2 (din 32): by default (ppc 0) the one-accumulator-set kernel with two waves per SIMD (conv_wino1.hip: the same
  // bits, 44.1 -> 42.8 us in the step); an explicit ppc selects the kernel below, bit 16 of ppc the other one
  if ((din == 32 || din == 16) && (ppc & 0x100ff) == 0 && wino1_default()) ppc |= 1 << 16;   // (conv1: 18.8 -> 16.7 us)
  if ((ppc >> 16) & 1) {
    if (din == 32) rc = nvf_wino1_bwd(dy, wp, dx, mask, batch, ppc & 255, bias_part, bias_nparts, nvf_stream(stream));
    else if (din == 16) rc = nvf_wino1_bwd16(dy, wp, dx, mask, batch, ppc & 255, bias_part, bias_nparts, nvf_stream(stream));
    else return NVF_EINVAL;
    if (rc != NVF_OK) return rc;
    NVF_LAUNCH_CHECK();
    return NVF_OK;
  }
  if (din == 32) rc = launch_wino<WCfg<32, 3>, 1>(dy, wp, dx, mask, batch, (ppc & 255) ? ppc : (ppc | 6), bias_part, bias_nparts, nvf_stream(stream));
  else if (din == 16) rc = launch_wino<WCfg<16, 3>, 1>(dy, wp, dx, mask, batch, (ppc & 255) ? ppc : (ppc | 2), bias_part, bias_nparts, nvf_stream(stream));
  else return NVF_EINVAL;
  if (rc != NVF_OK) return rc;
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// y = relu(conv3d(x, w) + bias): the FORWARD pass of the same layers in the Winograd form -- for training steps only (its
// results differ from the direct fixed-order kernel by fp32 rounding, 1e-6 of max |y|: the eval / encode / decode forward,
// whose occupancy must be batch-invariant bit for bit, never uses it).  x [batch, 8, din^3] (din = 35: conv2, 19: conv1),
// y [batch, 8, (din - 3)^3]; wp = kind 40 of the layer's w_fwd.
extern "C" int nvf_conv3d_k4_wino_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int din,
                                      int ppc, void* stream) {
  if (!x || !wp || !bias || !y || batch <= 0) return NVF_EINVAL;
  int rc;
  if (din == 35 && (ppc & 0x100ff) == 0 && wino1_default()) ppc |= 1 << 16;        // conv2's forward: as above (31.3 -> 30.4 us)
  if ((ppc >> 16) & 1) {
    if (din == 35) rc = nvf_wino1_fwd(x, wp, bias, y, batch, ppc & 255, nvf_stream(stream));
    else if (din == 19) rc = nvf_wino1_fwd19(x, wp, bias, y, batch, ppc & 255, nvf_stream(stream));
    else return NVF_EINVAL;
    if (rc != NVF_OK) return rc;
    NVF_LAUNCH_CHECK();
    return NVF_OK;
  }
  if (din == 35) rc = launch_wino<WCfg<35, 0>, 0>(x, wp, y, bias, batch, (ppc & 255) ? ppc : (ppc | 4), nullptr, nullptr, nvf_stream(stream));
  else if (din == 19) rc = launch_wino<WCfg<19, 0>, 0>(x, wp, y, bias, batch, (ppc & 255) ? ppc : (ppc | 2), nullptr, nullptr, nvf_stream(stream));
  else return NVF_EINVAL;
  if (rc != NVF_OK) return rc;
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
