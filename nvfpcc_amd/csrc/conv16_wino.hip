// The 4x4x4 convolutions of the WIDE decoder (chanstr 16,32,16,16: conv2 32^3 <-> 35^3, conv1 16^3 <-> 19^3; 16 -> 16
// channels) in the reduced-multiplication form of conv_wino.hip: Winograd F(2x2, 4x4) over (y, x), direct over z.
// Reference site: F.conv3d, utils/network.py:687, and its autograd backward (NVFPCC.py:197).  Training steps only: the
// eval / encode / decode forward keeps the direct fixed-order kernel (conv16_mfma.hip).
//
//   out[o, z, y, x] = sum_c sum_{tz,ty,tx} in[c, z + tz, y + ty, x + tx] w[c][tz,ty,tx][o]     (gather form; `in` zero-padded
//   by PAD).  For the 2 x 2 outputs of tile (R, X) on plane z:
//   out_tile = A^T [ sum_c sum_tz U[f][tz][c][o] * V[f][z + tz][c][tile] ] A,   V = B^T in_tile B,   U = G w_{tz} G^T
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32): rows = the 16 OUTPUT channels (every lane useful -- no plane pairing as
// in the 8-channel kernel), K = four input channels (four groups), columns = 16 tiles.  The accumulators of one output
// plane are 25 frequencies x 4 registers; two planes (a PAIR 2q, 2q + 1) are in flight -- the 200 accumulation registers --
// and the five input planes 2q .. 2q + 4 of a pair are walked once per pair: plane t feeds tap t of the first and tap
// t - 1 of the second plane of the pair.  (A plane therefore meets 2.5 pairs and is fetched and transformed once for
// each: with 16 x 16 channels a transform of 57 vector instructions feeds up to 50 MFMAs, so the repeated transforms are
// ~20 % on top of the MFMAs, against the 2.56 x fewer MFMAs of the form.)  The lane that owns column j (tile) and K index
// k (channel) transforms the window it feeds; raw planes pass through a per-wave LDS image of EIGHT channels (two phases
// per plane), fetched one phase ahead by 16-byte buffer loads.  U (100 KB: [g][tz][f][lane]) is copied to LDS once per
// workgroup by DMA.  No barrier after the prologue.
#include "wino_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned w16_u4 __attribute__((ext_vector_type(4)));
typedef unsigned w16_u2 __attribute__((ext_vector_type(2)));

constexpr int kWino16AFloats = 4 * 4 * 25 * 64;     // [g][tz][f][lane]

extern "C" size_t nvf_pack_wino16_k4_floats(void) { return (size_t)kWino16AFloats; }

struct W16Dims {
  int batch, units, ppc;        // work units = (block, z chunk, column group); ppc pairs of output planes per chunk
  float* bias_part;             // optional (EPI 1): per unit the 16 channel sums of what it stored
};

// DIN: input extent; PAD: zero padding of the gather (3: backward-data, 0: forward); output extent DIN + 2 PAD - 3.  The
// kernel works in PADDED input coordinates p = input index + PAD.  (Tile geometry as WCfg of conv_wino.hip.)
template <int DIN_, int PAD_>
struct W16Cfg {
  static constexpr int DIN = DIN_, PAD = PAD_, DOUT = DIN_ + 2 * PAD_ - 3, TPR = (DOUT + 1) / 2, NTILE = TPR * TPR;
  static constexpr int NCG = (NTILE + 15) / 16, NPAIR = TPR;
  static constexpr int SPAN = TPR % 16 == 0 ? 1 : (16 % TPR == 0 ? 16 / TPR : (14 + TPR) / TPR + 1);
  static constexpr int NR = 2 * SPAN + 3;                 // raw rows staged per plane and channel
  static constexpr int SEGS = (DIN + 3) / 4, RPI = 64 / SEGS, NROW = 8 * NR, NLD = (NROW + RPI - 1) / RPI;
  static constexpr int rs_for() {
    int r = 2 * TPR + 4 > PAD + 4 * SEGS ? 2 * TPR + 4 : PAD + 4 * SEGS;
    while (r % 32 != TPR % 32) ++r;
    return r;
  }
  static constexpr int RS = rs_for();
  static constexpr int cs_for() { int c = NR * RS; while (c % 64 != 32) ++c; return c; }
  static constexpr int CS = cs_for();
  static constexpr int BUF = 8 * CS;                      // eight channels of one plane
  static_assert(RS % 2 == 0 && CS % 2 == 0, "8-byte window reads");
  static_assert((kWino16AFloats + 4 * BUF) * 4 <= 160 * 1024, "LDS");
};

// EPI 1: y = mask > 0 ? acc : 0 (backward-data through the ReLU of the layer below; `mask` = that layer's output)
// EPI 0: y = relu(acc + bias[channel])  (forward; `mask` = the 16 biases)
// BIAS (EPI 1 only): also leave the channel sums of what was stored (d.bias_part); a template switch because the eight adds
// per emitted plane cost the backward-data kernel 2-3 us whether anyone reads the sums or not
template <class C, int EPI, bool BIAS = false>
__global__ __launch_bounds__(256) void conv16_k4_wino(const float* __restrict__ g, const float* __restrict__ wp,
                                                      float* __restrict__ y, const float* __restrict__ mask, W16Dims d) {
  constexpr int DIN = C::DIN, PAD = C::PAD, DOUT = C::DOUT, TPR = C::TPR, RS = C::RS, CS = C::CS, NLD = C::NLD;
  __shared__ __attribute__((aligned(16))) float lds[kWino16AFloats + 4 * C::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* raw = lds + kWino16AFloats + wave * C::BUF;
  for (int i = lane; i < C::BUF; i += 64) raw[i] = 0.f;          // the margins stay zero for the whole launch
  // XCD k (workgroups k, k + 8, ...) takes a CONTIGUOUS range of work units (each XCD has its own L2)
  const int per = (int)(gridDim.x >> 3);                         // the grid is a multiple of 8
  const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int unit_ = __builtin_amdgcn_readfirstlane(wg * 4 + wave);
  const int j = lane & 15, kq = lane >> 4;
  const bool idle = unit_ >= d.units;
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

  // staging descriptors: load k covers rows (k RPI + lane / SEGS) of the (channel, row) list of ONE eight-channel phase
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
      (void*)(g + (size_t)b * 16 * DIN * DIN * DIN), 0, 16 * DIN * DIN * DIN * 4, 0x00020000);
  w16_u4 st[NLD];
  auto fetch = [&](int p, int h) {                       // plane p (padded coordinate), channels 8 h .. 8 h + 7
    const int pz = p - PAD;
    const bool pin = pz >= 0 && pz < DIN;
    const int so = __builtin_amdgcn_readfirstlane(pin ? ((h * 8 * DIN + pz) * DIN * DIN) * 4 : 0);
#pragma unroll
    for (int k = 0; k < NLD; ++k) st[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pin ? voff[k] : 0x7ffffff0, so, 0);
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (ldst[k] < 0) continue;
      float* o = raw + ldst[k];
      if constexpr (PAD & 1) {                                    // odd word: 4 + 8 + 4 bytes
        o[0] = __uint_as_float(st[k].x);
        *(float2*)(o + 1) = float2{__uint_as_float(st[k].y), __uint_as_float(st[k].z)};
        o[3] = __uint_as_float(st[k].w);
      } else {
        *(float2*)o = float2{__uint_as_float(st[k].x), __uint_as_float(st[k].y)};
        *(float2*)(o + 2) = float2{__uint_as_float(st[k].z), __uint_as_float(st[k].w)};
      }
    }
  };

  // two accumulator sets = the two output planes of the pair in flight
  f32x4 acc[2][25];

  // V = B^T (5 x 5 window of channel 4 gl + kq of the staged phase) B
  auto transform = [&](auto gl, float (&V)[25]) {
    constexpr int GL = decltype(gl)::value;
    const float* p = win + GL * 4 * CS;
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
  // one block: the 25 frequencies of (output plane set S, tap TZ, channel group G)
  auto mfma25 = [&](auto slot, auto tzc, auto gi, auto firstc, const float (&V)[25]) {
    constexpr int S = decltype(slot)::value, TZ = decltype(tzc)::value, G = decltype(gi)::value;
    constexpr bool FIRST = decltype(firstc)::value;  // the first block of a plane starts from zero: no clearing pass
    const float* ap = abase + (G * 4 + TZ) * 25 * 64;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 25; ++f)
      acc[S][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[f * 64], V[f], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[S][f], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto clear = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int f = 0; f < 25; ++f) acc[S][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- a finished plane: lane holds channels 4 kq + r (r = 0..3) of its tile.  Mask / output addressing through buffer
  // descriptors (out-of-range lanes read 0 and store nothing): lane part per output row yo, (channel r, plane) part scalar
  const size_t cstride = (size_t)DOUT * DOUT * DOUT;
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(EPI == 1 ? mask + (size_t)b * 16 * cstride : mask), 0, EPI == 1 ? (int)(16 * cstride * 4) : 64, 0x00020000);
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = mask[4 * kq + r];
  }
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)b * 16 * cstride), 0,
                                                                         (int)(16 * cstride * 4), 0x00020000);
  constexpr int kOob = 0x7ffffff0;
  const bool full = 2 * X + 1 < DOUT;                     // the tile's second x output exists
  int vo[2], vs64[2], vs32[2];
#pragma unroll
  for (int yo = 0; yo < 2; ++yo) {
    const bool ok = tvalid && 2 * R + yo < DOUT;
    const int o = (int)(((size_t)(4 * kq) * cstride + (size_t)(2 * R + yo) * DOUT + 2 * X) * 4);
    vo[yo] = ok ? o : kOob;
    vs64[yo] = ok && full ? o : kOob;
    vs32[yo] = ok && !full ? o : kOob;
  }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};                   // channel sums of what this lane stored (a masked-out or
  w16_u2 mk[8];                                           // out-of-range output is 0); mk: the ReLU mask, a plane ahead
  auto mask_fetch = [&](int z) {
    if constexpr (EPI != 1) return;
    const bool zin = z < DOUT;                            // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int so = __builtin_amdgcn_readfirstlane(zin ? (int)(((size_t)r * cstride + (size_t)z * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) mk[2 * r + yo] = __builtin_amdgcn_raw_buffer_load_b64(rs_m, zin ? vo[yo] : kOob, so, 0);
    }
  };
  auto emit = [&](auto slot, int z) {
    constexpr int S = decltype(slot)::value;
    const bool zin = z < DOUT;                            // wave-uniform
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
      const int so = __builtin_amdgcn_readfirstlane(zin ? (int)(((size_t)r * cstride + (size_t)z * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        float o0 = (c[yo][0] + c[yo][1]) + (c[yo][2] + c[yo][3]);
        float o1 = (c[yo][1] - c[yo][2]) + fmaf(2.f, c[yo][3], c[yo][4]);
        if constexpr (EPI == 1) {
          const w16_u2 m = mk[2 * r + yo];
          o0 = __uint_as_float(m.x) > 0.f ? o0 : 0.f;
          o1 = (full && __uint_as_float(m.y) > 0.f) ? o1 : 0.f;
        } else {
          o0 = fmaxf(o0 + bias4[r], 0.f);
          o1 = fmaxf(o1 + bias4[r], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b64(w16_u2{__float_as_uint(o0), __float_as_uint(o1)}, rs_y, zin ? vs64[yo] : kOob, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rs_y, zin ? vs32[yo] : kOob, so, 0);
        if constexpr (BIAS) bsum[r] += o0 + o1;
      }
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using Yes = std::true_type;
  using No = std::false_type;
  // blocks of input plane T of a pair: tap T of the pair's first output plane (set 0), tap T - 1 of its second (set 1)
  auto blocks = [&](auto tc, auto gi, const float (&V)[25]) {
    constexpr int T = decltype(tc)::value, G = decltype(gi)::value;
    if constexpr (T <= 3) {
      if constexpr (T == 0 && G == 0) mfma25(I0{}, I0{}, gi, Yes{}, V);
      else mfma25(I0{}, std::integral_constant<int, T>{}, gi, No{}, V);
    }
    if constexpr (T >= 1) {
      if constexpr (T == 1 && G == 0) mfma25(I1{}, I0{}, gi, Yes{}, V);
      else mfma25(I1{}, std::integral_constant<int, (T >= 1 ? T - 1 : 0)>{}, gi, No{}, V);
    }
  };
  // one input plane = two staging phases of eight channels.  The data of phase (T, 0) is in `st` on entry; on exit `st`
  // holds phase (T + 1, 0) -- or the first phase of the next pair
  auto plane = [&](auto tc, int q) {
    constexpr int T = decltype(tc)::value;
    const int p = 2 * q + T;
    const bool pin = p - PAD >= 0 && p - PAD < DIN;      // wave-uniform
    if constexpr (T == 3) mask_fetch(2 * q);
    if constexpr (T == 4) mask_fetch(2 * q + 1);
    float V[25];
    commit();
    fetch(p, 1);
    if (pin) {
      transform(I0{}, V); blocks(tc, I0{}, V);
      transform(I1{}, V); blocks(tc, I1{}, V);
    } else {
      if constexpr (T == 0) clear(I0{});                  // the plane that would have started the set does not exist
      if constexpr (T == 1) clear(I1{});
    }
    commit();
    if constexpr (T < 4) fetch(p + 1, 0);
    else if (q + 1 < q1) fetch(2 * q + 2, 0);
    if (pin) {
      transform(I0{}, V); blocks(tc, I2{}, V);
      transform(I1{}, V); blocks(tc, I3{}, V);
    }
  };

  // prologue: the first phase's loads go out before the A fragments are copied (L2 -> LDS by DMA, 1 KB per instruction)
  if (!idle) fetch(2 * q0, 0);
  {
    constexpr int NV = kWino16AFloats / 4, NI = (NV + 255) / 256;
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
    if (BIAS && d.bias_part && j == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) d.bias_part[(size_t)unit_ * 16 + 4 * kq + r] = 0.f;
    }
    return;
  }
#pragma unroll 1
  for (int q = q0; q < q1; ++q) {
    plane(I0{}, q);
    plane(I1{}, q);
    plane(I2{}, q);
    plane(I3{}, q);
    emit(I0{}, 2 * q);
    plane(I4{}, q);
    emit(I1{}, 2 * q + 1);
  }
  if (BIAS && d.bias_part) {                              // the unit's channel sums (the bias gradient of the layer below)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = bsum[r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (j == 0) d.bias_part[(size_t)unit * 16 + 4 * kq + r] = v;
    }
  }
}

template <class C, int EPI>
static int launch_wino16(const float* x, const float* wp, float* y, const float* aux, int batch, int ppc, float* bias_part,
                         int* bias_nparts, hipStream_t s) {
  if (ppc <= 0) return NVF_EINVAL;
  const int nchunk = (C::NPAIR + ppc - 1) / ppc;
  W16Dims d{batch, batch * nchunk * C::NCG, ppc, bias_part};
  const int grid = ((d.units + 3) / 4 + 7) / 8 * 8;      // a multiple of the 8 XCDs (idle waves write zero partials)
  if (bias_nparts) *bias_nparts = grid * 4;
  if (EPI == 1 && bias_part) conv16_k4_wino<C, EPI, EPI == 1><<<grid, 256, 0, s>>>(x, wp, y, aux, d);
  else conv16_k4_wino<C, EPI, false><<<grid, 256, 0, s>>>(x, wp, y, aux, d);
  return NVF_OK;
}

// dx[b, ci, :] = relu-mask( sum_co conv_full(dy[b, co], w) ): backward-data of a valid 4^3 convolution with 16 -> 16
// channels through the ReLU of the layer below.  dy [batch, 16, din^3] (din = 32: conv2, 16: conv1), dx / mask
// [batch, 16, (din + 3)^3]; wp = nvf_pack_mfma_all kind 41 of the layer's w_bwd (nvf_pack_wino16_k4_floats() floats).
// ppc: pairs of output planes per work unit (0 = default).  bias_part (optional): *bias_nparts slabs of 16 channel sums of dx
// (the bias gradient of the layer below; a jtotal = 16 job of nvf_wgrad_reduce_multi*).  NVF_EINVAL for shapes without an
// instantiation.
// (a caller that wants the two-plane kernel for the backward-data as well passes an explicit ppc: no process-wide switch)
static constexpr bool wino161_default() { return true; }

int nvf_wino16_1_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, hipStream_t s);
int nvf_wino16_1_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s);

extern "C" int nvf_conv3d_k4_wino16_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int din,
                                        int ppc, float* bias_part, int* bias_nparts, void* stream) {
  if (!dy || !wp || !dx || !mask || batch <= 0 || ppc < 0 || (bias_part && !bias_nparts)) return NVF_EINVAL;
  int rc;
  // conv2's backward-data (din 32) without bias sums: by default the kernel with one output plane in flight and two waves
  // per SIMD (conv16_wino1.hip: the same bits, 140 -> 133 us at batch 16, 534 -> 479 at 64; the forward is faster in the
  // two-plane kernel below: 88 vs 100 us); an explicit ppc selects the kernel below, bit 16 of ppc the other one
  if (din == 32 && !bias_part && (ppc & 0x100ff) == 0 && wino161_default()) ppc |= 1 << 16;
  if ((ppc >> 16) & 1) {
    if (din != 32 || bias_part) return NVF_EINVAL;
    rc = nvf_wino16_1_bwd(dy, wp, dx, mask, batch, ppc & 255, nvf_stream(stream));
    if (rc != NVF_OK) return rc;
    NVF_LAUNCH_CHECK();
    return NVF_OK;
  }
  if (din == 32) rc = launch_wino16<W16Cfg<32, 3>, 1>(dy, wp, dx, mask, batch, ppc ? ppc : 6, bias_part, bias_nparts, nvf_stream(stream));
  else if (din == 16) rc = launch_wino16<W16Cfg<16, 3>, 1>(dy, wp, dx, mask, batch, ppc ? ppc : 1, bias_part, bias_nparts, nvf_stream(stream));
  else return NVF_EINVAL;
  if (rc != NVF_OK) return rc;
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// y = relu(conv3d(x, w) + bias): the FORWARD pass of the same layers in the Winograd form -- for training steps only.
// x [batch, 16, din^3] (din = 35: conv2, 19: conv1), y [batch, 16, (din - 3)^3]; wp = kind 41 of the layer's w_fwd.
extern "C" int nvf_conv3d_k4_wino16_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int din,
                                        int ppc, void* stream) {
  if (!x || !wp || !bias || !y || batch <= 0 || ppc < 0) return NVF_EINVAL;
  int rc;
  if ((ppc >> 16) & 1) {
    if (din != 35) return NVF_EINVAL;
    rc = nvf_wino16_1_fwd(x, wp, bias, y, batch, ppc & 255, nvf_stream(stream));
    if (rc != NVF_OK) return rc;
    NVF_LAUNCH_CHECK();
    return NVF_OK;
  }
  if (din == 35) rc = launch_wino16<W16Cfg<35, 0>, 0>(x, wp, y, bias, batch, ppc ? ppc : 4, nullptr, nullptr, nvf_stream(stream));
  else if (din == 19) rc = launch_wino16<W16Cfg<19, 0>, 0>(x, wp, y, bias, batch, ppc ? ppc : 1, nullptr, nullptr, nvf_stream(stream));
  else return NVF_EINVAL;
  if (rc != NVF_OK) return rc;
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
