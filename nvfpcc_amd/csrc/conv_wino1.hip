// conv_wino.hip's Winograd (y, x) convolution (conv2: 8 -> 8 channels, 32^3 <-> 35^3) with ONE accumulator set per wave and
// two waves per SIMD: a wave finishes one pair of output planes at a time from its five input planes (each plane is fetched
// and transformed for the 2.5 pairs it meets, as in conv16_wino.hip), which halves the accumulation registers (100) so that
// eight waves share a CU -- vector instructions then cost ~2.5 instead of ~5.3 cycles each (profiles/r04_mfma_valu_overlap.md)
// and one wave's LDS / memory latencies pass under the other's MFMAs.  The order of every output's sum -- taps zw = 0..4,
// channel group 0 then 1 -- is the two-set kernel's: the results are the same BITS (tests/test_gpu_ops.py).  With the 57-
// instruction transforms of wino_common.h the repeated transforms cost less than the cheaper issue returns: conv2
// backward-data 44.1 -> 42.8 us, forward 31.3 -> 30.4 us in the step (with the 115-instruction transforms the same idea --
// the "team" variant of DESIGN.md section 12(c) -- lost 15 %).  Default for conv2 (nvf_conv3d_k4_wino_*, ppc 0); conv1's
// backward-data keeps the two-set kernel (its 10 x 10 tiles need 13 KB of LDS per wave: eight do not fit beside the A fragments).
#include "wino_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wn1_u4 __attribute__((ext_vector_type(4)));
typedef unsigned wn1_u2 __attribute__((ext_vector_type(2)));

constexpr int kWino1AFloats = 2 * 5 * 25 * 64;     // [g][zw][f][lane] (pack kind 40)

struct W1Dims {
  int batch, units, ppc;
  float* bias_part;
};

template <int DIN_, int PAD_, int NWAVE_ = 8>
struct W1Cfg {
  static constexpr int DIN = DIN_, PAD = PAD_, DOUT = DIN_ + 2 * PAD_ - 3, TPR = (DOUT + 1) / 2, NTILE = TPR * TPR;
  static constexpr int NCG = (NTILE + 15) / 16, NPAIR = TPR;
  static constexpr int SPAN = TPR % 16 == 0 ? 1 : (16 % TPR == 0 ? 16 / TPR : (14 + TPR) / TPR + 1);
  static constexpr int NR = 2 * SPAN + 3;
  static constexpr int SEGS = (DIN + 3) / 4, RPI = 64 / SEGS, NROW = 8 * NR, NLD = (NROW + RPI - 1) / RPI;
  static constexpr int rs_for() {
    int r = 2 * TPR + 4 > PAD + 4 * SEGS ? 2 * TPR + 4 : PAD + 4 * SEGS;
    while (r % 32 != TPR % 32) ++r;
    return r;
  }
  static constexpr int RS = rs_for();
  static constexpr int cs_for() { int c = NR * RS; while (c % 64 != 32) ++c; return c; }
  static constexpr int CS = cs_for();
  static constexpr int BUF = 8 * CS;
  static constexpr int NWAVE = NWAVE_;
  static_assert((kWino1AFloats + NWAVE * BUF) * 4 <= 160 * 1024, "LDS");
};

template <class C, int EPI>
__global__ __launch_bounds__(C::NWAVE * 64, 2) void conv_k4_wino1(const float* __restrict__ g, const float* __restrict__ wp,
                                                        float* __restrict__ y, const float* __restrict__ mask, W1Dims d) {
  constexpr int DIN = C::DIN, PAD = C::PAD, DOUT = C::DOUT, TPR = C::TPR, RS = C::RS, CS = C::CS, NLD = C::NLD;
  __shared__ __attribute__((aligned(16))) float lds[kWino1AFloats + C::NWAVE * C::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* raw = lds + kWino1AFloats + wave * C::BUF;
  for (int i = lane; i < C::BUF; i += 64) raw[i] = 0.f;
  const int per = (int)(gridDim.x >> 3);
  const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int unit_ = __builtin_amdgcn_readfirstlane(wg * C::NWAVE + wave);
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
  int voff[NLD], ldst[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int ri = k * C::RPI + lane / C::SEGS, seg = lane % C::SEGS;
    const int co = ri / C::NR, row = ri % C::NR, yd = 2 * R0 + row - PAD;
    const bool live = ri < C::NROW && lane < C::RPI * C::SEGS;
    const bool ok = live && yd >= 0 && yd < DIN;
    voff[k] = ok ? ((co * DIN * DIN + yd) * DIN + 4 * seg) * 4 : 0x7ffffff0;
    ldst[k] = live ? co * CS + row * RS + PAD + 4 * seg : -1;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g + (size_t)b * 8 * DIN * DIN * DIN), 0, 8 * DIN * DIN * DIN * 4, 0x00020000);
  wn1_u4 st[NLD];
  auto fetch = [&](int p) {
    const int pz = p - PAD;
    const bool pin = pz >= 0 && pz < DIN;
    const int so = __builtin_amdgcn_readfirstlane(pin ? pz * DIN * DIN * 4 : 0);
#pragma unroll
    for (int k = 0; k < NLD; ++k) st[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pin ? voff[k] : 0x7ffffff0, so, 0);
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (ldst[k] < 0) continue;
      float* o = raw + ldst[k];
      if constexpr (PAD & 1) {
        o[0] = __uint_as_float(st[k].x);
        *(float2*)(o + 1) = float2{__uint_as_float(st[k].y), __uint_as_float(st[k].z)};
        o[3] = __uint_as_float(st[k].w);
      } else {
        *(float2*)o = float2{__uint_as_float(st[k].x), __uint_as_float(st[k].y)};
        *(float2*)(o + 2) = float2{__uint_as_float(st[k].z), __uint_as_float(st[k].w)};
      }
    }
  };
  f32x4 acc[25];
  float bsum[2] = {0.f, 0.f};
  auto transform = [&](auto gi, float (&V)[25]) {
    constexpr int G = decltype(gi)::value;
    const float* p = win + G * 4 * CS;
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
  auto mfma25 = [&](auto zwc, auto gi, auto firstc, const float (&V)[25]) {
    constexpr int ZW = decltype(zwc)::value, G = decltype(gi)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    const float* ap = abase + (G * 5 + ZW) * 25 * 64;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 25; ++f)
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[f * 64], V[f], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[f], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const size_t cstride = (size_t)DOUT * DOUT * DOUT;
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(EPI == 1 ? mask + (size_t)b * 8 * cstride : mask), 0, EPI == 1 ? (int)(8 * cstride * 4) : 32, 0x00020000);
  float bias2[2] = {0.f, 0.f};
  if constexpr (EPI == 0) { bias2[0] = mask[2 * kq]; bias2[1] = mask[2 * kq + 1]; }
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)b * 8 * cstride), 0,
                                                                         (int)(8 * cstride * 4), 0x00020000);
  constexpr int kOob = 0x7ffffff0;
  const bool full = 2 * X + 1 < DOUT;
  int vo[2], vs64[2], vs32[2];
#pragma unroll
  for (int yo = 0; yo < 2; ++yo) {
    const bool ok = tvalid && 2 * R + yo < DOUT;
    const int o = (int)(((size_t)(2 * kq) * cstride + (size_t)(2 * R + yo) * DOUT + 2 * X) * 4);
    vo[yo] = ok ? o : kOob;
    vs64[yo] = ok && full ? o : kOob;
    vs32[yo] = ok && !full ? o : kOob;
  }
  wn1_u2 mk[8];
  auto mask_fetch = [&](int q) {
    if constexpr (EPI != 1) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool zin = 2 * q + (r & 1) < DOUT;
      const int so = __builtin_amdgcn_readfirstlane(
          zin ? (int)(((size_t)(r >> 1) * cstride + (size_t)(2 * q + (r & 1)) * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) mk[2 * r + yo] = __builtin_amdgcn_raw_buffer_load_b64(rs_m, zin ? vo[yo] : kOob, so, 0);
    }
  };
  auto emit = [&](int q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float c[2][5];
#pragma unroll
      for (int fx = 0; fx < 5; ++fx) {
        const float m0 = acc[fx][r], m1 = acc[5 + fx][r], m2 = acc[10 + fx][r], m3 = acc[15 + fx][r], m4 = acc[20 + fx][r];
        c[0][fx] = (m0 + m1) + (m2 + m3);
        c[1][fx] = (m1 - m2) + fmaf(2.f, m3, m4);
      }
      const bool zin = 2 * q + (r & 1) < DOUT;
      const int so = __builtin_amdgcn_readfirstlane(
          zin ? (int)(((size_t)(r >> 1) * cstride + (size_t)(2 * q + (r & 1)) * DOUT * DOUT) * 4) : 0);
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        float o0 = (c[yo][0] + c[yo][1]) + (c[yo][2] + c[yo][3]);
        float o1 = (c[yo][1] - c[yo][2]) + fmaf(2.f, c[yo][3], c[yo][4]);
        if constexpr (EPI == 1) {
          const wn1_u2 m = mk[2 * r + yo];
          o0 = __uint_as_float(m.x) > 0.f ? o0 : 0.f;
          o1 = (full && __uint_as_float(m.y) > 0.f) ? o1 : 0.f;
        } else {
          o0 = fmaxf(o0 + bias2[r >> 1], 0.f);
          o1 = fmaxf(o1 + bias2[r >> 1], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b64(wn1_u2{__float_as_uint(o0), __float_as_uint(o1)}, rs_y, zin ? vs64[yo] : kOob, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rs_y, zin ? vs32[yo] : kOob, so, 0);
        bsum[r >> 1] += o0 + o1;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;
  // input plane T of pair q feeds tap zw = T of the pair's row pairs (ci, s)
  auto plane = [&](auto tc, int q) {
    constexpr int T = decltype(tc)::value;
    const int p = 2 * q + T;
    const bool pin = p - PAD >= 0 && p - PAD < DIN;
    if constexpr (T == 3) mask_fetch(q);
    float V[25];
    commit();
    if constexpr (T < 4) fetch(p + 1);
    else if (q + 1 < q1) fetch(2 * q + 2);
    if (pin) {
      transform(I0{}, V);
      if constexpr (T == 0) mfma25(tc, I0{}, Yes{}, V); else mfma25(tc, I0{}, No{}, V);
      transform(I1{}, V);
      mfma25(tc, I1{}, No{}, V);
    } else if constexpr (T == 0) {
#pragma unroll
      for (int f = 0; f < 25; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (!idle) fetch(2 * q0);
  {
    constexpr int NT = C::NWAVE * 64, NV = kWino1AFloats / 4, NI = (NV + NT - 1) / NT;
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * NT + tid < NV)
        __builtin_amdgcn_global_load_lds((glb_vp)(wp + (size_t)(i * NT + tid) * 4), (lds_vp)(lds + (i * NT + wave * 64) * 4), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (idle) {
    if (d.bias_part && j == 0) { d.bias_part[(size_t)unit_ * 8 + 2 * kq] = 0.f; d.bias_part[(size_t)unit_ * 8 + 2 * kq + 1] = 0.f; }
    return;
  }
#pragma unroll 1
  for (int q = q0; q < q1; ++q) {
    plane(std::integral_constant<int, 0>{}, q);
    plane(std::integral_constant<int, 1>{}, q);
    plane(std::integral_constant<int, 2>{}, q);
    plane(std::integral_constant<int, 3>{}, q);
    plane(std::integral_constant<int, 4>{}, q);
    emit(q);
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
static int launch_wino1(const float* x, const float* wp, float* y, const float* aux, int batch, int ppc, float* bias_part,
                        int* bias_nparts, hipStream_t s) {
  if (ppc <= 0) return NVF_EINVAL;
  const int nchunk = (C::NPAIR + ppc - 1) / ppc;
  W1Dims d{batch, batch * nchunk * C::NCG, ppc, bias_part};
  const int grid = ((d.units + C::NWAVE - 1) / C::NWAVE + 7) / 8 * 8;
  if (bias_nparts) *bias_nparts = grid * C::NWAVE;
  conv_k4_wino1<C, EPI><<<grid, C::NWAVE * 64, 0, s>>>(x, wp, y, aux, d);
  return NVF_OK;
}

// called by nvf_conv3d_k4_wino_bwd / _fwd (conv_wino.hip) for conv2's shapes: by default, or when bit 16 of ppc is set
int nvf_wino1_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, float* bias_part,
                  int* bias_nparts, hipStream_t s) {
  return launch_wino1<W1Cfg<32, 3>, 1>(dy, wp, dx, mask, batch, ppc ? ppc : 3, bias_part, bias_nparts, s);
}
int nvf_wino1_bwd16(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, float* bias_part,
                    int* bias_nparts, hipStream_t s) {       // conv1 (16^3 -> 19^3): six waves per workgroup fit the LDS
  return launch_wino1<W1Cfg<16, 3, 6>, 1>(dy, wp, dx, mask, batch, ppc ? ppc : 1, bias_part, bias_nparts, s);
}
int nvf_wino1_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s) {
  return launch_wino1<W1Cfg<35, 0>, 0>(x, wp, y, bias, batch, ppc ? ppc : 2, nullptr, nullptr, s);
}
int nvf_wino1_fwd19(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s) {
  return launch_wino1<W1Cfg<19, 0>, 0>(x, wp, y, bias, batch, ppc ? ppc : 1, nullptr, nullptr, s);
}
