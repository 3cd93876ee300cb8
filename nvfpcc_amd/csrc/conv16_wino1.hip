// conv16_wino.hip's Winograd (y, x) convolution for the wide decoder (16 -> 16 channels) with ONE output plane in flight
// per wave (100 accumulation registers) and two waves per SIMD -- the step conv_wino1.hip takes for the narrow decoder:
// an output plane reads its four input planes, each staged and transformed one channel GROUP at a time (a per-wave LDS image
// of four channels: eight waves fit beside the 100 KB of U), 25 MFMAs per (tap, group).  A plane is transformed for each of
// the four output planes it meets instead of 2.5 pairs, at half the price per vector instruction.  Per output the sum has
// conv16_k4_wino's order -- taps 0..3, groups 0..3 -- so the results are the same bits.
#include "wino_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned w161_u4 __attribute__((ext_vector_type(4)));
typedef unsigned w161_u2 __attribute__((ext_vector_type(2)));

constexpr int kWino161AFloats = 4 * 4 * 25 * 64;     // [g][tz][f][lane] (pack kind 41)

struct W161Dims {
  int batch, units, ppc;        // work units = (block, z chunk, column group); ppc OUTPUT PLANES per chunk
};

template <int DIN_, int PAD_>
struct W161Cfg {
  static constexpr int DIN = DIN_, PAD = PAD_, DOUT = DIN_ + 2 * PAD_ - 3, TPR = (DOUT + 1) / 2, NTILE = TPR * TPR;
  static constexpr int NCG = (NTILE + 15) / 16;
  static constexpr int SPAN = TPR % 16 == 0 ? 1 : (16 % TPR == 0 ? 16 / TPR : (14 + TPR) / TPR + 1);
  static constexpr int NR = 2 * SPAN + 3;
  static constexpr int SEGS = (DIN + 3) / 4, RPI = 64 / SEGS, NROW = 4 * NR, NLD = (NROW + RPI - 1) / RPI;
  static constexpr int rs_for() {
    int r = 2 * TPR + 4 > PAD + 4 * SEGS ? 2 * TPR + 4 : PAD + 4 * SEGS;
    while (r % 32 != TPR % 32) ++r;
    return r;
  }
  static constexpr int RS = rs_for();
  static constexpr int cs_for() { int c = NR * RS; while (c % 64 != 32) ++c; return c; }
  static constexpr int CS = cs_for();
  static constexpr int BUF = 4 * CS;                      // four channels of one plane
  static constexpr int NWAVE = 8;
  static_assert(RS % 2 == 0 && CS % 2 == 0, "8-byte window reads");
  static_assert((kWino161AFloats + NWAVE * BUF) * 4 <= 160 * 1024, "LDS");
};

template <class C, int EPI>
__global__ __launch_bounds__(512, 2) void conv16_k4_wino1(const float* __restrict__ g, const float* __restrict__ wp,
                                                          float* __restrict__ y, const float* __restrict__ mask, W161Dims d) {
  constexpr int DIN = C::DIN, PAD = C::PAD, DOUT = C::DOUT, TPR = C::TPR, RS = C::RS, CS = C::CS, NLD = C::NLD;
  __shared__ __attribute__((aligned(16))) float lds[kWino161AFloats + C::NWAVE * C::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* raw = lds + kWino161AFloats + wave * C::BUF;
  for (int i = lane; i < C::BUF; i += 64) raw[i] = 0.f;
  const int per = (int)(gridDim.x >> 3);
  const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int unit_ = __builtin_amdgcn_readfirstlane(wg * C::NWAVE + wave);
  const int j = lane & 15, kq = lane >> 4;
  const bool idle = unit_ >= d.units;
  const int unit = idle ? 0 : unit_;
  const int nchunk = (DOUT + d.ppc - 1) / d.ppc;
  const int cg = unit % C::NCG, zc = (unit / C::NCG) % nchunk, b = unit / (C::NCG * nchunk);
  const int z0 = zc * d.ppc, z1 = min(z0 + d.ppc, DOUT);
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
      (void*)(g + (size_t)b * 16 * DIN * DIN * DIN), 0, 16 * DIN * DIN * DIN * 4, 0x00020000);
  w161_u4 st[NLD];
  auto fetch = [&](int p, int grp) {                     // plane p (padded coordinate), channels 4 grp .. 4 grp + 3
    const int pz = p - PAD;
    const bool pin = pz >= 0 && pz < DIN;
    const int so = __builtin_amdgcn_readfirstlane(pin ? ((grp * 4 * DIN + pz) * DIN * DIN) * 4 : 0);
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
  auto transform = [&](float (&V)[25]) {
    const float* p = win;
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
  auto mfma25 = [&](auto tzc, auto gi, auto firstc, const float (&V)[25]) {
    constexpr int TZ = decltype(tzc)::value, G = decltype(gi)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    const float* ap = abase + (G * 4 + TZ) * 25 * 64;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 25; ++f)
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[f * 64], V[f], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[f], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
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
  const bool full = 2 * X + 1 < DOUT;
  int vo[2], vs64[2], vs32[2];
#pragma unroll
  for (int yo = 0; yo < 2; ++yo) {
    const bool ok = tvalid && 2 * R + yo < DOUT;
    const int o = (int)(((size_t)(4 * kq) * cstride + (size_t)(2 * R + yo) * DOUT + 2 * X) * 4);
    vo[yo] = ok ? o : kOob;
    vs64[yo] = ok && full ? o : kOob;
    vs32[yo] = ok && !full ? o : kOob;
  }
  w161_u2 mk[8];
  auto mask_fetch = [&](int z) {
    if constexpr (EPI != 1) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int so = __builtin_amdgcn_readfirstlane((int)(((size_t)r * cstride + (size_t)z * DOUT * DOUT) * 4));
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) mk[2 * r + yo] = __builtin_amdgcn_raw_buffer_load_b64(rs_m, vo[yo], so, 0);
    }
  };
  auto emit = [&](int z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float c[2][5];
#pragma unroll
      for (int fx = 0; fx < 5; ++fx) {
        const float m0 = acc[fx][r], m1 = acc[5 + fx][r], m2 = acc[10 + fx][r], m3 = acc[15 + fx][r], m4 = acc[20 + fx][r];
        c[0][fx] = (m0 + m1) + (m2 + m3);
        c[1][fx] = (m1 - m2) + fmaf(2.f, m3, m4);
      }
      const int so = __builtin_amdgcn_readfirstlane((int)(((size_t)r * cstride + (size_t)z * DOUT * DOUT) * 4));
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        float o0 = (c[yo][0] + c[yo][1]) + (c[yo][2] + c[yo][3]);
        float o1 = (c[yo][1] - c[yo][2]) + fmaf(2.f, c[yo][3], c[yo][4]);
        if constexpr (EPI == 1) {
          const w161_u2 m = mk[2 * r + yo];
          o0 = __uint_as_float(m.x) > 0.f ? o0 : 0.f;
          o1 = (full && __uint_as_float(m.y) > 0.f) ? o1 : 0.f;
        } else {
          o0 = fmaxf(o0 + bias4[r], 0.f);
          o1 = fmaxf(o1 + bias4[r], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b64(w161_u2{__float_as_uint(o0), __float_as_uint(o1)}, rs_y, vs64[yo], so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rs_y, vs32[yo], so, 0);
      }
    }
  };
  using Yes = std::true_type;
  using No = std::false_type;
  // one staging phase = (input plane z + T, channel group G): its data is in `st` on entry, the next phase's on exit
  auto phase = [&](auto tc, auto gc, int z) {
    constexpr int T = decltype(tc)::value, G = decltype(gc)::value;
    const int p = z + T;
    const bool pin = p - PAD >= 0 && p - PAD < DIN;      // wave-uniform
    if constexpr (T == 3 && G == 0) mask_fetch(z);
    commit();
    if constexpr (G < 3) fetch(p, G + 1);
    else if constexpr (T < 3) fetch(p + 1, 0);
    else if (z + 1 < z1) fetch(z + 1, 0);
    if (pin) {
      float V[25];
      transform(V);
      if constexpr (T == 0 && G == 0) mfma25(tc, gc, Yes{}, V); else mfma25(tc, gc, No{}, V);
    } else if constexpr (T == 0 && G == 0) {
#pragma unroll
      for (int f = 0; f < 25; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (!idle) fetch(z0, 0);
  {
    constexpr int NV = kWino161AFloats / 4, NI = (NV + 511) / 512;
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * 512 + tid < NV)
        __builtin_amdgcn_global_load_lds((glb_vp)(wp + (size_t)(i * 512 + tid) * 4), (lds_vp)(lds + (i * 512 + wave * 64) * 4), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (idle) return;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
#pragma unroll 1
  for (int z = z0; z < z1; ++z) {
    phase(I0{}, I0{}, z); phase(I0{}, I1{}, z); phase(I0{}, I2{}, z); phase(I0{}, I3{}, z);
    phase(I1{}, I0{}, z); phase(I1{}, I1{}, z); phase(I1{}, I2{}, z); phase(I1{}, I3{}, z);
    phase(I2{}, I0{}, z); phase(I2{}, I1{}, z); phase(I2{}, I2{}, z); phase(I2{}, I3{}, z);
    phase(I3{}, I0{}, z); phase(I3{}, I1{}, z); phase(I3{}, I2{}, z); phase(I3{}, I3{}, z);
    emit(z);
  }
}

template <class C, int EPI>
static int launch_wino161(const float* x, const float* wp, float* y, const float* aux, int batch, int ppc, hipStream_t s) {
  if (ppc <= 0) return NVF_EINVAL;
  const int nchunk = (C::DOUT + ppc - 1) / ppc;
  W161Dims d{batch, batch * nchunk * C::NCG, ppc};
  const int grid = ((d.units + C::NWAVE - 1) / C::NWAVE + 7) / 8 * 8;
  conv16_k4_wino1<C, EPI><<<grid, 512, 0, s>>>(x, wp, y, aux, d);
  return NVF_OK;
}

// called by nvf_conv3d_k4_wino16_bwd / _fwd (conv16_wino.hip) for conv2's shapes; ppc = output planes per work unit
int nvf_wino16_1_bwd(const float* dy, const float* wp, float* dx, const float* mask, int batch, int ppc, hipStream_t s) {
  return launch_wino161<W161Cfg<32, 3>, 1>(dy, wp, dx, mask, batch, ppc ? ppc : 6, s);
}
int nvf_wino16_1_fwd(const float* x, const float* wp, const float* bias, float* y, int batch, int ppc, hipStream_t s) {
  return launch_wino161<W161Cfg<35, 0>, 0>(x, wp, y, bias, batch, ppc ? ppc : 4, s);
}
