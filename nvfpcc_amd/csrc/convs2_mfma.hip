// Matrix-core (v_mfma_f32_16x16x4_f32, exact fp32) backward-data of the stride-2, 5^3, padding-0 transposed
// convolutions up1 / up2 (autograd backward of F.conv_transpose3d, utils/network.py:621):
//
//   dx[ci, i] = sum_{co, k} g[co, 2 i + k] w[ci][co][k]        -- a stride-2 gather convolution, 125 taps
//
// rows  : 16 = the layer's 16 input channels (up1), or (ci, s) for 8 channels (up2), where s picks one of two
//         x-adjacent outputs i_x = 2m + s; both read g[.., 4m + tx], tx = 2s + kx in 0..6, each row uses 5 of the
//         7 taps (zero weights on the other two);
// K     : four channels of g;
// cols  : 2 output rows x 8 cells (up2: 8 cells = 16 x outputs; up1: 8 x outputs).
// A wave stacks NT tiles along z; one B fragment (one input plane/row/tap) feeds every (tile, kz) that touches it.
// A workgroup owns a (4 rows x NZ planes) region; per group of four g channels it brings that group's A fragments
// (45 KB) and input tile into LDS -- all global loads of a pass are issued into registers before any is waited for,
// and the loads of the second pass are in flight while the first computes.  Fixed per-output accumulation order:
// (channel group, ky, tx, kz).  Epilogue: (+ addend) and the ReLU mask of the layer's input, as nvf_conv3d_gather.
#include "nvf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// PAIR = 1: 8 output channels, rows (ci, s); PAIR = 0: 16 output channels
template <int PAIR_, int NIN_, int NOUT_, int RY_, int NWZ_, int NT_, int RS_>
struct S2Cfg {
  static constexpr int PAIR = PAIR_, NIN = NIN_, NOUT = NOUT_, RY = RY_, NWZ = NWZ_, NT = NT_, RS = RS_;
  static constexpr int COG = PAIR ? 8 : 16;
  static constexpr int KEX = PAIR ? 7 : 5;                 // x taps of a row pair / of a row
  static constexpr int XSTEP = PAIR ? 4 : 2;               // input words between neighbouring columns
  static constexpr int NW = RY * NWZ, NTH = NW * 64;
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  static constexpr int OY = 2 * RY, OZ = NWZ * NT;         // output rows / planes of a workgroup
  static constexpr int IZ = 2 * (OZ - 1) + 5, IY = 2 * (OY - 1) + 5;
  static constexpr int IZW = 2 * (NT - 1) + 5;             // input planes one wave reads
  static constexpr int PS = IY * RS;
  static constexpr int CS = (IZ * PS) | 1;                 // odd: the second channel of a read group -> other banks
  static constexpr int XW = 4 * CS;                        // input tile of one channel group (words)
  static constexpr int NFR = 25 * KEX;                     // A fragments per channel group
  static constexpr int AW = NFR * 64;
  static constexpr int ROWCH = (NIN + 3) / 4;              // float4 chunks per input row
  static constexpr int XITEMS = 4 * IZ * IY * ROWCH;
  static constexpr int NX4 = (XITEMS + NTH - 1) / NTH, NA4 = (AW / 4 + NTH - 1) / NTH;
  static_assert(RS >= NIN && (XW + AW) * 4 <= 160 * 1024, "LDS");
};

__global__ void pack_s2k5_mfma_kernel(const float* __restrict__ gw /* [cig][125][cog] */, float* __restrict__ wp,
                                      int cig, int cog) {
  const int pair = cog == 8, KEX = pair ? 7 : 5;
  const int total = (cig / 4) * 25 * KEX * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int lane = r % 64; r /= 64;
    const int tx = r % KEX; r /= KEX;
    const int ky = r % 5; r /= 5;
    const int kz = r % 5;
    const int g = r / 5;
    const int i = lane & 15, ch = 4 * g + (lane >> 4);
    const int co = pair ? i >> 1 : i, kx = pair ? tx - 2 * (i & 1) : tx;
    wp[idx] = (kx >= 0 && kx < 5) ? gw[(ch * 125 + (kz * 5 + ky) * 5 + kx) * cog + co] : 0.f;
  }
}

template <class C>
__global__ __launch_bounds__(C::NTH) void conv_s2k5_mfma(const float* __restrict__ g, const float* __restrict__ wp,
                                                      float* __restrict__ dx, const float* __restrict__ addend,
                                                      const float* __restrict__ mask, int cig) {
  constexpr int NIN = C::NIN, NOUT = C::NOUT, RS = C::RS, PS = C::PS, CS = C::CS, KEX = C::KEX, NT = C::NT,
                IZ = C::IZ, IY = C::IY, ROWCH = C::ROWCH, NX4 = C::NX4, NA4 = C::NA4, COG = C::COG;
  __shared__ __attribute__((aligned(16))) float xs[C::XW];
  __shared__ __attribute__((aligned(16))) float as[C::AW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int TY = NOUT / C::OY, TZ = NOUT / C::OZ;
  const int tile = blockIdx.x % (TY * TZ), b = blockIdx.x / (TY * TZ);
  const int oy0 = (tile % TY) * C::OY, oz0 = (tile / TY) * C::OZ;
  const int wy = wave % C::RY, wz = wave / C::RY;
  const int j = lane & 15, kq = lane >> 4, yy = j >> 3, m = j & 7;
  // column (yy, m) of this wave's tile stack: input word of tap (zi, ky, tx) = colbase + zi PS + ky RS + tx
  const int colbase = kq * CS + (2 * wz * NT) * PS + (2 * (2 * wy + yy)) * RS + C::XSTEP * m;
  const float* gb = g + (size_t)b * cig * NIN * NIN * NIN;
  const int gz0 = 2 * oz0, gy0 = 2 * oy0;

  float4 xv[NX4], av[NA4];
  // input tile through buffer loads: descriptor = the channel group of this batch element (scalar), soffset = the
  // tile's origin (scalar), voffset = a per-thread constant (or beyond the range when the row lies outside the tensor:
  // the load returns 0) -- no per-element address arithmetic.  Rows are NIN = 35 / 19 words: 4-byte aligned 16-byte
  // loads; the last chunk of a row holds NIN % 4 elements, its other components are cleared.
  int xvoff[NX4];
#pragma unroll
  for (int u = 0; u < NX4; ++u) {
    const int i = tid + u * C::NTH;
    const int xq = i % ROWCH, r = i / ROWCH, yi = r % IY, t = r / IY, zi = t % IZ, c = t / IZ;
    const bool ok = i < C::XITEMS && gz0 + zi < NIN && gy0 + yi < NIN;
    xvoff[u] = ok ? (((c * NIN + zi) * NIN + yi) * NIN + 4 * xq) * 4 : 0x7ffffff0;
  }
  const int xsoff = ((gz0 * NIN + gy0) * NIN) * 4;
  auto load = [&](int grp) {
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gb + (size_t)grp * 4 * NIN * NIN * NIN), 0, 4 * NIN * NIN * NIN * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rg, xvoff[u], xsoff, 0);
      float4 v = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
      const int xq = (tid + u * C::NTH) % ROWCH;
      if (4 * xq + 1 >= NIN) v.y = 0.f;
      if (4 * xq + 2 >= NIN) v.z = 0.f;
      if (4 * xq + 3 >= NIN) v.w = 0.f;
      xv[u] = v;
    }
    const float4* ap = (const float4*)(wp + (size_t)grp * C::AW);
#pragma unroll
    for (int u = 0; u < NA4; ++u) {
      const int i = tid + u * C::NTH;
      av[u] = i < C::AW / 4 ? ap[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * C::NTH;
      if (i < C::XITEMS) {
        const int xq = i % ROWCH, r = i / ROWCH, yi = r % IY, t = r / IY, zi = t % IZ, c = t / IZ;
        float* d = xs + c * CS + zi * PS + yi * RS + 4 * xq;
        d[0] = xv[u].x;
        if (4 * xq + 1 < RS) d[1] = xv[u].y;
        if (4 * xq + 2 < RS) d[2] = xv[u].z;
        if (4 * xq + 3 < RS) d[3] = xv[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < NA4; ++u) {
      const int i = tid + u * C::NTH;
      if (i < C::AW / 4) ((float4*)as)[i] = av[u];
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ngroups = cig / 4;
  load(0);
#pragma unroll 1
  for (int grp = 0; grp < ngroups; ++grp) {
    if (grp) __syncthreads();                     // everyone is done reading the previous group's tile
    store();
    __syncthreads();
    if (grp + 1 < ngroups) load(grp + 1);         // in flight while this group computes
    const float* al = as + lane;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int tx = 0; tx < KEX; ++tx) {
        float a[5];
#pragma unroll
        for (int kz = 0; kz < 5; ++kz) a[kz] = al[((kz * 5 + ky) * KEX + tx) * 64];
#pragma unroll
        for (int zi = 0; zi < C::IZW; ++zi) {
          const float bv = xs[colbase + zi * PS + ky * RS + tx];
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            const int kz = zi - 2 * q;
            if (kz >= 0 && kz < 5) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kz], bv, acc[q], 0, 0, 0);
          }
        }
      }
  }
  // ---- epilogue
  const size_t vol = (size_t)NOUT * NOUT * NOUT;
  const int oy = oy0 + 2 * wy + yy;
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int oz = oz0 + wz * NT + q;
    if (C::PAIR) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                 // channel 2 kq + h, outputs x = 2m, 2m + 1
        const size_t o = ((size_t)b * COG + 2 * kq + h) * vol + ((size_t)oz * NOUT + oy) * NOUT + 2 * m;
        float v0 = acc[q][2 * h], v1 = acc[q][2 * h + 1];
        if (addend) { const float2 a2 = *(const float2*)(addend + o); v0 += a2.x; v1 += a2.y; }
        if (mask) { const float2 m2 = *(const float2*)(mask + o); v0 = m2.x > 0.f ? v0 : 0.f; v1 = m2.y > 0.f ? v1 : 0.f; }
        *(float2*)(dx + o) = make_float2(v0, v1);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {                 // channel 4 kq + r, output x = m
        const size_t o = ((size_t)b * COG + 4 * kq + r) * vol + ((size_t)oz * NOUT + oy) * NOUT + m;
        float v = acc[q][r];
        if (addend) v += addend[o];
        if (mask) v = mask[o] > 0.f ? v : 0.f;
        dx[o] = v;
      }
    }
  }
}

// ---- up1's backward-data with the two channel groups of g on DIFFERENT waves (training steps behind the engine's
// `winograd` switch; variant 7).  The kernel above runs up1 (16 output channels, 8^3 outputs, batch 16) as 128 workgroups of
// four waves, 250 MFMAs each, the two channel groups one after the other: half the chip idles and the MFMA phase is 3.3 us
// of a 12 us launch.  Here a workgroup owns 2 rows x 2 planes (256 workgroups at batch 16), wave = (plane, channel group):
// both groups' tiles and A fragments are staged at once, every wave issues 125 MFMAs, and the second group's sums are added
// to the first's through LDS -- (group 0's chain) + (group 1's chain) instead of one chain over both: other bits.
template <int NIN_, int NOUT_, int RS_>
struct S2KCfg {
  static constexpr int NIN = NIN_, NOUT = NOUT_, RS = RS_, COG = 16, KEX = 5, XSTEP = 2;
  static constexpr int NTH = 256, OY = 2, OZ = 2;
  static constexpr int IZ = 2 * (OZ - 1) + 5, IY = 2 * (OY - 1) + 5, IZW = 5;
  static constexpr int PS = IY * RS, CS = (IZ * PS) | 1, XW = 4 * CS, NFR = 25 * KEX, AW = NFR * 64;
  static constexpr int ROWCH = (NIN + 3) / 4, XITEMS = 4 * IZ * IY * ROWCH;
  static constexpr int NX4 = (XITEMS + NTH - 1) / NTH, NA4 = (AW / 4 + NTH - 1) / NTH;
  static_assert(RS >= NIN && 2 * (XW + AW) * 4 <= 160 * 1024 && NOUT % 2 == 0, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NTH) void conv_s2k5_mfma_ks2(const float* __restrict__ g, const float* __restrict__ wp,
                                                          float* __restrict__ dx, const float* __restrict__ addend,
                                                          const float* __restrict__ mask) {
  constexpr int NIN = C::NIN, NOUT = C::NOUT, RS = C::RS, PS = C::PS, CS = C::CS, KEX = C::KEX, IZ = C::IZ, IY = C::IY,
                ROWCH = C::ROWCH, NX4 = C::NX4, NA4 = C::NA4, COG = C::COG;
  __shared__ __attribute__((aligned(16))) float xs[2 * C::XW];
  __shared__ __attribute__((aligned(16))) float as[2 * C::AW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wz = wave & 1, kg = wave >> 1;                  // output plane of the pair, channel group of g
  constexpr int TY = NOUT / C::OY, TZ = NOUT / C::OZ;
  const int tile = blockIdx.x % (TY * TZ), b = blockIdx.x / (TY * TZ);
  const int oy0 = (tile % TY) * C::OY, oz0 = (tile / TY) * C::OZ;
  const int j = lane & 15, kq = lane >> 4, yy = j >> 3, m = j & 7;
  const int colbase = kq * CS + (2 * wz) * PS + (2 * yy) * RS + C::XSTEP * m;
  const float* gb = g + (size_t)b * 8 * NIN * NIN * NIN;
  const int gz0 = 2 * oz0, gy0 = 2 * oy0;
  // staging: as in the kernel above, for both channel groups at once
  float4 xv[2][NX4], av[2][NA4];
  const int xsoff = ((gz0 * NIN + gy0) * NIN) * 4;
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gb + (size_t)grp * 4 * NIN * NIN * NIN), 0, 4 * NIN * NIN * NIN * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * C::NTH;
      const int xq = i % ROWCH, r = i / ROWCH, yi = r % IY, t = r / IY, zi = t % IZ, c = t / IZ;
      const bool ok = i < C::XITEMS && gz0 + zi < NIN && gy0 + yi < NIN;
      const int voff = ok ? (((c * NIN + zi) * NIN + yi) * NIN + 4 * xq) * 4 : 0x7ffffff0;
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, xsoff, 0);
      float4 v = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
      if (4 * xq + 1 >= NIN) v.y = 0.f;
      if (4 * xq + 2 >= NIN) v.z = 0.f;
      if (4 * xq + 3 >= NIN) v.w = 0.f;
      xv[grp][u] = v;
    }
    const float4* ap = (const float4*)(wp + (size_t)grp * C::AW);
#pragma unroll
    for (int u = 0; u < NA4; ++u) {
      const int i = tid + u * C::NTH;
      av[grp][u] = i < C::AW / 4 ? ap[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * C::NTH;
      if (i < C::XITEMS) {
        const int xq = i % ROWCH, r = i / ROWCH, yi = r % IY, t = r / IY, zi = t % IZ, c = t / IZ;
        float* d = xs + grp * C::XW + c * CS + zi * PS + yi * RS + 4 * xq;
        d[0] = xv[grp][u].x;
        if (4 * xq + 1 < RS) d[1] = xv[grp][u].y;
        if (4 * xq + 2 < RS) d[2] = xv[grp][u].z;
        if (4 * xq + 3 < RS) d[3] = xv[grp][u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < NA4; ++u) {
      const int i = tid + u * C::NTH;
      if (i < C::AW / 4) ((float4*)(as + grp * C::AW))[i] = av[grp][u];
    }
  }
  __syncthreads();
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* al = as + kg * C::AW + lane;
    const float* xg = xs + kg * C::XW;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int tx = 0; tx < KEX; ++tx) {
        float a[5];
#pragma unroll
        for (int kz = 0; kz < 5; ++kz) a[kz] = al[((kz * 5 + ky) * KEX + tx) * 64];
#pragma unroll
        for (int zi = 0; zi < 5; ++zi)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[zi], xg[colbase + zi * PS + ky * RS + tx], acc, 0, 0, 0);
      }
  }
  // ---- the second group's sums join the first's: through LDS (the tiles are no longer read)
  __syncthreads();
  f32x4* red = (f32x4*)xs;
  if (kg == 1) red[wz * 64 + lane] = acc;
  __syncthreads();
  if (kg == 1) return;
  const f32x4 other = red[wz * 64 + lane];
  const size_t vol = (size_t)NOUT * NOUT * NOUT;
  const int oy = oy0 + yy, oz = oz0 + wz;
#pragma unroll
  for (int r = 0; r < 4; ++r) {                     // channel 4 kq + r, output x = m
    const size_t o = ((size_t)b * COG + 4 * kq + r) * vol + ((size_t)oz * NOUT + oy) * NOUT + m;
    float v = acc[r] + other[r];
    if (addend) v += addend[o];
    if (mask) v = mask[o] > 0.f ? v : 0.f;
    dx[o] = v;
  }
}

}  // namespace

extern "C" size_t nvf_pack_s2k5_mfma_floats(int cig, int cog) { return (size_t)(cig / 4) * 25 * (cog == 8 ? 7 : 5) * 64; }

// gather_w = the [cig][125][cog] gather-form weight (= w_bwd of a transposed convolution: cig = its output channels)
extern "C" int nvf_pack_s2k5_mfma(const float* gather_w, int cig, int cog, float* wp, void* stream) {
  if (!gather_w || !wp || cig <= 0 || cig % 4 || (cog != 8 && cog != 16)) return NVF_EINVAL;
  const int total = (int)nvf_pack_s2k5_mfma_floats(cig, cog);
  pack_s2k5_mfma_kernel<<<(total + 255) / 256, 256, 0, nvf_stream(stream)>>>(gather_w, wp, cig, cog);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// dx[b,ci,i] = (sum_{co,k} g[b,co,2i+k] w[co][k][ci] (+ addend)) (masked), din = 2 dout + 3.
// NVF_EINVAL = no instantiation for this shape (the caller then uses nvf_conv3d_gather).
extern "C" int nvf_conv3d_s2k5_mfma(const float* g, const float* wp, float* dx, const float* addend, const float* mask,
                                    int batch, int cig, int cog, int din, int dout, int variant, void* stream) {
  if (!g || !wp || !dx || batch <= 0 || din != 2 * dout + 3) return NVF_EINVAL;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
#define NVF_S2(VAR, CIG, COGV, NIN, NOUT, RY, NWZ, NT, RS)                                             \
  if (rc == 1 && variant == VAR && cig == CIG && cog == COGV && din == NIN) {                          \
    using C = S2Cfg<(COGV == 8), NIN, NOUT, RY, NWZ, NT, RS>;                                          \
    conv_s2k5_mfma<C><<<batch * (NOUT / C::OY) * (NOUT / C::OZ), C::NTH, 0, s>>>(g, wp, dx, addend, mask, cig); \
    rc = NVF_OK;                                                                                       \
  }
  NVF_S2(0, 8, 8, 35, 16, 2, 2, 2, 37)     // up2 backward-data: 4 rows x 4 planes per workgroup
  NVF_S2(0, 8, 16, 19, 8, 2, 2, 1, 24)     // up1 backward-data: 4 rows x 2 planes
  NVF_S2(2, 8, 8, 35, 16, 2, 2, 1, 37)
  NVF_S2(3, 8, 8, 35, 16, 1, 4, 2, 37)
  NVF_S2(2, 8, 16, 19, 8, 2, 2, 2, 24)
  NVF_S2(3, 8, 16, 19, 8, 1, 4, 1, 24)
  NVF_S2(5, 8, 8, 35, 16, 2, 4, 1, 37)     // up2: the same 4 rows x 4 planes on eight waves (one plane each)
  NVF_S2(6, 8, 8, 35, 16, 4, 2, 1, 37)     // up2: 8 rows x 2 planes on eight waves
  NVF_S2(5, 8, 16, 19, 8, 4, 2, 1, 24)     // up1: 8 rows x 2 planes on eight waves (64 workgroups)
#undef NVF_S2
  if (rc == 1 && variant == 7 && cig == 8 && cog == 16 && din == 19) {     // up1, channel groups on different waves
    using C = S2KCfg<19, 8, 24>;
    conv_s2k5_mfma_ks2<C><<<batch * (8 / C::OY) * (8 / C::OZ), C::NTH, 0, s>>>(g, wp, dx, addend, mask);
    rc = NVF_OK;
  }
  if (rc == 1) return NVF_EINVAL;
  NVF_LAUNCH_CHECK();
  return rc;
}
