// Matrix-core forward of the stride-2, 5^3 transposed convolutions with 16 OUTPUT channels and no padding -- the wide
// decoder's up1 (32 -> 16 channels, 8^3 -> 19^3) and up2 (16 -> 16, 16^3 -> 35^3); F.conv_transpose3d, network.py:621.
//
// Sub-pixel form: output o = 2c + e per axis (cell c, parity e) reads inputs i = c - j through taps k = e + 2j
// (j = 0..2 for e = 0, j = 0..1 for e = 1): all 125 taps do useful work, no inserted zeros.  MFMA mapping
// (v_mfma_f32_16x16x4_f32, an exact fp32 fmaf chain): rows = the 16 output channels, K = four input channels,
// columns = 16 consecutive cells of the FLATTENED (cy, cx) cell plane; the eight parity classes (ez, ey, ex) are separate
// accumulators fed by the same B fragment (27 LDS reads feed 125 MFMAs).  The LDS image of an input plane has row
// stride = cells per row (= input width + 2) with two zero words in front of every row and zero rows around it, so
//   address(cell p, jy, jx) = p - jy * NCELL - jx + const
// is linear in p: any 16 consecutive cells are one conflict-free ds_read_b32, and 18- / 10-cell rows cost no padding
// columns.  All input channels of the three planes cz-2 .. cz stay in LDS for an item; the A fragments (125 per
// channel group) come straight from the packed weights in L2 into registers, one (jy, jx) tap column ahead of their use.
// Per output the accumulation order is fixed: (channel group, jy, jx, jz) -- independent of batch and tiling.
#include "nvf_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kA16 = 125;     // A fragments per channel group: sum over classes of (3-ez)(3-ey)(3-ex)

__host__ __device__ constexpr int a16_index(int ez, int ey, int ex, int jz, int jy, int jx) {
  int base = 0;
  for (int c = 0; c < 4 * ez + 2 * ey + ex; ++c) base += (3 - (c >> 2)) * (3 - ((c >> 1) & 1)) * (3 - (c & 1));
  return base + (jz * (3 - ey) + jy) * (3 - ex) + jx;
}

__global__ void pack_convT16_kernel(const float* __restrict__ wf /* [cin][125][cout] */, float* __restrict__ wp, int cin,
                                    int cout) {
  const int total = (cout / 16) * (cin / 4) * kA16 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx % 64, f = (idx / 64) % kA16, gg = idx / (64 * kA16), g = gg % (cin / 4), cog = gg / (cin / 4);
    int cls = 0, r = f;
    for (;; ++cls) {
      const int n = (3 - (cls >> 2)) * (3 - ((cls >> 1) & 1)) * (3 - (cls & 1));
      if (r < n) break;
      r -= n;
    }
    const int ez = cls >> 2, ey = (cls >> 1) & 1, ex = cls & 1;
    const int jx = r % (3 - ex), jy = (r / (3 - ex)) % (3 - ey), jz = r / ((3 - ex) * (3 - ey));
    const int co = cog * 16 + (lane & 15), ci = 4 * g + (lane >> 4);
    const int kz = ez + 2 * jz, ky = ey + 2 * jy, kx = ex + 2 * jx;
    wp[idx] = wf[(ci * 125 + (kz * 5 + ky) * 5 + kx) * cout + co];
  }
}

// PAD_ = 0: outputs 2 NIN + 3 (up1 / up2); PAD_ = 2 (with output_padding 1): outputs 2 NIN, o = 2 c + e - 2 (conv0,
// up0): the same cells are computed and the out-of-range border is not stored.  COUT_ = 16 or 32 (two row groups).
template <int CIN_, int NIN_, int NCT_, int PAD_ = 0, int COUT_ = 16>
struct T16 {
  static constexpr int CIN = CIN_, NIN = NIN_, NCT = NCT_, PAD = PAD_, COUT = COUT_;
  static constexpr int NOUT = PAD == 0 ? 2 * NIN + 3 : 2 * NIN;
  static constexpr int NCELL = NIN + 2;                        // cells per axis
  static constexpr int NPT = (NCELL * NCELL + 15) / 16;        // column tiles of a cell plane
  static constexpr int NW = 4, CPW = NW * NCT;                 // column tiles per workgroup
  static constexpr int NSPLIT = (NPT + CPW - 1) / CPW;
  static constexpr int PLANE = (NCELL + 3) * NCELL + 18;       // LDS words per (channel, plane), zero margins
  static constexpr int cs_for(int v) { while (v % 32 != 16) ++v; return v; }
  static constexpr int CS = cs_for(3 * PLANE);                 // channel stride: second channel -> banks 16..31
  static constexpr int NG = CIN / 4;
  static constexpr int XS = CIN * CS;                          // input image (all channels, three planes)
  static_assert(XS * 4 <= 80 * 1024, "LDS: two workgroups per CU");
};

// item -> (batch element, cell plane, column split), heaviest planes first (convt_mfma.hip, convT_item)
template <int NIN>
__device__ __forceinline__ void convT16_item(int item, int nsplit, int batch, int& b, int& cz, int& split) {
  constexpr int NCELL = NIN + 2;
  split = item % nsplit;
  b = (item / nsplit) % batch;
  const int k = item / (nsplit * batch);
  if (NIN < 3) { cz = k; return; }                             // up0 (2^3): no full plane at all, keep the order
  cz = k < NIN - 2 ? k + 2 : (k == NIN - 2 ? 1 : (k == NIN - 1 ? NIN : (k == NIN ? 0 : NCELL - 1)));
}

template <class T>
__global__ __launch_bounds__(256, 2) void convT16_k5s2_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ y, int act,
                                                         int items, int batch) {
  constexpr int CIN = T::CIN, NIN = T::NIN, NCELL = T::NCELL, NCT = T::NCT, NPT = T::NPT, PLANE = T::PLANE, CS = T::CS,
                NG = T::NG, NOUT = T::NOUT, PAD = T::PAD, COUT = T::COUT;
  __shared__ __attribute__((aligned(16))) float xs[T::XS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // input rows travel as float4 where the row length allows (NIN % 4 == 0), as NIN-float rows of scalars otherwise
  constexpr int VW = NIN % 4 == 0 ? 4 : 1, RV = NIN / VW;      // vector width, vectors per row
  constexpr int ITEMS = CIN * 3 * NIN * RV;                    // vector loads of an item
  constexpr int NX4 = (ITEMS + 255) / 256;
  float4 xv[NX4];
  const int cog = blockIdx.y;                                  // group of 16 output channels
  auto load_x = [&](int item) {
    int b, cz, split_;
    convT16_item<NIN>(item, T::NSPLIT, batch, b, cz, split_);
    const float* xb = x + (size_t)b * CIN * NIN * NIN * NIN;
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * 256;
      const int xq = i % RV, iy = (i / RV) % NIN, pl = (i / (RV * NIN)) % 3, c = i / (RV * NIN * 3);
      const int zi = cz - 2 + pl;
      const bool ok = i < ITEMS && zi >= 0 && zi < NIN;
      const float* src = xb + (((size_t)c * NIN + zi) * NIN + iy) * NIN + VW * xq;
      if constexpr (VW == 4) xv[u] = ok ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
      else xv[u].x = ok ? src[0] : 0.f;
    }
  };
  auto store_x = [&]() {                                       // planes outside the input are written as zeros
#pragma unroll
    for (int u = 0; u < NX4; ++u) {
      const int i = tid + u * 256;
      if (i < ITEMS) {
        const int xq = i % RV, iy = (i / RV) % NIN, pl = (i / (RV * NIN)) % 3, c = i / (RV * NIN * 3);
        float* dst = xs + c * CS + pl * PLANE + (iy + 2) * NCELL + VW * xq + 2;
        dst[0] = xv[u].x;
        if constexpr (VW == 4) { dst[1] = xv[u].y; dst[2] = xv[u].z; dst[3] = xv[u].w; }
      }
    }
  };
  if ((int)blockIdx.x < items) load_x(blockIdx.x);
  for (int i = tid * 4; i < T::XS; i += 256 * 4) *(float4*)(xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  const int j = lane & 15, kq = lane >> 4;
  const size_t cstride = (size_t)NOUT * NOUT * NOUT;
  const float* wl = wp + (size_t)cog * NG * kA16 * 64 + lane;
  // A fragments of one (jy, jx) tap column: jz = 0..2 x the parity classes that use the tap (<= 20), straight from the
  // packed weights in L2 into registers, one tap column ahead of the MFMAs that use them (no LDS: the 77 KB input image
  // leaves room for a second workgroup on the CU, whose MFMAs cover this one's staging and epilogue)
  float ac[20], an[20];
  auto load_a = [&](auto jyc, auto jxc, int g, float (&dst)[20]) {
    constexpr int jy = decltype(jyc)::value, jx = decltype(jxc)::value;
    int n = 0;
#pragma unroll
    for (int jz = 0; jz < 3; ++jz)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ez = e >> 2, ey = (e >> 1) & 1, ex = e & 1;
        if (jz <= 2 - ez && jy <= 2 - ey && jx <= 2 - ex)
          dst[n++] = wl[((size_t)g * kA16 + a16_index(ez, ey, ex, jz, jy, jx)) * 64];
      }
  };
  // rounds of the stride loop: see convT_k5s2_mfma (convt_mfma.hip) -- items sorted by cost, odd rounds run the CUs
  // backwards on their second workgroup
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
    int b, cz, split;
    convT16_item<NIN>(item, T::NSPLIT, batch, b, cz, split);
    __syncthreads();                                           // zero fill done / the previous item's reads done
    store_x();
    load_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, ac);
    __syncthreads();
    if (item_of(r + 1) < items) load_x(item_of(r + 1));        // in flight under this item's MFMAs
    f32x4 acc[NCT][8];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[c][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    int colbase[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int tl = split * T::CPW + c * T::NW + wave;          // wave-uniform
      // cell p = 16 tl + j reads plane word (cy - jy + 2) NCELL + cx - jx + 2 = p + 2 NCELL + 2 - jy NCELL - jx
      // (a column slot past the last tile of the plane computes on the last tile's data and stores nothing)
      colbase[c] = kq * CS + 16 * min(tl, NPT - 1) + j + 2 * NCELL + 2;
    }
    // MASK: bit jz set = input plane cz - jz exists; a skipped block would have added exact zeros (same bits)
    auto phase = [&](auto maskc) {
    constexpr int MASK = decltype(maskc)::value;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      const float* xg = xs + g * 4 * CS;
      auto column = [&](auto jyc, auto jxc, auto nyc, auto nxc, bool last) {
        constexpr int jy = decltype(jyc)::value, jx = decltype(jxc)::value;
        // prefetch the next tap column's fragments: (jy, jx + 1) ... or the first column of the next channel group
        if (!last) load_a(nyc, nxc, g, an);
        else if (g + 1 < NG) load_a(nyc, nxc, g + 1, an);
        int n = 0;
#pragma unroll
        for (int jz = 0; jz < 3; ++jz) {
          float bv[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) bv[c] = xg[colbase[c] + (2 - jz) * PLANE - jy * NCELL - jx];   // plane cz - jz
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ez = e >> 2, ey = (e >> 1) & 1, ex = e & 1;
            if (jz <= 2 - ez && jy <= 2 - ey && jx <= 2 - ex) {
              const float a = ac[n++];
              if (!((MASK >> jz) & 1)) continue;
#pragma unroll
              for (int c = 0; c < NCT; ++c) acc[c][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[c], acc[c][e], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 20; ++i) ac[i] = an[i];
        __builtin_amdgcn_sched_barrier(0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      column(I0{}, I0{}, I0{}, I1{}, false);
      column(I0{}, I1{}, I0{}, I2{}, false);
      column(I0{}, I2{}, I1{}, I0{}, false);
      column(I1{}, I0{}, I1{}, I1{}, false);
      column(I1{}, I1{}, I1{}, I2{}, false);
      column(I1{}, I2{}, I2{}, I0{}, false);
      column(I2{}, I0{}, I2{}, I1{}, false);
      column(I2{}, I1{}, I2{}, I2{}, false);
      column(I2{}, I2{}, I0{}, I0{}, true);
    }
    };
    {
      // input plane zi = cz - jz (PAD 2 layers have the same cells)
      const int mask = (cz < NIN ? 1 : 0) | ((cz >= 1 && cz <= NIN) ? 2 : 0) | (cz >= 2 ? 4 : 0);     // wave-uniform
      switch (mask) {
        case 7: phase(std::integral_constant<int, 7>{}); break;
        case 3: phase(std::integral_constant<int, 3>{}); break;
        case 6: phase(std::integral_constant<int, 6>{}); break;
        case 1: phase(std::integral_constant<int, 1>{}); break;
        default: phase(std::integral_constant<int, 4>{}); break;
      }
    }
    // epilogue: lane holds rows co = 4 kq + r of cell p for each parity class
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int tl = split * T::CPW + c * T::NW + wave;
      const int p = 16 * tl + j;
      if (tl >= NPT || p >= NCELL * NCELL) continue;
      const int cy = p / NCELL, cx = p % NCELL;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {          // the two x parities of a cell are adjacent outputs: one 8-byte store
        const int oz = 2 * cz + (e >> 2) - PAD, oy = 2 * cy + ((e >> 1) & 1) - PAD, ox = 2 * cx - PAD;
        if (oz < 0 || oy < 0 || oz >= NOUT || oy >= NOUT) continue;
        const bool in0 = ox >= 0 && ox < NOUT, in1 = ox + 1 >= 0 && ox + 1 < NOUT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cog * 16 + 4 * kq + r;
          const float bv = bias ? bias[co] : 0.f;
          float* o = y + ((size_t)b * COUT + co) * cstride + ((size_t)oz * NOUT + oy) * NOUT + ox;
          const float v0 = nvf_act(acc[c][e][r] + bv, act), v1 = nvf_act(acc[c][e + 1][r] + bv, act);
          if (in0 && in1) *(nvf_f2u*)o = nvf_f2u{v0, v1};
          else if (in0) o[0] = v0;
          else if (in1) o[1] = v1;
        }
      }
    }
  }
}

}  // namespace

extern "C" size_t nvf_pack_convT16_mfma_floats(int cin, int cout) { return (size_t)(cout / 16) * (cin / 4) * kA16 * 64; }

// w_fwd = the [cin][125][cout] packed forward weight of a transposed convolution with 16 or 32 output channels
extern "C" int nvf_pack_convT16_mfma(const float* w_fwd, int cin, int cout, float* wp, void* stream) {
  if (!w_fwd || !wp || cin <= 0 || cin % 4 || cout <= 0 || cout % 16) return NVF_EINVAL;
  const int total = (int)nvf_pack_convT16_mfma_floats(cin, cout);
  pack_convT16_kernel<<<(total + 255) / 256, 256, 0, nvf_stream(stream)>>>(w_fwd, wp, cin, cout);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// y[b,co,o] = act(bias[co] + sum_{ci,k : o + pad - k = 2 i} x[b,ci,i] w[ci][k][co]); pad 0: dout = 2 din + 3; pad 2 (with
// output_padding 1): dout = 2 din; 16 or 32 output channels.  NVF_EINVAL = no instantiation for this shape (the caller
// then uses nvf_convT3d_k5s2_fwd).
extern "C" int nvf_convT3d_k5s2_mfma16(const float* x, const float* wp, const float* bias, float* y, int batch, int cin,
                                       int cout, int pad, int din, int act, int variant, void* stream) {
  if (!x || !wp || !y || batch <= 0 || (pad != 0 && pad != 2)) return NVF_EINVAL;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
#define NVF_T16(VAR, CI, CO, PADV, NIN, NCT)                                                           \
  if (rc == 1 && variant == VAR && cin == CI && cout == CO && pad == PADV && din == NIN) {             \
    using T = T16<CI, NIN, NCT, PADV, CO>;                                                             \
    const int items = batch * T::NCELL * T::NSPLIT;                                                    \
    const int cap = 512 / (CO / 16);                       /* two workgroups per CU */                  \
    convT16_k5s2_mfma<T><<<dim3(items < cap ? items : cap, CO / 16), 256, 0, s>>>(x, wp, bias, y, act, items, batch); \
    rc = NVF_OK;                                                                                       \
  }
  NVF_T16(0, 16, 16, 0, 16, 2)   // up2: 21 column tiles per cell plane, 8 per workgroup
  NVF_T16(0, 32, 16, 0, 8, 2)    // up1: all 7 column tiles of a cell plane in one workgroup
  NVF_T16(0, 16, 32, 2, 4, 1)    // conv0 (16 -> 32 channels, 4^3 -> 8^3, padding 2)
  NVF_T16(0, 8, 16, 2, 2, 1)     // up0 (8 -> 16 channels, 2^3 -> 4^3, padding 2)
  NVF_T16(2, 16, 16, 0, 16, 1)
  NVF_T16(3, 16, 16, 0, 16, 3)
  NVF_T16(2, 32, 16, 0, 8, 1)
#undef NVF_T16
  if (rc == 1) return NVF_EINVAL;
  NVF_LAUNCH_CHECK();
  return rc;
}
