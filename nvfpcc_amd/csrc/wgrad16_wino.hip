// Weight gradient of the WIDE decoder's valid 4x4x4 convolutions (16 -> 16 channels; conv2: dY 32^3, X 35^3; conv1: dY 16^3,
// X 19^3) in the reduced-multiplication form of wgrad_wino.h: Winograd F(4x4, 2x2) over (y, x) -- the 4 x 4 taps are the
// output, a 2 x 2 tile of dY the "filter", a 5 x 5 window of X the input -- direct over z on the matrix cores.
// Reference site: the weight half of the autograd backward of F.conv3d, utils/network.py:687 (NVFPCC.py:197).
//
//   dW[co][ci][kz][ky][kx] = sum_{n,z,y,x} dY[n,co,z,y,x] X[n,ci,z+kz,y+ky,x+kx]
//   M[f][kz][co][ci] = sum_{n,z,T} Gh[f][co][n,z,T] Xh[f][ci][n,z+kz,T]   (Gh = G g_T G^T, Xh = B^T x_T B),  dW = A^T M A
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32), every lane useful: rows = the 16 dY channels, columns = the 16 X channels,
// K = four tiles; lane (channel c, tile k) transforms the tile / window it feeds.  One z tap is 25 frequencies x 4
// accumulation registers, so a wave carries TWO taps (kz = 2 kp, 2 kp + 1: 200 registers) and a workgroup runs its items
// twice, once per tap pair.  A wave walks z for one group of four tiles of a tile row: at step z it needs Gh of dY plane
// z and Xh of the X planes z + 2 kp and z + 2 kp + 1 -- the first is the second of the step before and stays in registers,
// so a step is one dY transform (13 vector instructions), one X transform (57) and 50 MFMAs.  Planes arrive as 16-byte
// buffer loads two steps ahead (tile groups start on 32-byte boundaries) and are committed to a per-wave LDS ring of two
// slots after the step's reads; no barrier before the epilogue.  Epilogue per tap pair: A^T M A in registers, the four
// waves leave their tile groups' sums in padded LDS regions, which are added in a fixed order into the workgroup's half of a
// 16 x 16 x 64 slab; the caller's fixed-order reduction (nvf_wgrad_reduce_multi*) adds the slabs.
#include "wino_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ww_u4 __attribute__((ext_vector_type(4)));

namespace {

template <int W_>
struct WW16 {
  static constexpr int W = W_, WQ = W_ + 3, TPR = W_ / 2, NGR = TPR / 4;   // tile groups of four per tile row
  static constexpr int RPW = 4 / NGR;                                      // tile rows a workgroup's four waves cover
  static_assert(NGR == 4 || NGR == 2, "conv2 (32) / conv1 (16)");
  static constexpr int XRS = 12, XCS = 68, XPS = 16 * XCS;                 // X window rows of 11 (+1) words; channel
  static constexpr int GRS = 8, GCS = 20, GPS = 16 * GCS;                  // stride = 4 (mod 32): two passes per read
  static constexpr int WLDS = 2 * XPS + 2 * GPS;                           // floats of LDS per wave (two slots each)
  static constexpr int REGION = 8192 + 256;                                // [co][ci][32 taps], one pad word per 32
  static constexpr int LDSF = 4 * WLDS > 4 * REGION ? 4 * WLDS : 4 * REGION;
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

struct WW16Dims {
  int batch, items, items_per_wg, zsplit;     // items = (block, tile-row group, z part)
};

template <class C>
__global__ __launch_bounds__(256) void wgrad16_k4_wino(const float* __restrict__ g, const float* __restrict__ x,
                                                       float* __restrict__ slabs, WW16Dims d) {
  constexpr int W = C::W, WQ = C::WQ, XRS = C::XRS, XCS = C::XCS, XPS = C::XPS, GRS = C::GRS, GCS = C::GCS, GPS = C::GPS;
  __shared__ __attribute__((aligned(16))) float lds[C::LDSF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = lane & 15, k = lane >> 4;                      // MFMA row / column channel, K index = tile of the group
  float* xr = lds + wave * C::WLDS;
  float* gr = xr + 2 * XPS;
  constexpr int kOob = 0x7ffffff0;
  // staging descriptors: X plane slice = 80 (channel, row) rows of three 16-byte pieces; dY slice = 32 rows of two
  int voffx[4], ldsx[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = lane + 64 * u, r = e / 3, seg = e % 3, c = r / 5, row = r % 5;
    voffx[u] = e < 240 ? ((c * WQ * WQ + row) * WQ + 4 * seg) * 4 : kOob;
    ldsx[u] = e < 240 ? c * XCS + row * XRS + 4 * seg : -1;
  }
  const int voffg = (((lane >> 2) * W * W + ((lane >> 1) & 1)) * W + 4 * (lane & 1)) * 4;
  const int ldsg = (lane >> 2) * GCS + ((lane >> 1) & 1) * GRS + 4 * (lane & 1);
  const int tg = wave % C::NGR, trl = wave / C::NGR;            // this wave's tile group and tile row inside the item
  // XCD-local work: workgroups bx, bx + 8, ... share an XCD (and its L2); they take consecutive item ranges -- the
  // neighbouring tile rows of the same blocks, which walk z together and share every X row they read
  const int nwg = (int)gridDim.x, bx = (int)blockIdx.x;
  const int bxl = nwg % 8 == 0 ? (bx & 7) * (nwg >> 3) + (bx >> 3) : bx;
  const int first = bxl * d.items_per_wg, last = min(first + d.items_per_wg, d.items);
  const int nrg = C::TPR / C::RPW;                              // tile-row groups per block
  float* slab = slabs + (size_t)blockIdx.x * 16384;
  // A^T with B^T's factors folded in: rows = tap, columns = frequency
  const float AT[4][5] = {{0.5f, 0.5f, 1.f / 6.f, -1.f / 6.f, 0.f},
                          {0.f, 0.5f, -1.f / 6.f, -2.f / 6.f, 0.f},
                          {0.f, 0.5f, 1.f / 6.f, -4.f / 6.f, 0.f},
                          {0.f, 0.5f, -1.f / 6.f, -8.f / 6.f, 1.f}};
#pragma unroll 1
  for (int kp = 0; kp < 2; ++kp) {                              // tap pair kz = 2 kp, 2 kp + 1
    f32x4 acc[2][25];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int f = 0; f < 25; ++f) acc[s][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int item = first; item < last; ++item) {
      const int it = __builtin_amdgcn_readfirstlane(item);
      const int zs = it % d.zsplit, rg = (it / d.zsplit) % nrg, n = it / (d.zsplit * nrg);
      const int tr = rg * C::RPW + trl;
      const int per = (W + d.zsplit - 1) / d.zsplit;
      const int z0 = zs * per, z1 = min(z0 + per, W);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * 16 * WQ * WQ * WQ), 0,
                                                                           16 * WQ * WQ * WQ * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g + (size_t)n * 16 * W * W * W), 0,
                                                                            16 * W * W * W * 4, 0x00020000);
      // two register sets of staged planes: the loads of step z + 2's planes are issued at step z and committed at the
      // end of step z + 1 (one wave per SIMD: a plane needs more than one step's MFMAs to arrive)
      ww_u4 xvA[4], gvA, xvB[4], gvB;
      auto load_x = [&](int p, ww_u4 (&xv)[4]) {                // X plane p (always inside the tensor when called)
        const int so = ((p * WQ + 2 * tr) * WQ + 8 * tg) * 4;           // wave-uniform by construction
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx[u], so, 0);
      };
      auto load_g = [&](int p, ww_u4& gv) {
        const int so = ((p * W + 2 * tr) * W + 8 * tg) * 4;
        gv = __builtin_amdgcn_raw_buffer_load_b128(rg_, voffg, so, 0);
      };
      auto commit_x = [&](int slot, const ww_u4 (&xv)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ldsx[u] >= 0) *(ww_u4*)(xr + slot * XPS + ldsx[u]) = xv[u];
      };
      auto commit_g = [&](int slot, const ww_u4& gv) { *(ww_u4*)(gr + slot * GPS + ldsg) = gv; };
      auto xform = [&](int slot, float (&Xh)[25]) {             // Xh of this lane's (channel, tile) window in `slot`
        const float* xp = xr + slot * XPS + ch * XCS + 2 * k;
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
      };
      // one z step: dY plane z (slot z & 1), X plane z + 2 kp + 1 (slot sx) new, X plane z + 2 kp from the step before.
      // `cur` holds the planes of step z + 1 (committed at the end), `nxt` receives those of step z + 2
      auto step = [&](int z, int sx, const float (&Xo)[25], float (&Xn)[25], ww_u4 (&xc)[4], ww_u4& gc, ww_u4 (&xn)[4],
                      ww_u4& gn) {
        if (z + 2 < z1) { load_g(z + 2, gn); load_x(z + 2 * kp + 3, xn); }
        const float* gp = gr + (z & 1) * GPS + ch * GCS + 2 * k;
        const wino_f2 g0 = *(const wino_f2*)gp, g1 = *(const wino_f2*)(gp + GRS);
        float Gh[25];
        {                                                       // Gh = G g G^T, G = [1 0; 1 1; 1 -1; 1 2; 0 1]
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
        xform(sx, Xn);
#pragma unroll
        for (int f = 0; f < 25; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gh[f], Xo[f], acc[0][f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 25; ++f) acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gh[f], Xn[f], acc[1][f], 0, 0, 0);
        if (z + 1 < z1) {                                       // dY plane z is dead: plane z + 1 takes the other slot;
          commit_g((z + 1) & 1, gc);                            // X: the slot NOT read by this step takes plane z + 2 kp + 2
          commit_x(sx ^ 1, xc);
        }
      };
      // prologue: dY plane z0; X planes z0 + 2 kp (slot 0) and z0 + 2 kp + 1 (slot 1); the planes of step z0 + 1 in set A
      load_g(z0, gvA); commit_g(z0 & 1, gvA);
      load_x(z0 + 2 * kp, xvA); commit_x(0, xvA);
      load_x(z0 + 2 * kp + 1, xvA); commit_x(1, xvA);
      if (z0 + 1 < z1) { load_g(z0 + 1, gvA); load_x(z0 + 2 * kp + 2, xvA); }
      float XA[25], XB[25];
      xform(0, XA);
      int z = z0;
#pragma unroll 1
      for (;;) {                                                // two steps per trip: the X registers and the sets swap roles
        step(z, 1, XA, XB, xvA, gvA, xvB, gvB);
        if (++z >= z1) break;
        step(z, 0, XB, XA, xvB, gvB, xvA, gvA);
        if (++z >= z1) break;
      }
    }
    // ---- epilogue of the tap pair: A^T M A, the four waves' sums in fixed order, half a slab ----
    __syncthreads();                                            // every wave is done with its ring: the region overlays it
    {
      float* reg = lds + wave * C::REGION;                      // one region per wave, all four transform at once
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * k + r, ci = ch;
#pragma unroll
        for (int kzl = 0; kzl < 2; ++kzl) {
          float t[5][4];                                        // t[fy][kx] = sum_fx AT[kx][fx] M[fy][fx]
#pragma unroll
          for (int fy = 0; fy < 5; ++fy)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
              float s = 0.f;
#pragma unroll
              for (int fx = 0; fx < 5; ++fx)
                if (AT[kx][fx] != 0.f) s = fmaf(AT[kx][fx], acc[kzl][fy * 5 + fx][r], s);
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
              reg[(co * 16 + ci) * 33 + kzl * 16 + ky * 4 + kx] = s;           // 32 taps + one pad word per (co, ci)
            }
        }
      }
    }
    __syncthreads();
    for (int o = tid; o < 8192; o += 256) {
      const int p = (o >> 5) * 33 + (o & 31);
      slab[(o >> 5) * 64 + kp * 32 + (o & 31)] =
          ((lds[p] + lds[C::REGION + p]) + lds[2 * C::REGION + p]) + lds[3 * C::REGION + p];
    }
    __syncthreads();                                            // the region is the rings of the next tap pair
  }
}

template <class C>
static int launch_ww16(const float* dy, const float* x, float* slabs, int batch, int zsplit, int max_slabs, int* nslab,
                       hipStream_t s) {
  WW16Dims d{};
  d.batch = batch; d.zsplit = zsplit;
  d.items = batch * (C::TPR / C::RPW) * zsplit;
  const int cap = max_slabs < 256 ? max_slabs : 256;            // one workgroup per CU (200 accumulation registers a wave)
  int n = d.items < cap ? d.items : cap;
  d.items_per_wg = (d.items + n - 1) / n;
  n = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  *nslab = n;
  wgrad16_k4_wino<C><<<n, 256, 0, s>>>(dy, x, slabs, d);
  return NVF_OK;
}

}  // namespace

// Partial sums of dW[16][16][4][4][4] = the weight gradient of a valid 4^3 convolution with 16 -> 16 channels in the
// Winograd (y, x) form: dy [batch, 16, w^3] (w = 32: conv2, 16: conv1), x [batch, 16, (w + 3)^3].  Writes *nslab <= max_slabs
// slabs of 16384 floats (layout [co][ci][kz][ky][kx]) to `slabs`; the caller adds them in a fixed order (a jtotal = 16384
// job of nvf_wgrad_reduce_multi*).  zsplit: z steps of a (block, tile rows) item split over this many items (0: default).
// fp32 error against float64 ~4e-6 of max |dW| (direct form 1e-6; the gradient goldens are held to 2e-4).
extern "C" int nvf_wgrad16_k4_wino_partial(const float* dy, const float* x, float* slabs, int batch, int w, int zsplit,
                                           int max_slabs, int* nslab, void* stream) {
  if (!dy || !x || !slabs || !nslab || batch <= 0 || zsplit < 0 || zsplit > 8 || max_slabs <= 0) return NVF_EINVAL;
  int rc;
  if (w == 32) rc = launch_ww16<WW16<32>>(dy, x, slabs, batch, zsplit ? zsplit : 1, max_slabs, nslab, nvf_stream(stream));
  else if (w == 16) rc = launch_ww16<WW16<16>>(dy, x, slabs, batch, zsplit ? zsplit : 4, max_slabs, nslab, nvf_stream(stream));
  else return NVF_EINVAL;
  if (rc != NVF_OK) return rc;
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
