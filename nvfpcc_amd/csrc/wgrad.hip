// Weight gradients of the NVF decoder's convolutions on gfx950.
//
//   dw[a][b][k] = sum_{n,i} p[n,a,i] * q[n,b, S*i - pad + k]
//
// replaces the bwd-weight half of torch's convolution_backward behind
// F.conv3d / F.conv_transpose3d (utils/network.py:621, 687, 741) -- 66 % of the
// reference's CPU step time.
//
// Mapping: one lane per kernel tap k (4^3 = 64 taps fill a wave exactly; 5^3 = 125
// use two waves; 3^3 = 27 lanes of one wave), one wave group per q-channel b, and the
// A p-channels as per-lane accumulators.  For a fixed grid position i every lane needs
// the SAME p[a,i] -- so p is fetched with scalar loads into SGPRs and costs no VGPR/LDS
// traffic -- and its own q[b, S*i + k], a ds_read_b32 from the staged q tile whose row /
// plane strides are chosen (= K mod 32, K^2 mod 32) so the 32-lane groups hit distinct
// banks.  The reduction over positions and batch happens in registers; each workgroup
// writes one partial slab and a second kernel adds the slabs in a fixed order, so the
// result is reproducible run to run (no float atomics).
#include "nvf_common.h"
#include <cstdio>
#include <cstdlib>
#include "step_ctx.h"
#include "heads_wgrad_mfma.h"

static const int kMaxSlabs = 512;
static const int kSumChunks = 128;
struct MultiSumDesc;
__global__ void multi_channel_sum_final(MultiSumDesc d, const float* __restrict__ part);
extern "C" size_t nvf_multi_channel_sum_workspace(int total_channels);
__global__ void wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw, int nslab, int jtotal,
                             int accumulate);

struct WgDims {
  int batch, bc;           // batch, number of q channels
  int dp, hp, wp;          // p grid
  int dq, hq, wq;          // q grid
  int pad;
  int tiles_x, tiles_y, tiles_z;
  int items, items_per_wg; // work items (n, tile) and how many each workgroup walks
  int out_mode, jtotal;    // slab layout, slab length A*Bc*K^3
  float* bias_slab;        // matrix-core 4^3 gradient only, optional: per workgroup the 8 channel sums of the dY tiles it
                           // walked (the layer's bias gradient: the items partition dY) -- slab [workgroup][8]
};

// general shape, one WAVE per output: lanes stride over the (n, iz, iy, ix) positions, fixed-order wave sum
__global__ void wgrad_naive(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ dw, int a_ch,
                            int k, int stride, WgDims d, int accumulate) {
  int k3 = k * k * k;
  int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (j >= a_ch * d.bc * k3) return;
  int kk = j % k3, b = (j / k3) % d.bc, a = j / (k3 * d.bc);
  int kz = kk / (k * k), ky = (kk / k) % k, kx = kk % k;
  const long plane = (long)d.hp * d.wp, vol = (long)d.dp * plane, total = (long)d.batch * vol;
  float acc = 0.f;
  for (long e = lane; e < total; e += 64) {
    const int n = (int)(e / vol);
    const long r = e - (long)n * vol;
    const int iz = (int)(r / plane), iy = (int)((r - (long)iz * plane) / d.wp), ix = (int)(r % d.wp);
    const int qz = iz * stride - d.pad + kz, qy = iy * stride - d.pad + ky, qx = ix * stride - d.pad + kx;
    if (qz < 0 || qz >= d.dq || qy < 0 || qy >= d.hq || qx < 0 || qx >= d.wq) continue;
    const float pv = p[((size_t)n * a_ch + a) * vol + r];
    const float qv = q[(((size_t)n * d.bc + b) * d.dq + qz) * d.hq * d.wq + qy * d.wq + qx];
    acc = fmaf(pv, qv, acc);
  }
  acc = nvf_wave_sum(acc);
  if (lane != 0) return;
  int o = d.out_mode == 0 ? j : (b * a_ch + a) * k3 + (k3 - 1 - kk);
  dw[o] = accumulate ? dw[o] + acc : acc;
}

template <int A_, int KS_, int S_, int NB_, int TX_, int TY_, int TZ_, int IXU_ = 0>
struct WCfg {
  static constexpr int A = A_, KS = KS_, S = S_, NB = NB_, TX = TX_, TY = TY_, TZ = TZ_;
  static constexpr int K3 = KS * KS * KS;
  // x-loop unroll: each unrolled step holds 4*A scalar registers of p; keep the live set under ~64 SGPRs
  static constexpr int IXU = IXU_ > 0 ? IXU_ : (A_ <= 8 ? 2 : 1);
  static constexpr int WPB = (K3 + 63) / 64;          // waves per q channel
  static constexpr int NT = NB * WPB * 64;
  static constexpr int QX = (TX - 1) * S + KS, QY = (TY - 1) * S + KS, QZ = (TZ - 1) * S + KS;
  static constexpr int mod32(int v, int r) { return v + ((r - v % 32) + 32) % 32; }  // smallest >= v, == r (mod 32)
  static constexpr int QRS = mod32(QX, KS % 32);              // row stride
  static constexpr int QPS = mod32(QY * QRS, (KS * KS) % 32); // plane stride
  static constexpr int QCS = QZ * QPS;                        // channel stride
  static constexpr int PT = A * TZ * TY * TX;                  // the p tile [a][iz][iy][ix], read back as broadcasts
  static constexpr int POFF = (NB * QCS + 3) / 4 * 4;
  static constexpr int LDSF = POFF + PT;
  static_assert(TX % 4 == 0, "p rows are read four at a time");
  static_assert(NT <= 1024, "workgroup size");
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

template <class C>
__device__ __forceinline__ void wgrad_tiled_body(const float* __restrict__ p, const float* __restrict__ q,
                                                 float* __restrict__ slabs, const WgDims& d, int bx, int by, float* lds) {
  constexpr int A = C::A, KS = C::KS, S = C::S, NB = C::NB, TX = C::TX, TY = C::TY, TZ = C::TZ, K3 = C::K3;
  constexpr int QX = C::QX, QY = C::QY, QZ = C::QZ, QRS = C::QRS, QPS = C::QPS, QCS = C::QCS, NT = C::NT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bb = wave / C::WPB;                 // q channel within this workgroup
  const int kk = (wave % C::WPB) * 64 + lane;   // tap
  const bool valid = kk < K3;
  const int kz = kk / (KS * KS), ky = (kk / KS) % KS, kx = kk % KS;
  const int b0 = by * NB;
  const int lane_off = valid ? bb * QCS + kz * QPS + ky * QRS + kx : 0;
  float acc[A];
#pragma unroll
  for (int a = 0; a < A; ++a) acc[a] = 0.f;

  const int tiles = d.tiles_x * d.tiles_y * d.tiles_z;
  const int first = bx * d.items_per_wg;
  const int last = min(first + d.items_per_wg, d.items);
  const size_t pplane = (size_t)d.hp * d.wp, qplane = (size_t)d.hq * d.wq;
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    const int n = item / tiles, t = item % tiles;
    const int x0 = (t % d.tiles_x) * TX, y0 = ((t / d.tiles_x) % d.tiles_y) * TY, z0 = (t / (d.tiles_x * d.tiles_y)) * TZ;
    const int qx0 = x0 * S - d.pad, qy0 = y0 * S - d.pad, qz0 = z0 * S - d.pad;
    if (item != first) __syncthreads();
    nvf_stage_rows<NT, NB * QZ * QY, QX, QRS, 8>(
        q + (size_t)n * d.bc * d.dq * qplane, lds, tid,
        [&](int r, int xx, bool& ok) -> size_t {
          const int yy = r % QY, t2 = r / QY, zz = t2 % QZ, c = t2 / QZ;
          const int gx = qx0 + xx, gy = qy0 + yy, gz = qz0 + zz;
          ok = b0 + c < d.bc && gx >= 0 && gx < d.wq && gy >= 0 && gy < d.hq && gz >= 0 && gz < d.dq;
          return ((size_t)(b0 + c) * d.dq + gz) * qplane + (size_t)gy * d.wq + gx;
        },
        [&](int r, int xx) {
          const int yy = r % QY, t2 = r / QY, zz = t2 % QZ, c = t2 / QZ;
          return c * QCS + zz * QPS + yy * QRS + xx;
        });
    // the p tile goes through LDS as well: every lane needs the same p values, and fetching them with scalar loads
    // in the tap loop is a chain of exposed memory latencies (one per row)
    const float* pn = p + (size_t)n * A * d.dp * pplane;
    float* pl = lds + C::POFF;
    for (int i = tid; i < C::PT / 4; i += NT) {
      const int xq = i % (TX / 4), r = i / (TX / 4), iy = r % TY, t2 = r / TY, iz = t2 % TZ, a = t2 / TZ;
      *(float4*)(pl + 4 * i) =
          *(const float4*)(pn + ((size_t)a * d.dp + z0 + iz) * pplane + (size_t)(y0 + iy) * d.wp + x0 + 4 * xq);
    }
    __syncthreads();
#pragma unroll 1
    for (int iz = 0; iz < TZ; ++iz) {
#pragma unroll 1
      for (int iy = 0; iy < TY; ++iy) {
        const float* prow = pl + (iz * TY + iy) * TX;                                  // wave-uniform LDS address
        const float* qrow = lds + lane_off + iz * S * QPS + iy * S * QRS;
#pragma unroll C::IXU
        for (int ix = 0; ix < TX; ix += 4) {
          float qv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) qv[j] = qrow[(ix + j) * S];
#pragma unroll
          for (int a = 0; a < A; ++a) {
            const float4 pv = *(const float4*)(prow + a * TZ * TY * TX + ix);           // broadcast ds_read_b128
            acc[a] = fmaf(pv.x, qv[0], acc[a]);
            acc[a] = fmaf(pv.y, qv[1], acc[a]);
            acc[a] = fmaf(pv.z, qv[2], acc[a]);
            acc[a] = fmaf(pv.w, qv[3], acc[a]);
          }
        }
      }
    }
  }
  if (!valid || b0 + bb >= d.bc) return;
  float* slab = slabs + (size_t)bx * d.jtotal;
  const int b = b0 + bb;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int o = d.out_mode == 0 ? (a * d.bc + b) * K3 + kk : (b * A + a) * K3 + (K3 - 1 - kk);
    slab[o] = acc[a];
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) void wgrad_tiled(const float* __restrict__ p, const float* __restrict__ q,
                                                     float* __restrict__ slabs, WgDims d) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDSF];
  wgrad_tiled_body<C>(p, q, slabs, d, blockIdx.x, blockIdx.y, lds);
}

// up1's and conv0's weight gradients (the two small transposed convolutions of the narrow trunk) in one launch:
// each is latency-bound on part of the chip, together they cost the longer of the two.  Same bodies, same slabs.
struct WgTiled2 {
  const float* p[2];
  const float* q[2];
  float* slabs[2];
  WgDims d[2];
  int32_t nx[2], ny[2];
  int32_t conv0_mfma;      // > 0: job 1 (conv0) runs on the matrix cores instead: this many workgroups (5 per block), one slab per block
};

template <class C0, class C1>
__global__ __launch_bounds__(C0::NT) void wgrad_tiled2_kernel(WgTiled2 m) {
  static_assert(C0::NT == C1::NT, "one workgroup size");
  __shared__ __attribute__((aligned(16))) float lds[C0::LDSF > C1::LDSF ? C0::LDSF : C1::LDSF];
  int bid = blockIdx.x;
  if (bid < m.nx[0] * m.ny[0]) { wgrad_tiled_body<C0>(m.p[0], m.q[0], m.slabs[0], m.d[0], bid % m.nx[0], bid / m.nx[0], lds); return; }
  bid -= m.nx[0] * m.ny[0];
  wgrad_tiled_body<C1>(m.p[1], m.q[1], m.slabs[1], m.d[1], bid % m.nx[1], bid / m.nx[1], lds);
}

// ---------------------------------------------------------------------------
// MFMA weight gradient for the 4^3, 8 -> 8 channel convolutions (conv1, conv2: 75 % of all
// bwd-weight FLOPs).  v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain at the fp32 vector rate,
// and here it loses nothing to the tiny channel counts:
//   rows  m = (co, sA)  A[m][kk] = dY[co, z, y, x0 + kk - sA]          sA in {0,1}
//   cols  n = (ci, sB)  B[kk][n] = X [ci, z+kz, y+ky, x0 + kk + 2 sB]  sB in {0,1}
//   D[m][n] += sum_kk A B  =  dW[co][ci][kz][ky][kx = sA + 2 sB]
// so one 16x16 tile per (kz,ky) covers the four kx taps with every MFMA lane useful; the K
// dimension runs along x in groups of four (x padded to a multiple of 4, dY zero outside [0,W)).
// The A fragment is shared by the 16 (kz,ky) tiles: 17 ds_read_b32 per 16 MFMAs.  Both tiles sit in
// LDS planar [c][z][y][x] with the channel stride == 4 (mod 32), so the 32 lanes of a ds_read group
// (8 channels x 4 x-offsets) hit 32 distinct banks and staging stays a plain coalesced copy.
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Tile staging through buffer loads: the descriptor (base of the batch element, its size) and the per-segment offset
// are wave-uniform and live in scalar registers, the per-thread part is ONE constant voffset -- a load is an s_add and
// a buffer_load, no vector address arithmetic (flat loads cost ~4 VALU instructions + waits per element: 64-bit
// per-lane adds, 13 % of the conv2 gradient's time).  A lane that has nothing to fetch carries a voffset beyond the
// descriptor's range: the range check returns 0 for it.
#ifndef NVF_WG_BUF
#define NVF_WG_BUF 1
#endif
constexpr int kWgOob = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float wg_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wg_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const wg_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

#ifndef NVF_WG_EPI_REGIONS
#define NVF_WG_EPI_REGIONS 1
#endif
#ifndef NVF_WG_DBG
#define NVF_WG_DBG 0
#endif
constexpr int kWgRegion = 4096 + 64;                       // one wave's sums in the epilogue (padded)
constexpr int kWgEpiFloats = NVF_WG_EPI_REGIONS ? 4 * kWgRegion : 4096;

template <int W_, int TZ_, int TY_>
struct MCfg {
  static constexpr int W = W_, TZ = TZ_, TY = TY_;
  static constexpr int NG = (W + 1 + 3) / 4;             // x groups: x runs over [0, W] (W+1 values)
  static constexpr int GRS = NG * 4 + 2;                 // dY row: u = x + 1 in [0, 4 NG]
  static constexpr int XRS = NG * 4 + 2;                 // X row: x + 2 sB up to 4 NG + 1
  static constexpr int mod32(int v, int r) { return v + ((r - v % 32) + 32) % 32; }
  static constexpr int GCS = mod32(TZ * TY * GRS, 4);
  static constexpr int XCS = mod32((TZ + 3) * (TY + 3) * XRS, 4);
  static constexpr int GOFF = 0, XOFF = 8 * GCS;
  static constexpr int LDSF = 8 * GCS + 8 * XCS;
  static constexpr int NT = 256;
  static_assert(LDSF >= 4096, "the staging area doubles as the 4096-float reduction buffer");
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

template <class C>
__device__ __forceinline__ void wgrad_k4_mfma_body(const float* __restrict__ g, const float* __restrict__ x,
                                                   float* __restrict__ slabs, const WgDims& d, int bx, float* lds) {
  constexpr int W = C::W, TZ = C::TZ, TY = C::TY, NG = C::NG, GRS = C::GRS, XRS = C::XRS, GCS = C::GCS, XCS = C::XCS;
  float* ldsG = lds + C::GOFF;
  float* ldsX = lds + C::XOFF;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ch = lane & 7, sh = (lane >> 3) & 1, kk = lane >> 4;
  const int laneA = ch * GCS + kk - sh + 1;
  const int laneB = ch * XCS + kk + 2 * sh;
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int WQ = W + 3;                       // q grid extent (valid conv, k = 4)
  const int tiles_y = W / TY, tiles_z = W / TZ, tiles = tiles_y * tiles_z;
  const int first = bx * d.items_per_wg;
  const int last = min(first + d.items_per_wg, d.items);
  // Tile staging through registers: ALL global loads of an item are issued before any is waited for, and the next
  // item's loads are in flight while this item's MFMAs issue.  dY: aligned float4 rows.  X: for one (channel, plane)
  // the TY + 3 rows of W + 3 words are CONTIGUOUS in memory, so the tile is NSEG = 8 (TZ + 3) segments of SEG words;
  // thread t copies word t (+256 ..) of every segment: the global address is a per-item scalar base plus a
  // compile-time segment offset plus t, the LDS address a per-thread constant plus a compile-time offset -- no
  // per-element index arithmetic.  The zero padding of the LDS image (u = 0, u > W, xx >= WQ) is written once.
  constexpr int RX = (TZ + 3) * (TY + 3);                 // X rows per channel
  constexpr int NG4 = 8 * TZ * TY * (W / 4);              // float4 items of the dY tile
  constexpr int SEG = (TY + 3) * (W + 3), PPS = (SEG + 255) / 256, NSEG = 8 * (TZ + 3);
  constexpr int UG = (NG4 + 255) / 256, UX = NSEG * PPS;
  float4 gv[UG];
  float xv[UX];
  int xlo[PPS];                                           // LDS word of this thread's element inside a segment
#pragma unroll
  for (int part = 0; part < PPS; ++part) {
    const int e = tid + part * 256;
    xlo[part] = (e / (W + 3)) * XRS + e % (W + 3);
  }
#if NVF_WG_BUF
  // dY: thread t owns float4 (xq, yy, zz, c0) of the tile, c = c0 + u * CSTEP; X: word e = tid (+ 256 part) of a segment
  constexpr int PER = (W / 4) * TY * TZ, CSTEP = 256 / PER;
  static_assert(256 % PER == 0 && NG4 % 256 == 0, "a thread's dY float4s differ by whole channels");
  const int gxq = tid % (W / 4), gr = tid / (W / 4), gyy = gr % TY, gt2 = gr / TY, gzz = gt2 % TZ, gc0 = gt2 / TZ;
  const int gvoff = (((gc0 * W + gzz) * W + gyy) * W + 4 * gxq) * 4;
  int xvoff[PPS];
#pragma unroll
  for (int part = 0; part < PPS; ++part) xvoff[part] = tid + part * 256 < SEG ? (tid + part * 256) * 4 : kWgOob;
  auto load = [&](int item) {
    const int n = item / tiles, t = item % tiles;
    const int y0 = (t % tiles_y) * TY, z0 = (t / tiles_y) * TZ;
    const __amdgpu_buffer_rsrc_t rg = wg_rsrc(g + (size_t)n * 8 * W * W * W, 8 * W * W * W * 4);
    const int gs = ((z0 * W + y0) * W) * 4;
#pragma unroll
    for (int u = 0; u < UG; ++u) gv[u] = wg_ld4(rg, gvoff, gs + u * CSTEP * W * W * W * 4);
    const __amdgpu_buffer_rsrc_t rx = wg_rsrc(x + (size_t)n * 8 * WQ * WQ * WQ, 8 * WQ * WQ * WQ * 4);
    // a RUNNING scalar offset (one s_add between two loads): with independent offsets the compiler computes all 56
    // up front and spills scalar registers
    int xs = ((z0 * WQ + y0) * WQ) * 4;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int seg = u / PPS, part = u % PPS, c = seg / (TZ + 3), zz = seg % (TZ + 3);
      xv[u] = wg_ld(rx, xvoff[part], xs);
      if (part == PPS - 1) {
        const int nseg = seg + 1, nc = nseg / (TZ + 3), nz = nseg % (TZ + 3);
        xs += ((nc * WQ + nz) - (c * WQ + zz)) * WQ * WQ * 4;
        asm volatile("" : "+s"(xs));
      }
    }
  };
#else
  auto load = [&](int item) {
    const int n = item / tiles, t = item % tiles;
    const int y0 = (t % tiles_y) * TY, z0 = (t / tiles_y) * TZ;
    const float* gn = g + (size_t)n * 8 * W * W * W;
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int i = tid + u * 256;
      const int xq = i % (W / 4), r = i / (W / 4), yy = r % TY, t2 = r / TY, zz = t2 % TZ, c = t2 / TZ;
      gv[u] = i < NG4 ? *(const float4*)(gn + (((size_t)c * W + z0 + zz) * W + y0 + yy) * W + 4 * xq)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* xt = x + (((size_t)n * 8 * WQ + z0) * WQ + y0) * WQ;      // wave-uniform
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int seg = u / PPS, part = u % PPS, c = seg / (TZ + 3), zz = seg % (TZ + 3);
      const int e = tid + part * 256;
      xv[u] = e < SEG ? xt[(c * WQ + zz) * WQ * WQ + e] : 0.f;
    }
  };
#endif
  float bsum[UG];                                         // channel sums of dY (this thread's float4s), per u
#pragma unroll
  for (int u = 0; u < UG; ++u) bsum[u] = 0.f;
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int i = tid + u * 256;
      if (i < NG4) {
        const int xq = i % (W / 4), r = i / (W / 4), c = r / (TZ * TY);
        float* dst = ldsG + c * GCS + (r - c * TZ * TY) * GRS + 4 * xq + 1;      // u = x + 1
        dst[0] = gv[u].x; dst[1] = gv[u].y; dst[2] = gv[u].z; dst[3] = gv[u].w;
        bsum[u] += (gv[u].x + gv[u].y) + (gv[u].z + gv[u].w);
      }
    }
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int seg = u / PPS, part = u % PPS, c = seg / (TZ + 3), zz = seg % (TZ + 3);
      if (tid + part * 256 < SEG) ldsX[c * XCS + zz * (TY + 3) * XRS + xlo[part]] = xv[u];
    }
  };
  for (int i = tid; i < C::LDSF; i += 256) lds[i] = 0.f;
  if (first < last) load(first);
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    const int t = item % tiles;
    const int y0 = (t % tiles_y) * TY, z0 = (t / tiles_y) * TZ;
    (void)y0; (void)z0;
    __syncthreads();                                        // zero fill done / previous item no longer being read
#if NVF_WG_DBG != 1                                         // tuning builds: 1 = stage the first item only, 2 = no MFMAs
    store();
#else
    if (item == first) store();
#endif
    __syncthreads();
#if NVF_WG_DBG != 1
    if (item + 1 < last) load(item + 1);
#endif
#if NVF_WG_DBG == 2
    if (d.items >= 0) continue;
#endif
#pragma unroll 1
    for (int row = wave; row < TZ * TY; row += 4) {
      const int zz = row / TY, yy = row % TY;
      const float* pa = ldsG + laneA + row * GRS;
      const float* pb = ldsX + laneB + (zz * (TY + 3) + yy) * XRS;
      // operands of x-group xg+1 are fetched while the 16 MFMAs of group xg issue
      float a_cur = pa[0], b_cur[16];
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) b_cur[t2] = pb[((t2 >> 2) * (TY + 3) + (t2 & 3)) * XRS];
#pragma unroll
      for (int xg = 0; xg < NG; ++xg) {
        float a_nxt = 0.f, b_nxt[16];
        if (xg + 1 < NG) {
          a_nxt = pa[(xg + 1) * 4];
#pragma unroll
          for (int t2 = 0; t2 < 16; ++t2) b_nxt[t2] = pb[((t2 >> 2) * (TY + 3) + (t2 & 3)) * XRS + (xg + 1) * 4];
        }
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t2], acc[t2], 0, 0, 0);
        if (xg + 1 < NG) {
          a_cur = a_nxt;
#pragma unroll
          for (int t2 = 0; t2 < 16; ++t2) b_cur[t2] = b_nxt[t2];
        }
      }
    }
  }
  // the layer's bias gradient, this workgroup's share: thread t holds channels c0 + u * CSTEP (c0 = t / PER);
  // wave sums, then the waves of one c0 in ascending order -- a fixed order
  if (d.bias_slab) {
    constexpr int PERB = (W / 4) * TY * TZ, CSTEPB = 256 / PERB, WPC = PERB / 64;
    static_assert(PERB % 64 == 0 && 256 % PERB == 0, "whole waves per channel group");
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const float w = nvf_wave_sum(bsum[u]);
      if (lane == 0) lds[wave * UG + u] = w;
    }
    __syncthreads();
    if (tid < 8) {
      const int c0 = tid % CSTEPB, u = tid / CSTEPB;
      float t = 0.f;
      for (int w = 0; w < WPC; ++w) t += lds[(c0 * WPC + w) * UG + u];
      d.bias_slab[(size_t)bx * 8 + tid] = t;
    }
  }
  // cross-wave sum through LDS (reusing the staging area), then one slab per workgroup.  Every wave drops its 4096
  // sums into a region of its own (a word of padding per 64: the (co, ci) pairs of a store land on different banks),
  // then all threads add the four regions in wave order -- the same ((w0 + w1) + w2) + w3 as a serial pass, without
  // four rounds of dependent LDS read-modify-writes.
  __syncthreads();
  const int nn = lane & 15, ci = nn & 7, sB = nn >> 3;
#if NVF_WG_EPI_REGIONS
  {
    float* reg = lds + wave * kWgRegion;
#pragma unroll
    for (int t2 = 0; t2 < 16; ++t2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * (lane >> 4) + r, co = m & 7, sA = m >> 3;
        const int o = (co * 8 + ci) * 64 + (t2 >> 2) * 16 + (t2 & 3) * 4 + sA + 2 * sB;
        reg[o + (o >> 6)] = acc[t2][r];
      }
  }
  __syncthreads();
  float* slab = slabs + (size_t)bx * 4096;
  for (int o = tid; o < 4096; o += 256) {
    const int p = o + (o >> 6);
    slab[o] = ((lds[p] + lds[kWgRegion + p]) + lds[2 * kWgRegion + p]) + lds[3 * kWgRegion + p];
  }
#else
#pragma unroll 1
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * (lane >> 4) + r, co = m & 7, sA = m >> 3;
          const int o = (co * 8 + ci) * 64 + (t2 >> 2) * 16 + (t2 & 3) * 4 + sA + 2 * sB;
          lds[o] = wv == 0 ? acc[t2][r] : lds[o] + acc[t2][r];
        }
    }
    __syncthreads();
  }
  float* slab = slabs + (size_t)bx * 4096;
  for (int o = tid; o < 4096; o += 256) slab[o] = lds[o];
#endif
}

template <class C>
__global__ __launch_bounds__(256) void wgrad_k4_mfma(const float* __restrict__ g, const float* __restrict__ x,
                                                     float* __restrict__ slabs, WgDims d) {
  __shared__ float lds[C::LDSF > kWgEpiFloats ? C::LDSF : kWgEpiFloats];
  wgrad_k4_mfma_body<C>(g, x, slabs, d, blockIdx.x, lds);
}

template <class C>
static int launch_wgrad_mfma(const float* g, const float* x, float* dw, float* slabs, WgDims d, int accumulate,
                             hipStream_t s, int* defer_nslab) {
  d.items = d.batch * (C::W / C::TY) * (C::W / C::TZ);
  // one workgroup per CU (it holds 256 VGPRs per lane): each walks its items with the next item's loads in flight
  int nslab = d.items < 256 ? d.items : 256;
  d.items_per_wg = (d.items + nslab - 1) / nslab;
  nslab = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  wgrad_k4_mfma<C><<<nslab, 256, 0, s>>>(g, x, slabs, d);
  if (defer_nslab) *defer_nslab = nslab;
  else wgrad_reduce<<<(4096 + 63) / 64, 1024, 0, s>>>(slabs, dw, nslab, 4096, accumulate);
  return NVF_OK;
}

// ---- the same gradient in the Winograd (y, x) form (wgrad_wino.h): 2.56 x fewer multiplications ----
#include "wgrad_wino.h"
using WgWino2 = WWCfg<32>;
__global__ __launch_bounds__(256) void wgrad_k4_wino(const float* __restrict__ g, const float* __restrict__ x,
                                                     float* __restrict__ slabs, WgDims d) {
  __shared__ __attribute__((aligned(16))) float lds[WgWino2::LDSF > kWgEpiFloats ? WgWino2::LDSF : kWgEpiFloats];
  wgrad_k4_wino_body<WgWino2>(g, x, slabs, d, blockIdx.x, lds, kWgRegion);
}

using WgWino1 = WWCfg<16>;          // conv1 (dY 16^3, X 19^3)
// walkers of the Winograd gradient: (block, group of four tiles, z split); fills d.items / items_per_wg / tiles_z
static int wino_items(WgDims& d, int batch, int zsplit, int cap, int ngrp = WgWino2::NGRP) {
  d.tiles_z = zsplit < 1 ? 1 : zsplit;
  d.items = batch * ngrp * d.tiles_z;
  const int quads = (d.items + 3) / 4;                         // a workgroup's four waves take one walker each per round
  int n = quads < cap ? quads : cap;
  d.items_per_wg = (quads + n - 1) / n * 4;
  d.tiles_y = (d.items + d.items_per_wg - 1) / d.items_per_wg;      // workgroups of the job (XCD-local item ranges)
  return d.tiles_y;
}

// dW [8][8][4][4][4] (and, optional, db [8] = channel sums of dy) of conv2 from dy [batch, 8, 32^3] and x [batch, 8, 35^3]
// in the Winograd form: one launch of partial slabs + the fixed-order reduction.  workspace: 512 slabs of 4096 + 8 floats.
extern "C" int nvf_wgrad_k4_wino(const float* dy, const float* x, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, int batch, int zsplit, void* stream) {
  if (!dy || !x || !dw || !workspace || batch <= 0 || zsplit < 1 || zsplit > 8) return NVF_EINVAL;
  if (workspace_bytes < (size_t)kMaxSlabs * (4096 + 8) * sizeof(float)) return NVF_EWORKSPACE;
  WgDims d{};
  d.batch = batch; d.bc = 8;
  const int nslab = wino_items(d, batch, zsplit, kMaxSlabs);
  float* slabs = (float*)workspace;
  d.bias_slab = db ? slabs + (size_t)kMaxSlabs * 4096 : nullptr;
  hipStream_t s = nvf_stream(stream);
  wgrad_k4_wino<<<nslab, 256, 0, s>>>(dy, x, slabs, d);
  wgrad_reduce<<<(4096 + 63) / 64, 1024, 0, s>>>(slabs, dw, nslab, 4096, 0);
  if (db) wgrad_reduce<<<1, 1024, 0, s>>>(d.bias_slab, db, nslab, 8, 0);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------------------------------
// MFMA weight gradient of the stride-2, 5^3 transposed convolution with 8 -> 8 channels (up2):
//   dW[ci][co][kz][ky][kx] = sum_{n,i} x[ci, i] g[co, 2 i + k]
// K of the MFMA runs along four consecutive input positions ix; rows m = (ci, sA) take x shifted by sA, columns
// n = (co, sB) take g[.., 2 ix + sB (+ 2)]:
//   tile 1: D[(ci,sA)][(co,sB)]   = dW[kx = 2 sA + sB]        -- kx 0..3, every lane useful
//   tile 2: the same A with g two words further: kx = 2 sA + sB + 2 -- only (sA,sB) = (1,0), kx = 4, is kept
// so each (kz, ky) costs two accumulators and 5 of 8 lane-taps are useful.  The 25 (kz, ky) pairs are dealt
// round-robin to the four waves (no cross-wave reduction); a workgroup walks items (n, 2 input planes, 2 input
// rows) with the next item's global loads already in registers, and writes one slab at the end.
// ---------------------------------------------------------------------------------------------------
// W = input extent (16: up2, 8: up1), CIB = row blocks of eight input channels (1: up2, 2: up1's 16 channels -- two A
// fragments and two accumulator sets per step, the same B fragments)
template <int TZ_, int TY_, int W_ = 16, int CIB_ = 1>
struct TWCfg {
  static constexpr int W = W_, WG = 2 * W_ + 3, TZ = TZ_, TY = TY_, CIB = CIB_, CI = 8 * CIB_;
  static constexpr int GZ = 2 * TZ + 3, GY = 2 * TY + 3;
  static constexpr int GRS = WG + 1, XRS = W + 6;           // g row: WG words; x row: u = ix + 1 in 0..W + 4, zero outside 1..W
  static constexpr int mod32(int v, int r) { return v + ((r - v % 32) + 32) % 32; }
  static constexpr int GCS = mod32(GZ * GY * GRS, 8), XCS = mod32(TZ * TY * XRS, 8);
  static constexpr int GOFF = 0, XOFF = 8 * GCS, LDSF = 8 * GCS + CI * XCS;
  static constexpr int NXE = CI * TZ * TY * W;                            // x elements to load per item
  // g tile: for one (channel, plane) the GY rows of WG words are contiguous: NSEG segments of SEG words
  static constexpr int SEG = GY * WG, PPS = (SEG + 255) / 256, NSEG = 8 * GZ;
  static constexpr int UG = NSEG * PPS, UX = (NXE + 255) / 256;
  static constexpr int JT = CI * 8 * 125;                                 // slab length
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

template <class C>
__device__ __forceinline__ void wgrad_s2k5_mfma_body(const float* __restrict__ x, const float* __restrict__ g,
                                                     float* __restrict__ slabs, const WgDims& d, int bx, float* lds) {
  constexpr int W = C::W, WG = C::WG, TZ = C::TZ, TY = C::TY, GZ = C::GZ, GY = C::GY, GRS = C::GRS, XRS = C::XRS,
                GCS = C::GCS, XCS = C::XCS, UG = C::UG, UX = C::UX, CIB = C::CIB, CI = C::CI;
  float* ldsG = lds + C::GOFF;
  float* ldsX = lds + C::XOFF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 4, ch = (lane & 15) >> 1, sh = lane & 1;
  const int laneA = ch * XCS + kk - sh + 1;                  // x[ci = ch, ix0 + kk - sA], u = ix + 1
  // 40 tiles: per kz, ty = 0..4 are the (kz, ky = ty) tiles with kx = 2 sA + sB; ty = 5..7 hold kx = 4 for the
  // rows ky = 2 (ty - 5) + sB (sA = 0 rows only).  Wave w owns tiles j = w + 4 i, i < 10 -- the same instruction
  // stream for every wave, the per-lane LDS word of tile i sits in boff[i].
  constexpr int NTW = 10;
  f32x4 acc[CIB][NTW];
  int boff[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
#pragma unroll
    for (int rb = 0; rb < CIB; ++rb) acc[rb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int j = wave + 4 * i, kz = j >> 3, ty = j & 7;
    boff[i] = ty < 5 ? ch * GCS + 2 * kk + sh + (kz * GY + ty) * GRS
                     : ch * GCS + 2 * kk + 4 + (kz * GY + 2 * (ty - 5) + sh) * GRS;
  }
  const int tiles_y = W / TY, tiles_z = W / TZ, tiles = tiles_y * tiles_z;
  const int first = bx * d.items_per_wg, last = min(first + d.items_per_wg, d.items);
  float gv[UG], xv[UX];
  int glo[C::PPS];                                           // LDS word of this thread's element inside a g segment
#pragma unroll
  for (int part = 0; part < C::PPS; ++part) {
    const int e = tid + part * 256;
    glo[part] = (e / WG) * GRS + e % WG;
  }
#if NVF_WG_BUF
  // g: word e = tid (+ 256 part) of a segment; x: thread t owns element (xx, yy, zz, c0), c = c0 + u * CSTEP
  constexpr int PERX = W * TY * TZ, CSTEP = 256 / PERX;
  static_assert(256 % PERX == 0 && C::NXE % 256 == 0, "a thread's x elements differ by whole channels");
  const int xxx = tid % W, xr = tid / W, xyy = xr % TY, xt2 = xr / TY, xzz = xt2 % TZ, xc0 = xt2 / TZ;
  const int xvoff = (((xc0 * W + xzz) * W + xyy) * W + xxx) * 4;
  int gvoff[C::PPS];
#pragma unroll
  for (int part = 0; part < C::PPS; ++part) gvoff[part] = tid + part * 256 < C::SEG ? (tid + part * 256) * 4 : kWgOob;
  auto load = [&](int item) {
    const int n = item / tiles, t = item % tiles;
    const int y0 = (t % tiles_y) * TY, z0 = (t / tiles_y) * TZ;
    const __amdgpu_buffer_rsrc_t rg = wg_rsrc(g + (size_t)n * 8 * WG * WG * WG, 8 * WG * WG * WG * 4);
    int gs = ((2 * z0 * WG + 2 * y0) * WG) * 4;           // running scalar offset (see wgrad_k4_mfma_body)
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int seg = u / C::PPS, part = u % C::PPS, c = seg / GZ, zz = seg % GZ;
      gv[u] = wg_ld(rg, gvoff[part], gs);
      if (part == C::PPS - 1) {
        const int nseg = seg + 1, nc = nseg / GZ, nz = nseg % GZ;
        gs += ((nc * WG + nz) - (c * WG + zz)) * WG * WG * 4;
        asm volatile("" : "+s"(gs));
      }
    }
    const __amdgpu_buffer_rsrc_t rx = wg_rsrc(x + (size_t)n * CI * W * W * W, CI * W * W * W * 4);
    const int xs = ((z0 * W + y0) * W) * 4;
#pragma unroll
    for (int u = 0; u < UX; ++u) xv[u] = wg_ld(rx, xvoff, xs + u * CSTEP * W * W * W * 4);
  };
#else
  auto load = [&](int item) {
    const int n = item / tiles, t = item % tiles;
    const int y0 = (t % tiles_y) * TY, z0 = (t / tiles_y) * TZ;
    const float* gt = g + (((size_t)n * 8 * WG + 2 * z0) * WG + 2 * y0) * WG;       // wave-uniform
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int seg = u / C::PPS, part = u % C::PPS, c = seg / GZ, zz = seg % GZ;
      const int e = tid + part * 256;
      gv[u] = e < C::SEG ? gt[(c * WG + zz) * WG * WG + e] : 0.f;
    }
    const float* xn = x + (size_t)n * CI * W * W * W;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int e = tid + u * 256;
      const int xx = e % W, r = e / W, yy = r % TY, t2 = r / TY, zz = t2 % TZ, c = t2 / TZ;
      xv[u] = e < C::NXE ? xn[(((size_t)c * W + z0 + zz) * W + y0 + yy) * W + xx] : 0.f;
    }
  };
#endif
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int seg = u / C::PPS, part = u % C::PPS, c = seg / GZ, zz = seg % GZ;
      if (tid + part * 256 < C::SEG) ldsG[c * GCS + zz * GY * GRS + glo[part]] = gv[u];
    }
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int e = tid + u * 256;
      if (e < C::NXE) {
        const int xx = e % W, r = e / W, c = r / (TZ * TY);
        ldsX[c * XCS + (r - c * TZ * TY) * XRS + xx + 1] = xv[u];
      }
    }
  };
  for (int i = tid; i < C::LDSF; i += 256) lds[i] = 0.f;     // padding words stay zero for the whole launch
  if (first < last) load(first);
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    __syncthreads();
    store();
    __syncthreads();
    if (item + 1 < last) load(item + 1);
    {
      // 20 steps (row r, x group xg); the operands of step s + 1 are fetched while the 10 MFMAs of step s issue.
      // The rows shifted by sA = 1 need ix = 15 from a fifth x group.
      constexpr int NXG = W / 4 + 1, NSTEP = TZ * TY * NXG;
      const float* pa = ldsX + laneA;
      auto a_off = [](int st) { return (st / NXG) * XRS + 4 * (st % NXG); };
      auto b_off = [](int st) {
        const int r = st / NXG, zl = r / TY, yl = r % TY;
        return ((2 * zl) * GY + 2 * yl) * GRS + 8 * (st % NXG);
      };
      float a_cur[CIB], b_cur[NTW];
#pragma unroll
      for (int rb = 0; rb < CIB; ++rb) a_cur[rb] = pa[rb * 8 * XCS + a_off(0)];
#pragma unroll
      for (int i = 0; i < NTW; ++i) b_cur[i] = ldsG[boff[i] + b_off(0)];
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        float a_nxt[CIB], b_nxt[NTW];
        if (st + 1 < NSTEP) {
#pragma unroll
          for (int rb = 0; rb < CIB; ++rb) a_nxt[rb] = pa[rb * 8 * XCS + a_off(st + 1)];
#pragma unroll
          for (int i = 0; i < NTW; ++i) b_nxt[i] = ldsG[boff[i] + b_off(st + 1)];
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
          for (int rb = 0; rb < CIB; ++rb)
            acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rb], b_cur[i], acc[rb][i], 0, 0, 0);
        if (st + 1 < NSTEP) {
#pragma unroll
          for (int rb = 0; rb < CIB; ++rb) a_cur[rb] = a_nxt[rb];
#pragma unroll
          for (int i = 0; i < NTW; ++i) b_cur[i] = b_nxt[i];
        }
      }
    }
  }
  // lane holds column n = (co, sB) = lane & 15 and rows m = 4 (lane >> 4) + r = (ci, sA)
  float* slab = slabs + (size_t)bx * C::JT;
  const int co = (lane & 15) >> 1, sB = lane & 1;
#pragma unroll
  for (int rb = 0; rb < CIB; ++rb)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int j = wave + 4 * i, kz = j >> 3, ty = j & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * (lane >> 4) + r, ci = 8 * rb + (m >> 1), sA = m & 1;
        float* o = slab + (ci * 8 + co) * 125 + kz * 25;
        if (ty < 5) o[ty * 5 + 2 * sA + sB] = acc[rb][i][r];
        else if (sA == 0 && 2 * (ty - 5) + sB < 5) o[(2 * (ty - 5) + sB) * 5 + 4] = acc[rb][i][r];
      }
    }
}

template <class C>
__global__ __launch_bounds__(256) void wgrad_s2k5_mfma(const float* __restrict__ x, const float* __restrict__ g,
                                                       float* __restrict__ slabs, WgDims d) {
  __shared__ float lds[C::LDSF];
  wgrad_s2k5_mfma_body<C>(x, g, slabs, d, blockIdx.x, lds);
}

// The three matrix-core weight gradients of the narrow trunk (conv2, up2, conv1) in ONE launch.  Each of them keeps
// its MFMA pipes busy about half of the time (tile staging, LDS waits); 256 VGPRs and <= 80 KB of LDS per workgroup
// let two workgroups share a CU, so the second kernel's workgroups run in the first one's bubbles.  Same bodies, same
// slabs, same results as three launches.
// the latent tail queued in the caller's NvfStepCtx by nvf_latent_tail_queue: consumed by the next
// nvf_wgrad_mfma3_partial / nvf_wgrad_trunk5_partial or nvf_wgrad_reduce_multi_and_sums call with that context

#ifndef NVF_SUM_T
#define NVF_SUM_T 512      // threads that load a row (256: 17.6 us for the step's reduction launch, 512 / 1024: 15.9)
#endif
// `T` threads do the work (the arithmetic does not depend on the launch's workgroup size: in the one-launch tail the
// workgroups have 1024 threads, the extra ones only take part in the block sum with zeros)
template <int T>
__device__ __forceinline__ void multi_channel_sum_partial_body(const MultiSumDesc& d, float* __restrict__ part, int gch,
                                                               int g, float* red) {
  int t = 0;
  while (t + 1 < d.ntensors && gch >= d.chan_base[t + 1]) ++t;
  const int ch = gch - d.chan_base[t], c = d.c[t], spatial = d.spatial[t];
  const int per = (d.batch + d.nchunk - 1) / d.nchunk;
  const int n_lo = g * per, n_hi = min(n_lo + per, d.batch);
  const float* x = d.x[t];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int n = n_lo; n < n_hi && (int)threadIdx.x < T; ++n) {
    const float* row = x + ((size_t)n * c + ch) * spatial;     // one contiguous row per (n, channel)
    // rows of 35^3 or 19^3 floats start at any 4-byte phase: a scalar head up to the next 16-byte boundary, then
    // aligned float4s (four per thread in flight), then a scalar tail
    const int head = (int)((4 - (((uintptr_t)row >> 2) & 3)) & 3);
    const int body = (spatial - head) & ~3;
    if ((int)threadIdx.x < head) s0 += row[threadIdx.x];
    const float4* r4 = (const float4*)(row + head);
    const int n4 = body >> 2;
    int i = threadIdx.x;
    for (; i + 3 * (int)T < n4; i += 4 * T) {
      const float4 a = r4[i], b4 = r4[i + T], c4 = r4[i + 2 * T], d4 = r4[i + 3 * T];
      s0 += (a.x + b4.x) + (c4.x + d4.x); s1 += (a.y + b4.y) + (c4.y + d4.y);
      s2 += (a.z + b4.z) + (c4.z + d4.z); s3 += (a.w + b4.w) + (c4.w + d4.w);
    }
    for (; i < n4; i += T) {
      const float4 v = r4[i];
      s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    const int tail = head + body + threadIdx.x;
    if (tail < spatial) s1 += row[tail];
  }
  const float s = nvf_block_sum((s0 + s1) + (s2 + s3), red);
  if (threadIdx.x == 0) part[(size_t)g * d.total_channels + gch] = s;
}

struct WgMfma3 {                 // jobs 0-2: conv2, up2, conv1; job 3 (n[3] may be 0): up1 on the matrix cores
  const float* p[4];
  const float* q[4];
  float* slabs[4];
  WgDims d[4];
  int32_t n[4];
};
#ifndef NVF_UP1_MFMA
#define NVF_UP1_MFMA 1           // 0: up1's gradient as a VALU tile job (wgrad_tiled), as before round 3
#endif
using WgUp1 = TWCfg<2, 2, 8, 2>;
constexpr int wg_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

#ifdef NVF_WG_STAMP
// tuning builds (tools/wg_timeline.py): start / end of every workgroup of the five-gradient launch on the 100 MHz clock
__device__ unsigned long long g_wg_stamps[2 * 4096];
struct WgStamp {
  int b;
  __device__ explicit WgStamp(int b_) : b(b_) { if (threadIdx.x == 0 && b < 4096) g_wg_stamps[2 * b] = __builtin_amdgcn_s_memrealtime(); }
  __device__ ~WgStamp() { if (threadIdx.x == 0 && b < 4096) g_wg_stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime(); }
};
extern "C" int nvf_debug_wg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * (size_t)(n < 8192 ? n : 8192));
}
#endif

// ---- conv0's weight gradient (8 -> 16 channels, 4^3 -> 8^3, k 5 s 2 padding 2) on the matrix cores -------------------------
//   dw[ci][co][k] = sum_{b, i} h0[b, ci, i] g1[b, co, 2 i - 2 + k]
// 16.4 M multiply-adds, but as the VALU tile job of this launch it held 128 slots for 16 us each (6 % of the launch's slot
// time).  Five workgroups (256 threads; one per ky) and one slab per block; v_mfma_f32_16x16x4_f32 with rows = the 16 output channels,
// K = four positions i, columns = (ci, s): column s = 1 reads h0 one z plane lower, which makes its product the tap kz + 2
// of the same (ky, kx) -- the pair trick of the other layers, 75 products for the 125 taps.  A wave owns 3-4 of them for
// all 20 K steps (accumulators in registers); an A operand is g1 through a bounds check (no padded copy: 33 KB of LDS, not
// 110).  Per output the sum runs over the block's positions in ascending order; the slab reduction adds the blocks.
// Another summation order than the tile job's: the caller's context keeps the tile job for the direct forms.
struct Conv0WLds { static constexpr int GCS = 513, HCS = 65, FLOATS = 16 * GCS + 8 * HCS; };   // odd strides: no bank conflicts

__device__ __forceinline__ void conv0_wgrad_mfma_body(const float* __restrict__ h0, const float* __restrict__ g1,
                                                      float* __restrict__ slab, int b, int ky, float* lds) {
  typedef float c0_f32x4 __attribute__((ext_vector_type(4)));
  constexpr int GCS = Conv0WLds::GCS, HCS = Conv0WLds::HCS;
  float* s_g = lds;
  float* s_h = lds + 16 * GCS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < 16 * 512; e += 256) s_g[(e >> 9) * GCS + (e & 511)] = g1[(size_t)b * 16 * 512 + e];
  for (int e = tid; e < 8 * 64; e += 256) s_h[(e >> 6) * HCS + (e & 63)] = h0[(size_t)b * 8 * 64 + e];
  __syncthreads();
  const int j = lane & 15, kq = lane >> 4;
  const int ci = j >> 1, sft = j & 1;
  constexpr int NPROD = 15, PER = (NPROD + 3) / 4;              // this workgroup's (kx, z pair) products; per wave
  const int t0 = wave * PER, t1 = min(t0 + PER, NPROD);
  c0_f32x4 acc[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) acc[u] = c0_f32x4{0.f, 0.f, 0.f, 0.f};
  // (five z planes of positions: the shifted column s = 1 meets h0's last plane at iz = 4, where column s = 0 has none)
#pragma unroll 1
  for (int ks = 0; ks < 20; ++ks) {
    const int i = 4 * ks + kq, iz = i >> 4, iy = (i >> 2) & 3, ix = i & 3;
    const float bv = (unsigned)(iz - sft) < 4u ? s_h[ci * HCS + i - 16 * sft] : 0.f;      // B[k = kq][col = (ci, s)]
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = t0 + u;                                     // wave-uniform
      if (t < t1) {
        const int zp = t % 3, kx = t / 3;
        const int kz = zp == 2 ? 4 : zp;                        // pairs (0, 2) (1, 3) (4, -)
        const int qz = 2 * iz - 2 + kz, qy = 2 * iy - 2 + ky, qx = 2 * ix - 2 + kx;
        const bool ok = (unsigned)qz < 8u && (unsigned)qy < 8u && (unsigned)qx < 8u;
        const float av = ok ? s_g[j * GCS + (qz * 8 + qy) * 8 + qx] : 0.f;     // A[row = co = j][k = kq]
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u], 0, 0, 0);
      }
    }
  }
  float* out = slab + (size_t)b * (8 * 16 * 125);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int t = t0 + u;
    if (t < t1) {
      const int zp = t % 3, kx = t / 3;
      const int kz = (zp == 2 ? 4 : zp) + 2 * sft;
      if (kz < 5) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(ci * 16 + 4 * kq + r) * 125 + kz * 25 + ky * 5 + kx] = acc[u][r];
      }
    }
  }
}

// floats of LDS the latent tail needs when the stem's backward rides in the same launch: + its copy of dx0
constexpr int kTailStemLds = kTailLds + kStemCoopMaxBatch * kStemMaxCh * 8;

template <class C0, class T1, class C2, class U0, class U1, bool TAIL, bool STEM>
__global__ __launch_bounds__(256) void wgrad_mfma3_kernel(WgMfma3 m, WgTiled2 u, LatentTail tail, HeadsW3 hw,
                                                          MultiSumDesc sums, float* __restrict__ sum_part,
                                                          const float* __restrict__ coef_src, float* coef_live,
                                                          StemBwdJob sj) {
  static_assert(U0::NT == 256 && U1::NT == 256, "one workgroup size");
  static_assert(TAIL || !STEM, "the stem's backward feeds the latent tail of the same launch");
  __shared__ __attribute__((aligned(16))) float lds[wg_max3(wg_max3(C0::LDSF, T1::LDSF, C2::LDSF), kWgEpiFloats,
                                                            wg_max3(wg_max3(kTailStemLds, U0::LDSF, U1::LDSF),
                                                                    wg_max3(HeadW0::SMEM, HeadW1::SMEM, HeadW2::SMEM),
                                                                    wg_max3(WgUp1::LDSF, StemBwdLds<8>::FLOATS,
                                                                            wg_max3(StemDhLds<8, 16>::FLOATS,
                                                                                    Conv0WLds::FLOATS, 0))))];
#ifndef NVF_WG_SKIP
#define NVF_WG_SKIP 0                  // tuning builds: bit j set = job j does nothing (results are then meaningless)
#endif
  int bid = blockIdx.x;
#ifdef NVF_WG_STAMP
  WgStamp stamp_(blockIdx.x);
#endif
  if (STEM) {
    // the stem's backward (stem_bwd.h): conv0's backward-data partials over (block, channel pair) workgroups, then one
    // workgroup per block for IGDN / up0 -- producers first (lowest ids), each stage signals the next through arrival
    // counters; they need only g1 and feed only the tail below, and are hidden behind the gradients like it
    const int ndh = sj.batch * 8;
    if (bid < ndh) {
      stem_bwd_dh_body<8, 16, false, true>(sj.g1, sj.w1b, sj.part, nullptr, nullptr, bid >> 3, bid & 7, lds, sj.coop);
      return;
    }
    bid -= ndh;
    if (bid < sj.nwg) {
      stem_bwd_body<8, 16, 256, true>(sj.part, sj.x0, sj.a0, sj.w0b, sj.beta_hat, sj.gamma_hat, sj.da0, sj.dx0,
                                      sj.slab_gdn, sj.slab_w, sj.batch, sj.ch, 1, bid, sj.nwg, lds, sj.coop);
      return;
    }
    bid -= sj.nwg;
  }
  if (TAIL) {                          // the latent tail: one workgroup, dispatched first, hidden behind the gradients
    if (bid == 0) {
      if (NVF_WG_SKIP & 32) return;
      if (STEM) {                      // dx0 comes out of this launch: wait for the per-block workgroups, fetch it with
        nvf_coop_wait(sj.coop.stem_done, (unsigned)sj.nwg);      // device-scope loads into LDS and run the tail on that copy
        float* s_dx = lds + kTailLds;
        for (int e = threadIdx.x; e < sj.batch * sj.ch * 8; e += 256) s_dx[e] = nvf_load_dev(sj.dx0 + e);
        __syncthreads();
        latent_tail_body(tail, lds, s_dx);
      } else {
        latent_tail_body(tail, lds, tail.dx_addend);
      }
      return;
    }
    --bid;
  }
  // conv0 on the matrix cores (five short workgroups per block: in the first dispatch round, not behind the long ones)
  if (bid < u.conv0_mfma) { conv0_wgrad_mfma_body(u.p[1], u.q[1], u.slabs[1], bid / 5, bid % 5, lds); return; }
  bid -= u.conv0_mfma;
  if (bid < m.n[0]) {                  // conv2: the Winograd (y, x) form when the job says so (tiles_z = its z split)
    if (NVF_WG_SKIP & 1) return;
    if (m.d[0].tiles_z > 0) wgrad_k4_wino_body<WgWino2>(m.p[0], m.q[0], m.slabs[0], m.d[0], bid, lds, kWgRegion);
    else wgrad_k4_mfma_body<C0>(m.p[0], m.q[0], m.slabs[0], m.d[0], bid, lds);
    return;
  }
  bid -= m.n[0];
  if (bid < m.n[1]) { if (!(NVF_WG_SKIP & 2)) wgrad_s2k5_mfma_body<T1>(m.p[1], m.q[1], m.slabs[1], m.d[1], bid, lds); return; }
  bid -= m.n[1];
  if (bid < m.n[2]) {                  // conv1: likewise
    if (NVF_WG_SKIP & 4) return;
    if (m.d[2].tiles_z > 0) wgrad_k4_wino_body<WgWino1>(m.p[2], m.q[2], m.slabs[2], m.d[2], bid, lds, kWgRegion);
    else wgrad_k4_mfma_body<C2>(m.p[2], m.q[2], m.slabs[2], m.d[2], bid, lds);
    return;
  }
  bid -= m.n[2];
  // up1 (16 -> 8 channels, 8^3 -> 19^3) on the matrix cores: as a VALU tile job it cost 13 us of this launch for 4 % of
  // its multiply-adds (measured by skipping it); the stride-2 body with two row blocks of eight input channels
  if (bid < m.n[3]) { if (!(NVF_WG_SKIP & 8)) wgrad_s2k5_mfma_body<WgUp1>(m.p[3], m.q[3], m.slabs[3], m.d[3], bid, lds); return; }
  bid -= m.n[3];

  if (NVF_WG_SKIP & (8 | 64)) { if (bid < u.nx[0] * u.ny[0] + u.nx[1] * u.ny[1]) return; }     // 64: the tile jobs only
  if ((NVF_WG_SKIP & 16) && bid >= u.nx[0] * u.ny[0] + u.nx[1] * u.ny[1]) return;
  // the two small transposed convolutions' gradients (VALU kernels, latency-bound on their own) fill the slots the
  // short matrix-core workgroups leave while conv2's are still running
  if (bid < u.nx[0] * u.ny[0]) {
    wgrad_tiled_body<U0>(u.p[0], u.q[0], u.slabs[0], u.d[0], bid % u.nx[0], bid / u.nx[0], lds);
    return;
  }
  bid -= u.nx[0] * u.ny[0];
  if (bid < u.nx[1] * u.ny[1]) {
    wgrad_tiled_body<U1>(u.p[1], u.q[1], u.slabs[1], u.d[1], bid % u.nx[1], bid / u.nx[1], lds);
    return;
  }
  bid -= u.nx[1] * u.ny[1];
  // the three classifier heads' weight gradients (matrix cores, 22 KB of LDS, short): they depend on nothing this
  // launch produces and run in the slots the other jobs have left by then
  const int nheads = hw.n[0] + hw.n[1] + hw.n[2];
  if (bid < nheads) { heads3_wgrad_mfma_dispatch<HeadW0, HeadW1, HeadW2>(hw, bid, lds); return; }
  bid -= nheads;
  // partial channel sums of the (small) tensors whose bias gradients no other kernel leaves behind: with these here, the
  // final passes of the step depend on nothing the slab reduction produces and share its launch
  // (and the optimiser's two step coefficients leave the step buffer for a place the schedule hand-over does not touch:
  // the slab reduction that shares the hand-over's launch reads them there -- nvf_wgrad_reduce_finals_tail)
  if (bid == 0 && threadIdx.x < 2 && coef_live) coef_live[threadIdx.x] = coef_src[threadIdx.x];
  multi_channel_sum_partial_body<256>(sums, sum_part, bid % sums.total_channels, bid / sums.total_channels, lds);
}

// geometry of the up1 / conv0 jobs (nvf_wgrad_up1_conv0_partial, nvf_wgrad_trunk5_partial)
template <class U0, class U1>
static void fill_up1_conv0(WgTiled2& m, const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                           int* nslabs) {
  const int geo[2][6] = {{8, 8, 19, 0, 16, 8}, {4, 16, 8, 2, 8, 16}};   // wp, bc, wq, pad, a, b
  for (int j = 0; j < 2; ++j) {
    WgDims d{};
    d.batch = batch; d.bc = geo[j][1]; d.dp = d.hp = d.wp = geo[j][0]; d.dq = d.hq = d.wq = geo[j][2]; d.pad = geo[j][3];
    d.out_mode = 0; d.jtotal = geo[j][4] * geo[j][5] * 125;
    const int tx = j == 0 ? U0::TX : U1::TX, ty = j == 0 ? U0::TY : U1::TY, tz = j == 0 ? U0::TZ : U1::TZ;
    const int nb = j == 0 ? U0::NB : U1::NB;
    d.tiles_x = d.wp / tx; d.tiles_y = d.hp / ty; d.tiles_z = d.dp / tz;
    d.items = batch * d.tiles_x * d.tiles_y * d.tiles_z;
    int n = d.items < kMaxSlabs ? d.items : kMaxSlabs;
    d.items_per_wg = (d.items + n - 1) / n;
    n = (d.items + d.items_per_wg - 1) / d.items_per_wg;
    m.p[j] = ps[j]; m.q[j] = qs[j]; m.slabs[j] = slabs[j]; m.d[j] = d; m.nx[j] = n; m.ny[j] = (d.bc + nb - 1) / nb;
    nslabs[j] = n;
  }
}

// njobs = 3: conv2, up2, conv1 (matrix cores); njobs = 5: + up1, conv0 (the VALU tile kernel with 256-thread workgroups)
struct HeadsJob {                       // optional: the heads' gradients in the same launch (njobs = 5 only)
  const float* const* dls;
  const float* const* xs;
  float* const* slabs;
  int max_slabs;
  int* nslabs;
};
struct SumsJob {                        // optional: partial channel sums (nvf_multi_channel_sum's first pass) in the launch
  MultiSumDesc d;
  float* part;
  const float* coef_src;                // optional: two floats copied to coef_live by the launch
  float* coef_live;
};

static int launch_trunk_wgrads(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                               int* nslabs, int njobs, NvfStepCtx* ctx, void* stream,
                               float* const* bias_slabs = nullptr, const HeadsJob* heads = nullptr,
                               const SumsJob* sums = nullptr) {
  if (!ps || !qs || !slabs || !nslabs || batch <= 0) return NVF_EINVAL;
  using C0 = MCfg<32, 4, 4>; using T1 = TWCfg<2, 2>; using C2 = MCfg<16, 2, 8>;
  using U0 = WCfg<16, 5, 2, 2, 8, 4, 2, 0>; using U1 = WCfg<8, 5, 2, 2, 4, 4, 4, 0>;
  for (int j = 0; j < njobs; ++j)
    if (!ps[j] || !qs[j] || !slabs[j]) return NVF_EINVAL;
  WgMfma3 m{};
  const int items[3] = {batch * (32 / 4) * (32 / 4), batch * (16 / 2) * (16 / 2), batch * (16 / 8) * (16 / 2)};
  for (int j = 0; j < 3; ++j) {
    m.p[j] = ps[j]; m.q[j] = qs[j]; m.slabs[j] = slabs[j];
    WgDims d{};
    d.batch = batch; d.bc = 8; d.items = items[j];
    // slabs (= workgroups) per job: two workgroups share a CU, so conv2 (the longest) gets one per CU and the other
    // two half of that -- fewer slabs to add up afterwards, same launch time
    // conv1: 64 (with up1 at 64: 86.9 us for the launch against 92-93 with 128 / 128); tuning builds: NVF_WG_CAP0..2
    const int caps[3] = {nvf_tune_int("NVF_WG_CAP0", 256), nvf_tune_int("NVF_WG_CAP1", 128), nvf_tune_int("NVF_WG_CAP2", 64)};
    const int cap = caps[j] < 1 ? 1 : (caps[j] > 512 ? 512 : caps[j]);
    int n = items[j] < cap ? items[j] : cap;
    d.items_per_wg = (items[j] + n - 1) / n;
    n = (items[j] + d.items_per_wg - 1) / d.items_per_wg;
    // conv2 in the Winograd (y, x) form (wgrad_wino.h) unless the caller's context asks for the direct forms
    // (nvf_step_ctx_set_direct); nvf_step_ctx_set_wgrad_forms: z steps split over `conv2_zsplit` work items (default 1 =
    // 256 workgroups at batch 16: step 0.375 ms against 0.3825 with 2 and 0.3886 for the direct form).  The choice of
    // arithmetic is the CALLER's, carried by its context -- no environment read, no process-wide state.
    const bool direct = nvf_ctx_ok(ctx) && ctx->direct_forms;
    int wino = nvf_ctx_ok(ctx) && ctx->wg_conv2_zsplit > 0 ? ctx->wg_conv2_zsplit : 1;
    if (wino > 8) wino = 8;
    if (j == 0 && !direct) {
      int wcap = nvf_tune_int("NVF_WGRAD_WINO_CAP", 512);
      if (wcap < 1 || wcap > 512) wcap = 512;
      n = wino_items(d, batch, wino, wcap);
    }
    // conv1's gradient in the same form is opt-in (nvf_step_ctx_set_wgrad_forms(ctx, z, 1)): 0.3485 against 0.3517 ms per
    // step, but with it the three-epoch trajectory golden holds 98.9 % instead of >= 99 % of the sampled parameters within
    // 2e-5 of the reference's (entries whose gradient is rounding noise take Adam's first steps with the other sign)
    const bool wino1 = nvf_ctx_ok(ctx) && ctx->wg_conv1_wino;
    if (j == 2 && wino1 && !direct) n = wino_items(d, batch, 1, cap, WgWino1::NGRP);
    if (bias_slabs && j != 1) d.bias_slab = bias_slabs[j];     // conv2 (job 0) and conv1 (job 2): p = dY
    m.d[j] = d; m.n[j] = n; nslabs[j] = n;
  }
  WgTiled2 u{};
  int grid = m.n[0] + m.n[1] + m.n[2];
  if (njobs == 5) {
    fill_up1_conv0<U0, U1>(u, ps + 3, qs + 3, slabs + 3, batch, nslabs + 3);
#if NVF_UP1_MFMA
    {
      WgDims d{};
      d.batch = batch; d.bc = 8; d.items = batch * (WgUp1::W / WgUp1::TZ) * (WgUp1::W / WgUp1::TY);
      int up1_cap = nvf_tune_int("NVF_UP1_CAP", 64);
      if (up1_cap < 1) up1_cap = 64;
      int n = d.items < up1_cap ? d.items : up1_cap;
      d.items_per_wg = (d.items + n - 1) / n;
      n = (d.items + d.items_per_wg - 1) / d.items_per_wg;
      m.p[3] = ps[3]; m.q[3] = qs[3]; m.slabs[3] = slabs[3]; m.d[3] = d; m.n[3] = n;
      nslabs[3] = n;
      u.nx[0] = 0;                       // the tile job of up1 is not dispatched
      grid += n;
    }
#endif
    // conv0's gradient on the matrix cores too (conv0_wgrad_mfma_body: one workgroup and one slab per block) unless the
    // caller's context asks for the direct forms (another summation order)
    if (!(nvf_ctx_ok(ctx) && ctx->direct_forms) && nvf_tune_int("NVF_CONV0_WG_MFMA", 1)) {
      u.conv0_mfma = 5 * batch;
      u.nx[1] = 0;
      nslabs[4] = batch;
      grid += 5 * batch;
    }
    grid += u.nx[0] * u.ny[0] + u.nx[1] * u.ny[1];
  }
  HeadsW3 hw{};
  if (heads) {
    if (njobs != 5) return NVF_EINVAL;
    const int rc = heads3_wgrad_mfma_fill<HeadW0, HeadW1, HeadW2>(hw, heads->dls, heads->xs, heads->slabs, batch,
                                                                  heads->max_slabs, heads->nslabs);
    if (rc != NVF_OK) return rc;
    grid += hw.n[0] + hw.n[1] + hw.n[2];
  }
  MultiSumDesc sd{};
  float* spart = nullptr;
  const float* csrc = nullptr;
  float* clive = nullptr;
  if (sums) {
    if (!heads) return NVF_EINVAL;        // the block ranges assume the heads' job in front of the sums
    sd = sums->d; spart = sums->part;
    grid += sd.total_channels * sd.nchunk;
    csrc = sums->coef_src; clive = sums->coef_live;
  }
  if (nvf_ctx_ok(ctx) && ctx->stem_pending) {
    // a queued stem backward (nvf_stem_bwd_queue) rides in front of the tail it feeds: both or neither
    if (!ctx->tail_pending || njobs != 5 || ctx->stem.batch != batch ||
        ctx->tail.batch * ctx->tail.c * ctx->tail.spatial != batch * ctx->stem.ch * 8)
      return NVF_EINVAL;
    ctx->stem_pending = ctx->tail_pending = 0;
    const int nstem = ctx->stem.batch * 8 + ctx->stem.nwg;
#ifdef NVF_WG_STAMP
    fprintf(stderr, "[wg_stamp] grid %d: stem_dh %d stem %d tail 1 conv2 %d up2 %d conv1 %d up1 %d tiles %d %d heads %d sums %d\n",
            nstem + 1 + grid, ctx->stem.batch * 8, ctx->stem.nwg, m.n[0], m.n[1], m.n[2], m.n[3], u.nx[0] * u.ny[0],
            u.nx[1] * u.ny[1], hw.n[0] + hw.n[1] + hw.n[2], sd.total_channels * sd.nchunk);
#endif
    wgrad_mfma3_kernel<C0, T1, C2, U0, U1, true, true><<<nstem + 1 + grid, 256, 0, nvf_stream(stream)>>>(
        m, u, ctx->tail, hw, sd, spart, csrc, clive, ctx->stem);
  } else if (nvf_ctx_ok(ctx) && ctx->tail_pending) {
    ctx->tail_pending = 0;
    wgrad_mfma3_kernel<C0, T1, C2, U0, U1, true, false><<<1 + grid, 256, 0, nvf_stream(stream)>>>(m, u, ctx->tail, hw, sd, spart, csrc, clive, StemBwdJob{});
  } else {
    wgrad_mfma3_kernel<C0, T1, C2, U0, U1, false, false><<<grid, 256, 0, nvf_stream(stream)>>>(m, u, LatentTail{}, hw, sd, spart, csrc, clive, StemBwdJob{});
  }
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// job 0: conv2 (p = dY [B,8,32^3], q = X [B,8,35^3]); job 1: up2 (p = X [B,8,16^3], q = dY [B,8,35^3]);
// job 2: conv1 (p = dY [B,8,16^3], q = X [B,8,19^3]).  slabs[j] must hold 512 slabs of 4096 / 8000 / 4096 floats;
// nslabs[j] receives the number written (to be added by nvf_wgrad_reduce_multi).
extern "C" int nvf_wgrad_mfma3_partial(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                                       int* nslabs, NvfStepCtx* ctx, void* stream) {
  return launch_trunk_wgrads(ps, qs, slabs, batch, nslabs, 3, ctx, stream);
}

// ... and jobs 3, 4 = up1 (p = X [B,16,8^3], q = dY [B,8,19^3]), conv0 (p = X [B,8,4^3], q = dY [B,16,8^3]) of
// nvf_wgrad_up1_conv0_partial in the same launch: all five weight gradients of the narrow trunk above the stem.
// slabs[3], slabs[4]: up to 512 slabs of 16000 floats.  Same results as the two separate launches.
extern "C" int nvf_wgrad_trunk5_partial(const float* const* ps, const float* const* qs, float* const* slabs, int batch,
                                        int* nslabs, NvfStepCtx* ctx, void* stream) {
  return launch_trunk_wgrads(ps, qs, slabs, batch, nslabs, 5, ctx, stream);
}

// ... and, with bias_slabs[0] / bias_slabs[2] (either may be NULL; entry 1 is ignored), the channel sums of conv2's and
// conv1's dY per workgroup: nslabs[j] slabs of 8 floats whose sum (nvf_wgrad_reduce_multi, jtotal 8) is the bias
// gradient of that layer -- the kernel holds every dY tile in registers anyway, and the tiles partition dY.
extern "C" int nvf_wgrad_trunk5_partial_bias(const float* const* ps, const float* const* qs, float* const* slabs,
                                             float* const* bias_slabs, int batch, int* nslabs, NvfStepCtx* ctx,
                                             void* stream) {
  return launch_trunk_wgrads(ps, qs, slabs, batch, nslabs, 5, ctx, stream, bias_slabs);
}

// ... and the weight gradients of the narrow decoder's three classifier heads (nvf_heads3_wgrad_partial's contract:
// head_dls / head_xs / head_slabs / head_nslabs have three entries, at most head_max_slabs slabs each) as further
// workgroups of the same launch.
extern "C" int nvf_wgrad_trunk5_heads_partial(const float* const* ps, const float* const* qs, float* const* slabs,
                                              float* const* bias_slabs, const float* const* head_dls,
                                              const float* const* head_xs, float* const* head_slabs,
                                              int head_max_slabs, int batch, int* nslabs, int* head_nslabs,
                                              NvfStepCtx* ctx, void* stream) {
  if (!head_dls || !head_xs || !head_slabs || !head_nslabs || head_max_slabs <= 0) return NVF_EINVAL;
  const HeadsJob h{head_dls, head_xs, head_slabs, head_max_slabs, head_nslabs};
  return launch_trunk_wgrads(ps, qs, slabs, batch, nslabs, 5, ctx, stream, bias_slabs, &h);
}

static int fill_sum_desc(const float* const* xs, float* const* outs, const int* channels, const int* spatials,
                         int ntensors, int batch, MultiSumDesc& d) {
  if (!xs || !outs || !channels || !spatials || ntensors <= 0 || ntensors > 12 || batch <= 0) return NVF_EINVAL;
  int cb = 0;
  for (int i = 0; i < ntensors; ++i) {
    if (!xs[i] || !outs[i] || channels[i] <= 0 || spatials[i] <= 0) return NVF_EINVAL;
    d.x[i] = xs[i]; d.out[i] = outs[i]; d.c[i] = channels[i]; d.spatial[i] = spatials[i]; d.chan_base[i] = cb;
    cb += channels[i];
  }
  d.ntensors = ntensors; d.batch = batch; d.total_channels = cb;
  d.nchunk = batch < kSumChunks ? batch : kSumChunks;
  return NVF_OK;
}

// ... and the first pass of nvf_multi_channel_sum over sum_xs (the bias gradients sum_outs no other kernel leaves
// behind) as further workgroups of the launch; its final pass is queued in ctx (nvf_finals_begin must be open) or
// launched here.  sum_workspace: nvf_multi_channel_sum_workspace(total channels) bytes, untouched until the flush.
// coef_src / coef_live (both or neither): two floats copied by the launch (see nvf_wgrad_reduce_finals_tail).
extern "C" int nvf_wgrad_trunk5_heads_sums_partial(const float* const* ps, const float* const* qs, float* const* slabs,
                                                   float* const* bias_slabs, const float* const* head_dls,
                                                   const float* const* head_xs, float* const* head_slabs,
                                                   int head_max_slabs, const float* const* sum_xs,
                                                   float* const* sum_outs, const int* sum_channels,
                                                   const int* sum_spatials, int sum_n, void* sum_workspace,
                                                   size_t sum_workspace_bytes, const float* coef_src,
                                                   float* coef_live, int batch, int* nslabs, int* head_nslabs,
                                                   NvfStepCtx* ctx, void* stream) {
  if (!head_dls || !head_xs || !head_slabs || !head_nslabs || head_max_slabs <= 0 || !sum_workspace) return NVF_EINVAL;
  if ((coef_src == nullptr) != (coef_live == nullptr)) return NVF_EINVAL;
  const HeadsJob h{head_dls, head_xs, head_slabs, head_max_slabs, head_nslabs};
  SumsJob sj{};
  const int rc = fill_sum_desc(sum_xs, sum_outs, sum_channels, sum_spatials, sum_n, batch, sj.d);
  if (rc != NVF_OK) return rc;
  if (sum_workspace_bytes < nvf_multi_channel_sum_workspace(sj.d.total_channels)) return NVF_EWORKSPACE;
  sj.part = (float*)sum_workspace;
  sj.coef_src = coef_src; sj.coef_live = coef_live;
  const int rc2 = launch_trunk_wgrads(ps, qs, slabs, batch, nslabs, 5, ctx, stream, bias_slabs, &h, &sj);
  if (rc2 != NVF_OK) return rc2;
  if (!nvf_finals_push_sums(ctx, sj.d, sj.part))
    multi_channel_sum_final<<<(sj.d.total_channels + 63) / 64, 64, 0, nvf_stream(stream)>>>(sj.d, sj.part);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

template <class C>
static int launch_wgrad_s2k5(const float* x, const float* g, float* dw, float* slabs, WgDims d, int accumulate,
                             hipStream_t s, int* defer_nslab) {
  d.items = d.batch * (C::W / C::TY) * (C::W / C::TZ);
  int nslab = d.items < 256 ? d.items : 256;                 // one workgroup per CU
  d.items_per_wg = (d.items + nslab - 1) / nslab;
  nslab = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  wgrad_s2k5_mfma<C><<<nslab, 256, 0, s>>>(x, g, slabs, d);
  if (defer_nslab) *defer_nslab = nslab;
  else wgrad_reduce<<<(8000 + 63) / 64, 1024, 0, s>>>(slabs, dw, nslab, 8000, accumulate);
  return NVF_OK;
}

// dw[j] (+)= sum_g slabs[g][j]: 16 interleaved slices of g per output (each summed in ascending g), then the
// slices added in order 0..15 -- a fixed order, so the result is reproducible
__global__ __launch_bounds__(1024) void wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw,
                                                     int nslab, int jtotal, int accumulate) {
  __shared__ float part[16][64];
  const int jl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  // slice sl of 16 takes slabs sl, sl + 16, ...: up to 16 of them are fetched before the first add (one round trip
  // for <= 256 slabs), then added in ascending order in four interleaved chains and a fixed tree
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < jtotal) {
    for (int g0 = sl; g0 < nslab; g0 += 256) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = g0 + 16 * i < nslab ? slabs[(size_t)(g0 + 16 * i) * jtotal + j] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
  }
  const float s = (s0 + s1) + (s2 + s3);
  part[sl][jl] = s;
  __syncthreads();
  if (sl == 0 && j < jtotal) {
    float t = part[0][jl];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][jl];
    dw[j] = accumulate ? dw[j] + t : t;
  }
}

extern "C" size_t nvf_wgrad_workspace(int batch, int a, int b, int k, int dp, int hp, int wp) {
  (void)batch; (void)dp; (void)hp; (void)wp;
  return (size_t)kMaxSlabs * a * b * k * k * k * sizeof(float);
}

template <class C>
static int launch_wgrad(const float* p, const float* q, float* dw, float* slabs, WgDims d, int accumulate,
                        hipStream_t s, int* defer_nslab) {
  d.tiles_x = d.wp / C::TX;
  d.tiles_y = d.hp / C::TY;
  d.tiles_z = d.dp / C::TZ;
  d.items = d.batch * d.tiles_x * d.tiles_y * d.tiles_z;
  const int ygroups = (d.bc + C::NB - 1) / C::NB;
  int nslab = d.items < kMaxSlabs ? d.items : kMaxSlabs;
  d.items_per_wg = (d.items + nslab - 1) / nslab;
  nslab = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  wgrad_tiled<C><<<dim3(nslab, ygroups), C::NT, 0, s>>>(p, q, slabs, d);
  if (defer_nslab) *defer_nslab = nslab;
  else wgrad_reduce<<<(d.jtotal + 63) / 64, 1024, 0, s>>>(slabs, dw, nslab, d.jtotal, accumulate);
  return NVF_OK;
}

static int wgrad_dispatch(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes, int batch,
                          int a, int b, int k, int stride, int pad, int dp, int hp, int wp, int dq, int hq, int wq,
                          int out_mode, int accumulate, int variant, void* stream, int* defer_nslab) {
  if (!p || !q || !dw || batch <= 0 || a <= 0 || b <= 0 || k <= 0 || stride <= 0) return NVF_EINVAL;
  if (dp <= 0 || hp <= 0 || wp <= 0 || dq <= 0 || hq <= 0 || wq <= 0) return NVF_EINVAL;
  if (out_mode != 0 && out_mode != 1) return NVF_EINVAL;
  WgDims d{};
  d.batch = batch; d.bc = b; d.dp = dp; d.hp = hp; d.wp = wp; d.dq = dq; d.hq = hq; d.wq = wq; d.pad = pad;
  d.out_mode = out_mode; d.jtotal = a * b * k * k * k;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
  if (variant != 1 && workspace) {
    if (workspace_bytes < nvf_wgrad_workspace(batch, a, b, k, dp, hp, wp)) return NVF_EWORKSPACE;
    float* slabs = (float*)workspace;
#define NVF_W(VAR, AA, KS, ST, WP, NB, TX, TY, TZ, IXU)                                                          \
  if (rc == 1 && variant == VAR && a == AA && k == KS && stride == ST && wp == WP && hp % TY == 0 && dp % TZ == 0 && \
      b % NB == 0)                                                                                                 \
    rc = launch_wgrad<WCfg<AA, KS, ST, NB, TX, TY, TZ, IXU>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    if (variant == 0 && a == 1 && k == 3 && stride == 1 && pad == 1 && out_mode == 0 && dp == hp && hp == wp &&
        dq == dp && hq == dp && wq == dp) {     // classifier heads (heads.hip): p = dlogit, q = X
      int n = 0;
      if (nvf_head_wgrad_launch(p, q, slabs, 256, batch, b, wp, &n, s) == 0) {   // 256 slabs: see WgradBatch.add_heads3
        if (defer_nslab) *defer_nslab = n;
        else wgrad_reduce<<<(d.jtotal + 63) / 64, 1024, 0, s>>>(slabs, dw, n, d.jtotal, accumulate);
        rc = NVF_OK;
      }
    }
    // wide decoder: 16 q-channels are the MFMA columns, 16 / 32 p-channels the rows (wgrad16_mfma.hip)
    if (rc == 1 && variant == 0 && b == 16 && a % 16 == 0 && out_mode == 0 && dp == hp && hp == wp && dq == hq &&
        hq == wq && ((k == 4 && stride == 1 && pad == 0 && dq == dp + 3) || (k == 5 && stride == 2 && pad == 0 && dq == 2 * dp + 3))) {
      int n = 0;
      if (nvf_wgrad16_launch(p, q, slabs, batch, a, k, stride, pad, dp, dq, kMaxSlabs, &n, s) == 0) {
        if (defer_nslab) *defer_nslab = n;
        else wgrad_reduce<<<(d.jtotal + 63) / 64, 1024, 0, s>>>(slabs, dw, n, d.jtotal, accumulate);
        rc = NVF_OK;
      }
    }
    const bool cube_k4 = a == 8 && b == 8 && k == 4 && stride == 1 && pad == 0 && out_mode == 0 && dp == wp &&
                         hp == wp && dq == wp + 3 && hq == wp + 3 && wq == wp + 3;
    if (rc == 1 && variant == 0 && cube_k4 && wp == 32) rc = launch_wgrad_mfma<MCfg<32, 4, 4>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    if (rc == 1 && variant == 0 && cube_k4 && wp == 16) rc = launch_wgrad_mfma<MCfg<16, 2, 8>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    if (rc == 1 && variant == 7 && cube_k4 && wp == 32) rc = launch_wgrad_mfma<MCfg<32, 2, 8>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    if (rc == 1 && variant == 8 && cube_k4 && wp == 32) rc = launch_wgrad_mfma<MCfg<32, 2, 4>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    const bool cube_t5 = a == 8 && b == 8 && k == 5 && stride == 2 && pad == 0 && out_mode == 0 && dp == 16 && hp == 16 &&
                         wp == 16 && dq == 35 && hq == 35 && wq == 35;      // up2: p = X [8,16^3], q = dY [8,35^3]
    if (rc == 1 && variant == 0 && cube_t5) rc = launch_wgrad_s2k5<TWCfg<2, 2>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    if (rc == 1 && variant == 7 && cube_t5) rc = launch_wgrad_s2k5<TWCfg<1, 4>>(p, q, dw, slabs, d, accumulate, s, defer_nslab);
    NVF_W(9, 8, 4, 1, 32, 8, 32, 8, 2, 0)    // conv2 narrow, VALU form: p = dY [8,32^3], q = X [8,35^3]
    NVF_W(9, 8, 4, 1, 16, 8, 16, 8, 2, 0)    // conv1 narrow, VALU form
    NVF_W(0, 8, 5, 2, 16, 4, 16, 4, 2, 0)    // up2 narrow, VALU form (variant 9 below selects it explicitly)
    NVF_W(9, 8, 5, 2, 16, 4, 16, 4, 2, 0)
    NVF_W(0, 16, 5, 2, 8, 4, 8, 4, 2, 0)     // up1 narrow: p = X [16,8^3], q = dY [8,19^3]
    NVF_W(0, 8, 5, 2, 4, 4, 4, 4, 4, 0)      // conv0 narrow: p = X [8,4^3], q = dY [16,8^3]
    NVF_W(0, 16, 4, 1, 32, 8, 32, 4, 2, 0)   // conv2 wide (415 vs 547 us at batch 16, tools/wide_sweep.py)
    NVF_W(0, 16, 4, 1, 16, 8, 16, 8, 2, 0)   // conv1 wide
    NVF_W(0, 16, 5, 2, 16, 2, 16, 4, 2, 0)   // up2 wide (134 vs 156 us)
    NVF_W(0, 32, 5, 2, 8, 4, 8, 4, 2, 0)     // up1 wide
    NVF_W(0, 16, 5, 2, 4, 4, 4, 4, 4, 0)     // conv0 wide
    // wide-decoder tuning candidates (tools/wide_sweep.py)
    NVF_W(40, 16, 4, 1, 32, 16, 32, 8, 2, 0)
    NVF_W(41, 16, 4, 1, 32, 8, 32, 4, 2, 0)
    NVF_W(42, 16, 4, 1, 32, 8, 32, 8, 4, 0)
    NVF_W(43, 16, 4, 1, 32, 4, 32, 8, 2, 0)
    NVF_W(40, 16, 5, 2, 16, 8, 16, 4, 2, 0)
    NVF_W(41, 16, 5, 2, 16, 4, 16, 8, 2, 0)
    NVF_W(42, 16, 5, 2, 16, 4, 16, 4, 4, 0)
    NVF_W(43, 16, 5, 2, 16, 2, 16, 4, 2, 0)
    NVF_W(0, 8, 3, 1, 32, 1, 32, 8, 4, 0)    // conv2_cls: p = X [8,32^3], q = dlogit [1,32^3] (out_mode 1)
    NVF_W(0, 8, 3, 1, 16, 1, 16, 8, 4, 0)    // conv1_cls
    NVF_W(0, 16, 3, 1, 8, 1, 8, 8, 8, 0)     // conv0_cls narrow
    NVF_W(0, 16, 3, 1, 32, 1, 32, 8, 4, 0)   // wide heads
    NVF_W(0, 16, 3, 1, 16, 1, 16, 8, 4, 0)
    NVF_W(0, 32, 3, 1, 8, 1, 8, 8, 8, 0)
    NVF_W(0, 1, 3, 1, 32, 8, 32, 8, 4, 8)    // heads in the plain orientation: p = dlogit [1,32^3], q = X [8,32^3]
    NVF_W(0, 1, 3, 1, 16, 8, 16, 8, 4, 4)
    NVF_W(0, 1, 3, 1, 8, 8, 8, 8, 8, 2)
    // tuning alternatives
    NVF_W(2, 1, 3, 1, 32, 8, 32, 8, 2, 8)
    NVF_W(3, 1, 3, 1, 32, 4, 32, 8, 4, 8)
    NVF_W(2, 8, 4, 1, 32, 8, 32, 8, 2, 1)
    NVF_W(3, 8, 4, 1, 32, 8, 32, 8, 2, 8)
    NVF_W(4, 8, 4, 1, 32, 8, 32, 4, 2, 2)
    NVF_W(5, 8, 4, 1, 32, 4, 32, 8, 4, 2)
    NVF_W(6, 8, 4, 1, 32, 8, 32, 8, 4, 2)
    NVF_W(2, 8, 3, 1, 32, 1, 32, 8, 4, 1)
    NVF_W(3, 8, 3, 1, 32, 1, 32, 4, 2, 2)
    NVF_W(2, 8, 5, 2, 16, 4, 16, 4, 2, 1)
    NVF_W(3, 8, 5, 2, 16, 8, 16, 4, 2, 2)
    NVF_W(4, 8, 5, 2, 16, 4, 16, 2, 2, 2)
#undef NVF_W
  }
  if (rc == 1) {
    wgrad_naive<<<(d.jtotal + 3) / 4, 256, 0, s>>>(p, q, dw, a, k, stride, d, accumulate);
    if (defer_nslab) *defer_nslab = 0;   // dw is already final
    rc = NVF_OK;
  }
  NVF_LAUNCH_CHECK();
  return rc;
}

// job 0: up1 (p = X [B,16,8^3], q = dY [B,8,19^3]); job 1: conv0 (p = X [B,8,4^3], q = dY [B,16,8^3]); both k 5,
// stride 2, out_mode 0.  slabs[j] holds up to 512 slabs of 16000 floats; nslabs[j] = number written.
extern "C" int nvf_wgrad_up1_conv0_partial(const float* const* ps, const float* const* qs, float* const* slabs,
                                           int batch, int* nslabs, void* stream) {
  if (!ps || !qs || !slabs || !nslabs || batch <= 0) return NVF_EINVAL;
  using C0 = WCfg<16, 5, 2, 4, 8, 4, 2, 0>; using C1 = WCfg<8, 5, 2, 4, 4, 4, 4, 0>;
  WgTiled2 m{};
  for (int j = 0; j < 2; ++j)
    if (!ps[j] || !qs[j] || !slabs[j]) return NVF_EINVAL;
  fill_up1_conv0<C0, C1>(m, ps, qs, slabs, batch, nslabs);
  wgrad_tiled2_kernel<C0, C1><<<m.nx[0] * m.ny[0] + m.nx[1] * m.ny[1], C0::NT, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_wgrad(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes, int batch,
                         int a, int b, int k, int stride, int pad, int dp, int hp, int wp, int dq, int hq, int wq,
                         int out_mode, int accumulate, int variant, void* stream) {
  return wgrad_dispatch(p, q, dw, workspace, workspace_bytes, batch, a, b, k, stride, pad, dp, hp, wp, dq, hq, wq,
                        out_mode, accumulate, variant, stream, nullptr);
}

// The partial-sum launch only: slabs land in `workspace` (which must stay untouched until the reduction), *nslab
// tells how many (0: no slabs, dw was written directly).  nvf_wgrad_reduce_multi then finishes up to 16 such
// gradients in ONE launch -- a backward pass has ten, and ten tiny reductions were ten launch gaps.
extern "C" int nvf_wgrad_partial(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes,
                                 int batch, int a, int b, int k, int stride, int pad, int dp, int hp, int wp, int dq,
                                 int hq, int wq, int out_mode, int variant, int* nslab, void* stream) {
  if (!nslab) return NVF_EINVAL;
  return wgrad_dispatch(p, q, dw, workspace, workspace_bytes, batch, a, b, k, stride, pad, dp, hp, wp, dq, hq, wq,
                        out_mode, 0, variant, stream, nslab);
}

struct WgReduceMulti {
  const float* slabs[16];
  float* dw[16];
  const float* add[16];        // optional addend per gradient (the weight-rate term's gradient), or null
  int32_t nslab[16], jtotal[16], blk_base[17];
  int32_t n, fuse;             // fuse: apply `adam` to every element written
  NvfAdamFuse adam;
};

// same arithmetic as wgrad_reduce (16 interleaved slices in ascending slab order, then slices 0..15)
__device__ __forceinline__ void wgrad_reduce_multi_body(const WgReduceMulti& d, int bid, float (*part)[64]) {
  int t = 0;
  while (t + 1 < d.n && bid >= d.blk_base[t + 1]) ++t;
  const float* slabs = d.slabs[t];
  const int nslab = d.nslab[t], jtotal = d.jtotal[t];
  const int jl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int j = (bid - d.blk_base[t]) * 64 + jl;
  // four independent partial sums keep four loads in flight; the order (4 strided chains, then a fixed tree) is
  // the same in wgrad_reduce and wgrad_reduce_multi
  // (slice sl of 16 takes slabs sl, sl + 16, ...: up to 16 of them are fetched before the first add -- one round trip
  // for <= 256 slabs -- then added in ascending order in four interleaved chains)
  // (fetching the optimiser state of the output here, under the slab loads, was slower: 15.3 vs 12.3 us)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < jtotal) {
    for (int g0 = sl; g0 < nslab; g0 += 256) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = g0 + 16 * i < nslab ? slabs[(size_t)(g0 + 16 * i) * jtotal + j] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
  }
  const float s = (s0 + s1) + (s2 + s3);
  part[sl][jl] = s;
  __syncthreads();
  if (sl == 0) {
    int bad = 0;
    if (j < jtotal) {
      float v = part[0][jl];
#pragma unroll
      for (int k = 1; k < 16; ++k) v += part[k][jl];
      if (d.add[t]) v += d.add[t][j];
      d.dw[t][j] = v;
      if (d.fuse) bad = adam_fused_elem(d.adam, d.dw[t] + j, v);
    }
    if (d.fuse && d.adam.bad_count) {          // wave 0 of the workgroup: integer-valued, order-free
      const unsigned long long any = __ballot(bad);
      if (any && jl == 0) atomicAdd(d.adam.bad_count, (float)__popcll(any));
    }
  }
}

__global__ __launch_bounds__(1024) void wgrad_reduce_multi(WgReduceMulti d) {
  __shared__ float part[16][64];
  wgrad_reduce_multi_body(d, blockIdx.x, part);
}

extern "C" int nvf_wgrad_reduce_multi(const float* const* slabs, float* const* dws, const int* nslabs,
                                      const int* jtotals, int n, void* stream) {
  if (!slabs || !dws || !nslabs || !jtotals || n <= 0 || n > 16) return NVF_EINVAL;
  WgReduceMulti d{};
  int base = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    if (nslabs[i] == 0) continue;     // written directly by the partial launch
    if (!slabs[i] || !dws[i] || nslabs[i] < 0 || jtotals[i] <= 0) return NVF_EINVAL;
    d.slabs[m] = slabs[i]; d.dw[m] = dws[i]; d.nslab[m] = nslabs[i]; d.jtotal[m] = jtotals[i];
    d.blk_base[m] = base;
    base += (jtotals[i] + 63) / 64;
    ++m;
  }
  d.blk_base[m] = base;
  d.n = m;
  if (m == 0) return NVF_OK;
  wgrad_reduce_multi<<<base, 1024, 0, nvf_stream(stream)>>>(d);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// per-channel sums (bias gradients): out[c] (+)= sum_{n,s} x[n,c,s]
// stage 1: grid (c, G) partial sums over contiguous chunks; stage 2: fixed-order add.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void channel_sum_partial(const float* __restrict__ x, float* __restrict__ part,
                                                           int batch, int c, int spatial, int chunk) {
  __shared__ float red[16];
  const int ch = blockIdx.x, g = blockIdx.y;
  const long total = (long)batch * spatial;   // flattened (n, s) index space of this channel
  const long lo = (long)g * chunk;
  long hi = lo + chunk;
  if (hi > total) hi = total;
  float s = 0.f;
  for (long e = lo + threadIdx.x; e < hi; e += blockDim.x) {
    const long n = e / spatial, sp = e % spatial;
    s += x[((size_t)n * c + ch) * spatial + sp];
  }
  s = nvf_block_sum(s, red);
  if (threadIdx.x == 0) part[(size_t)g * c + ch] = s;
}

__global__ void channel_sum_final(const float* __restrict__ part, float* __restrict__ out, int c, int nchunk,
                                  int accumulate) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float s = 0.f;
  for (int g = 0; g < nchunk; ++g) s += part[(size_t)g * c + ch];
  out[ch] = accumulate ? out[ch] + s : s;
}

extern "C" size_t nvf_channel_sum_workspace(int c) { return (size_t)kSumChunks * c * sizeof(float); }

extern "C" int nvf_channel_sum(const float* x, float* out, void* workspace, size_t workspace_bytes, int batch, int c,
                               int spatial, int accumulate, void* stream) {
  if (!x || !out || !workspace || batch <= 0 || c <= 0 || spatial <= 0) return NVF_EINVAL;
  if (workspace_bytes < nvf_channel_sum_workspace(c)) return NVF_EWORKSPACE;
  const long total = (long)batch * spatial;
  long chunk = (total + kSumChunks - 1) / kSumChunks;
  if (chunk < 1024) chunk = 1024;
  const int nchunk = (int)((total + chunk - 1) / chunk);
  hipStream_t s = nvf_stream(stream);
  channel_sum_partial<<<dim3(c, nchunk), 256, 0, s>>>(x, (float*)workspace, batch, c, spatial, (int)chunk);
  channel_sum_final<<<(c + 63) / 64, 64, 0, s>>>((const float*)workspace, out, c, nchunk, accumulate);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// all bias gradients of a backward pass in two launches: out_i[c] = sum_{n,s} x_i[n,c,s] for up to 12 tensors
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(NVF_SUM_T) void multi_channel_sum_partial(MultiSumDesc d, float* __restrict__ part) {
  __shared__ float red[16];
  multi_channel_sum_partial_body<NVF_SUM_T>(d, part, blockIdx.x, blockIdx.y, red);
}

// the slab reduction of all weight gradients and the partial bias sums are independent: one launch
__global__ __launch_bounds__(1024) void wgrad_reduce_and_sums(WgReduceMulti r, int r_blocks, MultiSumDesc m,
                                                              float* __restrict__ part) {
  __shared__ float sm[16][64];
  const int bid = blockIdx.x;
  if (bid < r_blocks) { wgrad_reduce_multi_body(r, bid, sm); return; }
  const int q = bid - r_blocks;
  multi_channel_sum_partial_body<NVF_SUM_T>(m, part, q % m.total_channels, q / m.total_channels, &sm[0][0]);
}

// ... and with the latent tail (latent_tail.h) as one more workgroup, the first one dispatched
__global__ __launch_bounds__(1024, 8) void wgrad_reduce_sums_tail(WgReduceMulti r, int r_blocks, MultiSumDesc m,
                                                               float* __restrict__ part, LatentTail t) {
  __shared__ float sm[kTailLds > 16 * 64 ? kTailLds : 16 * 64];
  if (blockIdx.x == 0) { latent_tail_body(t, sm, t.dx_addend); return; }
  const int bid = blockIdx.x - 1;
  if (bid < r_blocks) { wgrad_reduce_multi_body(r, bid, (float(*)[64])sm); return; }
  const int q = bid - r_blocks;
  multi_channel_sum_partial_body<NVF_SUM_T>(m, part, q % m.total_channels, q / m.total_channels, sm);
}

// The slab reduction (+ fused optimiser) and the step's final passes + tail (finals_tail_body) in ONE launch: possible
// when no final pass reads anything this launch's reduction writes (the partial bias sums were made by an earlier
// launch: nvf_wgrad_trunk5_heads_sums_partial).  The reduction's workgroups read NOTHING from the step buffer (their
// optimiser coefficients are the copy that earlier launch staged), so the schedule hand-over waits only for the final
// passes' workgroups.  (With the ~800 reduction workgroups in the arrival count -- on one counter or through ~sqrt(n)
// group counters -- the launch took 18 us instead of 13: the device-scope atomics' round trips end up behind the last
// of them.)
__global__ __launch_bounds__(1024) void wgrad_reduce_finals_tail(WgReduceMulti r, int f_blocks, FinalsArgs a,
                                                                 int sum_blocks, NvfStepTail t, TailRanges rg) {
  __shared__ float sm[16][64];
  // the final passes are dependent-load chains (~10 us alone): they take the FIRST workgroups so that they start with
  // the launch and run beside the reduction, not after its first wave of workgroups
  const int bid = blockIdx.x;
  if (bid < f_blocks) finals_tail_body(a, sum_blocks, t, rg, bid, f_blocks);
  else wgrad_reduce_multi_body(r, bid - f_blocks, sm);
}

// ... and without the optimiser (data-parallel steps): the slab reduction with its addends + nvf_finals_flush in one launch
__global__ __launch_bounds__(1024) void wgrad_reduce_finals(WgReduceMulti r, int f_blocks, FinalsArgs a, int sum_blocks) {
  __shared__ float sm[16][64];
  const int bid = blockIdx.x;
  if (bid < f_blocks) finals_plain_body(a, sum_blocks, bid);
  else wgrad_reduce_multi_body(r, bid - f_blocks, sm);
}

extern "C" int nvf_wgrad_reduce_finals(const float* const* slabs, float* const* dws, const int* nslabs,
                                       const int* jtotals, int n, const float* const* addends, NvfStepCtx* ctx,
                                       void* stream) {
  if (!slabs || !dws || !nslabs || !jtotals || n <= 0 || n > 16 || !nvf_ctx_ok(ctx)) return NVF_EINVAL;
  WgReduceMulti r{};
  int base = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    if (nslabs[i] == 0) continue;
    if (!slabs[i] || !dws[i] || nslabs[i] < 0 || jtotals[i] <= 0) return NVF_EINVAL;
    r.slabs[m] = slabs[i]; r.dw[m] = dws[i]; r.nslab[m] = nslabs[i]; r.jtotal[m] = jtotals[i];
    r.add[m] = addends ? addends[i] : nullptr;
    r.blk_base[m] = base;
    base += (jtotals[i] + 63) / 64;
    ++m;
  }
  r.blk_base[m] = base;
  r.n = m;
  const FinalsArgs a = ctx->args;
  ctx->args = FinalsArgs{};
  ctx->deferring = 0;
  if (a.has_f && a.f_nterm > 3) return NVF_EINVAL;
  const int sum_blocks = finals_sum_blocks(a);
  const int f_blocks = 2 + sum_blocks + (a.has_m ? 1 : 0);
  wgrad_reduce_finals<<<f_blocks + base, 1024, 0, nvf_stream(stream)>>>(r, f_blocks, a, sum_blocks);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_wgrad_reduce_finals_tail(const float* const* slabs, float* const* dws, const int* nslabs,
                                            const int* jtotals, int n, const float* const* addends,
                                            const NvfAdamFuse* adam, NvfStepCtx* ctx, const NvfStepTail* tail,
                                            const int64_t* ranges, int nranges, void* stream) {
  if (!slabs || !dws || !nslabs || !jtotals || n <= 0 || n > 16 || !adam || !tail || !nvf_ctx_ok(ctx)) return NVF_EINVAL;
  if (nranges < 0 || nranges > 16 || (nranges > 0 && !ranges)) return NVF_EINVAL;
  const NvfStepTail t = *tail;
  if (!t.p || !t.g || !t.m || !t.v || t.n <= 0) return NVF_EINVAL;
  if (t.acc && (!t.loss_terms || !t.lbits || !t.nbits || t.nnb <= 0 || t.nnb > 16)) return NVF_EINVAL;
  if (t.sched_rows && (!t.sched_buf || !t.sched_cursor || t.sched_words <= 0 || !t.done)) return NVF_EINVAL;
  if (!adam->g_base || !adam->p_base || !adam->m_base || !adam->v_base || adam->n <= 0) return NVF_EINVAL;
  if (t.sched_rows && adam->coef_dev) {     // the reduction must not read what the hand-over of this launch overwrites
    const char* c = (const char*)adam->coef_dev;
    const char* b = (const char*)t.sched_buf;
    if (c + 2 * sizeof(float) > b && c < b + (size_t)t.sched_words * sizeof(int64_t)) return NVF_EINVAL;
  }
  WgReduceMulti r{};
  int base = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    if (nslabs[i] == 0) continue;
    if (!slabs[i] || !dws[i] || nslabs[i] < 0 || jtotals[i] <= 0) return NVF_EINVAL;
    r.slabs[m] = slabs[i]; r.dw[m] = dws[i]; r.nslab[m] = nslabs[i]; r.jtotal[m] = jtotals[i];
    r.add[m] = addends ? addends[i] : nullptr;
    r.blk_base[m] = base;
    base += (jtotals[i] + 63) / 64;
    ++m;
  }
  r.blk_base[m] = base;
  r.n = m;
  r.fuse = 1;
  r.adam = *adam;
  TailRanges rg{};
  for (int q = 0; q < nranges; ++q) {
    if (ranges[2 * q] < 0 || ranges[2 * q + 1] > t.n || ranges[2 * q] > ranges[2 * q + 1]) return NVF_EINVAL;
    rg.lo[q] = (long)ranges[2 * q]; rg.hi[q] = (long)ranges[2 * q + 1];
  }
  rg.n = nranges;
  const FinalsArgs a = ctx->args;
  ctx->args = FinalsArgs{};
  ctx->deferring = 0;
  if (a.has_f && a.f_nterm > 3) return NVF_EINVAL;
  const int sum_blocks = finals_sum_blocks(a);
  const int f_blocks = 2 + sum_blocks + 1;
  wgrad_reduce_finals_tail<<<f_blocks + base, 1024, 0, nvf_stream(stream)>>>(r, f_blocks, a, sum_blocks, t, rg);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// Queue the latent tail of a training step (NVFPCC.py:186-196 backward of the latent generator): the gradient of the
// latent rate (+ dx_addend, the decoder's gradient) -> GDN backward -> 1x1x1 weight and bias gradients, on
// [batch, c <= 8, spatial] tensors.  It runs as one workgroup of the NEXT nvf_wgrad_reduce_multi_and_sums launch on
// the same stream (all of its inputs must already be enqueued there); results = nvf_latent_rate + nvf_gdn_bwd +
// nvf_wgrad (+ the bias sum, whose summation order differs from nvf_multi_channel_sum).
extern "C" int nvf_latent_tail_queue(NvfStepCtx* ctx, const float* lat, const int64_t* block_ids, const float* sigma, const float* mu,
                                     const float* dx_addend, float* dlat, float* dsigma, float* dmu,
                                     const float* g_dev, float g_host, int mode, uint64_t seed, uint64_t step,
                                     const uint64_t* step_dev, const float* h, const float* beta_hat,
                                     const float* gamma_hat, float* dh, float* dbeta_hat, float* dgamma_hat,
                                     const float* e, float* dw, float* db, int batch, int c, int spatial) {
  if (!lat || !sigma || !mu || !dlat || !dsigma || !dmu || !h || !beta_hat || !gamma_hat || !dh || !dbeta_hat ||
      !dgamma_hat || !e || !dw || !db)
    return NVF_EINVAL;
  if (!nvf_ctx_ok(ctx) || ctx->tail_pending) return NVF_EINVAL;
  if (batch <= 0 || c <= 0 || c > kTailMaxC || spatial <= 0 || (mode != 0 && mode != 1)) return NVF_EINVAL;
  LatentTail t{};
  t.lat = lat; t.block_ids = block_ids; t.sigma = sigma; t.mu = mu; t.dx_addend = dx_addend; t.dlat = dlat;
  t.dsigma = dsigma; t.dmu = dmu; t.g_dev = g_dev; t.step_dev = step_dev; t.seed = seed; t.step = step;
  t.h = h; t.beta_hat = beta_hat; t.gamma_hat = gamma_hat; t.dh = dh; t.dbeta_hat = dbeta_hat; t.dgamma_hat = dgamma_hat;
  t.e = e; t.dw = dw; t.db = db; t.g_host = g_host; t.batch = batch; t.c = c; t.spatial = spatial; t.mode = mode;
  ctx->tail = t;
  ctx->tail_pending = 1;
  return NVF_OK;
}

extern "C" int nvf_latent_tail_pending(const NvfStepCtx* ctx) { return nvf_ctx_ok(ctx) && ctx->tail_pending ? 1 : 0; }
extern "C" void nvf_latent_tail_cancel(NvfStepCtx* ctx) {
  if (nvf_ctx_ok(ctx)) ctx->tail_pending = ctx->stem_pending = 0;      // (and a stem backward queued beside it)
}

__global__ void multi_channel_sum_final(MultiSumDesc d, const float* __restrict__ part) {
  multi_channel_sum_final_body(d, part, blockIdx.x * blockDim.x + threadIdx.x);
}

extern "C" size_t nvf_multi_channel_sum_workspace(int total_channels) {
  return (size_t)kSumChunks * total_channels * sizeof(float);
}

extern "C" int nvf_multi_channel_sum(const float* const* xs, float* const* outs, const int* channels,
                                     const int* spatials, int ntensors, int batch, void* workspace,
                                     size_t workspace_bytes, NvfStepCtx* ctx, void* stream) {
  if (!xs || !outs || !channels || !spatials || ntensors <= 0 || ntensors > 12 || batch <= 0 || !workspace)
    return NVF_EINVAL;
  MultiSumDesc d{};
  int base = 0;
  long biggest = 0;
  for (int i = 0; i < ntensors; ++i) {
    if (!xs[i] || !outs[i] || channels[i] <= 0 || spatials[i] <= 0) return NVF_EINVAL;
    d.x[i] = xs[i]; d.out[i] = outs[i]; d.c[i] = channels[i]; d.spatial[i] = spatials[i]; d.chan_base[i] = base;
    base += channels[i];
    if ((long)batch * spatials[i] > biggest) biggest = (long)batch * spatials[i];
  }
  d.ntensors = ntensors; d.batch = batch; d.total_channels = base;
  (void)biggest;
  long nchunk = batch < kSumChunks ? batch : kSumChunks;   // groups of consecutive batch entries
  d.nchunk = (int)nchunk;
  if (workspace_bytes < nvf_multi_channel_sum_workspace(base)) return NVF_EWORKSPACE;
  hipStream_t s = nvf_stream(stream);
  multi_channel_sum_partial<<<dim3(base, d.nchunk), NVF_SUM_T, 0, s>>>(d, (float*)workspace);
  if (!nvf_finals_push_sums(ctx, d, (const float*)workspace))
    multi_channel_sum_final<<<(base + 63) / 64, 64, 0, s>>>(d, (const float*)workspace);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// nvf_wgrad_reduce_multi and the partial pass of nvf_multi_channel_sum in ONE launch (they are independent), then the
// final pass of the bias sums: the tail of a backward pass in two launches instead of three.  Results are those of
// the two separate calls, bit for bit.
extern "C" int nvf_wgrad_reduce_multi_and_sums(const float* const* slabs, float* const* dws, const int* nslabs,
                                               const int* jtotals, int n, const float* const* xs, float* const* outs,
                                               const int* channels, const int* spatials, int ntensors, int batch,
                                               void* workspace, size_t workspace_bytes, NvfStepCtx* ctx, void* stream) {
  return nvf_wgrad_reduce_multi_and_sums_fused(slabs, dws, nslabs, jtotals, n, nullptr, nullptr, xs, outs, channels,
                                               spatials, ntensors, batch, workspace, workspace_bytes, ctx, stream);
}

extern "C" int nvf_wgrad_reduce_multi_and_sums_fused(const float* const* slabs, float* const* dws, const int* nslabs,
                                                     const int* jtotals, int n, const float* const* addends,
                                                     const NvfAdamFuse* adam, const float* const* xs,
                                                     float* const* outs, const int* channels, const int* spatials,
                                                     int ntensors, int batch, void* workspace, size_t workspace_bytes,
                                                     NvfStepCtx* ctx, void* stream) {
  if (!slabs || !dws || !nslabs || !jtotals || n <= 0 || n > 16) return NVF_EINVAL;
  if (!xs || !outs || !channels || !spatials || ntensors <= 0 || ntensors > 12 || batch <= 0 || !workspace)
    return NVF_EINVAL;
  WgReduceMulti r{};
  int base = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    if (nslabs[i] == 0) continue;
    if (!slabs[i] || !dws[i] || nslabs[i] < 0 || jtotals[i] <= 0) return NVF_EINVAL;
    r.slabs[m] = slabs[i]; r.dw[m] = dws[i]; r.nslab[m] = nslabs[i]; r.jtotal[m] = jtotals[i];
    r.add[m] = addends ? addends[i] : nullptr;
    r.blk_base[m] = base;
    base += (jtotals[i] + 63) / 64;
    ++m;
  }
  r.blk_base[m] = base;
  r.n = m;
  if (adam) {
    if (!adam->g_base || !adam->p_base || !adam->m_base || !adam->v_base || adam->n <= 0) return NVF_EINVAL;
    r.fuse = 1;
    r.adam = *adam;
  }
  MultiSumDesc d{};
  int cb = 0;
  for (int i = 0; i < ntensors; ++i) {
    if (!xs[i] || !outs[i] || channels[i] <= 0 || spatials[i] <= 0) return NVF_EINVAL;
    d.x[i] = xs[i]; d.out[i] = outs[i]; d.c[i] = channels[i]; d.spatial[i] = spatials[i]; d.chan_base[i] = cb;
    cb += channels[i];
  }
  d.ntensors = ntensors; d.batch = batch; d.total_channels = cb;
  d.nchunk = batch < kSumChunks ? batch : kSumChunks;
  if (workspace_bytes < nvf_multi_channel_sum_workspace(cb)) return NVF_EWORKSPACE;
  hipStream_t s = nvf_stream(stream);
  if (nvf_ctx_ok(ctx) && ctx->tail_pending) {
    ctx->tail_pending = 0;
    wgrad_reduce_sums_tail<<<1 + base + cb * d.nchunk, 1024, 0, s>>>(r, base, d, (float*)workspace, ctx->tail);
  } else {
    wgrad_reduce_and_sums<<<base + cb * d.nchunk, 1024, 0, s>>>(r, base, d, (float*)workspace);
  }
  if (!nvf_finals_push_sums(ctx, d, (const float*)workspace))
    multi_channel_sum_final<<<(cb + 63) / 64, 64, 0, s>>>(d, (const float*)workspace);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

