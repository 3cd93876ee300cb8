// Matrix-core weight gradients for layers with 16 q-channels and 16 / 32 p-channels -- the wide decoder
// (chanstr 16,32,16,16; autograd backward of F.conv3d network.py:687 and F.conv_transpose3d network.py:621):
//
//   dw[a][b][tap] = sum_{n, pos} p[n, a, pos] * q[n, b, S * pos - pad + tap]
//     conv  (k 4, S 1): p = dY [B,16,n^3],  q = X  [B,16,(n+3)^3]   -> dw[cout][cin][k]
//     convT (k 5, S 2): p = X  [B,a,n^3],   q = dY [B,16,(2n+3)^3]  -> dw[cin][cout][k]
//
// One v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain) per (tap, four consecutive x positions):
//   D_tap[a][b] += sum_{k=0..3} A[a][k] * B[k][b],   A[a][k] = p[a, z, y, x0 + k],  B[k][b] = q[b, S(z,y,x0+k) - pad + tap]
// rows = 16 p-channels, columns = 16 q-channels, every lane useful.  A workgroup has NW waves, wave w owns a run of the
// K*K (kz, ky) tap rows (K taps each; its accumulator tiles stay in registers for the whole launch: 8 waves = two per
// SIMD with 2 rows each for k = 4, 3 or 4 rows for k = 5 -- K waves of one kz each left a SIMD with one wave, or with
// two of five) and walks its share of the items
// (batch element, z plane, TY rows); p / q tiles of the next item stream into the second LDS buffer by LDS-DMA while
// this item's MFMAs issue.  An A fragment (one ds_read_b32) feeds a wave's 2 K .. 4 K MFMAs, a B fragment one.  Each workgroup leaves
// one slab of 16 x 16 x K^3 partial sums; the caller's fixed-order reduction (nvf_wgrad_reduce_multi*) adds them.
#include "nvf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct W16Dims {
  int batch, ac, dp, dq, pad, items, items_per_wg, tiles_y;
};

// 32 lanes of a read group: channel c = 0..15 at stride cs, kq = 0..1 at stride s
constexpr int w16_conflicts(int cs, int s) {
  int cnt[32] = {};
  int worst = 0;
  for (int kq = 0; kq < 2; ++kq)
    for (int c = 0; c < 16; ++c) {
      const int bank = (c * cs + kq * s) % 32;
      if (++cnt[bank] > worst) worst = cnt[bank];
    }
  return worst;
}
constexpr int w16_stride(int least, int s) {
  int best = least, bw = 99;
  for (int cs = least; cs < least + 64; ++cs) {
    const int w = w16_conflicts(cs, s);
    if (w < bw) { bw = w; best = cs; }
  }
  return best;
}

template <int K_, int S_, int WP_, int TY_, int NW_ = K_>
struct W16 {
  static constexpr int K = K_, S = S_, WP = WP_, TY = TY_, NW = NW_, NT = NW * 64, KK = K * K, K3 = K * K * K;
  static_assert(WP % 4 == 0, "four x positions per MFMA");
  // the K*K (kz, ky) tap rows (K taps along x each) are dealt to the waves in runs: wave w owns rows
  // [w KK / NW, (w + 1) KK / NW) -- MAXR or MAXR - 1 of them
  static constexpr int MAXR = (KK + NW - 1) / NW;
  static constexpr bool EVEN = KK % NW == 0;
  static constexpr int QY = S * (TY - 1) + K, QX = S * (WP - 1) + K;
  static constexpr int QRS = QX, QPS = QY * QRS;
  static constexpr int QCS = w16_stride(K * QPS, S);       // q channel stride (bank spread of the B reads)
  static constexpr int PCS = w16_stride(TY * WP, 1);       // p channel stride
  static constexpr int BUF = 16 * QCS + 16 * PCS;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two item buffers in LDS");
  static constexpr int QIT = (K * QPS + NT - 1) / NT, PIT = (TY * WP + NT - 1) / NT;   // DMA instructions per channel
};

// the whole item loop for a wave that owns NR tap rows starting at row0 (every wave of the workgroup runs the same
// number of barriers whichever NR it has)
template <class C, int NR>
__device__ __forceinline__ void w16_run(const float* __restrict__ p, const float* __restrict__ q,
                                        float* __restrict__ slabs, const W16Dims& d, float* lds, int row0) {
  constexpr int K = C::K, S = C::S, WP = C::WP, TY = C::TY, K3 = C::K3, QRS = C::QRS, QPS = C::QPS,
                QCS = C::QCS, PCS = C::PCS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int ag = blockIdx.y;                                       // group of 16 p-channels
  const int first = blockIdx.x * d.items_per_wg, last = min(first + d.items_per_wg, d.items);
  const size_t pvol = (size_t)d.dp * d.dp * d.dp, qvol = (size_t)d.dq * d.dq * d.dq;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;

  // Staging by LDS-DMA.  The launcher only takes shapes whose q tiles lie inside the tensor (pad 0, dq = S (dp - 1) + K)
  // and whose p rows divide evenly (dp % TY == 0), so a lane's source offset inside the tile is the same for every item
  // and channel: it is computed once, an instruction then costs a scalar base and one DMA issue (per-element address
  // arithmetic and bounds checks had cost 40 of conv2's 172 us).
  unsigned qoff[C::QIT], poff[C::PIT];
  bool qlive[C::QIT], plive[C::PIT];
#pragma unroll
  for (int i = 0; i < C::QIT; ++i) {
    const int w = (i * C::NW + wave) * 64 + lane;
    const int zz = w / QPS, rem = w - zz * QPS, yy = rem / QRS, xx = rem - yy * QRS;
    qlive[i] = w < K * QPS;
    qoff[i] = qlive[i] ? (unsigned)((zz * d.dq + yy) * d.dq + xx) * 4u : 0u;
  }
#pragma unroll
  for (int i = 0; i < C::PIT; ++i) {
    const int w = (i * C::NW + wave) * 64 + lane;
    plive[i] = w < TY * WP;
    poff[i] = plive[i] ? (unsigned)w * 4u : 0u;
  }
  auto stage = [&](int item, int buf) {
    const int ty = item % d.tiles_y, t = item / d.tiles_y, z = t % d.dp, n = t / d.dp;
    const int y0 = ty * TY;
    const float* qb = q + (size_t)n * 16 * qvol + ((size_t)(S * z) * d.dq + S * y0) * d.dq;
    const float* pb = p + ((size_t)n * d.ac + ag * 16) * pvol + ((size_t)z * d.dp + y0) * d.dp;
#pragma unroll 2
    for (int c = 0; c < 16; ++c) {
      const unsigned cb = lds0 + (unsigned)(buf * C::BUF + c * QCS) * 4u;
#pragma unroll
      for (int i = 0; i < C::QIT; ++i)
        if (qlive[i]) nvf_glds_row(qb + (size_t)c * qvol, qoff[i], cb + (unsigned)((i * C::NW + wave) * 64) * 4u);
      const unsigned pc = lds0 + (unsigned)(buf * C::BUF + 16 * QCS + c * PCS) * 4u;
#pragma unroll
      for (int i = 0; i < C::PIT; ++i)
        if (plive[i]) nvf_glds_row(pb + (size_t)c * pvol, poff[i], pc + (unsigned)((i * C::NW + wave) * 64) * 4u);
    }
  };

  // LDS word of tap row r (kz, ky) relative to the (y, x group) origin of a q tile
  int roff[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) roff[r] = ((row0 + r) / K) * QPS + ((row0 + r) % K) * QRS;

  f32x4 acc[NR * K];
#pragma unroll
  for (int t = 0; t < NR * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (first < last) stage(first, 0);
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    const int buf = (item - first) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the item has landed
    __syncthreads();                                              // ... everyone's; the other buffer is free
    if (item + 1 < last) stage(item + 1, buf ^ 1);
    const float* qs = lds + buf * C::BUF + j * QCS + S * kq;
    const float* ps = lds + buf * C::BUF + 16 * QCS + j * PCS + kq;
#pragma unroll 1
    for (int y = 0; y < TY; ++y) {
      const float* qr = qs + S * y * QRS;
      const float* pr = ps + y * WP;
      float bc[K], bn[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) bc[kx] = qr[roff[0] + kx];
#pragma unroll
      for (int xg = 0; xg < WP / 4; ++xg) {
        const float a = pr[4 * xg];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          // next row of taps: (xg, r + 1), or (xg + 1, 0)
          const int nr = r + 1 < NR ? r + 1 : 0, nxg = r + 1 < NR ? xg : xg + 1;
          if (nxg < WP / 4) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) bn[kx] = qr[roff[nr] + S * 4 * nxg + kx];
          }
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
            acc[r * K + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[kx], acc[r * K + kx], 0, 0, 0);
#pragma unroll
          for (int kx = 0; kx < K; ++kx) bc[kx] = bn[kx];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // slab [ac][16][K^3]: lane holds D[a = 4 kq + r4][b = j] of each of this wave's taps
  float* slab = slabs + (size_t)blockIdx.x * d.ac * 16 * K3;
#pragma unroll
  for (int t = 0; t < NR * K; ++t)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      slab[((size_t)(ag * 16 + 4 * kq + r4) * 16 + j) * K3 + row0 * K + t] = acc[t][r4];
}

template <class C>
__global__ __launch_bounds__(C::NT) void wgrad16_mfma(const float* __restrict__ p, const float* __restrict__ q,
                                                      float* __restrict__ slabs, W16Dims d) {
  __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = wave * C::KK / C::NW, row1 = (wave + 1) * C::KK / C::NW;
  if constexpr (C::EVEN) {
    w16_run<C, C::MAXR>(p, q, slabs, d, lds, row0);
  } else {
    if (row1 - row0 == C::MAXR) w16_run<C, C::MAXR>(p, q, slabs, d, lds, row0);
    else w16_run<C, C::MAXR - 1>(p, q, slabs, d, lds, row0);
  }
}

template <class C>
int launch_w16(const float* p, const float* q, float* slabs, W16Dims d, int max_slabs, int* nslab, hipStream_t s) {
  if (d.dp % C::TY != 0 || d.dp != C::WP) return 1;
  d.tiles_y = d.dp / C::TY;
  d.items = d.batch * d.dp * d.tiles_y;
  int n = d.items < max_slabs ? d.items : max_slabs;
  d.items_per_wg = (d.items + n - 1) / n;
  n = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  wgrad16_mfma<C><<<dim3(n, d.ac / 16), C::NT, 0, s>>>(p, q, slabs, d);
  *nslab = n;
  return NVF_OK;
}

// ---- up1 of the wide decoder (k 5, S 2: p = X [B,32,8^3], q = dY [B,16,19^3]) ---------------------------------------------
// The kernel above walks four consecutive x positions per MFMA and reuses an A fragment along a tap row: with 8-wide rows
// that is two uses, and the VALU tile kernel was faster (36 us against 50 at batch 16).  This one turns the loops around:
// K = four positions of an (iy, ix) plane, rows = 16 p-channels, columns = the 16 q-channels -- every lane useful, no bounds
// (2 i + k <= 18) -- and the ACCUMULATORS are the taps: a workgroup owns (block, half of the z planes, p-channel group, kz),
// its waves three or four (ky, kx) taps each for all positions, so an A read feeds three or four MFMAs and a B read one.  One x plane (4 KB) and
// one dY plane (23 KB) per step, the next pair fetched into registers under this step's MFMAs.  320 workgroups at batch 16,
// two slabs per block (the ten workgroups of a (block, half) write disjoint parts of one slab): 8 MB of slabs instead of 33.
struct U1W {
  static constexpr int XS = 65, DS = 361, XW = 16 * XS, DW = 16 * DS, BUF = XW + DW, NTH = 512;
  static constexpr int NX = (16 * 64 + NTH - 1) / NTH, ND = (16 * 361 + NTH - 1) / NTH;
};

template <int NH>      // z halves per block (workgroups per (block, channel group, kz)): NP = 8 / NH planes each
__global__ __launch_bounds__(U1W::NTH) void wgrad16_up1_mfma(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ slabs) {
  constexpr int NP = 8 / NH;
  constexpr int XS = U1W::XS, DS = U1W::DS, XW = U1W::XW, BUF = U1W::BUF, NTH = U1W::NTH, NX = U1W::NX, ND = U1W::ND;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / NH, half = blockIdx.x % NH;       // block, z planes NP half .. NP half + NP - 1 of x
  const int ag = blockIdx.y / 5, kz = blockIdx.y % 5;          // group of 16 p-channels, z tap
  const float* xb = x + ((size_t)b * 32 + ag * 16) * 512;
  const float* db = dy + (size_t)b * 16 * 6859;
  float xv[NX], dv[ND];
  auto load = [&](int iz) {
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int e = tid + u * NTH;                               // (c, position of the plane)
      xv[u] = xb[(size_t)(e >> 6) * 512 + iz * 64 + (e & 63)];
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int e = tid + u * NTH;
      const int c = e / 361, r = e - c * 361;
      dv[u] = e < 16 * 361 ? db[((size_t)c * 19 + 2 * iz + kz) * 361 + r] : 0.f;
    }
  };
  auto store = [&](int buf) {
    float* xs = lds + buf * BUF;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int e = tid + u * NTH;
      xs[(e >> 6) * XS + (e & 63)] = xv[u];
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int e = tid + u * NTH;
      if (e < 16 * 361) xs[XW + e] = dv[u];                      // [c][361]: DS = 361
    }
  };
  // this wave's taps (ky, kx) = t0 .. t0 + nt - 1: 4 for wave 0, 3 for the others -- 7 / 6 / 6 / 6 on the four SIMDs
  const int t0 = wave == 0 ? 0 : 3 * wave + 1, nt = wave == 0 ? 4 : 3;
  int toff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = t0 + u < 25 ? t0 + u : 24;
    toff[u] = (t / 5) * 19 + t % 5;
  }
  f32x4 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  load(NP * half);
  store(0);
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < NP; ++st) {
    if (st + 1 < NP) load(NP * half + st + 1);
    const float* xs = lds + (st & 1) * BUF;
    const float* ds = xs + XW;
    if (nt > 0) {
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int i = 4 * ks + kq, iy = i >> 3, ix = i & 7;
        const float av = xs[j * XS + i];                                    // A[row = p-channel j][k = kq]
        const float* dp = ds + j * DS + (2 * iy) * 19 + 2 * ix;             // B[k = kq][col = q-channel j]
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < nt) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dp[toff[u]], acc[u], 0, 0, 0);
      }
    }
    if (st + 1 < NP) {
      store((st + 1) & 1);                                       // (the other buffer: last read one step ago, behind a barrier)
      __syncthreads();
    }
  }
  float* out = slabs + (size_t)(NH * b + half) * (32 * 16 * 125);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (u < nt) {
      const int t = t0 + u, tap = kz * 25 + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((size_t)(ag * 16 + 4 * kq + r) * 16 + j) * 125 + tap] = acc[u][r];
    }
}

}  // namespace

// Partial sums of dw[a][16][k^3] (a = 16 or 32) into `slabs` (*nslab slabs of a * 16 * k^3 floats, <= max_slabs);
// returns 1 when there is no instantiation for the shape.  Cubic tensors: p [B,a,dp^3], q [B,16,dq^3].
int nvf_wgrad16_launch(const float* p, const float* q, float* slabs, int batch, int a, int k, int stride, int pad, int dp,
                       int dq, int max_slabs, int* nslab, hipStream_t s) {
  W16Dims d{batch, a, dp, dq, pad, 0, 0, 0};
  if (a % 16 != 0 || max_slabs <= 0) return 1;
  if (pad != 0 || dq != stride * (dp - 1) + k) return 1;        // q tiles inside the tensor (see the staging)
  if (max_slabs > 256) max_slabs = 256;                         // one workgroup per CU and p-channel group
  if (k == 4 && stride == 1 && dp == 32) return launch_w16<W16<4, 1, 32, 4, 8>>(p, q, slabs, d, max_slabs, nslab, s);
  if (k == 4 && stride == 1 && dp == 16) return launch_w16<W16<4, 1, 16, 4, 8>>(p, q, slabs, d, max_slabs, nslab, s);
  if (k == 5 && stride == 2 && dp == 16) return launch_w16<W16<5, 2, 16, 2, 8>>(p, q, slabs, d, max_slabs, nslab, s);
  // (up1's 8-wide rows give an A fragment only two uses per tap row: 50 us against 36 us for the VALU tile kernel --
  // wgrad16_up1_mfma keeps the taps in the accumulators instead)
  if (k == 5 && stride == 2 && dp == 8 && a == 32 && max_slabs >= 2 * batch) {
    const int nh = nvf_tune_int("NVF_U1W_NH", 2);
    if (nh == 1) wgrad16_up1_mfma<1><<<dim3(batch, 10), U1W::NTH, 0, s>>>(p, q, slabs);
    else wgrad16_up1_mfma<2><<<dim3(2 * batch, 10), U1W::NTH, 0, s>>>(p, q, slabs);
    *nslab = (nh == 1 ? 1 : 2) * batch;
    return NVF_OK;
  }
  return 1;
}
