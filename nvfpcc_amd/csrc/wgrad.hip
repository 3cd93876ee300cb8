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

struct WgDims {
  int batch, bc;           // batch, number of q channels
  int dp, hp, wp;          // p grid
  int dq, hq, wq;          // q grid
  int pad;
  int tiles_x, tiles_y, tiles_z;
  int items, items_per_wg; // work items (n, tile) and how many each workgroup walks
  int out_mode, jtotal;    // slab layout, slab length A*Bc*K^3
};

__global__ void wgrad_naive(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ dw, int a_ch,
                            int k, int stride, WgDims d, int accumulate) {
  int k3 = k * k * k;
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a_ch * d.bc * k3) return;
  int kk = j % k3, b = (j / k3) % d.bc, a = j / (k3 * d.bc);
  int kz = kk / (k * k), ky = (kk / k) % k, kx = kk % k;
  float acc = 0.f;
  for (int n = 0; n < d.batch; ++n)
    for (int iz = 0; iz < d.dp; ++iz) {
      int qz = iz * stride - d.pad + kz;
      if (qz < 0 || qz >= d.dq) continue;
      for (int iy = 0; iy < d.hp; ++iy) {
        int qy = iy * stride - d.pad + ky;
        if (qy < 0 || qy >= d.hq) continue;
        for (int ix = 0; ix < d.wp; ++ix) {
          int qx = ix * stride - d.pad + kx;
          if (qx < 0 || qx >= d.wq) continue;
          float pv = p[(((size_t)n * a_ch + a) * d.dp + iz) * d.hp * d.wp + iy * d.wp + ix];
          float qv = q[(((size_t)n * d.bc + b) * d.dq + qz) * d.hq * d.wq + qy * d.wq + qx];
          acc = fmaf(pv, qv, acc);
        }
      }
    }
  int o = d.out_mode == 0 ? j : (b * a_ch + a) * k3 + (k3 - 1 - kk);
  dw[o] = accumulate ? dw[o] + acc : acc;
}

template <int A_, int KS_, int S_, int NB_, int TX_, int TY_, int TZ_>
struct WCfg {
  static constexpr int A = A_, KS = KS_, S = S_, NB = NB_, TX = TX_, TY = TY_, TZ = TZ_;
  static constexpr int K3 = KS * KS * KS;
  static constexpr int WPB = (K3 + 63) / 64;          // waves per q channel
  static constexpr int NT = NB * WPB * 64;
  static constexpr int QX = (TX - 1) * S + KS, QY = (TY - 1) * S + KS, QZ = (TZ - 1) * S + KS;
  static constexpr int mod32(int v, int r) { return v + ((r - v % 32) + 32) % 32; }  // smallest >= v, == r (mod 32)
  static constexpr int QRS = mod32(QX, KS % 32);              // row stride
  static constexpr int QPS = mod32(QY * QRS, (KS * KS) % 32); // plane stride
  static constexpr int QCS = QZ * QPS;                        // channel stride
  static constexpr int LDSF = NB * QCS;
  static_assert(TX % 4 == 0, "p rows are read four at a time");
  static_assert(NT <= 1024, "workgroup size");
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NT) void wgrad_tiled(const float* __restrict__ p, const float* __restrict__ q,
                                                     float* __restrict__ slabs, WgDims d) {
  constexpr int A = C::A, KS = C::KS, S = C::S, NB = C::NB, TX = C::TX, TY = C::TY, TZ = C::TZ, K3 = C::K3;
  constexpr int QX = C::QX, QY = C::QY, QZ = C::QZ, QRS = C::QRS, QPS = C::QPS, QCS = C::QCS, NT = C::NT;
  __shared__ float lds[C::LDSF];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bb = wave / C::WPB;                 // q channel within this workgroup
  const int kk = (wave % C::WPB) * 64 + lane;   // tap
  const bool valid = kk < K3;
  const int kz = kk / (KS * KS), ky = (kk / KS) % KS, kx = kk % KS;
  const int b0 = blockIdx.y * NB;
  const int lane_off = valid ? bb * QCS + kz * QPS + ky * QRS + kx : 0;
  float acc[A];
#pragma unroll
  for (int a = 0; a < A; ++a) acc[a] = 0.f;

  const int tiles = d.tiles_x * d.tiles_y * d.tiles_z;
  const int first = blockIdx.x * d.items_per_wg;
  const int last = min(first + d.items_per_wg, d.items);
  const size_t pplane = (size_t)d.hp * d.wp, qplane = (size_t)d.hq * d.wq;
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    const int n = item / tiles, t = item % tiles;
    const int x0 = (t % d.tiles_x) * TX, y0 = ((t / d.tiles_x) % d.tiles_y) * TY, z0 = (t / (d.tiles_x * d.tiles_y)) * TZ;
    const int qx0 = x0 * S - d.pad, qy0 = y0 * S - d.pad, qz0 = z0 * S - d.pad;
    if (item != first) __syncthreads();
    for (int e = tid; e < NB * QZ * QY * QX; e += NT) {
      int xx = e % QX;
      int r = e / QX;
      int yy = r % QY;
      r /= QY;
      int zz = r % QZ;
      int c = r / QZ;
      int gx = qx0 + xx, gy = qy0 + yy, gz = qz0 + zz;
      float v = 0.f;
      if (b0 + c < d.bc && gx >= 0 && gx < d.wq && gy >= 0 && gy < d.hq && gz >= 0 && gz < d.dq)
        v = q[(((size_t)n * d.bc + b0 + c) * d.dq + gz) * qplane + (size_t)gy * d.wq + gx];
      lds[c * QCS + zz * QPS + yy * QRS + xx] = v;
    }
    __syncthreads();
    const float* pn = p + (size_t)n * A * d.dp * pplane;
#pragma unroll 1
    for (int iz = 0; iz < TZ; ++iz) {
#pragma unroll 1
      for (int iy = 0; iy < TY; ++iy) {
        const float* prow = pn + ((size_t)(z0 + iz) * d.hp + (y0 + iy)) * d.wp + x0;  // wave-uniform
        const float* qrow = lds + lane_off + iz * S * QPS + iy * S * QRS;
#pragma unroll
        for (int ix = 0; ix < TX; ix += 4) {
          float qv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) qv[j] = qrow[(ix + j) * S];
#pragma unroll
          for (int a = 0; a < A; ++a) {
            const float4 pv = *(const float4*)(prow + (size_t)a * d.dp * pplane + ix);  // scalar load
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
  float* slab = slabs + (size_t)blockIdx.x * d.jtotal;
  const int b = b0 + bb;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int o = d.out_mode == 0 ? (a * d.bc + b) * K3 + kk : (b * A + a) * K3 + (K3 - 1 - kk);
    slab[o] = acc[a];
  }
}

// dw[j] (+)= sum_g slabs[g][j], g ascending inside each of 4 interleaved slices, slices added 0..3
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw, int nslab,
                                                    int jtotal, int accumulate) {
  __shared__ float part[4][64];
  const int jl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  float s = 0.f;
  if (j < jtotal)
    for (int g = sl; g < nslab; g += 4) s += slabs[(size_t)g * jtotal + j];
  part[sl][jl] = s;
  __syncthreads();
  if (sl == 0 && j < jtotal) {
    float t = ((part[0][jl] + part[1][jl]) + part[2][jl]) + part[3][jl];
    dw[j] = accumulate ? dw[j] + t : t;
  }
}

static const int kMaxSlabs = 512;

extern "C" size_t nvf_wgrad_workspace(int batch, int a, int b, int k, int dp, int hp, int wp) {
  (void)batch; (void)dp; (void)hp; (void)wp;
  return (size_t)kMaxSlabs * a * b * k * k * k * sizeof(float);
}

template <class C>
static int launch_wgrad(const float* p, const float* q, float* dw, float* slabs, WgDims d, int accumulate,
                        hipStream_t s) {
  d.tiles_x = d.wp / C::TX;
  d.tiles_y = d.hp / C::TY;
  d.tiles_z = d.dp / C::TZ;
  d.items = d.batch * d.tiles_x * d.tiles_y * d.tiles_z;
  const int ygroups = (d.bc + C::NB - 1) / C::NB;
  int nslab = d.items < kMaxSlabs ? d.items : kMaxSlabs;
  d.items_per_wg = (d.items + nslab - 1) / nslab;
  nslab = (d.items + d.items_per_wg - 1) / d.items_per_wg;
  wgrad_tiled<C><<<dim3(nslab, ygroups), C::NT, 0, s>>>(p, q, slabs, d);
  wgrad_reduce<<<(d.jtotal + 63) / 64, 256, 0, s>>>(slabs, dw, nslab, d.jtotal, accumulate);
  return NVF_OK;
}

extern "C" int nvf_wgrad(const float* p, const float* q, float* dw, void* workspace, size_t workspace_bytes, int batch,
                         int a, int b, int k, int stride, int pad, int dp, int hp, int wp, int dq, int hq, int wq,
                         int out_mode, int accumulate, int naive, void* stream) {
  if (!p || !q || !dw || batch <= 0 || a <= 0 || b <= 0 || k <= 0 || stride <= 0) return NVF_EINVAL;
  if (dp <= 0 || hp <= 0 || wp <= 0 || dq <= 0 || hq <= 0 || wq <= 0) return NVF_EINVAL;
  if (out_mode != 0 && out_mode != 1) return NVF_EINVAL;
  WgDims d{};
  d.batch = batch; d.bc = b; d.dp = dp; d.hp = hp; d.wp = wp; d.dq = dq; d.hq = hq; d.wq = wq; d.pad = pad;
  d.out_mode = out_mode; d.jtotal = a * b * k * k * k;
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
  if (!naive && workspace) {
    if (workspace_bytes < nvf_wgrad_workspace(batch, a, b, k, dp, hp, wp)) return NVF_EWORKSPACE;
    float* slabs = (float*)workspace;
#define NVF_W(AA, KS, ST, WP, NB, TX, TY, TZ)                                                          \
  if (rc == 1 && a == AA && k == KS && stride == ST && wp == WP && hp % TY == 0 && dp % TZ == 0)        \
    rc = launch_wgrad<WCfg<AA, KS, ST, NB, TX, TY, TZ>>(p, q, dw, slabs, d, accumulate, s);
    if (b % 8 == 0) {
      NVF_W(8, 4, 1, 32, 8, 32, 8, 2)    // conv2 narrow: p = dY [8,32^3], q = X [8,35^3]
      NVF_W(8, 4, 1, 16, 8, 16, 8, 2)    // conv1 narrow
      NVF_W(8, 5, 2, 16, 4, 16, 4, 2)    // up2 narrow: p = X [8,16^3], q = dY [8,35^3]
      NVF_W(16, 5, 2, 8, 4, 8, 4, 2)     // up1 narrow: p = X [16,8^3], q = dY [8,19^3]
      NVF_W(16, 4, 1, 32, 8, 32, 8, 2)   // conv2 wide
      NVF_W(16, 4, 1, 16, 8, 16, 8, 2)   // conv1 wide
      NVF_W(16, 5, 2, 16, 4, 16, 4, 2)   // up2 wide
      NVF_W(32, 5, 2, 8, 4, 8, 4, 2)     // up1 wide
    }
    if (b == 1) {
      NVF_W(8, 3, 1, 32, 1, 32, 8, 4)    // conv2_cls: p = X [8,32^3], q = dlogit [1,32^3] (out_mode 1)
      NVF_W(8, 3, 1, 16, 1, 16, 8, 4)    // conv1_cls
      NVF_W(16, 3, 1, 32, 1, 32, 8, 4)   // wide heads
      NVF_W(16, 3, 1, 16, 1, 16, 8, 4)
    }
#undef NVF_W
  }
  if (rc == 1) {
    wgrad_naive<<<(d.jtotal + 63) / 64, 64, 0, s>>>(p, q, dw, a, k, stride, d, accumulate);
    rc = NVF_OK;
  }
  NVF_LAUNCH_CHECK();
  return rc;
}

// ---------------------------------------------------------------------------
// per-channel sums (bias gradients): out[c] (+)= sum_{n,s} x[n,c,s]
// stage 1: grid (c, G) partial sums over contiguous chunks; stage 2: fixed-order add.
// ---------------------------------------------------------------------------
static const int kSumChunks = 128;

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
