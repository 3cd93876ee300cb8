// Direct 3-D convolutions for the NVF decoder on gfx950.
//
// Reference call sites replaced: F.conv3d (utils/network.py:687, 741),
// F.conv_transpose3d (network.py:621) and their autograd backward-data passes.
//
// Design (MI355X): the decoder's channel counts are 1..32, so an implicit GEMM
// would leave most of an MFMA tile's N dimension empty (Cout = 8).  Instead
// each thread register-tiles VX consecutive x-outputs x all Cout channels, the
// input tile (with halo) is staged once per workgroup in LDS, and the weights
// are read through *scalar* loads (s_load_dwordx16) so every v_pk_fma_f32 takes
// its weight operand from SGPRs: no LDS or VGPR traffic for weights at all, and
// each LDS input value feeds K*Cout FMAs.  Per output element the accumulation
// order is fixed (ci, kz, ky, kx; one fmaf chain), independent of batch size,
// tile or grid, so encode at any batch size equals decode at batch 1 bit for bit.
#include "nvf_common.h"

struct ConvDims {
  int din, hin, win, dout, hout, wout, pad, act, tiles_x, tiles_y, tiles_z;
};

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, int cout, int cin, int k3, float* __restrict__ wf,
                                 float* __restrict__ wb) {
  int n = cout * cin * k3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int t = i % k3, ci = (i / k3) % cin, co = i / (k3 * cin);
    float v = w[i];
    if (wf) wf[(ci * k3 + t) * cout + co] = v;
    if (wb) wb[(co * k3 + (k3 - 1 - t)) * cin + ci] = v;
  }
}

__global__ void pack_convT_kernel(const float* __restrict__ w, int cin, int cout, int k3, float* __restrict__ wf,
                                  float* __restrict__ wb) {
  int n = cout * cin * k3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int t = i % k3, co = (i / k3) % cout, ci = i / (k3 * cout);
    float v = w[i];
    if (wf) wf[(ci * k3 + t) * cout + co] = v;
    if (wb) wb[(co * k3 + t) * cin + ci] = v;
  }
}

extern "C" int nvf_pack_conv_weight(const float* w, int cout, int cin, int k, float* w_fwd, float* w_bwd,
                                    void* stream) {
  if (!w || cout <= 0 || cin <= 0 || k <= 0) return NVF_EINVAL;
  int n = cout * cin * k * k * k;
  pack_conv_kernel<<<(n + 255) / 256, 256, 0, nvf_stream(stream)>>>(w, cout, cin, k * k * k, w_fwd, w_bwd);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_pack_convT_weight(const float* w, int cin, int cout, int k, float* w_fwd, float* w_bwd,
                                     void* stream) {
  if (!w || cout <= 0 || cin <= 0 || k <= 0) return NVF_EINVAL;
  int n = cout * cin * k * k * k;
  pack_convT_kernel<<<(n + 255) / 256, 256, 0, nvf_stream(stream)>>>(w, cin, cout, k * k * k, w_fwd, w_bwd);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// ---------------------------------------------------------------------------
// gather convolution, one thread per output element (any shape; also the
// bit-exact cross-check of the tiled kernel: same fmaf order)
// ---------------------------------------------------------------------------
__global__ void conv_gather_naive(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y,
                                  const float* __restrict__ addend, const float* __restrict__ mask, int cin,
                                  int cout, int k, int stride, ConvDims d, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int ox = idx % d.wout;
  long r = idx / d.wout;
  int oy = r % d.hout;
  r /= d.hout;
  int oz = r % d.dout;
  r /= d.dout;
  int co = r % cout;
  long b = r / cout;
  const float* xb = x + b * cin * (long)d.din * d.hin * d.win;
  int k3 = k * k * k;
  float acc = 0.f;
  for (int ci = 0; ci < cin; ++ci)
    for (int kz = 0; kz < k; ++kz) {
      int iz = oz * stride - d.pad + kz;
      for (int ky = 0; ky < k; ++ky) {
        int iy = oy * stride - d.pad + ky;
        for (int kx = 0; kx < k; ++kx) {
          int ix = ox * stride - d.pad + kx;
          float xv = 0.f;
          if (iz >= 0 && iz < d.din && iy >= 0 && iy < d.hin && ix >= 0 && ix < d.win)
            xv = xb[((long)ci * d.din + iz) * d.hin * d.win + iy * d.win + ix];
          acc = fmaf(xv, w[(ci * k3 + (kz * k + ky) * k + kx) * cout + co], acc);
        }
      }
    }
  float o = nvf_act(acc + (bias ? bias[co] : 0.f), d.act);
  if (addend) o += addend[idx];
  if (mask) o = mask[idx] > 0.f ? o : 0.f;
  y[idx] = o;
}

// ---------------------------------------------------------------------------
// gather convolution, LDS-tiled, register-tiled, scalar-register weights
// ---------------------------------------------------------------------------
template <int CIN_, int COUT_, int KS_, int S_, int VX_, int NCX_, int TY_, int TZ_, int CC_, int KYU_ = 0, int COG_ = 0>
struct GCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, KS = KS_, S = S_, VX = VX_, NCX = NCX_, TY = TY_, TZ = TZ_, CC = CC_;
  // output channels per workgroup: small layers split Cout over workgroups (grid.x carries the group) to get
  // enough threads in flight; the input tile is then staged once per group (L2 hits)
  static constexpr int COG = COG_ > 0 ? COG_ : COUT_;
  static constexpr int NCOG = COUT_ / COG;
  static constexpr int LV = (VX_ * S_) % 4 == 0 ? 4 : ((VX_ * S_) % 2 == 0 ? 2 : 1);  // LDS read width (floats)
  // ky-loop unroll: the weights of one (c,kz,ky) step are KS*COUT scalar registers; unrolling ky multiplies
  // that, and past ~100 SGPRs the compiler spills them to VGPR lanes (v_readlane per FMA operand).
  static constexpr int KYU = KYU_ > 0 ? KYU_ : (KS * COG <= 32 ? KS : 1);
  static constexpr int TX = NCX * VX;
  static constexpr int NIN = (VX - 1) * S + KS;   // inputs one thread reads per (c,kz,ky)
  static constexpr int NIN4 = (NIN + LV - 1) / LV * LV;  // rounded to whole LDS reads
  static constexpr int IX = (TX - 1) * S + KS;    // staged tile extents
  static constexpr int IY = (TY - 1) * S + KS;
  static constexpr int IZ = (TZ - 1) * S + KS;
  static constexpr int RS = (NCX - 1) * VX * S + NIN4;  // LDS row stride (>= IX, 16-B aligned reads)
  static constexpr int NACT = NCX * TY * TZ;
  static constexpr int LDSF = CC * IZ * IY * RS;
  // threads beyond NACT only help staging the tile (keeps enough global loads in flight for small tiles)
  static constexpr int NT = ((NACT + 63) / 64 * 64 < 256 && LDSF >= 2048) ? 256 : (NACT + 63) / 64 * 64;
  static_assert(COUT_ % COG == 0, "channel groups");
  static_assert(RS >= IX, "row stride");
  static_assert(CIN % CC == 0, "channel chunk");
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

// One (ci,kz,ky) step of a thread's register tile: acc[co][v] += in[v*S+kx] * w[kx][co], kx ascending (the
// canonical per-output order).  For even COG the FMAs are written on channel PAIRS: the weight pair is an aligned
// SGPR pair straight out of s_load_dwordx16 and the input is one half of a VGPR pair picked by op_sel, so
// v_pk_fma_f32 needs no operand shuffling (packing over x pairs instead costs an s_mov/v_mov per odd operand).
typedef float nvf_f2 __attribute__((ext_vector_type(2)));
template <int COG, int VX>
struct NvfAcc {
  static constexpr bool PAIR = COG % 2 == 0;
  nvf_f2 p[PAIR ? COG / 2 : 1][VX];
  float s[PAIR ? 1 : COG][VX];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < (PAIR ? COG / 2 : 1); ++i)
#pragma unroll
      for (int j = 0; j < VX; ++j) p[i][j] = (nvf_f2){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (PAIR ? 1 : COG); ++i)
#pragma unroll
      for (int j = 0; j < VX; ++j) s[i][j] = 0.f;
  }
  __device__ __forceinline__ float get(int co, int v) const {
    if constexpr (PAIR) return (co & 1) ? p[co / 2][v].y : p[co / 2][v].x;
    else return s[co][v];
  }
};

template <int COG, int VX, int KS, int S, int COUT, int NIN>
__device__ __forceinline__ void nvf_mac_row(const float (&in)[NIN], const float* __restrict__ wr,
                                            NvfAcc<COG, VX>& acc) {
  if constexpr (COG % 2 == 0) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int co = 0; co < COG; co += 2) {
        const nvf_f2 wv = {wr[kx * COUT + co], wr[kx * COUT + co + 1]};
#pragma unroll
        for (int v = 0; v < VX; ++v) {
          const float xv = in[v * S + kx];
          acc.p[co / 2][v] = __builtin_elementwise_fma((nvf_f2){xv, xv}, wv, acc.p[co / 2][v]);
        }
      }
  } else {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int co = 0; co < COG; ++co) {
        const float wv = wr[kx * COUT + co];
#pragma unroll
        for (int v = 0; v < VX; ++v) acc.s[co][v] = fmaf(in[v * S + kx], wv, acc.s[co][v]);
      }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) void conv_gather_tiled(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           const float* __restrict__ addend,
                                                           const float* __restrict__ mask, ConvDims d) {
  constexpr int CIN = C::CIN, COUT = C::COUT, KS = C::KS, S = C::S, VX = C::VX, NCX = C::NCX, TY = C::TY, TZ = C::TZ,
                CC = C::CC;
  constexpr int RS = C::RS, IY = C::IY, IZ = C::IZ, IX = C::IX, NT = C::NT, NIN4 = C::NIN4, COG = C::COG, LV = C::LV;
  __shared__ __attribute__((aligned(16))) float lds[C::LDSF];
  const int ntile = d.tiles_x * d.tiles_y * d.tiles_z;
  const int co0 = (blockIdx.x % C::NCOG) * COG;            // wave-uniform
  const int wg = blockIdx.x / C::NCOG;
  const int tile = wg % ntile, b = wg / ntile;
  const int tx_i = tile % d.tiles_x, ty_i = (tile / d.tiles_x) % d.tiles_y, tz_i = tile / (d.tiles_x * d.tiles_y);
  const int ox0 = tx_i * C::TX, oy0 = ty_i * TY, oz0 = tz_i * TZ;
  const int tid = threadIdx.x;
  const int cx = tid % NCX, ty = (tid / NCX) % TY, tz = tid / (NCX * TY);
  const bool active = tid < C::NACT;
  NvfAcc<COG, VX> acc;
  acc.zero();
  const float* xb = x + (size_t)b * CIN * d.din * d.hin * d.win;
  const int gz0 = oz0 * S - d.pad, gy0 = oy0 * S - d.pad, gx0 = ox0 * S - d.pad;
  const int plane = d.hin * d.win;
#pragma unroll 1
  for (int c0 = 0; c0 < CIN; c0 += CC) {
    if (c0) __syncthreads();
    nvf_stage_rows<NT, CC * IZ * IY, RS, RS, 8>(
        xb, lds, tid,
        [&](int r, int xx, bool& ok) -> size_t {
          const int yy = r % IY, t = r / IY, zz = t % IZ, c = t / IZ;
          const int gx = gx0 + xx, gy = gy0 + yy, gz = gz0 + zz;
          ok = xx < IX && gx >= 0 && gx < d.win && gy >= 0 && gy < d.hin && gz >= 0 && gz < d.din;
          return ((size_t)(c0 + c) * d.din + gz) * plane + (size_t)gy * d.win + gx;
        },
        [&](int r, int xx) { return r * RS + xx; });
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int c = 0; c < CC; ++c) {
#pragma unroll 1
        for (int kz = 0; kz < KS; ++kz) {
#pragma unroll C::KYU
          for (int ky = 0; ky < KS; ++ky) {
            const float* rowp = lds + ((c * IZ + tz * S + kz) * IY + ty * S + ky) * RS + cx * VX * S;
            float in[NIN4];
            nvf_lds_row<NIN4, LV>(rowp, in);
            const float* wr = w + (size_t)((((c0 + c) * KS + kz) * KS + ky) * KS) * COUT + co0;  // wave-uniform
            nvf_mac_row<COG, VX, KS, S, COUT>(in, wr, acc);
          }
        }
      }
    }
  }
  if (!active) return;
  const int oz = oz0 + tz, oy = oy0 + ty;
  if (oz >= d.dout || oy >= d.hout) return;
#pragma unroll
  for (int co = 0; co < COG; ++co) {
    const float bv = bias ? bias[co0 + co] : 0.f;
    const size_t base = (((size_t)b * COUT + co0 + co) * d.dout + oz) * d.hout * d.wout + (size_t)oy * d.wout;
#pragma unroll
    for (int v = 0; v < VX; ++v) {
      const int ox = ox0 + cx * VX + v;
      if (ox < d.wout) {
        float o = nvf_act(acc.get(co, v) + bv, d.act);
        if (addend) o += addend[base + ox];
        if (mask) o = mask[base + ox] > 0.f ? o : 0.f;
        y[base + ox] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Same tile / register / accumulation scheme, but the input tile is filled by LDS-DMA
// (global_load_lds_dword: HBM/L2 -> LDS with no VGPR destination) into TWO tile buffers:
// the loads of channel chunk c+1 are in flight while chunk c is being multiplied, so a
// workgroup that is alone on its CU (batch 16: 1-2 waves per SIMD) no longer idles through
// every staging phase.  One wave-instruction moves whole rows (lane = x): the row base is
// scalar (SALU), the only per-lane quantity is the constant x offset, and halo / padding
// elements are simply never written -- both buffers are zeroed once, and the out-of-range
// pattern of a tile is the same for every channel chunk.  The DMA is issued from inline
// asm, so hipcc does not count it: the explicit vmcnt(0) before the barrier is the wait.
// Results are bit-identical to conv_gather_tiled (same LDS image, same fmaf chain).
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void conv_glds_issue(const float* __restrict__ xc, unsigned lds_byte, int wave, int lane,
                                                int gz0, int gy0, int gx0, const ConvDims& d) {
  constexpr int RS = C::RS, IX = C::IX, IY = C::IY, IZ = C::IZ, R = C::CC * C::IZ * C::IY, NW = C::NT / 64;
  constexpr int RPI = RS <= 32 ? 64 / RS : 1;                    // LDS rows are contiguous: several per instruction
  const int sub = RPI > 1 ? lane / RS : 0;
  const int xx = lane - sub * RS;
  const int gx = gx0 + xx;
  const bool xok = sub < RPI && xx < IX && gx >= 0 && gx < d.win;
#pragma unroll 1
  for (int r0 = wave * RPI; r0 < R; r0 += NW * RPI) {            // wave-uniform
    if constexpr (RPI == 1) {
      const int yy = r0 % IY, t = r0 / IY, zz = t % IZ, c = t / IZ;
      const int gy = gy0 + yy, gz = gz0 + zz;
      if (gy < 0 || gy >= d.hin || gz < 0 || gz >= d.din) continue;
      const float* rb = xc + ((c * d.din + gz) * d.hin + gy) * d.win;   // < 2^31 elements per batch item
      if (xok) nvf_glds_row(rb, (unsigned)gx * 4u, lds_byte + (unsigned)r0 * RS * 4u);
    } else {
      int off = 0;
      bool ok = false;
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const int r = r0 + u;
        const int yy = r % IY, t = r / IY, zz = t % IZ, c = t / IZ;
        const int gy = gy0 + yy, gz = gz0 + zz;
        const bool rok = r < R && gy >= 0 && gy < d.hin && gz >= 0 && gz < d.din;
        const int o = ((c * d.din + gz) * d.hin + gy) * d.win;
        if (sub == u) { off = o; ok = rok; }
      }
      if (xok && ok) nvf_glds_lane(xc + off + gx, lds_byte + (unsigned)r0 * RS * 4u);
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) void conv_gather_glds(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          const float* __restrict__ addend,
                                                          const float* __restrict__ mask, ConvDims d) {
  constexpr int CIN = C::CIN, COUT = C::COUT, KS = C::KS, S = C::S, VX = C::VX, NCX = C::NCX, TY = C::TY, TZ = C::TZ,
                CC = C::CC;
  constexpr int RS = C::RS, IY = C::IY, IZ = C::IZ, NT = C::NT, NIN4 = C::NIN4, COG = C::COG, LV = C::LV;
  constexpr int LDSF = (C::LDSF + 3) / 4 * 4;
  static_assert(2 * LDSF * 4 <= 160 * 1024, "two tile buffers");
  __shared__ __attribute__((aligned(16))) float lds[2 * LDSF];
  const int ntile = d.tiles_x * d.tiles_y * d.tiles_z;
  const int co0 = (blockIdx.x % C::NCOG) * COG;            // wave-uniform
  const int wg = blockIdx.x / C::NCOG;
  const int tile = wg % ntile, b = wg / ntile;
  const int tx_i = tile % d.tiles_x, ty_i = (tile / d.tiles_x) % d.tiles_y, tz_i = tile / (d.tiles_x * d.tiles_y);
  const int ox0 = tx_i * C::TX, oy0 = ty_i * TY, oz0 = tz_i * TZ;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int cx = tid % NCX, ty = (tid / NCX) % TY, tz = tid / (NCX * TY);
  const bool active = tid < C::NACT;
  NvfAcc<COG, VX> acc;
  acc.zero();
  const size_t vol = (size_t)d.din * d.hin * d.win;
  const float* xb = x + (size_t)b * CIN * vol;
  const int gz0 = oz0 * S - d.pad, gy0 = oy0 * S - d.pad, gx0 = ox0 * S - d.pad;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  for (int i = tid * 4; i < 2 * LDSF; i += NT * 4) *(float4*)(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  conv_glds_issue<C>(xb, lds0, wave, lane, gz0, gy0, gx0, d);
#pragma unroll 1
  for (int c0 = 0; c0 < CIN; c0 += CC) {
    const int buf = (c0 / CC) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA rows of chunk c0 have landed
    __syncthreads();                                        // ... everyone's; and chunk c0-CC is no longer being read
    if (c0 + CC < CIN)
      conv_glds_issue<C>(xb + (size_t)(c0 + CC) * vol, lds0 + (unsigned)(buf ^ 1) * LDSF * 4u, wave, lane, gz0, gy0,
                         gx0, d);
    if (active) {
      const float* tile_lds = lds + buf * LDSF;
#pragma unroll 1
      for (int c = 0; c < CC; ++c) {
#pragma unroll 1
        for (int kz = 0; kz < KS; ++kz) {
#pragma unroll C::KYU
          for (int ky = 0; ky < KS; ++ky) {
            const float* rowp = tile_lds + ((c * IZ + tz * S + kz) * IY + ty * S + ky) * RS + cx * VX * S;
            float in[NIN4];
            nvf_lds_row<NIN4, LV>(rowp, in);
            const float* wr = w + (size_t)((((c0 + c) * KS + kz) * KS + ky) * KS) * COUT + co0;  // wave-uniform
            nvf_mac_row<COG, VX, KS, S, COUT>(in, wr, acc);
          }
        }
      }
    }
  }
  if (!active) return;
  const int oz = oz0 + tz, oy = oy0 + ty;
  if (oz >= d.dout || oy >= d.hout) return;
#pragma unroll
  for (int co = 0; co < COG; ++co) {
    const float bv = bias ? bias[co0 + co] : 0.f;
    const size_t base = (((size_t)b * COUT + co0 + co) * d.dout + oz) * d.hout * d.wout + (size_t)oy * d.wout;
#pragma unroll
    for (int v = 0; v < VX; ++v) {
      const int ox = ox0 + cx * VX + v;
      if (ox < d.wout) {
        float o = nvf_act(acc.get(co, v) + bv, d.act);
        if (addend) o += addend[base + ox];
        if (mask) o = mask[base + ox] > 0.f ? o : 0.f;
        y[base + ox] = o;
      }
    }
  }
}

template <class C>
static int launch_gather_glds(const float* x, const float* w, const float* bias, float* y, const float* addend,
                              const float* mask, int batch, ConvDims d, hipStream_t s) {
  d.tiles_x = (d.wout + C::TX - 1) / C::TX;
  d.tiles_y = (d.hout + C::TY - 1) / C::TY;
  d.tiles_z = (d.dout + C::TZ - 1) / C::TZ;
  dim3 grid((unsigned)(d.tiles_x * d.tiles_y * d.tiles_z) * batch * C::NCOG);
  conv_gather_glds<C><<<grid, C::NT, 0, s>>>(x, w, bias, y, addend, mask, d);
  return NVF_OK;
}

template <class C>
static int launch_gather(const float* x, const float* w, const float* bias, float* y, const float* addend,
                         const float* mask, int batch, ConvDims d, hipStream_t s) {
  d.tiles_x = (d.wout + C::TX - 1) / C::TX;
  d.tiles_y = (d.hout + C::TY - 1) / C::TY;
  d.tiles_z = (d.dout + C::TZ - 1) / C::TZ;
  dim3 grid((unsigned)(d.tiles_x * d.tiles_y * d.tiles_z) * batch * C::NCOG);
  conv_gather_tiled<C><<<grid, C::NT, 0, s>>>(x, w, bias, y, addend, mask, d);
  return NVF_OK;
}

extern "C" int nvf_conv3d_gather(const float* x, const float* w, const float* bias, float* y, const float* addend,
                                 const float* mask, int batch, int cin, int cout, int k, int stride, int pad, int din,
                                 int hin, int win, int dout, int hout, int wout, int act, int variant, void* stream) {
  if (!x || !w || !y || batch <= 0 || cin <= 0 || cout <= 0 || k <= 0 || stride <= 0) return NVF_EINVAL;
  if (din <= 0 || hin <= 0 || win <= 0 || dout <= 0 || hout <= 0 || wout <= 0) return NVF_EINVAL;
  ConvDims d{din, hin, win, dout, hout, wout, pad, act, 0, 0, 0};
  hipStream_t s = nvf_stream(stream);
  int rc = 1;  // 1 = not dispatched yet
  // the one-channel classifier heads (3^3, padding 1, cubes) have their own kernels (heads.hip): same fmaf order
  if (variant == 0 && k == 3 && stride == 1 && pad == 1 && din == hin && hin == win && dout == din && hout == din &&
      wout == din) {
    if (cout == 1 && cin > 1) rc = nvf_head_fwd_launch(x, w, bias, y, addend, mask, batch, cin, win, act, s);
    else if (cin == 1 && cout > 1) rc = nvf_head_bwd_data_launch(x, w, bias, y, addend, mask, batch, cout, win, act, s);
    if (rc == 0) {
      NVF_LAUNCH_CHECK();
      return NVF_OK;
    }
  }
  // variant 0: the tuned configuration; 1: one-thread-per-output kernel; >= 2: alternatives kept for tuning runs.
  // Small batches cannot fill 256 CUs with whole-Cout tiles, so they take the Cout-split (COG) instantiations.
  if (variant == 0 && batch <= 64) {
    if (cin == 8 && cout == 8 && k == 5 && stride == 2 && wout >= 9 && wout <= 16) variant = 30;  // up2 backward-data
    if (cin == 8 && cout == 16 && k == 5 && stride == 2 && wout >= 5 && wout <= 8) variant = 9;   // up1 backward-data
    if (cin == 16 && cout == 8 && k == 5 && stride == 2 && wout >= 3 && wout <= 4) variant = 9;   // conv0 backward-data
    if (cin == 8 && cout == 8 && k == 4 && stride == 1 && wout >= 9 && wout <= 40) variant = 30;  // conv1, conv2: both
    if (cin == 1 && cout == 16 && k == 3 && wout <= 8) variant = 1;                               // conv0_cls backward-data
  } else if (variant == 0) {
    if (cin == 8 && cout == 8 && k == 4 && stride == 1 && wout >= 17 && wout <= 40) variant = 31; // conv1 bwd, conv2: both
    if (cin == 8 && cout == 16 && k == 5 && stride == 2 && wout >= 5 && wout <= 8) variant = 31;  // up1 backward-data
    if (cin == 16 && cout == 16 && k == 4 && stride == 1 && wout >= 21 && wout <= 36) variant = 31;  // wide conv2: both
  }
  // a tuned id (>= 9) without an instantiation for this shape falls back to the variant-0 table
  for (int pass = 0; pass < 2 && rc == 1 && variant != 1; ++pass, variant = 0) {
#define NVF_GC(VAR, CI, CO, KS, ST, WLO, WHI, VX, NCX, TY, TZ, CC, KYU, COG)                                        \
  if (rc == 1 && variant == VAR && cin == CI && cout == CO && k == KS && stride == ST && wout >= WLO && wout <= WHI) \
    rc = launch_gather<GCfg<CI, CO, KS, ST, VX, NCX, TY, TZ, CC, KYU, COG>>(x, w, bias, y, addend, mask, batch, d, s);
#define NVF_G(VAR, CI, CO, KS, ST, WLO, WHI, VX, NCX, TY, TZ, CC, KYU) \
  NVF_GC(VAR, CI, CO, KS, ST, WLO, WHI, VX, NCX, TY, TZ, CC, KYU, 0)
#define NVF_GD(VAR, CI, CO, KS, ST, WLO, WHI, VX, NCX, TY, TZ, CC, KYU, COG)                                        \
  if (rc == 1 && variant == VAR && cin == CI && cout == CO && k == KS && stride == ST && wout >= WLO && wout <= WHI) \
    rc = launch_gather_glds<GCfg<CI, CO, KS, ST, VX, NCX, TY, TZ, CC, KYU, COG>>(x, w, bias, y, addend, mask, batch, \
                                                                                 d, s);
    // LDS-DMA double-buffered kernels: the tuned choice for batch <= 64 (30) and above (31)
    NVF_GD(30, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 0)   // conv2 forward
    NVF_GD(30, 8, 8, 4, 1, 33, 40, 4, 9, 5, 4, 2, 0, 0)   // conv2 backward-data
    NVF_GD(30, 8, 8, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 4)    // conv1 forward
    NVF_GD(30, 8, 8, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 4)   // conv1 backward-data
    NVF_GD(30, 8, 8, 5, 2, 9, 16, 2, 8, 4, 4, 2, 0, 4)    // up2 backward-data
    NVF_GD(31, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 0)   // conv2 forward
    NVF_GD(31, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0, 0)   // conv2 backward-data
    NVF_GD(31, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 0)  // conv1 backward-data
    NVF_GD(31, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 0)    // up1 backward-data
    NVF_GD(31, 16, 16, 4, 1, 33, 36, 4, 9, 7, 4, 2, 0, 0) // wide conv2 backward-data
    NVF_GD(31, 16, 16, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 0) // wide conv2 forward
    NVF_GD(20, 16, 16, 4, 1, 33, 36, 4, 9, 7, 4, 2, 0, 0)
    NVF_GD(20, 16, 16, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 0)
    // LDS-DMA candidates kept for tuning runs
    NVF_GD(20, 8, 8, 4, 1, 21, 32, 4, 8, 8, 8, 2, 0, 0)
    NVF_GD(21, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 0)
    NVF_GD(22, 8, 8, 4, 1, 21, 32, 4, 8, 8, 8, 2, 0, 4)
    NVF_GD(23, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 4)
    NVF_GD(24, 8, 8, 4, 1, 21, 32, 4, 8, 4, 4, 2, 0, 0)
    NVF_GD(25, 8, 8, 4, 1, 21, 32, 4, 8, 8, 2, 2, 0, 0)
    NVF_GD(20, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0, 0)
    NVF_GD(21, 8, 8, 4, 1, 33, 40, 4, 9, 7, 8, 2, 0, 0)
    NVF_GD(22, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0, 4)
    NVF_GD(23, 8, 8, 4, 1, 33, 40, 4, 9, 7, 8, 2, 0, 4)
    NVF_GD(24, 8, 8, 4, 1, 33, 40, 4, 9, 5, 4, 2, 0, 0)
    NVF_GD(25, 8, 8, 4, 1, 33, 40, 4, 9, 7, 2, 2, 0, 0)
    NVF_GD(20, 8, 8, 4, 1, 9, 16, 4, 4, 16, 4, 2, 0, 0)
    NVF_GD(21, 8, 8, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 0)
    NVF_GD(22, 8, 8, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 4)
    NVF_GD(23, 8, 8, 4, 1, 9, 16, 4, 4, 16, 4, 2, 0, 4)
    NVF_GD(24, 8, 8, 4, 1, 9, 16, 4, 4, 8, 2, 2, 0, 4)
    NVF_GD(25, 8, 8, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 2)
    NVF_GD(20, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 0)
    NVF_GD(21, 8, 8, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 0)
    NVF_GD(22, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 4)
    NVF_GD(23, 8, 8, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 4)
    NVF_GD(24, 8, 8, 4, 1, 17, 20, 4, 5, 10, 2, 2, 0, 4)
    NVF_GD(25, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 2)
    NVF_GD(20, 8, 8, 5, 2, 9, 16, 4, 4, 16, 4, 1, 0, 0)
    NVF_GD(21, 8, 8, 5, 2, 9, 16, 2, 8, 8, 4, 2, 0, 0)
    NVF_GD(22, 8, 8, 5, 2, 9, 16, 2, 8, 8, 2, 2, 0, 4)
    NVF_GD(23, 8, 8, 5, 2, 9, 16, 2, 8, 4, 4, 2, 0, 4)
    NVF_GD(24, 8, 8, 5, 2, 9, 16, 4, 4, 8, 2, 2, 0, 4)
    NVF_GD(25, 8, 8, 5, 2, 9, 16, 2, 8, 8, 2, 2, 0, 2)
    NVF_GD(20, 8, 16, 5, 2, 5, 8, 2, 4, 8, 8, 2, 0, 0)
    NVF_GD(21, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 0)
    NVF_GD(22, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 4)
    NVF_GD(23, 8, 16, 5, 2, 5, 8, 2, 4, 4, 4, 4, 0, 2)
    NVF_GD(24, 8, 16, 5, 2, 5, 8, 2, 4, 8, 2, 2, 0, 4)
    NVF_GD(25, 8, 16, 5, 2, 5, 8, 2, 4, 4, 4, 2, 0, 4)
    // ---- narrow decoder (chanstr 8,16,8,8)
    NVF_G(0, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0)    // conv2 backward-data (35^3)
    NVF_G(0, 8, 8, 4, 1, 21, 32, 4, 8, 8, 8, 2, 0)    // conv2 forward (32^3)
    NVF_G(0, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0)   // conv1 backward-data (19^3)
    NVF_G(0, 8, 8, 4, 1, 9, 16, 4, 4, 16, 4, 2, 0)    // conv1 forward (16^3)
    NVF_G(0, 8, 1, 3, 1, 17, 32, 8, 4, 16, 4, 4, 0)   // conv2_cls forward (32^3)
    NVF_G(0, 8, 1, 3, 1, 9, 16, 4, 4, 16, 4, 4, 0)    // conv1_cls forward (16^3)
    NVF_G(0, 16, 1, 3, 1, 5, 8, 4, 2, 8, 8, 4, 0)     // conv0_cls forward (8^3)
    NVF_G(0, 1, 8, 3, 1, 17, 32, 8, 4, 16, 4, 1, 0)   // conv2_cls backward-data
    NVF_G(0, 1, 8, 3, 1, 9, 16, 4, 4, 16, 4, 1, 0)    // conv1_cls backward-data
    NVF_G(0, 1, 16, 3, 1, 5, 8, 4, 2, 8, 8, 1, 0)     // conv0_cls backward-data
    NVF_G(0, 8, 8, 5, 2, 9, 16, 4, 4, 16, 4, 1, 0)    // up2 backward-data (35^3 -> 16^3)
    NVF_G(0, 8, 16, 5, 2, 5, 8, 2, 4, 8, 8, 2, 0)     // up1 backward-data (19^3 -> 8^3)
    NVF_GC(0, 16, 8, 5, 2, 3, 4, 2, 2, 4, 4, 4, 0, 2) // conv0 backward-data (8^3 -> 4^3), Cout split 4 ways
    NVF_GC(0, 8, 3, 5, 2, 2, 2, 2, 1, 2, 2, 8, 0, 1)  // up0 backward-data (4^3 -> 2^3), ch = 3
    NVF_GC(0, 8, 8, 5, 2, 2, 2, 2, 1, 2, 2, 8, 0, 1)  // up0 backward-data, ch = 8 (narrow c0 with wide latent)
    // tuning alternatives
    NVF_G(2, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0)
    NVF_G(3, 8, 8, 4, 1, 21, 32, 8, 4, 8, 8, 2, 0)
    NVF_G(4, 8, 8, 4, 1, 21, 32, 4, 8, 8, 8, 2, 0)
    NVF_G(5, 8, 8, 4, 1, 21, 32, 8, 4, 16, 8, 1, 0)
    NVF_G(6, 8, 8, 4, 1, 21, 32, 8, 4, 16, 4, 2, 2)
    NVF_G(7, 8, 8, 4, 1, 21, 32, 8, 4, 16, 4, 2, 1)
    NVF_G(8, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 2)
    NVF_G(2, 8, 8, 4, 1, 33, 40, 4, 10, 6, 4, 2, 0)
    NVF_G(3, 8, 8, 4, 1, 33, 40, 8, 5, 12, 4, 2, 2)
    NVF_G(4, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0)
    NVF_G(5, 8, 8, 4, 1, 33, 40, 4, 9, 7, 8, 2, 0)
    NVF_G(2, 8, 8, 5, 2, 9, 16, 4, 4, 16, 4, 1, 5)
    NVF_G(3, 8, 8, 5, 2, 9, 16, 4, 4, 8, 4, 2, 0)
    NVF_G(4, 8, 8, 5, 2, 9, 16, 2, 8, 8, 4, 2, 0)
    NVF_G(2, 8, 16, 5, 2, 5, 8, 2, 4, 8, 8, 2, 0)
    NVF_G(3, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 4, 0)
    NVF_G(4, 8, 16, 5, 2, 5, 8, 4, 2, 8, 4, 4, 0)
    NVF_GC(9, 8, 8, 4, 1, 21, 32, 4, 8, 8, 8, 2, 0, 4)
    NVF_GC(10, 8, 8, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0, 4)
    NVF_GC(11, 8, 8, 4, 1, 21, 32, 8, 4, 8, 8, 2, 0, 4)
    NVF_GC(9, 8, 8, 4, 1, 33, 40, 4, 9, 7, 4, 2, 0, 4)
    NVF_GC(10, 8, 8, 4, 1, 33, 40, 4, 9, 7, 8, 2, 0, 4)
    NVF_GC(9, 8, 8, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 4)
    NVF_GC(10, 8, 8, 4, 1, 9, 16, 4, 4, 16, 4, 2, 0, 4)
    NVF_GC(9, 8, 8, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 4)
    NVF_GC(10, 8, 8, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 4)
    NVF_GC(7, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 4)
    NVF_GC(8, 8, 16, 5, 2, 5, 8, 2, 4, 8, 2, 2, 0, 4)
    NVF_GC(9, 8, 16, 5, 2, 5, 8, 2, 4, 4, 4, 4, 0, 2)
    NVF_GC(7, 8, 8, 5, 2, 9, 16, 2, 8, 8, 2, 2, 0, 4)
    NVF_GC(8, 8, 8, 5, 2, 9, 16, 4, 4, 8, 2, 2, 0, 4)
    NVF_GC(9, 8, 8, 5, 2, 9, 16, 2, 8, 4, 4, 2, 0, 4)
    NVF_GC(7, 16, 8, 5, 2, 3, 4, 2, 2, 4, 2, 4, 0, 1)
    NVF_GC(8, 16, 8, 5, 2, 3, 4, 2, 2, 4, 4, 8, 0, 1)
    NVF_GC(9, 16, 8, 5, 2, 3, 4, 2, 2, 2, 2, 4, 0, 1)
    NVF_GC(5, 8, 16, 5, 2, 5, 8, 2, 4, 8, 8, 2, 0, 4)
    NVF_GC(6, 8, 16, 5, 2, 5, 8, 2, 4, 8, 4, 4, 0, 4)
    NVF_GC(5, 8, 8, 5, 2, 9, 16, 2, 8, 8, 4, 2, 0, 4)
    NVF_GC(6, 8, 8, 5, 2, 9, 16, 4, 4, 8, 4, 2, 0, 4)
    NVF_GC(5, 16, 8, 5, 2, 3, 4, 2, 2, 4, 4, 4, 0, 1)
    NVF_GC(6, 16, 8, 5, 2, 3, 4, 2, 2, 4, 4, 8, 0, 4)
    // ---- wide decoder (chanstr 16,32,16,16)
    NVF_G(0, 16, 16, 4, 1, 33, 36, 4, 9, 7, 4, 2, 0)  // conv2 backward-data
    NVF_G(0, 16, 16, 4, 1, 21, 32, 4, 8, 8, 4, 2, 0)  // conv2 forward
    // (tile shapes of the small wide layers: tools/wide_sweep.py, batch 16 -- the first choices left most CUs idle)
    NVF_G(0, 16, 16, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0)  // conv1 backward-data (150 vs 158 us)
    NVF_G(0, 16, 16, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0)   // conv1 forward (138 vs 155 us)
    NVF_G(0, 16, 1, 3, 1, 17, 32, 8, 4, 16, 4, 4, 0)  // conv2_cls forward
    NVF_G(0, 16, 1, 3, 1, 9, 16, 4, 4, 16, 4, 4, 0)   // conv1_cls forward
    NVF_G(0, 32, 1, 3, 1, 5, 8, 4, 2, 8, 8, 4, 0)     // conv0_cls forward
    NVF_G(0, 1, 16, 3, 1, 17, 32, 4, 8, 8, 4, 1, 0)   // conv2_cls backward-data
    NVF_G(0, 1, 16, 3, 1, 9, 16, 4, 4, 16, 4, 1, 0)   // conv1_cls backward-data
    NVF_G(0, 1, 32, 3, 1, 5, 8, 4, 2, 8, 8, 1, 0)     // conv0_cls backward-data
    NVF_GC(0, 16, 16, 5, 2, 9, 16, 2, 8, 8, 2, 2, 0, 8) // up2 backward-data (233 vs 498 us)
    NVF_GC(0, 16, 32, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 4)  // up1 backward-data (164 vs 376 us)
    NVF_GC(0, 32, 16, 5, 2, 3, 4, 2, 2, 4, 2, 4, 0, 2)  // conv0 backward-data (136 vs 166 us)
    NVF_GC(0, 16, 8, 5, 2, 2, 2, 2, 1, 2, 2, 8, 0, 1) // up0 backward-data (ch = 8)
    // wide-decoder tuning candidates (tools/wide_sweep.py)
    NVF_GC(40, 16, 16, 5, 2, 9, 16, 2, 8, 8, 4, 2, 0, 8)
    NVF_GC(41, 16, 16, 5, 2, 9, 16, 4, 4, 8, 4, 2, 0, 8)
    NVF_GC(42, 16, 16, 5, 2, 9, 16, 2, 8, 8, 2, 2, 0, 8)
    NVF_GC(43, 16, 16, 5, 2, 9, 16, 2, 8, 8, 4, 2, 0, 4)
    NVF_GC(40, 16, 32, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 8)
    NVF_GC(41, 16, 32, 5, 2, 5, 8, 2, 4, 8, 2, 2, 0, 8)
    NVF_GC(42, 16, 32, 5, 2, 5, 8, 2, 4, 4, 4, 4, 0, 8)
    NVF_GC(43, 16, 32, 5, 2, 5, 8, 2, 4, 8, 4, 2, 0, 4)
    NVF_GC(40, 32, 16, 5, 2, 3, 4, 2, 2, 4, 4, 8, 0, 1)
    NVF_GC(41, 32, 16, 5, 2, 3, 4, 2, 2, 4, 2, 4, 0, 2)
    NVF_GC(42, 32, 16, 5, 2, 3, 4, 2, 2, 4, 4, 8, 0, 2)
    NVF_GC(40, 16, 16, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 8)
    NVF_GC(41, 16, 16, 4, 1, 9, 16, 4, 4, 16, 2, 2, 0, 8)
    NVF_GC(42, 16, 16, 4, 1, 9, 16, 4, 4, 8, 2, 2, 0, 8)
    NVF_GC(43, 16, 16, 4, 1, 9, 16, 4, 4, 8, 4, 2, 0, 0)
    NVF_GC(40, 16, 16, 4, 1, 17, 20, 4, 5, 10, 5, 2, 0, 8)
    NVF_GC(41, 16, 16, 4, 1, 17, 20, 4, 5, 10, 2, 2, 0, 8)
    NVF_GC(42, 16, 16, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 8)
    NVF_GC(43, 16, 16, 4, 1, 17, 20, 4, 5, 5, 5, 2, 0, 0)
#undef NVF_G
#undef NVF_GC
#undef NVF_GD
  }
  if (rc == 1) {
    long total = (long)batch * cout * dout * hout * wout;
    conv_gather_naive<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(x, w, bias, y, addend, mask, cin, cout, k,
                                                                      stride, d, total);
    rc = NVF_OK;
  }
  NVF_LAUNCH_CHECK();
  return rc;
}

// ---------------------------------------------------------------------------
// transposed convolution k=5 stride=2, forward
//   y[co,o] = bias + sum_ci sum_{k : (o+pad-k) even} x[ci,(o+pad-k)/2] w[ci][k][co]
// ---------------------------------------------------------------------------
__global__ void convT_k5s2_naive(const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ y, int cin, int cout, ConvDims d,
                                 long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int ox = idx % d.wout;
  long r = idx / d.wout;
  int oy = r % d.hout;
  r /= d.hout;
  int oz = r % d.dout;
  r /= d.dout;
  int co = r % cout;
  long b = r / cout;
  const float* xb = x + b * cin * (long)d.din * d.hin * d.win;
  float acc = 0.f;
  for (int ci = 0; ci < cin; ++ci)
    for (int kz = 0; kz < 5; ++kz) {
      int uz = oz + d.pad - kz;
      if (uz < 0 || (uz & 1) || (uz >> 1) >= d.din) continue;
      for (int ky = 0; ky < 5; ++ky) {
        int uy = oy + d.pad - ky;
        if (uy < 0 || (uy & 1) || (uy >> 1) >= d.hin) continue;
        for (int kx = 0; kx < 5; ++kx) {
          int ux = ox + d.pad - kx;
          if (ux < 0 || (ux & 1) || (ux >> 1) >= d.win) continue;
          float xv = xb[((long)ci * d.din + (uz >> 1)) * d.hin * d.win + (uy >> 1) * d.win + (ux >> 1)];
          acc = fmaf(xv, w[(ci * 125 + (kz * 5 + ky) * 5 + kx) * cout + co], acc);
        }
      }
    }
  y[idx] = nvf_act(acc + (bias ? bias[co] : 0.f), d.act);
}

// Tiled form.  Work is indexed by "cells" m = (o + pad) >> 1 per axis: a cell owns the
// two outputs o = 2m - pad + e (e = parity) and reads inputs i = m - j, j = 0..2 (even
// parity: taps k = 0,2,4) or j = 0..1 (odd parity: taps 1,3): 125 taps per cell, none
// multiplied by an inserted zero.  A thread owns VX consecutive cells along x, both x
// parities and all Cout channels, and walks the four (z,y) parity classes in turn.
template <int CIN_, int COUT_, int VX_, int NCX_, int TY_, int TZ_, int COG_ = 0>
struct TCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, VX = VX_, NCX = NCX_, TY = TY_, TZ = TZ_;
  static constexpr int COG = COG_ > 0 ? COG_ : COUT_;
  static constexpr int NCOG = COUT_ / COG;
  static constexpr int LV = VX_ % 4 == 0 ? 4 : (VX_ % 2 == 0 ? 2 : 1);
  static constexpr int TX = NCX * VX;
  static constexpr int NIN = VX + 2;
  static constexpr int NIN4 = (NIN + LV - 1) / LV * LV;
  static constexpr int IX = TX + 2, IY = TY + 2, IZ = TZ + 2;
  static constexpr int RS = (NCX - 1) * VX + NIN4;
  static constexpr int NACT = NCX * TY * TZ;
  static constexpr int LDSF = CIN * IZ * IY * RS;
  static constexpr int NT = ((NACT + 63) / 64 * 64 < 256 && LDSF >= 2048) ? 256 : (NACT + 63) / 64 * 64;
  static_assert(COUT_ % COG == 0, "channel groups");
  static_assert(RS >= IX, "row stride");
  static_assert(LDSF * 4 <= 160 * 1024, "LDS");
};

template <class C, int EZ, int EY>
__device__ __forceinline__ void convT_class(const float* lds, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ y, const ConvDims& d,
                                            int b, int co0, int cx, int ty, int tz, int mx0, int my, int mz) {
  constexpr int CIN = C::CIN, COUT = C::COUT, COG = C::COG, VX = C::VX, RS = C::RS, IY = C::IY, IZ = C::IZ,
                NIN4 = C::NIN4;
  float acc[2][COG][VX];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < COG; ++i)
#pragma unroll
      for (int j = 0; j < VX; ++j) acc[e][i][j] = 0.f;
#pragma unroll 1
  for (int c = 0; c < CIN; ++c) {
#pragma unroll
    for (int jz = 0; jz < 3 - EZ; ++jz) {
#pragma unroll
      for (int jy = 0; jy < 3 - EY; ++jy) {
        float in[NIN4];
        nvf_lds_row<NIN4, C::LV>(lds + ((c * IZ + tz + 2 - jz) * IY + ty + 2 - jy) * RS + cx * VX, in);
        const int kz = EZ + 2 * jz, ky = EY + 2 * jy;
        const float* wr = w + (size_t)((c * 5 + kz) * 5 + ky) * 5 * COUT + co0;  // wave-uniform
#pragma unroll
        for (int ex = 0; ex < 2; ++ex)
#pragma unroll
          for (int jx = 0; jx < 3 - ex; ++jx)
#pragma unroll
            for (int co = 0; co < COG; ++co) {
              const float wv = wr[(ex + 2 * jx) * COUT + co];
#pragma unroll
              for (int v = 0; v < VX; ++v) acc[ex][co][v] = fmaf(in[v + 2 - jx], wv, acc[ex][co][v]);
            }
      }
    }
  }
  const int oz = 2 * mz + EZ - d.pad, oy = 2 * my + EY - d.pad;
  if (oz < 0 || oz >= d.dout || oy < 0 || oy >= d.hout) return;
#pragma unroll
  for (int co = 0; co < COG; ++co) {
    const float bv = bias ? bias[co0 + co] : 0.f;
    const size_t base = (((size_t)b * COUT + co0 + co) * d.dout + oz) * d.hout * d.wout + (size_t)oy * d.wout;
#pragma unroll
    for (int v = 0; v < VX; ++v)
#pragma unroll
      for (int ex = 0; ex < 2; ++ex) {
        const int ox = 2 * (mx0 + v) + ex - d.pad;
        if (ox >= 0 && ox < d.wout) y[base + ox] = nvf_act(acc[ex][co][v] + bv, d.act);
      }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) void convT_k5s2_tiled(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          ConvDims d) {
  constexpr int CIN = C::CIN, VX = C::VX, NCX = C::NCX, TY = C::TY, TZ = C::TZ, RS = C::RS, IY = C::IY, IZ = C::IZ,
                IX = C::IX, NT = C::NT;
  __shared__ __attribute__((aligned(16))) float lds[C::LDSF];
  const int ntile = d.tiles_x * d.tiles_y * d.tiles_z;
  const int co0 = (blockIdx.x % C::NCOG) * C::COG;
  const int wg = blockIdx.x / C::NCOG;
  const int tile = wg % ntile, b = wg / ntile;
  const int tx_i = tile % d.tiles_x, ty_i = (tile / d.tiles_x) % d.tiles_y, tz_i = tile / (d.tiles_x * d.tiles_y);
  const int mlo = d.pad >> 1;
  const int cx0 = mlo + tx_i * C::TX, cy0 = mlo + ty_i * TY, cz0 = mlo + tz_i * TZ;  // first cell of the tile
  const int tid = threadIdx.x;
  const float* xb = x + (size_t)b * CIN * d.din * d.hin * d.win;
  const int plane = d.hin * d.win;
  nvf_stage_rows<NT, CIN * IZ * IY, RS, RS, 8>(
      xb, lds, tid,
      [&](int r, int xx, bool& ok) -> size_t {
        const int yy = r % IY, t = r / IY, zz = t % IZ, c = t / IZ;
        const int gx = cx0 - 2 + xx, gy = cy0 - 2 + yy, gz = cz0 - 2 + zz;
        ok = xx < IX && gx >= 0 && gx < d.win && gy >= 0 && gy < d.hin && gz >= 0 && gz < d.din;
        return ((size_t)c * d.din + gz) * plane + (size_t)gy * d.win + gx;
      },
      [&](int r, int xx) { return r * RS + xx; });
  __syncthreads();
  if (tid >= C::NACT) return;
  const int cx = tid % NCX, ty = (tid / NCX) % TY, tz = tid / (NCX * TY);
  const int mx0 = cx0 + cx * VX, my = cy0 + ty, mz = cz0 + tz;
  convT_class<C, 0, 0>(lds, w, bias, y, d, b, co0, cx, ty, tz, mx0, my, mz);
  convT_class<C, 0, 1>(lds, w, bias, y, d, b, co0, cx, ty, tz, mx0, my, mz);
  convT_class<C, 1, 0>(lds, w, bias, y, d, b, co0, cx, ty, tz, mx0, my, mz);
  convT_class<C, 1, 1>(lds, w, bias, y, d, b, co0, cx, ty, tz, mx0, my, mz);
}

template <class C>
static int launch_convT(const float* x, const float* w, const float* bias, float* y, int batch, ConvDims d,
                        hipStream_t s) {
  const int mlo = d.pad >> 1;
  // cells per axis: m in [mlo, (dim_out - 1 + pad) >> 1]
  const int ncx = ((d.wout - 1 + d.pad) >> 1) - mlo + 1;
  const int ncy = ((d.hout - 1 + d.pad) >> 1) - mlo + 1;
  const int ncz = ((d.dout - 1 + d.pad) >> 1) - mlo + 1;
  d.tiles_x = (ncx + C::TX - 1) / C::TX;
  d.tiles_y = (ncy + C::TY - 1) / C::TY;
  d.tiles_z = (ncz + C::TZ - 1) / C::TZ;
  dim3 grid((unsigned)(d.tiles_x * d.tiles_y * d.tiles_z) * batch * C::NCOG);
  convT_k5s2_tiled<C><<<grid, C::NT, 0, s>>>(x, w, bias, y, d);
  return NVF_OK;
}

extern "C" int nvf_convT3d_k5s2_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int cin,
                                    int cout, int pad, int din, int hin, int win, int dout, int hout, int wout,
                                    int act, int variant, void* stream) {
  if (!x || !w || !y || batch <= 0 || cin <= 0 || cout <= 0) return NVF_EINVAL;
  if (pad != 0 && pad != 2) return NVF_EINVAL;
  const int extra = pad == 0 ? 3 : 0;  // (in-1)*2 - 2*pad + 5 + output_padding
  if (dout != 2 * din + extra || hout != 2 * hin + extra || wout != 2 * win + extra) return NVF_EINVAL;
  ConvDims d{din, hin, win, dout, hout, wout, pad, act, 0, 0, 0};
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
  if (variant == 0 && batch <= 64) {
    if (cin == 16 && cout == 8 && win == 8) variant = 7;    // up1 forward
    if (cin == 8 && cout == 8 && win == 16) variant = 7;    // up2 forward
    if (cin == 8 && cout == 16 && win == 4) variant = 8;    // conv0 forward
  }
  if (variant != 1) {
#define NVF_T(VAR, CI, CO, WIN, VX, NCX, TY, TZ, COG)                      \
  if (rc == 1 && variant == VAR && cin == CI && cout == CO && win == WIN) \
    rc = launch_convT<TCfg<CI, CO, VX, NCX, TY, TZ, COG>>(x, w, bias, y, batch, d, s);
    NVF_T(0, 8, 8, 16, 2, 9, 6, 3, 4)     // up2 narrow: 16^3 -> 35^3 (18 cells / axis), Cout split 2 ways
    NVF_T(0, 16, 8, 8, 2, 5, 5, 5, 0)     // up1 narrow: 8^3 -> 19^3 (10 cells / axis)
    NVF_T(6, 16, 8, 8, 2, 5, 5, 5, 4)     // up1 narrow, small batch
    NVF_T(0, 3, 8, 2, 2, 1, 2, 2, 2)      // up0 narrow (ch = 3): 2^3 -> 4^3 (2 cells / axis)
    NVF_T(0, 8, 16, 2, 2, 1, 2, 2, 4)     // up0 wide (ch = 8)
    NVF_T(0, 8, 16, 4, 2, 2, 4, 4, 2)     // conv0 narrow: 4^3 -> 8^3 (4 cells / axis), Cout split 8 ways
    NVF_T(0, 16, 16, 16, 2, 9, 6, 3, 8)   // up2 wide (182 vs 480 us at batch 16, tools/wide_sweep.py)
    NVF_T(0, 32, 16, 8, 2, 5, 5, 2, 4)    // up1 wide (139 vs 643 us)
    NVF_T(0, 16, 32, 4, 2, 2, 4, 4, 2)    // conv0 wide (55 vs 74 us)
    NVF_T(7, 16, 8, 8, 2, 5, 5, 2, 2)
    NVF_T(8, 16, 8, 8, 2, 5, 5, 1, 4)
    NVF_T(9, 16, 8, 8, 2, 5, 10, 1, 4)
    NVF_T(7, 8, 8, 16, 2, 9, 6, 2, 4)
    NVF_T(8, 8, 8, 16, 2, 9, 3, 3, 4)
    NVF_T(9, 8, 8, 16, 2, 9, 6, 3, 2)
    NVF_T(7, 8, 16, 4, 2, 2, 4, 2, 2)
    NVF_T(8, 8, 16, 4, 2, 2, 4, 4, 1)
    NVF_T(2, 8, 8, 16, 4, 5, 6, 3, 0)
    NVF_T(3, 8, 8, 16, 2, 9, 6, 3, 0)
    NVF_T(4, 8, 8, 16, 2, 9, 6, 3, 4)
    NVF_T(5, 8, 8, 16, 4, 5, 6, 6, 4)
    NVF_T(2, 16, 8, 8, 2, 5, 5, 5, 0)
    NVF_T(3, 16, 8, 8, 2, 5, 5, 5, 4)
    NVF_T(4, 16, 8, 8, 2, 5, 10, 5, 2)
    NVF_T(5, 16, 8, 8, 4, 3, 5, 5, 4)
    NVF_T(2, 8, 16, 4, 2, 2, 4, 4, 2)
    NVF_T(3, 8, 16, 4, 2, 2, 4, 4, 8)
    // wide-decoder tuning candidates (tools/wide_sweep.py)
    NVF_T(40, 32, 16, 8, 2, 5, 5, 5, 4)
    NVF_T(41, 32, 16, 8, 2, 5, 5, 2, 4)
    NVF_T(42, 32, 16, 8, 2, 5, 10, 1, 4)
    NVF_T(43, 32, 16, 8, 2, 5, 5, 5, 8)
    NVF_T(40, 16, 16, 16, 2, 9, 6, 3, 4)
    NVF_T(41, 16, 16, 16, 2, 9, 6, 3, 8)
    NVF_T(42, 16, 16, 16, 2, 9, 6, 6, 4)
    NVF_T(43, 16, 16, 16, 2, 9, 3, 3, 8)
    NVF_T(40, 16, 32, 4, 2, 2, 4, 2, 4)
    NVF_T(41, 16, 32, 4, 2, 2, 4, 4, 2)
    NVF_T(42, 16, 32, 4, 2, 2, 4, 4, 8)
#undef NVF_T
  }
  if (rc == 1) {
    long total = (long)batch * cout * dout * hout * wout;
    convT_k5s2_naive<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(x, w, bias, y, cin, cout, d, total);
    rc = NVF_OK;
  }
  NVF_LAUNCH_CHECK();
  return rc;
}
