// Shared device helpers for libnvf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nvf_hip.h"

#define NVF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline hipStream_t nvf_stream(void* s) { return (hipStream_t)s; }

// Tuning knobs.  A default build has NO environment reads and no mutable process-wide state ("no hidden
// global state"): nvf_tune_int is the compile-time default.  A throw-away build with -DNVF_TUNING (tools/ab_build.py)
// reads the knob from the environment on every call -- launch geometry only, never a choice of arithmetic: the forms
// that change bits are selected by the caller through NvfStepCtx (nvf_step_ctx_set_direct / _set_wgrad_forms) or by an
// explicit argument of the entry point.
#ifdef NVF_TUNING
#include <stdlib.h>
static inline int nvf_tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int nvf_tune_int(const char*, int dflt) { return dflt; }
#endif

__device__ __forceinline__ float nvf_act(float v, int act) {
  if (act == NVF_ACT_RELU) return fmaxf(v, 0.f);
  if (act == NVF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// two floats at any 4-byte phase (rows of 35 or 19 floats): the compiler may use one 8-byte access, not an aligned one
struct __attribute__((packed, aligned(4))) nvf_f2u { float a, b; };
struct __attribute__((packed, aligned(4))) nvf_f4u { float a, b, c, d; };   // four, likewise (parameter slices of a flat buffer)

// ---- Philox4x32-10 counter RNG ------------------------------------------------
// key = seed (64 bit), counter = (index lo, index hi, stream lo, stream hi).
__device__ __forceinline__ void nvf_philox(uint64_t seed, uint64_t stream_id, uint64_t index, uint32_t out[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
  uint32_t c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one U[0,1) float per element index (24 random mantissa bits, like torch's uniform)
__device__ __forceinline__ float nvf_uniform01(uint64_t seed, uint64_t stream_id, uint64_t index) {
  uint32_t r[4];
  nvf_philox(seed, stream_id, index >> 2, r);
  uint32_t v = r[index & 3];
  return (float)(v >> 8) * (1.0f / 16777216.0f);
}

// wave64 sum via DPP-free shuffles (width 64)
__device__ __forceinline__ float nvf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block sum of up to 1024 threads; result valid in thread 0. `red` holds >= 16 floats.
__device__ __forceinline__ float nvf_block_sum(float v, float* red) {
  v = nvf_wave_sum(v);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) s += red[i];
  }
  return s;
}

// ---- global -> LDS staging of a tile made of ROWS rows of RS floats ------------------------------
// One wave handles whole rows (lane = x), so the row -> (channel, z, y) decomposition is wave-uniform
// (scalar ALU) and each row is one coalesced load.  U rows are issued back to back as UNCONDITIONAL
// loads (out-of-range elements read a safe address and are zeroed afterwards), so U loads are in flight
// per lane instead of one load / one s_waitcnt vmcnt(0) per element.
// `src(row, x, ok)` returns the element's offset into `base` and sets ok = element is inside the tensor.
template <int NT, int ROWS, int RS, int LDS_RS, int U, class Src, class Dst>
__device__ __forceinline__ void nvf_stage_rows(const float* __restrict__ base, float* lds, int tid, Src src, Dst dst) {
  static_assert(RS <= 64, "one wave covers a row");
  constexpr int NW = NT / 64;
  constexpr int RPW = RS <= 32 ? 2 : 1;                    // short rows: two rows per wave pass
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // SGPR: row math on the SALU
  const int sub = RPW == 2 ? lane >> 5 : 0;
  const int xx = RPW == 2 ? lane & 31 : lane;
#pragma unroll 1
  for (int r0 = wave * RPW; r0 < ROWS; r0 += NW * RPW * U) {
    float v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * NW * RPW + sub;
      const bool live = r < ROWS && xx < RS;
      bool inside = false;
      const size_t off = src(live ? r : 0, live ? xx : 0, inside);
      ok[u] = live && inside;
      v[u] = base[ok[u] ? off : 0];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * NW * RPW + sub;
      if (r < ROWS && xx < RS) lds[dst(r, xx)] = ok[u] ? v[u] : 0.f;
    }
  }
}

// read N floats (N a multiple of LV) from an LDS row whose start is LV*4-byte aligned
template <int N, int LV>
__device__ __forceinline__ void nvf_lds_row(const float* p, float* out) {
  if constexpr (LV == 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      float4 t = ((const float4*)p)[i];
      out[4 * i] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = t.w;
    }
  } else if constexpr (LV == 2) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      float2 t = ((const float2*)p)[i];
      out[2 * i] = t.x; out[2 * i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = p[i];
  }
}

// ---- LDS-DMA (global_load_lds_dword: HBM/L2 -> LDS with no VGPR destination) ----------------------
// One wave-instruction writes LDS[m0 + 4*lane] for every active lane.  M0 is compiler-reserved, so it is
// saved, set and restored inside the one asm statement that uses it; hipcc does not count these loads,
// the caller waits with an explicit s_waitcnt vmcnt(0).
// a pointer the compiler can keep in scalar registers (the "s" operand of nvf_glds_row)
__device__ __forceinline__ const float* nvf_uniform_ptr(const float* p) {
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const float*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void nvf_glds_row(const float* row_base, unsigned voff_bytes, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "s"(row_base)
      : "memory");
}
__device__ __forceinline__ void nvf_glds_lane(const float* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_byte))
      : "memory");
}


// ---- classifier-head kernels (heads.hip), reached through nvf_conv3d_gather / nvf_wgrad; 1 = no instantiation ----
int nvf_head_fwd_launch(const float* x, const float* w, const float* bias, float* y, const float* addend,
                        const float* mask, int batch, int c, int s, int act, hipStream_t st);
int nvf_head_bwd_data_launch(const float* dl, const float* wb, const float* bias, float* dx, const float* addend,
                             const float* mask, int batch, int c, int s, int act, hipStream_t st);
int nvf_head_wgrad_launch(const float* dl, const float* x, float* slabs, int max_slabs, int batch, int c, int s,
                          int* nslab, hipStream_t st);
// ---- matrix-core weight gradients with 16 q-channels (wgrad16_mfma.hip), reached through nvf_wgrad*; 1 = no instantiation
int nvf_wgrad16_launch(const float* p, const float* q, float* slabs, int batch, int a, int k, int stride, int pad, int dp,
                       int dq, int max_slabs, int* nslab, hipStream_t s);
