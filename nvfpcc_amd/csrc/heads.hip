// The three one-channel classifier heads of the NVF decoder (conv0_cls / conv1_cls / conv2_cls:
// Conv3d(C -> 1, k = 3, padding 1) + sigmoid, utils/network.py:4761-4768 through IConv3d :735-742 and
// QConv3d :677-687) and their autograd backward passes.
//
// They are 4 % of the FLOPs, but as instances of the general tiled kernels they were a fifth of the
// batch-16 step: 27-tap stencils over tensors that are read once are bound by memory latency, and
// the general kernels keep only a few loads in flight per wave and fetch weights through chains of
// scalar loads.  These kernels (reached through nvf_conv3d_gather / nvf_wgrad, same contract, same
// per-output fmaf order as the one-thread-per-output kernels) instead
//   * take full-width tiles (TZ x TY rows of S voxels), so every global access is an aligned float4
//     and a thread issues all of its tile loads before the first one is consumed,
//   * keep the 27 C weights in LDS (broadcast ds_read_b128) -- no scalar-load chains,
//   * give each thread four consecutive x outputs, so one row segment (a b128 plus two b32 LDS
//     reads) feeds 12 FMAs per input channel.
#include "nvf_common.h"
#include "step_ctx.h"
#include "stem_bwd.h"      // nvf_coop_signal / device-scope accesses
#include <cstdlib>

int nvf_heads3_wgrad_mfma_launch(const float* const* dls, const float* const* xs, float* const* slabs, int narrow,
                                 int batch, int max_slabs, int* nslabs, hipStream_t s);

namespace {

template <int C_, int S_, int TZ_, int TY_>
struct HCfg {
  static constexpr int C = C_, S = S_, TZ = TZ_, TY = TY_;
  static constexpr int XG = S / 4;                 // float4 groups per row
  static constexpr int RS = S + 8;                 // LDS row: x = -1 at word 3, x = 0 at word 4 (16-B aligned)
  static constexpr int IZ = TZ + 2, IY = TY + 2;
  static constexpr int NACT = TZ * TY * XG;        // one thread per four outputs
  static constexpr int NT = NACT < 256 ? 256 : NACT;   // small tiles: the other threads only help staging
  static_assert(S % 4 == 0 && NACT >= 64 && NT <= 1024 && NACT % 64 == 0, "tile");
};

// C channels of the x tile (with a one-voxel halo, zero outside the tensor) -> LDS [c][IZ][IY][RS]
template <int CH, int S, int IZ, int IY, int RS, int NT>
__device__ __forceinline__ void head_stage(const float* __restrict__ xb, float* xs, int tid, int z0, int y0) {
  constexpr int XG = S / 4, ITEMS = CH * IZ * IY * XG, U = 16;
#pragma unroll 1
  for (int i0 = tid; i0 < ITEMS; i0 += NT * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NT;
      const int xq = i % XG, r = i / XG, yi = r % IY, t = r / IY, zi = t % IZ, c = t / IZ;
      const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
      const bool ok = i < ITEMS && gz >= 0 && gz < S && gy >= 0 && gy < S;
      v[u] = ok ? *(const float4*)(xb + (((size_t)c * S + gz) * S + gy) * S + 4 * xq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NT;
      if (i < ITEMS) {
        const int xq = i % XG, r = i / XG;
        float* row = xs + (size_t)r * RS;
        *(float4*)(row + 4 + 4 * xq) = v[u];
        if (xq == 0) row[3] = 0.f;
        if (xq == XG - 1) row[S + 4] = 0.f;
      }
    }
  }
}

// Six consecutive words x-1 .. x+4 of a staged row for the thread that owns x .. x+3 (xg = its float4 group, threads of
// a row are consecutive lanes): ONE aligned ds_read_b128; the two halo words come from the neighbour lanes by DPP (the
// row ends are the zero padding).  Scalar LDS reads at a lane stride of four words would hit 8 of the 32 banks.
typedef float hf4 __attribute__((ext_vector_type(4)));
template <int XG>
__device__ __forceinline__ void head_row6(const float* row, int xg, float (&v)[6]) {
  const hf4 m = *(const hf4*)(row + 4);
  const int left = __builtin_amdgcn_update_dpp(0, __float_as_int(m.w), 0x111, 0xf, 0xf, true);    // row_shr:1
  const int right = __builtin_amdgcn_update_dpp(0, __float_as_int(m.x), 0x101, 0xf, 0xf, true);   // row_shl:1
  v[0] = xg == 0 ? 0.f : __int_as_float(left);
  v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
  v[5] = xg == XG - 1 ? 0.f : __int_as_float(right);
}

// ---- forward, channel-pipelined: the tile of ONE input channel (with its y / z halo rows; the x halo comes from the
// neighbour lanes, head_row6) is staged by LDS-DMA into one of two buffers while the previous channel's 27 taps are
// being accumulated.  A workgroup needs 8-20 KB of LDS instead of the whole C-channel tile (61-102 KB), so up to
// eight of them share a CU and one's staging overlaps another's arithmetic; tiles are TZ x TY = 4 x 8 / 8 x 8 whatever
// C is.  Rows are S words apart (a wave reads 1 KB of consecutive LDS per ds_read_b128, a DMA instruction fills two
// rows); rows outside the tensor are zeroed once.  Per output the fmaf order is (c, kz, ky, kx) as before.
template <int C_, int S_, int TZ_, int TY_, int CPS_ = 1>
struct HPCfg {
  static constexpr int C = C_, S = S_, TZ = TZ_, TY = TY_, CPS = CPS_;   // CPS channels per pipeline step (small tiles:
  static_assert(C % CPS == 0, "whole steps");                            //  a step costs ~1 us of latency whatever its size)
  static constexpr int XG = S / 4, IZ = TZ + 2, IY = TY + 2;
  static constexpr int WORDS = IZ * IY * S;            // one channel's tile
  static constexpr int NACT = TZ * TY * XG;            // one thread per four outputs
  static constexpr int NT = 256, NW = 4;
  static constexpr int NIT = (WORDS + NT - 1) / NT;    // (dword DMA instructions per wave per channel: no longer used)
  static_assert(S % 4 == 0 && NACT <= NT && NACT % 64 == 0 && WORDS % 4 == 0, "tile");
};
template <class H>
struct HFwdSmem { static constexpr int WORDS = 2 * H::CPS * H::WORDS + H::C * 9 * 4; };

// 16-byte device-scope accesses (sc1: the store goes through to memory, the load never hits a stale line of this XCD's
// L2) for values that cross workgroups INSIDE a launch (heads3_fwd_loss_bwd_data_kernel).  The compiler does not track an
// asm load: the caller waits (nvf_wait_dev4) before it touches the registers.
constexpr int kHeadFlagStride = 64;     // words between two counters: 256 B, so the pollers of different blocks hit different lines
#ifndef NVF_HC_SLEEP
#define NVF_HC_SLEEP 32                 // s_sleep argument of a polling consumer (x 64 cycles)
#endif

#ifndef NVF_HC_DBG
#define NVF_HC_DBG 0     // tuning builds (wrong results): 1 = plain stores / loads, 2 = consumers do not wait, 4 = no signal
#endif
__device__ __forceinline__ void nvf_store_dev4(float* p, hf4 v) {
  if (NVF_HC_DBG & 1) { *(hf4*)p = v; return; }
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void nvf_load_dev4_issue(hf4& v, const float* p) {
  if (NVF_HC_DBG & 1) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); return; }
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void nvf_wait_dev4(hf4 (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));      // (ordered behind the wait: volatile asms keep their order)
}

// COOP (the forward inside the launch that also runs the loss and the backward-data): p leaves with device-scope stores
// and the workgroup signals done[b] -- every thread reaches the signal, so no early return.
template <class H, bool COOP = false>
__device__ __forceinline__ void head_fwd_body(const float* __restrict__ x, const float* __restrict__ w,
                                              const float* __restrict__ bias, float* __restrict__ y,
                                              const float* __restrict__ addend, const float* __restrict__ mask, int act,
                                              int bid, float* smem, unsigned* done = nullptr) {
  constexpr int C = H::C, S = H::S, TZ = H::TZ, TY = H::TY, IY = H::IY, XG = H::XG, NT = H::NT, NIT = H::NIT,
                WORDS = H::WORDS, CPS = H::CPS;
  float* xs = smem;
  float* ws = smem + 2 * CPS * WORDS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int TILES_Y = S / TY, TILES_Z = S / TZ;
  const int tile = bid % (TILES_Y * TILES_Z), b = bid / (TILES_Y * TILES_Z);
  const int y0 = (tile % TILES_Y) * TY, z0 = (tile / TILES_Y) * TZ;
  for (int i = tid; i < C * 9 * 4; i += NT) ws[i] = (i & 3) < 3 ? w[(i >> 2) * 3 + (i & 3)] : 0.f;
  // which 16-byte piece of a channel volume each of this lane's DMA instructions fetches (the same for every channel).
  // global_load_lds_dwordx4: a wave-instruction moves 1 KiB (64 lanes x 16 B, LDS destination = wave base + 16 lane); the
  // tile image is rows of S words with nothing between them, so 64 consecutive pieces are 64 consecutive LDS quadwords;
  // the per-lane SOURCE follows the (z, y) of the piece's row.  (The dword form was four times the instructions: the
  // DMA issue rate, not its latency, bound this kernel.)
  constexpr int NPC = WORDS / 4, NI4 = (NPC + NT - 1) / NT;       // pieces per channel tile, instructions per wave
  unsigned soff[NI4];
  bool sok[NI4];
#pragma unroll
  for (int i = 0; i < NI4; ++i) {
    const int pc = i * NT + tid, wd = 4 * pc;
    const int r = wd / S, xx = wd - r * S, yi = r % IY, zi = r / IY;
    const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
    const bool live = pc < NPC;
    sok[i] = live && gz >= 0 && gz < S && gy >= 0 && gy < S;
    soff[i] = sok[i] ? (unsigned)((gz * S + gy) * S + xx) : 0u;
    if (live && !sok[i]) {
#pragma unroll
      for (int q = 0; q < 2 * CPS; ++q) *(float4*)(xs + q * WORDS + wd) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float* xb = x + (size_t)b * C * S * S * S;
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  auto stage = [&](int step, int buf) {
#pragma unroll
    for (int cc = 0; cc < CPS; ++cc) {
      const float* src = xb + (size_t)(step * CPS + cc) * S * S * S;
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        // wave-uniform LDS base of this instruction's 64 pieces
        float* dst = xs + (buf * CPS + cc) * WORDS + (i * NT + wave * 64) * 4;
        if (sok[i]) __builtin_amdgcn_global_load_lds((glb_vp)(src + soff[i]), (lds_vp)dst, 16, 0, 0);
      }
    }
  };
  stage(0, 0);
  const bool active = tid < H::NACT;
  const int xg = tid % XG, ty = (tid / XG) % TY, tz = tid / (XG * TY);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int step = 0; step < C / CPS; ++step) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the step has landed
    __syncthreads();                                              // ... everyone's; the other buffer is free
    if (step + 1 < C / CPS) stage(step + 1, (step + 1) & 1);
    if (active) {
#pragma unroll 1
      for (int cc = 0; cc < CPS; ++cc) {
        const int c = step * CPS + cc;
        const float* xc = xs + ((step & 1) * CPS + cc) * WORDS;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float* row = xc + ((tz + kz) * IY + ty + ky) * S + 4 * xg - 4;     // head_row6 reads row + 4
            float v[6];
            head_row6<XG>(row, xg, v);
            const float4 wv = *(const float4*)(ws + (c * 9 + kz * 3 + ky) * 4);
            const float wk[3] = {wv.x, wv.y, wv.z};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int o = 0; o < 4; ++o) acc[o] = fmaf(v[o + kx], wk[kx], acc[o]);
          }
      }
    }
  }
  if (COOP) {
    if (active) {
      const float bv = bias ? bias[0] : 0.f;
      const size_t off = (((size_t)b * S + z0 + tz) * S + y0 + ty) * S + 4 * xg;
      nvf_store_dev4(y + off, hf4{nvf_act(acc[0] + bv, act), nvf_act(acc[1] + bv, act), nvf_act(acc[2] + bv, act),
                                  nvf_act(acc[3] + bv, act)});
    }
    if (!(NVF_HC_DBG & 4)) nvf_coop_signal(done + b * kHeadFlagStride);
    return;
  }
  if (!active) return;
  const float bv = bias ? bias[0] : 0.f;
  const size_t off = (((size_t)b * S + z0 + tz) * S + y0 + ty) * S + 4 * xg;
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = nvf_act(acc[i] + bv, act);
  if (addend) {
    const float4 a = *(const float4*)(addend + off);
    o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
  }
  if (mask) {
    const float4 m = *(const float4*)(mask + off);
    o[0] = m.x > 0.f ? o[0] : 0.f; o[1] = m.y > 0.f ? o[1] : 0.f;
    o[2] = m.z > 0.f ? o[2] : 0.f; o[3] = m.w > 0.f ? o[3] : 0.f;
  }
  *(float4*)(y + off) = make_float4(o[0], o[1], o[2], o[3]);
}

template <class H>
__global__ __launch_bounds__(H::NT) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const float* __restrict__ addend,
                                                         const float* __restrict__ mask, int act) {
  __shared__ __attribute__((aligned(16))) float smem[HFwdSmem<H>::WORDS];
  head_fwd_body<H>(x, w, bias, y, addend, mask, act, blockIdx.x, smem);
}

// Focal inputs of one head: the backward-data kernel can compute dlogit = d focal / d logit on the fly while it
// stages its tile (the halo is recomputed, the tile's own voxels are also written to `dl` for the weight gradient
// and summed into one loss partial per workgroup) -- the separate loss launch disappears.
struct HeadLoss {
  const float* p;
  const float* gt;
  const float* dist;       // or null
  float* dl;               // [B, 1, S^3] out
  float* part;             // loss partials of this term: one per workgroup
  float* bias_part;        // or null: the sum of the workgroup's own dl values (the head's bias gradient), one per workgroup
  float alpha, beta;
  // COOP only (p is produced by the forward workgroups of the SAME launch): done[b] counts the nprod forward workgroups of
  // block b, used[b] the ncons workgroups that have seen them all -- the last of those zeroes both for the next launch
  unsigned* done;
  unsigned* used;
  unsigned nprod, ncons;
};

// every thread of a consumer workgroup calls it before its first device-scope load of what the nprod producers wrote
__device__ __forceinline__ void nvf_coop_wait_many(unsigned* counter, unsigned target, unsigned* used, unsigned ncons) {
  if (threadIdx.x == 0 && !(NVF_HC_DBG & 2)) {
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
      __builtin_amdgcn_s_sleep(NVF_HC_SLEEP);
    // (a consumer counts itself only after its wait has ended, so when the last one arrives nobody polls `counter` any more)
    if (__hip_atomic_fetch_add(used, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ncons - 1) {
      __hip_atomic_store(used, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

template <int S, int TZ, int TY, int RS, int NT, bool COOP = false>
__device__ __forceinline__ void head_stage_loss(const HeadLoss& f, float* ds, float* red, int tid, int b, int z0, int y0,
                                                int wg) {
  constexpr int XG = S / 4, IZ = TZ + 2, IY = TY + 2, ITEMS = IZ * IY * XG;
  const size_t vol = (size_t)S * S * S;
  const float a1 = f.alpha, a0 = 1.f - f.alpha;
  float s = 0.f, sb = 0.f;
  constexpr int U = (ITEMS + NT - 1) / NT;                    // every load of the thread in flight before any is used
  float4 pv[U], gv[U], dv[U];
  hf4 pdev[COOP ? U : 1];
  if (COOP) {
    // the targets and distances do not depend on the forward: their loads are in flight while the workgroup waits for
    // block b's forward workgroups; then p by device-scope loads
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * NT;
      const int xq = i % XG, r = i / XG, yi = r % IY, zi = r / IY;
      const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
      const bool ok = i < ITEMS && gz >= 0 && gz < S && gy >= 0 && gy < S;
      const size_t off = ok ? b * vol + ((size_t)gz * S + gy) * S + 4 * xq : 0;
      gv[u] = *(const float4*)(f.gt + off);
      dv[u] = *(const float4*)((f.dist ? f.dist : f.gt) + off);
    }
    nvf_coop_wait_many(f.done + b * kHeadFlagStride, f.nprod, f.used + b * kHeadFlagStride, f.ncons);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * NT;
      const int xq = i % XG, r = i / XG, yi = r % IY, zi = r / IY;
      const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
      const bool ok = i < ITEMS && gz >= 0 && gz < S && gy >= 0 && gy < S;
      nvf_load_dev4_issue(pdev[u], f.p + (ok ? b * vol + ((size_t)gz * S + gy) * S + 4 * xq : 0));
    }
    nvf_wait_dev4(pdev);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = tid + u * NT;
    const int xq = i % XG, r = i / XG, yi = r % IY, zi = r / IY;
    const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
    const bool ok = i < ITEMS && gz >= 0 && gz < S && gy >= 0 && gy < S;
    const size_t off = ok ? b * vol + ((size_t)gz * S + gy) * S + 4 * xq : 0;
    // (unconditional loads from a valid address, then a select: `ok ? *p : zero` becomes a select of ADDRESSES -- a
    // private zero against the global pointer -- i.e. four flat_load_dword per element instead of one global_load_dwordx4)
    const float4 pl = COOP ? make_float4(pdev[COOP ? u : 0].x, pdev[COOP ? u : 0].y, pdev[COOP ? u : 0].z, pdev[COOP ? u : 0].w)
                           : *(const float4*)(f.p + off);
    const float4 gl = COOP ? gv[u] : *(const float4*)(f.gt + off);
    const float4 dl4 = COOP ? dv[u] : *(const float4*)((f.dist ? f.dist : f.p) + off);
    // (component selects: `ok ? pl : z4` on the float4 STRUCT is a select of two memory copies -- both values went
    // through scratch memory and came back by a flat load)
    const bool okd = ok && f.dist;
    pv[u] = make_float4(ok ? pl.x : 0.f, ok ? pl.y : 0.f, ok ? pl.z : 0.f, ok ? pl.w : 0.f);
    gv[u] = make_float4(ok ? gl.x : 0.f, ok ? gl.y : 0.f, ok ? gl.z : 0.f, ok ? gl.w : 0.f);
    dv[u] = make_float4(okd ? dl4.x : 0.f, okd ? dl4.y : 0.f, okd ? dl4.z : 0.f, okd ? dl4.w : 0.f);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = tid + u * NT;
    if (i >= ITEMS) continue;
    const int xq = i % XG, r = i / XG, yi = r % IY, zi = r / IY;
    const int gz = z0 - 1 + zi, gy = y0 - 1 + yi;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gz >= 0 && gz < S && gy >= 0 && gy < S) {
      const size_t off = b * vol + ((size_t)gz * S + gy) * S + 4 * xq;
      const float t0 = focal_elem(pv[u].x, gv[u].x, dv[u].x, f.dist != nullptr, a1, a0, f.beta, 1, o.x);
      const float t1 = focal_elem(pv[u].y, gv[u].y, dv[u].y, f.dist != nullptr, a1, a0, f.beta, 1, o.y);
      const float t2 = focal_elem(pv[u].z, gv[u].z, dv[u].z, f.dist != nullptr, a1, a0, f.beta, 1, o.z);
      const float t3 = focal_elem(pv[u].w, gv[u].w, dv[u].w, f.dist != nullptr, a1, a0, f.beta, 1, o.w);
      if (zi >= 1 && zi <= TZ && yi >= 1 && yi <= TY) {       // this tile's own voxels
        *(float4*)(f.dl + off) = o;
        s += (t0 + t1) + (t2 + t3);
        sb += (o.x + o.y) + (o.z + o.w);
      }
    }
    float* row = ds + (size_t)r * RS;
    *(float4*)(row + 4 + 4 * xq) = o;
    if (xq == 0) row[3] = 0.f;
    if (xq == XG - 1) row[S + 4] = 0.f;
  }
  const float tot = nvf_block_sum(s, red);
  if (tid == 0) f.part[wg] = tot;
  if (f.bias_part) {
    const float totb = nvf_block_sum(sb, red + 8);
    if (tid == 0) f.bias_part[wg] = totb;
  }
}

// ---- backward-data: dx[c, i] = sum_k' dl[i - 1 + k'] wb[k'][c]  (wb = w_bwd: taps flipped) ---------------------
template <class H>
struct HBwdSmem { static constexpr int WORDS = (H::IZ * H::IY * H::RS + 3) / 4 * 4 + 27 * H::C + 16; };   // + block-sum scratch

template <class H, bool COOP = false>
__device__ __forceinline__ void head_bwd_data_body(const float* __restrict__ dl, const float* __restrict__ wb,
                                                   const float* __restrict__ bias, float* __restrict__ dx,
                                                   const float* __restrict__ addend, const float* __restrict__ mask,
                                                   int act, int bid, float* smem, const HeadLoss* loss = nullptr) {
  constexpr int C = H::C, S = H::S, TZ = H::TZ, TY = H::TY, RS = H::RS, IZ = H::IZ, IY = H::IY, XG = H::XG, NT = H::NT;
  float* ds = smem;
  float* ws = smem + (IZ * IY * RS + 3) / 4 * 4;
  const int tid = threadIdx.x;
  constexpr int TILES_Y = S / TY, TILES_Z = S / TZ;
  const int tile = bid % (TILES_Y * TILES_Z), b = bid / (TILES_Y * TILES_Z);
  const int y0 = (tile % TILES_Y) * TY, z0 = (tile / TILES_Y) * TZ;
  for (int i = tid; i < 27 * C; i += NT) ws[i] = wb[i];
  if (loss) head_stage_loss<S, TZ, TY, RS, NT, COOP>(*loss, ds, ws + 27 * C, tid, b, z0, y0, bid);
  else head_stage<1, S, IZ, IY, RS, NT>(dl + (size_t)b * S * S * S, ds, tid, z0, y0);
  __syncthreads();
  if (tid >= H::NACT) return;
  const int xg = tid % XG, ty = (tid / XG) % TY, tz = tid / (XG * TY);
  float acc[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[c][o] = 0.f;
#pragma unroll 1
  for (int kz = 0; kz < 3; ++kz)
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = ds + ((size_t)(tz + kz) * IY + ty + ky) * RS + 4 * xg;
      float v[6];
      head_row6<XG>(row, xg, v);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* wr = ws + ((kz * 3 + ky) * 3 + kx) * C;
#pragma unroll
        for (int c4 = 0; c4 < C; c4 += 4) {
          const float4 wv = *(const float4*)(wr + c4);
          const float wk[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[c4 + cc][o] = fmaf(v[o + kx], wk[cc], acc[c4 + cc][o]);
        }
      }
    }
  const size_t vol = (size_t)S * S * S;
  const size_t off = (size_t)b * C * vol + (((size_t)(z0 + tz)) * S + y0 + ty) * S + 4 * xg;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float bv = bias ? bias[c] : 0.f;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = nvf_act(acc[c][i] + bv, act);
    if (addend) {
      const float4 a = *(const float4*)(addend + off + c * vol);
      o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
    }
    if (mask) {
      const float4 m = *(const float4*)(mask + off + c * vol);
      o[0] = m.x > 0.f ? o[0] : 0.f; o[1] = m.y > 0.f ? o[1] : 0.f;
      o[2] = m.z > 0.f ? o[2] : 0.f; o[3] = m.w > 0.f ? o[3] : 0.f;
    }
    *(float4*)(dx + off + c * vol) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <class H>
__global__ __launch_bounds__(H::NT) void head_bwd_data_kernel(const float* __restrict__ dl, const float* __restrict__ wb,
                                                              const float* __restrict__ bias, float* __restrict__ dx,
                                                              const float* __restrict__ addend,
                                                              const float* __restrict__ mask, int act) {
  __shared__ __attribute__((aligned(16))) float smem[HBwdSmem<H>::WORDS];
  head_bwd_data_body<H>(dl, wb, bias, dx, addend, mask, act, blockIdx.x, smem);
}

// ---- weight gradient: dw[c][k] = sum_{b,i} dl[b,i] x[b,c,i + k - 1] -------------------------------------------
// lane = (row of the tile, float4 group along x): every LDS access of a wave is a run of consecutive float4s, and a
// lane keeps the 27 tap sums of one channel for its four x positions; a wave takes C / 4 channels in turn and adds
// its lanes' sums by shuffles at the end (fixed order), one slab per workgroup.
template <int C_, int S_, int TZ_, int TY_, int CSPLIT_>
struct HWCfg {
  static constexpr int C = C_, S = S_, TZ = TZ_, TY = TY_, CSPLIT = CSPLIT_;
  static constexpr int XG = S / 4, RS = S + 8, IZ = TZ + 2, IY = TY + 2;
  static constexpr int NW = 4, NT = NW * 64;
  static constexpr int RPW = 64 / XG;                  // tile rows a wave covers at once
  static constexpr int CW = C / CSPLIT;                // channels per workgroup (grid.y picks the group)
  static constexpr int CPW = CW / NW;                  // channels per wave
  static_assert((TZ * TY) % RPW == 0 && C % CSPLIT == 0 && CW % NW == 0, "tile rows / channels split evenly");
};

template <class H>
struct HWSmem { static constexpr int WORDS = H::CW * H::IZ * H::IY * H::RS + H::TZ * H::TY * H::S; };

template <class H>
__device__ __forceinline__ void head_wgrad_body(const float* __restrict__ dl, const float* __restrict__ x,
                                                float* __restrict__ slabs, int items, int items_per_wg, int bx, int by,
                                                float* smem) {
  constexpr int C = H::C, S = H::S, TZ = H::TZ, TY = H::TY, RS = H::RS, IZ = H::IZ, IY = H::IY, XG = H::XG, NT = H::NT,
                RPW = H::RPW, CPW = H::CPW, CW = H::CW;
  float* xs = smem;
  const int cg0 = by * CW;                             // first channel of this workgroup
  float* dls = smem + CW * IZ * IY * RS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xq = lane % XG, rl = lane / XG;
  constexpr int TILES_Y = S / TY, TILES_Z = S / TZ, TILES = TILES_Y * TILES_Z;
  float acc[CPW][27];
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[cc][t] = 0.f;
  const int first = bx * items_per_wg, last = min(first + items_per_wg, items);
#pragma unroll 1
  for (int item = first; item < last; ++item) {
    const int tile = item % TILES, b = item / TILES;
    const int y0 = (tile % TILES_Y) * TY, z0 = (tile / TILES_Y) * TZ;
    if (item != first) __syncthreads();
    head_stage<CW, S, IZ, IY, RS, NT>(x + ((size_t)b * C + cg0) * S * S * S, xs, tid, z0, y0);
    for (int i = tid; i < TZ * TY * XG; i += NT) {
      const int q = i % XG, r = i / XG, ty = r % TY, tz = r / TY;
      *(float4*)(dls + r * S + 4 * q) = *(const float4*)(dl + (((size_t)b * S + z0 + tz) * S + y0 + ty) * S + 4 * q);
    }
    __syncthreads();
#pragma unroll 1
    for (int r0 = 0; r0 < TZ * TY; r0 += RPW) {
      const int r = r0 + rl, ty = r % TY, tz = r / TY;
      const float4 d4 = *(const float4*)(dls + r * S + 4 * xq);
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int c = wave * CPW + cc;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float* row = xs + ((size_t)(c * IZ + tz + kz) * IY + ty + ky) * RS + 4 * xq;
            float v[6];
            head_row6<XG>(row, xq, v);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx)
                acc[cc][(kz * 3 + ky) * 3 + kx] = fmaf(d[o], v[o + kx], acc[cc][(kz * 3 + ky) * 3 + kx]);
          }
      }
    }
  }
  float* slab = slabs + (size_t)bx * (C * 27);
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float sum = nvf_wave_sum(acc[cc][t]);
      if (lane == 0) slab[(cg0 + wave * CPW + cc) * 27 + t] = sum;
    }
}

template <class H>
__global__ __launch_bounds__(H::NT) void head_wgrad_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                           float* __restrict__ slabs, int items, int items_per_wg) {
  __shared__ __attribute__((aligned(16))) float smem[HWSmem<H>::WORDS];
  head_wgrad_body<H>(dl, x, slabs, items, items_per_wg, blockIdx.x, blockIdx.y, smem);
}

// ---- the three heads of the narrow decoder in ONE launch each (forward / backward-data / weight gradient): the two
// small heads are latency-bound on a handful of CUs; beside the big one they cost nothing.  Workgroups
// [0, n0) run head 0, [n0, n0 + n1) head 1, the rest head 2; same bodies, so results are identical.
struct Heads3 {
  const float* a[3];      // x      | dlogit | dlogit
  const float* w[3];      // w_fwd  | w_bwd  | x
  const float* bias[3];
  float* out[3];          // p      | dx     | slabs
  const float* mask[3];
  int32_t n[3];           // workgroups (x dimension) per head
  int32_t items[3], per[3];
  int32_t act;
};
constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

template <class H0, class H1, class H2>
__global__ __launch_bounds__(256) void heads3_fwd_kernel(Heads3 m) {
  __shared__ __attribute__((aligned(16))) float smem[cmax3(HFwdSmem<H0>::WORDS, HFwdSmem<H1>::WORDS, HFwdSmem<H2>::WORDS)];
  const int bid = blockIdx.x;       // the small heads' few workgroups first: theirs is the longest chain of steps
  if (bid < m.n[0]) head_fwd_body<H0>(m.a[0], m.w[0], m.bias[0], m.out[0], nullptr, nullptr, m.act, bid, smem);
  else if (bid < m.n[0] + m.n[1])
    head_fwd_body<H1>(m.a[1], m.w[1], m.bias[1], m.out[1], nullptr, nullptr, m.act, bid - m.n[0], smem);
  else head_fwd_body<H2>(m.a[2], m.w[2], m.bias[2], m.out[2], nullptr, nullptr, m.act, bid - m.n[0] - m.n[1], smem);
}

template <class H0, class H1, class H2>
__global__ __launch_bounds__(256) void heads3_bwd_data_kernel(Heads3 m) {
  __shared__ __attribute__((aligned(16))) float smem[cmax3(HBwdSmem<H0>::WORDS, HBwdSmem<H1>::WORDS, HBwdSmem<H2>::WORDS)];
  const int bid = blockIdx.x;
  if (bid < m.n[2]) head_bwd_data_body<H2>(m.a[2], m.w[2], nullptr, m.out[2], nullptr, m.mask[2], 0, bid, smem);
  else if (bid < m.n[2] + m.n[1])
    head_bwd_data_body<H1>(m.a[1], m.w[1], nullptr, m.out[1], nullptr, m.mask[1], 0, bid - m.n[2], smem);
  else head_bwd_data_body<H0>(m.a[0], m.w[0], nullptr, m.out[0], nullptr, m.mask[0], 0, bid - m.n[2] - m.n[1], smem);
}

template <class H0, class H1, class H2>
__global__ __launch_bounds__(256) void heads3_wgrad_kernel(Heads3 m) {
  __shared__ __attribute__((aligned(16))) float smem[cmax3(HWSmem<H0>::WORDS, HWSmem<H1>::WORDS, HWSmem<H2>::WORDS)];
  int bid = blockIdx.x;                               // head h owns n[h] * CSPLIT_h workgroups: (slab, channel group)
  if (bid < m.n[2] * H2::CSPLIT) {                    // the big head first
    head_wgrad_body<H2>(m.a[2], m.w[2], m.out[2], m.items[2], m.per[2], bid % m.n[2], bid / m.n[2], smem);
    return;
  }
  bid -= m.n[2] * H2::CSPLIT;
  if (bid < m.n[1] * H1::CSPLIT) {
    head_wgrad_body<H1>(m.a[1], m.w[1], m.out[1], m.items[1], m.per[1], bid % m.n[1], bid / m.n[1], smem);
    return;
  }
  bid -= m.n[1] * H1::CSPLIT;
  head_wgrad_body<H0>(m.a[0], m.w[0], m.out[0], m.items[0], m.per[0], bid % m.n[0], bid / m.n[0], smem);
}

}  // namespace

// ---- launchers used by the dispatchers of nvf_conv3d_gather / nvf_wgrad ----------------------------------------
// return 1 when there is no instantiation for the shape (the caller falls back to the general kernels)
int nvf_head_fwd_launch(const float* x, const float* w, const float* bias, float* y, const float* addend,
                        const float* mask, int batch, int c, int s, int act, hipStream_t st) {
#define NVF_H(CC, SS, TZ, TY, CPS)                                                                              \
  if (c == CC && s == SS) {                                                                                     \
    using H = HPCfg<CC, SS, TZ, TY, CPS>;                                                                       \
    head_fwd_kernel<H><<<batch * (SS / TZ) * (SS / TY), H::NT, 0, st>>>(x, w, bias, y, addend, mask, act);      \
    return 0;                                                                                                   \
  }
  NVF_H(8, 32, 4, 8, 1)
  NVF_H(8, 16, 8, 8, 2)
  NVF_H(16, 8, 8, 8, 4)
  NVF_H(16, 32, 4, 8, 1)
  NVF_H(16, 16, 8, 8, 2)
  NVF_H(32, 8, 8, 8, 4)
#undef NVF_H
  return 1;
}

int nvf_head_bwd_data_launch(const float* dl, const float* wb, const float* bias, float* dx, const float* addend,
                             const float* mask, int batch, int c, int s, int act, hipStream_t st) {
#define NVF_H(CC, SS, TZ, TY)                                                                                        \
  if (c == CC && s == SS) {                                                                                          \
    using H = HCfg<CC, SS, TZ, TY>;                                                                                  \
    head_bwd_data_kernel<H><<<batch * (SS / TZ) * (SS / TY), H::NT, 0, st>>>(dl, wb, bias, dx, addend, mask, act);   \
    return 0;                                                                                                        \
  }
  NVF_H(8, 32, 4, 8)
  NVF_H(8, 16, 4, 4)
  NVF_H(16, 8, 4, 8)
  NVF_H(16, 32, 4, 8)
  NVF_H(16, 16, 4, 4)
  NVF_H(32, 8, 4, 8)
#undef NVF_H
  return 1;
}

// slabs: up to max_slabs partial results of c*27 floats each; *nslab receives the number written
int nvf_head_wgrad_launch(const float* dl, const float* x, float* slabs, int max_slabs, int batch, int c, int s,
                          int* nslab, hipStream_t st) {
#define NVF_H(CC, SS, TZ, TY, CSPLIT)                                                          \
  if (c == CC && s == SS) {                                                                    \
    using H = HWCfg<CC, SS, TZ, TY, CSPLIT>;                                                   \
    const int items = batch * (SS / TZ) * (SS / TY);                                           \
    int n = items < max_slabs ? items : max_slabs;                                             \
    const int per = (items + n - 1) / n;                                                       \
    n = (items + per - 1) / per;                                                               \
    head_wgrad_kernel<H><<<dim3(n, CSPLIT), H::NT, 0, st>>>(dl, x, slabs, items, per);         \
    *nslab = n;                                                                                \
    return 0;                                                                                  \
  }
  NVF_H(8, 32, 2, 8, 2)
  NVF_H(8, 16, 4, 4, 2)
  NVF_H(16, 8, 4, 8, 4)
  NVF_H(16, 32, 2, 8, 4)
  NVF_H(16, 16, 4, 4, 4)
  NVF_H(32, 8, 4, 8, 8)
#undef NVF_H
  return 1;
}

// ---- three-head launches: conv0_cls, conv1_cls, conv2_cls of the narrow decoder ([16, 8^3], [8, 16^3], [8, 32^3]) or
// of the wide one ([32, 8^3], [16, 16^3], [16, 32^3]) ----
static int heads3_tuple(const int* cs, const int* ss) {      // 0 narrow, 1 wide, -1 neither
  if (ss[0] != 8 || ss[1] != 16 || ss[2] != 32) return -1;
  if (cs[0] == 16 && cs[1] == 8 && cs[2] == 8) return 0;
  if (cs[0] == 32 && cs[1] == 16 && cs[2] == 16) return 1;
  return -1;
}
template <class H>
static constexpr int head_tiles() { return (H::S / H::TZ) * (H::S / H::TY); }

template <class H0, class H1, class H2>
static int heads3_fwd_t(const float* const* xs, const float* const* ws, const float* const* biases, float* const* ps,
                        int batch, int act, void* stream) {
  static_assert(H0::NT == 256 && H1::NT == 256 && H2::NT == 256, "one workgroup size");
  Heads3 m{};
  for (int h = 0; h < 3; ++h) {
    if (!xs[h] || !ws[h] || !ps[h]) return NVF_EINVAL;
    m.a[h] = xs[h]; m.w[h] = ws[h]; m.bias[h] = biases[h]; m.out[h] = ps[h];
  }
  m.n[0] = batch * head_tiles<H0>(); m.n[1] = batch * head_tiles<H1>(); m.n[2] = batch * head_tiles<H2>();
  m.act = act;
  heads3_fwd_kernel<H0, H1, H2><<<m.n[0] + m.n[1] + m.n[2], 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_heads3_fwd(const float* const* xs, const float* const* ws, const float* const* biases,
                              float* const* ps, const int* cs, const int* ss, int batch, int act, void* stream) {
  if (!xs || !ws || !biases || !ps || !cs || !ss || batch <= 0) return NVF_EINVAL;
  const int t = heads3_tuple(cs, ss);
  if (t == 0) return heads3_fwd_t<HPCfg<16, 8, 8, 8, 4>, HPCfg<8, 16, 8, 8, 2>, HPCfg<8, 32, 4, 8>>(xs, ws, biases, ps, batch, act, stream);
  if (t == 1) return heads3_fwd_t<HPCfg<32, 8, 8, 8, 4>, HPCfg<16, 16, 8, 8, 2>, HPCfg<16, 32, 4, 8>>(xs, ws, biases, ps, batch, act, stream);
  return NVF_EINVAL;
}

template <class H0, class H1, class H2>
static int heads3_bwd_data_t(const float* const* dls, const float* const* wbs, float* const* dxs,
                             const float* const* masks, int batch, void* stream) {
  static_assert(H0::NT == 256 && H1::NT == 256 && H2::NT == 256, "one workgroup size");
  Heads3 m{};
  for (int h = 0; h < 3; ++h) {
    if (!dls[h] || !wbs[h] || !dxs[h]) return NVF_EINVAL;
    m.a[h] = dls[h]; m.w[h] = wbs[h]; m.out[h] = dxs[h]; m.mask[h] = masks[h];
  }
  m.n[0] = batch * head_tiles<H0>(); m.n[1] = batch * head_tiles<H1>(); m.n[2] = batch * head_tiles<H2>();
  heads3_bwd_data_kernel<H0, H1, H2><<<m.n[0] + m.n[1] + m.n[2], 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// dxs[h] = backward-data of head h from dls[h] with w_bwd wbs[h]; masks[h] (may be NULL) is the ReLU mask
extern "C" int nvf_heads3_bwd_data(const float* const* dls, const float* const* wbs, float* const* dxs,
                                   const float* const* masks, const int* cs, const int* ss, int batch, void* stream) {
  if (!dls || !wbs || !dxs || !masks || !cs || !ss || batch <= 0) return NVF_EINVAL;
  const int t = heads3_tuple(cs, ss);
  if (t == 0) return heads3_bwd_data_t<HCfg<16, 8, 4, 8>, HCfg<8, 16, 4, 4>, HCfg<8, 32, 4, 8>>(dls, wbs, dxs, masks, batch, stream);
  if (t == 1) return heads3_bwd_data_t<HCfg<32, 8, 4, 8>, HCfg<16, 16, 4, 4>, HCfg<16, 32, 4, 8>>(dls, wbs, dxs, masks, batch, stream);
  return NVF_EINVAL;
}

struct Heads3Loss { HeadLoss h[3]; };

template <class H0, class H1, class H2>
__global__ __launch_bounds__(256) void heads3_loss_bwd_data_kernel(Heads3 m, Heads3Loss f) {
  __shared__ __attribute__((aligned(16))) float smem[cmax3(HBwdSmem<H0>::WORDS, HBwdSmem<H1>::WORDS, HBwdSmem<H2>::WORDS)];
  const int bid = blockIdx.x;
  if (bid < m.n[2]) head_bwd_data_body<H2>(nullptr, m.w[2], nullptr, m.out[2], nullptr, m.mask[2], 0, bid, smem, &f.h[2]);
  else if (bid < m.n[2] + m.n[1])
    head_bwd_data_body<H1>(nullptr, m.w[1], nullptr, m.out[1], nullptr, m.mask[1], 0, bid - m.n[2], smem, &f.h[1]);
  else
    head_bwd_data_body<H0>(nullptr, m.w[0], nullptr, m.out[0], nullptr, m.mask[0], 0, bid - m.n[2] - m.n[1], smem, &f.h[0]);
}

extern "C" size_t nvf_reduce_workspace(void);

// The focal terms of the three heads (NVFPCC.py:166-184), their gradients w.r.t. the logits and the heads'
// backward-data in ONE launch (+ the one-block final pass of the loss sums, which nvf_finals_begin defers):
// dls[h] = d term_h / d logit_h (also consumed in place: dxs[h] is nvf_heads3_bwd_data of it), loss[slots[h]] = term_h.
// Heads as in nvf_heads3_fwd; dists[h] may be NULL; batch * 32 workgroups of the big head must fit the reduction
// workspace (batch <= 32), otherwise NVF_EINVAL and the caller uses nvf_focal_loss_multi + nvf_heads3_bwd_data.
template <class H0, class H1, class H2>
static int heads3_loss_bwd_data_t(const float* const* ps, const float* const* gts, const float* const* dists,
                                  const float* alphas, const float* betas, const int* slots, float* loss,
                                  float* const* dls, const float* const* wbs, float* const* dxs,
                                  const float* const* masks, int batch, void* workspace, NvfStepCtx* ctx, void* stream,
                                  float* const* bias_outs) {
  static_assert(H0::NT == 256 && H1::NT == 256 && H2::NT == 256, "one workgroup size");
  Heads3 m{};
  Heads3Loss f{};
  FocalMulti fm{};
  m.n[0] = batch * head_tiles<H0>(); m.n[1] = batch * head_tiles<H1>(); m.n[2] = batch * head_tiles<H2>();
  for (int h = 0; h < 3; ++h) {
    if (!ps[h] || !gts[h] || !dls[h] || !wbs[h] || !dxs[h] || slots[h] < 0 || slots[h] > 2 || m.n[h] > kLossMaxWG)
      return NVF_EINVAL;
    m.w[h] = wbs[h]; m.out[h] = dxs[h]; m.mask[h] = masks[h];
    f.h[h].p = ps[h]; f.h[h].gt = gts[h]; f.h[h].dist = dists[h]; f.h[h].dl = dls[h];
    f.h[h].part = (float*)workspace + slots[h] * kLossMaxWG;
    f.h[h].bias_part = bias_outs ? (float*)workspace + (3 + h) * kLossMaxWG : nullptr;
    f.h[h].alpha = alphas[h]; f.h[h].beta = betas[h];
    fm.nwg[slots[h]] = m.n[h];
  }
  if (slots[0] == slots[1] || slots[0] == slots[2] || slots[1] == slots[2]) return NVF_EINVAL;
  heads3_loss_bwd_data_kernel<H0, H1, H2><<<m.n[0] + m.n[1] + m.n[2], 256, 0, nvf_stream(stream)>>>(m, f);
  NVF_LAUNCH_CHECK();
  if (bias_outs) {
    const int rc = nvf_finals_run_head_bias(ctx, (const float*)workspace + 3 * kLossMaxWG, bias_outs, m.n, stream);
    if (rc != NVF_OK) return rc;
  }
  return nvf_finals_run_focal(ctx, fm, (const float*)workspace, loss, 3, stream);
}

// ... and, with bias_outs (three pointers), the heads' bias gradients bias_outs[h][0] = sum of dls[h]: one partial per
// workgroup from the values it writes anyway, added by the deferred finals (or a launch of their own).
extern "C" int nvf_heads3_loss_bwd_data_bias(const float* const* ps, const float* const* gts, const float* const* dists,
                                             const float* alphas, const float* betas, const int* slots, float* loss,
                                             float* const* dls, const float* const* wbs, float* const* dxs,
                                             const float* const* masks, const int* cs, const int* ss, int batch,
                                             float* const* bias_outs, void* workspace, size_t workspace_bytes,
                                             NvfStepCtx* ctx, void* stream) {
  if (!ps || !gts || !dists || !alphas || !betas || !slots || !loss || !dls || !wbs || !dxs || !masks || !cs || !ss ||
      !workspace || batch <= 0)
    return NVF_EINVAL;
  if (bias_outs && (!bias_outs[0] || !bias_outs[1] || !bias_outs[2])) return NVF_EINVAL;
  if (workspace_bytes < nvf_reduce_workspace()) return NVF_EWORKSPACE;
  const int t = heads3_tuple(cs, ss);
  if (t == 0)
    return heads3_loss_bwd_data_t<HCfg<16, 8, 4, 8>, HCfg<8, 16, 4, 4>, HCfg<8, 32, 4, 8>>(
        ps, gts, dists, alphas, betas, slots, loss, dls, wbs, dxs, masks, batch, workspace, ctx, stream, bias_outs);
  if (t == 1)
    return heads3_loss_bwd_data_t<HCfg<32, 8, 4, 8>, HCfg<16, 16, 4, 4>, HCfg<16, 32, 4, 8>>(
        ps, gts, dists, alphas, betas, slots, loss, dls, wbs, dxs, masks, batch, workspace, ctx, stream, bias_outs);
  return NVF_EINVAL;
}

extern "C" int nvf_heads3_loss_bwd_data(const float* const* ps, const float* const* gts, const float* const* dists,
                                        const float* alphas, const float* betas, const int* slots, float* loss,
                                        float* const* dls, const float* const* wbs, float* const* dxs,
                                        const float* const* masks, const int* cs, const int* ss, int batch,
                                        void* workspace, size_t workspace_bytes, NvfStepCtx* ctx, void* stream) {
  return nvf_heads3_loss_bwd_data_bias(ps, gts, dists, alphas, betas, slots, loss, dls, wbs, dxs, masks, cs, ss, batch,
                                       nullptr, workspace, workspace_bytes, ctx, stream);
}

// ---- forward + loss + backward-data of the three heads in ONE launch ---------------------------------------------------
// The forward workgroups (lowest ids: they wait for nothing) write p with device-scope stores and count themselves into
// done[head][block]; a loss / backward-data workgroup of (head, block) issues its target / distance loads, waits until all
// of that block's forward workgroups have arrived, and reads p with device-scope loads.  Same bodies, same arithmetic:
// p, dls, dxs, the loss and the bias partials are the BITS of nvf_heads3_fwd + nvf_heads3_loss_bwd_data_bias.  Any number
// of resident slots is enough: consumers are dispatched after every producer (in-order dispatch), so a waiting consumer
// never holds a slot a producer needs.
constexpr int cmax2(int a, int b) { return a > b ? a : b; }

// MINW = waves per SIMD the register allocation must allow: 4 for the narrow heads (128 registers, no spill -- the loss
// workgroups then all fit beside the forward ones: 1024 slots for 592 + 800 workgroups at batch 16, the 800 in ONE round
// once the forward has left; with the forward body's 134 registers there were 768 slots and a second round of 32), 2 for
// the wide ones (212 registers)
template <class F0, class F1, class F2, class H0, class H1, class H2, int MINW>
__global__ __launch_bounds__(256, MINW) void heads3_fwd_loss_bwd_data_kernel(Heads3 mf, Heads3 m, Heads3Loss f, unsigned* done,
                                                                        int batch) {
  __shared__ __attribute__((aligned(16))) float
      smem[cmax2(cmax3(HFwdSmem<F0>::WORDS, HFwdSmem<F1>::WORDS, HFwdSmem<F2>::WORDS),
                 cmax3(HBwdSmem<H0>::WORDS, HBwdSmem<H1>::WORDS, HBwdSmem<H2>::WORDS))];
  int bid = blockIdx.x;
  const int nf = mf.n[0] + mf.n[1] + mf.n[2];
  if (bid < nf) {                   // forward: the small heads' few workgroups first (the longest chains of steps)
    if (bid < mf.n[0])
      head_fwd_body<F0, true>(mf.a[0], mf.w[0], mf.bias[0], mf.out[0], nullptr, nullptr, mf.act, bid, smem, done);
    else if (bid < mf.n[0] + mf.n[1])
      head_fwd_body<F1, true>(mf.a[1], mf.w[1], mf.bias[1], mf.out[1], nullptr, nullptr, mf.act, bid - mf.n[0], smem,
                              done + batch * kHeadFlagStride);
    else
      head_fwd_body<F2, true>(mf.a[2], mf.w[2], mf.bias[2], mf.out[2], nullptr, nullptr, mf.act, bid - mf.n[0] - mf.n[1],
                              smem, done + 2 * batch * kHeadFlagStride);
    return;
  }
  bid -= nf;
  if (bid < m.n[2])
    head_bwd_data_body<H2, true>(nullptr, m.w[2], nullptr, m.out[2], nullptr, m.mask[2], 0, bid, smem, &f.h[2]);
  else if (bid < m.n[2] + m.n[1])
    head_bwd_data_body<H1, true>(nullptr, m.w[1], nullptr, m.out[1], nullptr, m.mask[1], 0, bid - m.n[2], smem, &f.h[1]);
  else
    head_bwd_data_body<H0, true>(nullptr, m.w[0], nullptr, m.out[0], nullptr, m.mask[0], 0, bid - m.n[2] - m.n[1], smem,
                                 &f.h[0]);
}

template <class F0, class F1, class F2, class H0, class H1, class H2, int MINW>
static int heads3_fwd_loss_bwd_data_t(const float* const* xs, const float* const* ws, const float* const* biases,
                                      float* const* ps, int act, const float* const* gts, const float* const* dists,
                                      const float* alphas, const float* betas, const int* slots, float* loss,
                                      float* const* dls, const float* const* wbs, float* const* dxs,
                                      const float* const* masks, int batch, void* workspace, unsigned* flags,
                                      NvfStepCtx* ctx, void* stream, float* const* bias_outs) {
  static_assert(H0::NT == 256 && H1::NT == 256 && H2::NT == 256 && F0::NT == 256 && F1::NT == 256 && F2::NT == 256,
                "one workgroup size");
  Heads3 mf{}, m{};
  Heads3Loss f{};
  FocalMulti fm{};
  const int nprod[3] = {head_tiles<F0>(), head_tiles<F1>(), head_tiles<F2>()};
  const int ncons[3] = {head_tiles<H0>(), head_tiles<H1>(), head_tiles<H2>()};
  for (int h = 0; h < 3; ++h) {
    mf.n[h] = batch * nprod[h];
    m.n[h] = batch * ncons[h];
    if (!xs[h] || !ws[h] || !ps[h] || !gts[h] || !dls[h] || !wbs[h] || !dxs[h] || slots[h] < 0 || slots[h] > 2 ||
        m.n[h] > kLossMaxWG)
      return NVF_EINVAL;
    mf.a[h] = xs[h]; mf.w[h] = ws[h]; mf.bias[h] = biases[h]; mf.out[h] = ps[h];
    m.w[h] = wbs[h]; m.out[h] = dxs[h]; m.mask[h] = masks[h];
    f.h[h].p = ps[h]; f.h[h].gt = gts[h]; f.h[h].dist = dists[h]; f.h[h].dl = dls[h];
    f.h[h].part = (float*)workspace + slots[h] * kLossMaxWG;
    f.h[h].bias_part = bias_outs ? (float*)workspace + (3 + h) * kLossMaxWG : nullptr;
    f.h[h].alpha = alphas[h]; f.h[h].beta = betas[h];
    f.h[h].done = flags + h * batch * kHeadFlagStride; f.h[h].used = flags + (3 + h) * batch * kHeadFlagStride;
    f.h[h].nprod = (unsigned)nprod[h]; f.h[h].ncons = (unsigned)ncons[h];
    fm.nwg[slots[h]] = m.n[h];
  }
  mf.act = act;
  if (slots[0] == slots[1] || slots[0] == slots[2] || slots[1] == slots[2]) return NVF_EINVAL;
  const int grid = mf.n[0] + mf.n[1] + mf.n[2] + m.n[0] + m.n[1] + m.n[2];
  heads3_fwd_loss_bwd_data_kernel<F0, F1, F2, H0, H1, H2, MINW><<<grid, 256, 0, nvf_stream(stream)>>>(mf, m, f, flags, batch);
  NVF_LAUNCH_CHECK();
  if (bias_outs) {
    const int rc = nvf_finals_run_head_bias(ctx, (const float*)workspace + 3 * kLossMaxWG, bias_outs, m.n, stream);
    if (rc != NVF_OK) return rc;
  }
  return nvf_finals_run_focal(ctx, fm, (const float*)workspace, loss, 3, stream);
}

// nvf_heads3_fwd (ps[h] = act(conv(xs[h], ws[h]) + biases[h])) and nvf_heads3_loss_bwd_data_bias on those ps in one
// launch.  flags: 6 * batch * 64 unsigned words (one 256-byte line per counter), ZERO before the first call; every call leaves them zero (the last consumer of
// a block resets its counters), so one buffer serves every step of a stream -- not two streams at once.
extern "C" int nvf_heads3_fwd_loss_bwd_data(const float* const* xs, const float* const* ws, const float* const* biases,
                                            float* const* ps, int act, const float* const* gts,
                                            const float* const* dists, const float* alphas, const float* betas,
                                            const int* slots, float* loss, float* const* dls, const float* const* wbs,
                                            float* const* dxs, const float* const* masks, const int* cs, const int* ss,
                                            int batch, float* const* bias_outs, void* workspace, size_t workspace_bytes,
                                            uint32_t* flags, NvfStepCtx* ctx, void* stream) {
  if (!xs || !ws || !biases || !ps || !gts || !dists || !alphas || !betas || !slots || !loss || !dls || !wbs || !dxs ||
      !masks || !cs || !ss || !workspace || !flags || batch <= 0)
    return NVF_EINVAL;
  if (bias_outs && (!bias_outs[0] || !bias_outs[1] || !bias_outs[2])) return NVF_EINVAL;
  if (workspace_bytes < nvf_reduce_workspace()) return NVF_EWORKSPACE;
  const int t = heads3_tuple(cs, ss);
  if (t == 0)
#ifndef NVF_H2_CPS
#define NVF_H2_CPS 1      // channels per pipeline step of the big head's forward inside the merged launch (tuning)
#endif
    return heads3_fwd_loss_bwd_data_t<HPCfg<16, 8, 8, 8, 4>, HPCfg<8, 16, 8, 8, 2>, HPCfg<8, 32, 4, 8, NVF_H2_CPS>, HCfg<16, 8, 4, 8>,
                                      HCfg<8, 16, 4, 4>, HCfg<8, 32, 4, 8>, 4>(
        xs, ws, biases, ps, act, gts, dists, alphas, betas, slots, loss, dls, wbs, dxs, masks, batch, workspace, flags, ctx,
        stream, bias_outs);
  if (t == 1)
    return heads3_fwd_loss_bwd_data_t<HPCfg<32, 8, 8, 8, 4>, HPCfg<16, 16, 8, 8, 2>, HPCfg<16, 32, 4, 8, NVF_H2_CPS>,
                                      HCfg<32, 8, 4, 8>, HCfg<16, 16, 4, 4>, HCfg<16, 32, 4, 8>, 2>(
        xs, ws, biases, ps, act, gts, dists, alphas, betas, slots, loss, dls, wbs, dxs, masks, batch, workspace, flags, ctx,
        stream, bias_outs);
  return NVF_EINVAL;
}

// partial sums of the three weight gradients: slabs[h] receives nslabs[h] slabs of cs[h] * 27 floats (<= max_slabs)
template <class H0, class H1, class H2>
static int launch_heads3_wgrad(const float* const* dls, const float* const* xs, float* const* slabs, int batch,
                               int max_slabs, int* nslabs, void* stream) {
  static_assert(H0::NT == 256 && H1::NT == 256 && H2::NT == 256, "one workgroup size");
  Heads3 m{};
  const int items[3] = {batch * (H0::S / H0::TZ) * (H0::S / H0::TY), batch * (H1::S / H1::TZ) * (H1::S / H1::TY),
                        batch * (H2::S / H2::TZ) * (H2::S / H2::TY)};
  for (int h = 0; h < 3; ++h) {
    if (!dls[h] || !xs[h] || !slabs[h]) return NVF_EINVAL;
    m.a[h] = dls[h]; m.w[h] = xs[h]; m.out[h] = slabs[h];
    int n = items[h] < max_slabs ? items[h] : max_slabs;
    const int per = (items[h] + n - 1) / n;
    n = (items[h] + per - 1) / per;
    m.n[h] = n; m.items[h] = items[h]; m.per[h] = per;
    nslabs[h] = n;
  }
  const int grid = m.n[0] * H0::CSPLIT + m.n[1] * H1::CSPLIT + m.n[2] * H2::CSPLIT;
  heads3_wgrad_kernel<H0, H1, H2><<<grid, 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_heads3_wgrad_partial(const float* const* dls, const float* const* xs, float* const* slabs,
                                        const int* cs, const int* ss, int batch, int max_slabs, int* nslabs,
                                        void* stream) {
  if (!dls || !xs || !slabs || !cs || !ss || !nslabs || batch <= 0 || max_slabs <= 0) return NVF_EINVAL;
  const int t = heads3_tuple(cs, ss);
  // on the matrix cores (heads_wgrad_mfma.hip; NVF_HEADS_WG_VALU=1 keeps the VALU kernels: tuning)
  const bool valu = nvf_tune_int("NVF_HEADS_WG_VALU", 0) != 0;
  if ((t == 0 || t == 1) && !valu) {
    const int rc = nvf_heads3_wgrad_mfma_launch(dls, xs, slabs, t == 0, batch, max_slabs, nslabs, nvf_stream(stream));
    if (rc != 1) {
      if (rc != NVF_OK) return rc;
      NVF_LAUNCH_CHECK();
      return NVF_OK;
    }
  }
  // the big head in 2 x 8-row tiles: larger tiles (4 x 8, 2 x 16, 8 x 8) re-read less halo but were 10-30 % slower --
  // the kernel is bound by how many staging round trips are in flight, not by bytes
  if (t == 0)
    return launch_heads3_wgrad<HWCfg<16, 8, 4, 8, 4>, HWCfg<8, 16, 4, 4, 2>, HWCfg<8, 32, 2, 8, 2>>(
        dls, xs, slabs, batch, max_slabs, nslabs, stream);
  if (t == 1)
    return launch_heads3_wgrad<HWCfg<32, 8, 4, 8, 8>, HWCfg<16, 16, 4, 4, 4>, HWCfg<16, 32, 2, 8, 4>>(
        dls, xs, slabs, batch, max_slabs, nslabs, stream);
  return NVF_EINVAL;
}
