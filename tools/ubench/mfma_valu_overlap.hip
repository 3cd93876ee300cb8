// Micro-benchmark: do f32 VALU instructions issue in the shadow of v_mfma_f32_16x16x4_f32 (32 cycles per MFMA per SIMD)?
// Each wave runs `iters` x 32 MFMAs on 8 independent accumulators with NV independent v_fma_f32 (own registers) behind
// every MFMA; one or two waves per SIMD.  Reported: shader cycles per MFMA (s_memtime of wave 0 / MFMAs per wave).
// If the VALU work hides, cycles per MFMA stay at 32 (one wave) / 64 per wave (two waves); if it does not, they grow
// by 4 x NV (one wave; a VALU instruction of a lone wave issues every 4 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void k(float* out, unsigned long long* stamps, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{(float)threadIdx.x, (float)i, 1.f, 2.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
  const float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.9999f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + n) & 7]) : "v"(b), "v"(a));
      }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
  if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}

template <int NV>
static void run(int wps, int iters) {
  const int nb = 256, nt = 256 * wps;
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)nb * nt * 4);
  hipMalloc(&st, (size_t)nb * 8);
  for (int i = 0; i < 2; ++i) k<NV><<<nb, nt>>>(out, st, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NV><<<nb, nt>>>(out, st, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nb);
  hipMemcpy(h.data(), st, nb * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto x : h) cyc += x; cyc /= nb;
  printf("| %d | %d | %.1f | %.3f |\n", NV, wps, cyc / (iters * 32.0), ms);
  hipFree(out); hipFree(st);
}

int main() {
  printf("| v_fma_f32 per MFMA | waves / SIMD | shader cycles per MFMA (per wave) | wall ms |\n|---|---|---|---|\n");
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, 2000); run<1>(w, 2000); run<2>(w, 2000); run<4>(w, 2000); run<6>(w, 2000); run<8>(w, 2000); run<12>(w, 2000);
  }
  return 0;
}
