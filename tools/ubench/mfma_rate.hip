// Micro-benchmark: what rate v_mfma_f32_16x16x4_f32 sustains on a whole MI355X, and at which shader clock -- the
// "attainable" roof the step's matrix-core kernels are priced against in DESIGN.md section 9 (the data-sheet peak,
// 157.3 TFLOP/s, is 256 CUs x 256 FLOP/clk x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Every wave issues back-to-back MFMAs on eight independent accumulators (no LDS, no memory); one, two or four waves
// per SIMD on every CU; short (~0.1 ms, the length of the step's big launches) and long (~5 ms) runs.
// Shader clock = s_memtime ticks (shader cycles) / s_memrealtime ticks (100 MHz) measured by wave 0 of every workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operands in registers.  MODE 1: the B operand of every MFMA is a fresh conflict-free ds_read_b32 (the inner
// loop of the convolution kernels: ~0.4-1 LDS reads per MFMA).
template <int MODE>
__global__ void mfma_rate_kernel(float* out, unsigned long long* stamps, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.9999f + i * 1e-7f;
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{(float)threadIdx.x, (float)i, 1.f, 2.f};
  const float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.9999f;
  const float* lp = lds + (threadIdx.x & 63);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float bb = MODE == 1 ? lp[((it * 32 + r * 8 + i) & 127) * 64] : b;
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static void run(int waves_per_simd, int iters) {
  int ncu = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ncu = p.multiProcessorCount;
  const int nb = ncu, nt = 256 * waves_per_simd;       // one workgroup per CU, 4 SIMDs x waves_per_simd waves
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)nb * nt * 4);
  hipMalloc(&st, (size_t)nb * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 3; ++k) mfma_rate_kernel<MODE><<<nb, nt>>>(out, st, iters);
  hipEventRecord(e0);
  mfma_rate_kernel<MODE><<<nb, nt>>>(out, st, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * nb);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < nb; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  cyc /= nb; rt /= nb;
  const double mfma_per_wave = (double)iters * 32, ghz = cyc / (rt * 10.0);            // rt ticks are 10 ns
  const double flops = mfma_per_wave * 2048.0 * nb * (nt / 64);
  // (a wave's own stamps span only ITS MFMAs -- the SIMD serves its oldest wave first -- so throughput is priced on the
  // wall time of the launch, which includes ~4 us of launch ramp and drain)
  printf("| %s | %d | %d | %.3f | %.2f | %.1f | %.2f |\n", MODE ? "B from LDS" : "registers", waves_per_simd, iters * 32, ms, ghz,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
  hipFree(out); hipFree(st);
}

int main() {
  printf("| operands | waves / SIMD | MFMAs per wave | wall ms (HIP events) | shader clock GHz | TFLOP/s | of 157.3 |\n"
         "|---|---|---|---|---|---|---|\n");
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, 100 / w);        // ~3200 MFMAs per SIMD: the size of the step's big launches
    run<0>(w, 4000 / w);       // ~128 000 MFMAs per SIMD: a long run
    run<1>(w, 100 / w);
    run<1>(w, 4000 / w);
  }
  return 0;
}
