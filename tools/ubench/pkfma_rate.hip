// Micro-benchmark: issue rate of v_pk_fma_f32 on gfx950 with the operand forms the conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pkfma_rate.hip -o /tmp/pkfma_rate && /tmp/pkfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ void rate_kernel(float* out, long long* cycles, int iters, float s0, float s1) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)threadIdx.x, (float)i};
  f2 x = {(float)threadIdx.x * 0.001f, 0.5f};
  f2 w = {s0, s1};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0)        // VGPR x pair (op_sel broadcast lo), SGPR weight pair, VGPR acc
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w));
        else if (MODE == 1)   // all VGPR
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "v"(w));
        else if (MODE == 2)   // plain v_fma_f32 with SGPR
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "s"(s0));
        else if (MODE == 3)   // x pair packing with scalar broadcast (old form)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x), "s"(w));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  f2 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  int nb = 256, nt = 256 * waves_per_simd, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, nb * nt * 4);
  hipMalloc(&cyc, nb * (nt / 64) * 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) rate_kernel<MODE><<<nb, nt>>>(out, cyc, iters, 1.0001f, 0.9999f);
  hipEventRecord(a);
  rate_kernel<MODE><<<nb, nt>>>(out, cyc, iters, 1.0001f, 0.9999f);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(nb * (nt / 64));
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= h.size();
  double ninstr = (double)iters * 64;
  double flops = ninstr * 64 * (MODE == 2 ? 2 : 4) * (double)nb * (nt / 64);
  printf("%-34s waves/SIMD %d: %.2f memtime-ticks/instr/wave (100MHz ticks x clock ratio), wall %.3f ms, %.1f TFLOP/s\n",
         name, waves_per_simd, mean / ninstr, ms, flops / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("pk_fma v,s-pair,v (co pairs)", w);
    run<1>("pk_fma v,v,v", w);
    run<3>("pk_fma vpair,s-bcast,v (x pairs)", w);
    run<2>("v_fma_f32 v,s,v", w);
  }
  return 0;
}
