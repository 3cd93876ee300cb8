// All MFMA A-fragment packings of a step in ONE launch (the step engine re-packs the weights of conv1, conv2,
// up1 and up2 after every weight preparation; separate launches were a fifth of a microsecond-scale budget).
// Job kinds mirror nvf_pack_mfma_k4 (kind 0 / 2 = pair axis), nvf_pack_convT_mfma (kind 10) and
// nvf_pack_s2k5_mfma (kind 20): identical outputs, tested against them.
#include "nvf_common.h"

struct PackJobs {
  const float* src[8];
  float* dst[8];
  int32_t kind[8], c0[8], c1[8], total[8];
  int32_t n;
};

__global__ void pack_mfma_all_kernel(PackJobs m) {
  const int job = blockIdx.y;
  const float* __restrict__ w = m.src[job];
  float* __restrict__ wp = m.dst[job];
  const int kind = m.kind[job], total = m.total[job];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int lane = r % 64; r /= 64;
    const int i = lane & 15, k = lane >> 4;
    float v = 0.f;
    if (kind == 0 || kind == 2) {                    // 4^3 conv, 8 output channels: [g][ty][tx][tz][lane]
      const int KEZ = kind == 2 ? 5 : 4, KEX = kind == 0 ? 5 : 4;
      const int tz = r % KEZ; r /= KEZ;
      const int tx = r % KEX; r /= KEX;
      const int ty = r % 4, g = r / 4;
      const int cog = i >> 1, s = i & 1, ci = 4 * g + k;
      const int kz = kind == 2 ? tz - s : tz, kx = kind == 0 ? tx - s : tx;
      if (kz >= 0 && kz < 4 && kx >= 0 && kx < 4) v = w[(ci * 64 + (kz * 4 + ty) * 4 + kx) * 8 + cog];
    } else if (kind == 10) {                         // transposed conv forward: [g][75 class/tap fragments][lane]
      int f = r % 75;
      const int g = r / 75;
      int ez = 0, ey = 0;
      if (f >= 63) { ez = 1; ey = 1; f -= 63; }
      else if (f >= 45) { ez = 1; f -= 45; }
      else if (f >= 27) { ey = 1; f -= 27; }
      const int jx = f % 3, jy = (f / 3) % (3 - ey), jz = f / (3 * (3 - ey));
      const int co = i >> 1, ex = i & 1, ci = 4 * g + k;
      const int kz = ez + 2 * jz, ky = ey + 2 * jy, kx = ex + 2 * jx;
      if (kx < 5) v = w[(ci * 125 + (kz * 5 + ky) * 5 + kx) * 8 + co];
    } else {                                         // stride-2 gather (transposed conv backward-data)
      const int cog = m.c1[job], pair = cog == 8, KEX = pair ? 7 : 5;
      const int tx = r % KEX; r /= KEX;
      const int ky = r % 5; r /= 5;
      const int kz = r % 5, g = r / 5;
      const int ch = 4 * g + k;
      const int co = pair ? i >> 1 : i, kx = pair ? tx - 2 * (i & 1) : tx;
      if (kx >= 0 && kx < 5) v = w[(ch * 125 + (kz * 5 + ky) * 5 + kx) * cog + co];
    }
    wp[idx] = v;
  }
}

// kinds: 0 / 2 = nvf_pack_mfma_k4 with that pair axis (c0 = cin); 10 = nvf_pack_convT_mfma (c0 = cin);
// 20 = nvf_pack_s2k5_mfma (c0 = cig, c1 = cog)
extern "C" int nvf_pack_mfma_all(const float* const* srcs, float* const* dsts, const int* kinds, const int* c0s,
                                 const int* c1s, int n, void* stream) {
  if (!srcs || !dsts || !kinds || !c0s || !c1s || n <= 0 || n > 8) return NVF_EINVAL;
  PackJobs m{};
  for (int j = 0; j < n; ++j) {
    if (!srcs[j] || !dsts[j] || c0s[j] <= 0 || c0s[j] % 4) return NVF_EINVAL;
    m.src[j] = srcs[j]; m.dst[j] = dsts[j]; m.kind[j] = kinds[j]; m.c0[j] = c0s[j]; m.c1[j] = c1s[j];
    if (kinds[j] == 0 || kinds[j] == 2) m.total[j] = (c0s[j] / 4) * 4 * (kinds[j] == 0 ? 5 : 4) * (kinds[j] == 2 ? 5 : 4) * 64;
    else if (kinds[j] == 10) m.total[j] = (c0s[j] / 4) * 75 * 64;
    else if (kinds[j] == 20 && (c1s[j] == 8 || c1s[j] == 16)) m.total[j] = (c0s[j] / 4) * 25 * (c1s[j] == 8 ? 7 : 5) * 64;
    else return NVF_EINVAL;
  }
  m.n = n;
  pack_mfma_all_kernel<<<dim3(16, n), 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
