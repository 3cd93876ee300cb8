// All MFMA A-fragment packings of a step in ONE launch (the step engine re-packs the weights of conv1, conv2,
// up1 and up2 after every weight preparation; separate launches were a fifth of a microsecond-scale budget).
// Job kinds mirror nvf_pack_mfma_k4 (kind 0 / 2 = pair axis), nvf_pack_convT_mfma (kind 10) and
// nvf_pack_s2k5_mfma (kind 20): identical outputs, tested against them.
#include "pack_mfma.h"

__global__ void pack_mfma_all_kernel(PackJobs m) {
  pack_mfma_body(m, blockIdx.y, blockIdx.x, gridDim.x, [&](int job, int i) { return m.src[job][i]; });
}

// kinds: 0 / 2 = nvf_pack_mfma_k4 with that pair axis (c0 = cin); 10 = nvf_pack_convT_mfma (c0 = cin);
// 20 = nvf_pack_s2k5_mfma (c0 = cig, c1 = cog)
extern "C" int nvf_pack_mfma_all(const float* const* srcs, float* const* dsts, const int* kinds, const int* c0s,
                                 const int* c1s, int n, void* stream) {
  if (!srcs) return NVF_EINVAL;
  PackJobs m{};
  const int rc = pack_jobs_desc(srcs, dsts, kinds, c0s, c1s, n, m);
  if (rc != NVF_OK) return rc;
  for (int j = 0; j < n; ++j)
    if (!srcs[j]) return NVF_EINVAL;
  pack_mfma_all_kernel<<<dim3(16, n), 256, 0, nvf_stream(stream)>>>(m);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
