// The queue behind finals.h and the one launch that runs every queued final pass.
#include "finals.h"

namespace {

struct FinalsArgs {
  FocalMulti f;
  const float* f_part;
  float* f_loss;
  MultiSumDesc s;
  const float* s_part;
  WeightRateBatch r;
  const float* r_part;
  const float* r_sigma;
  float* r_bits;
  float* r_dsigma;
  float* r_dmu;
  const float* r_gdev;
  float r_ghost;
  StemGdnFinal g;
  int32_t f_nterm, has_f, has_s, has_r, has_g;
};

// workgroup 0: the focal terms (one wave each); workgroup 1: the weight-rate term (wave 0) and the stem's IGDN
// parameter gradients (waves 1-2); workgroups 2..: 64 bias channels each
__global__ __launch_bounds__(192) void finals_kernel(FinalsArgs a) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    if (a.has_f) focal_multi_final_body(a.f, a.f_part, a.f_loss, a.f_nterm, tid);
  } else if (blockIdx.x == 1) {
    if (a.has_r && tid < 64)
      weight_rate_batch_final_body(a.r, a.r_part, a.r_sigma, a.r_bits, a.r_dsigma, a.r_dmu, a.r_gdev, a.r_ghost, tid);
    if (a.has_g && tid >= 64) stem_gdn_final_body(a.g, tid - 64);
  } else if (a.has_s && tid < 64) {
    multi_channel_sum_final_body(a.s, a.s_part, ((int)blockIdx.x - 2) * 64 + tid);
  }
}

FinalsArgs g_args{};
bool g_deferring = false;

}  // namespace

bool nvf_finals_push_focal(const FocalMulti& m, const float* part, float* loss, int nterm) {
  if (!g_deferring || g_args.has_f) return false;
  g_args.f = m; g_args.f_part = part; g_args.f_loss = loss; g_args.f_nterm = nterm; g_args.has_f = 1;
  return true;
}

bool nvf_finals_push_sums(const MultiSumDesc& d, const float* part) {
  if (!g_deferring || g_args.has_s) return false;
  g_args.s = d; g_args.s_part = part; g_args.has_s = 1;
  return true;
}

bool nvf_finals_push_rate(const WeightRateBatch& b, const float* part, const float* sigma, float* bits, float* dsigma,
                          float* dmu, const float* g_dev, float g_host) {
  if (!g_deferring || g_args.has_r) return false;
  g_args.r = b; g_args.r_part = part; g_args.r_sigma = sigma; g_args.r_bits = bits; g_args.r_dsigma = dsigma;
  g_args.r_dmu = dmu; g_args.r_gdev = g_dev; g_args.r_ghost = g_host; g_args.has_r = 1;
  return true;
}

__global__ void stem_gdn_final_kernel(StemGdnFinal f) { stem_gdn_final_body(f, threadIdx.x); }

int nvf_finals_run_stem_gdn(const StemGdnFinal& f, void* stream) {
  if (f.c0 + f.c0 * f.c0 > 128) return NVF_EINVAL;
  if (g_deferring && !g_args.has_g) {
    g_args.g = f; g_args.has_g = 1;
    return NVF_OK;
  }
  stem_gdn_final_kernel<<<1, 128, 0, nvf_stream(stream)>>>(f);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void focal_final_kernel(FocalMulti m, const float* __restrict__ part, float* __restrict__ loss, int nterm) {
  focal_multi_final_body(m, part, loss, nterm, threadIdx.x);
}

int nvf_finals_run_focal(const FocalMulti& m, const float* part, float* loss, int nterm, void* stream) {
  if (nvf_finals_push_focal(m, part, loss, nterm)) return NVF_OK;
  focal_final_kernel<<<1, 64 * nterm, 0, nvf_stream(stream)>>>(m, part, loss, nterm);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// Start queueing the final passes of nvf_focal_loss_multi, nvf_wgrad_reduce_multi_and_sums / nvf_multi_channel_sum
// and nvf_weight_rate_batch (at most one of each kind; a second one is launched as usual).
extern "C" void nvf_finals_begin(void) {
  g_args = FinalsArgs{};
  g_deferring = true;
}

// Drop whatever is queued and stop queueing (error paths).
extern "C" void nvf_finals_cancel(void) {
  g_args = FinalsArgs{};
  g_deferring = false;
}

// Run the queued final passes in one launch on `stream` (the stream their partial passes ran on) and stop queueing.
// Nothing queued: no launch.
extern "C" int nvf_finals_flush(void* stream) {
  const FinalsArgs a = g_args;
  g_args = FinalsArgs{};
  g_deferring = false;
  if (!a.has_f && !a.has_s && !a.has_r && !a.has_g) return NVF_OK;
  const int sum_blocks = a.has_s ? (a.s.total_channels + 63) / 64 : 0;
  finals_kernel<<<2 + sum_blocks, 192, 0, nvf_stream(stream)>>>(a);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
