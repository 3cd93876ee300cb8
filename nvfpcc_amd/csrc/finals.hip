// The queue behind finals.h and the one launch that runs every queued final pass.
#include "step_ctx.h"

namespace {

__global__ __launch_bounds__(256) void finals_kernel(FinalsArgs a, int sum_blocks) {
  finals_plain_body(a, sum_blocks, blockIdx.x);
}

__global__ __launch_bounds__(256) void finals_tail_kernel(FinalsArgs a, int sum_blocks, NvfStepTail t, TailRanges rg) {
  finals_tail_body(a, sum_blocks, t, rg, blockIdx.x, gridDim.x);
}

}  // namespace

bool nvf_finals_push_focal(NvfStepCtx* ctx, const FocalMulti& m, const float* part, float* loss, int nterm) {
  if (!nvf_ctx_ok(ctx) || !ctx->deferring || ctx->args.has_f) return false;
  FinalsArgs& q = ctx->args;
  q.f = m; q.f_part = part; q.f_loss = loss; q.f_nterm = nterm; q.has_f = 1;
  return true;
}

bool nvf_finals_push_sums(NvfStepCtx* ctx, const MultiSumDesc& d, const float* part) {
  if (!nvf_ctx_ok(ctx) || !ctx->deferring || ctx->args.has_s) return false;
  FinalsArgs& q = ctx->args;
  q.s = d; q.s_part = part; q.has_s = 1;
  return true;
}

bool nvf_finals_push_metrics(NvfStepCtx* ctx, const float* part, float* out, const int* nwg, int nterm,
                              int accumulate) {
  if (!nvf_ctx_ok(ctx) || !ctx->deferring || ctx->args.has_m || nterm < 1 || nterm > 3) return false;
  FinalsArgs& q = ctx->args;
  q.m_part = part; q.m_out = out; q.m_nterm = nterm; q.m_accumulate = accumulate; q.has_m = 1;
  for (int t = 0; t < nterm; ++t) q.m_nwg[t] = nwg[t];
  return true;
}

bool nvf_finals_push_rate(NvfStepCtx* ctx, const WeightRateBatch& b, const float* part, const float* sigma, float* bits,
                          float* dsigma, float* dmu, const float* g_dev, float g_host) {
  if (!nvf_ctx_ok(ctx) || !ctx->deferring || ctx->args.has_r) return false;
  FinalsArgs& q = ctx->args;
  q.r = b; q.r_part = part; q.r_sigma = sigma; q.r_bits = bits; q.r_dsigma = dsigma;
  q.r_dmu = dmu; q.r_gdev = g_dev; q.r_ghost = g_host; q.has_r = 1;
  return true;
}

__global__ void head_bias_final_kernel(FinalsArgs a) { head_bias_final_body(a, threadIdx.x >> 6, threadIdx.x & 63); }

int nvf_finals_run_head_bias(NvfStepCtx* ctx, const float* part, float* const* outs, const int* n, void* stream) {
  FinalsArgs one{};
  FinalsArgs& q = (nvf_ctx_ok(ctx) && ctx->deferring && !ctx->args.has_hb) ? ctx->args : one;
  q.hb_part = part; q.has_hb = 1;
  for (int h = 0; h < 3; ++h) { q.hb_out[h] = outs[h]; q.hb_n[h] = n[h]; }
  if (&q != &one) return NVF_OK;
  head_bias_final_kernel<<<1, 192, 0, nvf_stream(stream)>>>(one);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void stem_gdn_final_kernel(StemGdnFinal f) { stem_gdn_final_body(f, threadIdx.x, 128); }

int nvf_finals_run_stem_gdn(NvfStepCtx* ctx, const StemGdnFinal& f, void* stream) {
  if (nvf_ctx_ok(ctx) && ctx->deferring && !ctx->args.has_g) {
    ctx->args.g = f; ctx->args.has_g = 1;
    return NVF_OK;
  }
  stem_gdn_final_kernel<<<1, 128, 0, nvf_stream(stream)>>>(f);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

__global__ void focal_final_kernel(FocalMulti m, const float* __restrict__ part, float* __restrict__ loss, int nterm) {
  focal_multi_final_body(m, part, loss, nterm, threadIdx.x);
}

int nvf_finals_run_focal(NvfStepCtx* ctx, const FocalMulti& m, const float* part, float* loss, int nterm, void* stream) {
  if (nvf_finals_push_focal(ctx, m, part, loss, nterm)) return NVF_OK;
  focal_final_kernel<<<1, 64 * nterm, 0, nvf_stream(stream)>>>(m, part, loss, nterm);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" size_t nvf_step_ctx_bytes(void) { return sizeof(NvfStepCtx); }

extern "C" int nvf_step_ctx_init(NvfStepCtx* ctx) {
  if (!ctx) return NVF_EINVAL;
  *ctx = NvfStepCtx{};
  ctx->magic = kStepCtxMagic;
  return NVF_OK;
}

// on != 0: every launch that takes this context keeps the DIRECT arithmetic (no Winograd form of a weight gradient,
// whatever nvf_step_ctx_set_wgrad_forms asked for) -- the summation order that reproduces the reference's training trajectory to 1e-7
// (tests/test_gpu_engine.py, the trajectory golden); 0 (the state after nvf_step_ctx_init): the faster forms.
extern "C" int nvf_step_ctx_set_direct(NvfStepCtx* ctx, int on) {
  if (!nvf_ctx_ok(ctx)) return NVF_EINVAL;
  ctx->direct_forms = on ? 1 : 0;
  return NVF_OK;
}

// Which reduced-multiplication forms the merged weight-gradient launches use when this context does NOT ask for the
// direct forms: conv2_zsplit (1..8, 0 = the default 1) z work items per conv2 plane range; conv1_wino != 0: conv1's gradient
// in the Winograd form too.  Both change the summation order (results agree to fp32 rounding), which is why they are the
// caller's per-context choice and not an environment switch of the library.
extern "C" int nvf_step_ctx_set_wgrad_forms(NvfStepCtx* ctx, int conv2_zsplit, int conv1_wino) {
  if (!nvf_ctx_ok(ctx) || conv2_zsplit < 0 || conv2_zsplit > 8) return NVF_EINVAL;
  ctx->wg_conv2_zsplit = conv2_zsplit;
  ctx->wg_conv1_wino = conv1_wino ? 1 : 0;
  return NVF_OK;
}

// Start queueing the final passes of nvf_focal_loss_multi, nvf_wgrad_reduce_multi_and_sums / nvf_multi_channel_sum
// and nvf_weight_rate_batch issued with this context (at most one of each kind; a second one is launched as usual).
// NVF_EINVAL: not an initialised context, or a queue is already open on it.
extern "C" int nvf_finals_begin(NvfStepCtx* ctx) {
  if (!nvf_ctx_ok(ctx) || ctx->deferring) return NVF_EINVAL;
  ctx->args = FinalsArgs{};
  ctx->deferring = 1;
  return NVF_OK;
}

// Drop whatever is queued and stop queueing (error paths).
extern "C" void nvf_finals_cancel(NvfStepCtx* ctx) {
  if (!nvf_ctx_ok(ctx)) return;
  ctx->args = FinalsArgs{};
  ctx->deferring = 0;
}

// Run the queued final passes in one launch on `stream` (the stream their partial passes ran on) and stop queueing.
// Nothing queued: no launch.
extern "C" int nvf_finals_flush_tail(NvfStepCtx* ctx, const NvfStepTail* tail, const int64_t* ranges, int nranges,
                                     void* stream) {
  if (!nvf_ctx_ok(ctx) || !tail || nranges < 0 || nranges > 16 || (nranges > 0 && !ranges)) return NVF_EINVAL;
  const NvfStepTail t = *tail;
  if (!t.p || !t.g || !t.m || !t.v || t.n <= 0) return NVF_EINVAL;
  if (t.acc && (!t.loss_terms || !t.lbits || !t.nbits || t.nnb <= 0 || t.nnb > 16)) return NVF_EINVAL;
  if (t.sched_rows && (!t.sched_buf || !t.sched_cursor || t.sched_words <= 0 || !t.done)) return NVF_EINVAL;
  TailRanges rg{};
  for (int r = 0; r < nranges; ++r) {
    if (ranges[2 * r] < 0 || ranges[2 * r + 1] > t.n || ranges[2 * r] > ranges[2 * r + 1]) return NVF_EINVAL;
    rg.lo[r] = (long)ranges[2 * r]; rg.hi[r] = (long)ranges[2 * r + 1];
  }
  rg.n = nranges;
  const FinalsArgs a = ctx->args;
  ctx->args = FinalsArgs{};
  ctx->deferring = 0;
  if (a.has_f && a.f_nterm > 3) return NVF_EINVAL;
  const int sum_blocks = finals_sum_blocks(a);
  finals_tail_kernel<<<2 + sum_blocks + 1, 256, 0, nvf_stream(stream)>>>(a, sum_blocks, t, rg);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

extern "C" int nvf_finals_flush(NvfStepCtx* ctx, void* stream) {
  if (!nvf_ctx_ok(ctx)) return NVF_EINVAL;
  const FinalsArgs a = ctx->args;
  ctx->args = FinalsArgs{};
  ctx->deferring = 0;
  if (!a.has_f && !a.has_s && !a.has_r && !a.has_g && !a.has_m && !a.has_hb) return NVF_OK;
  const int sum_blocks = finals_sum_blocks(a);
  finals_kernel<<<2 + sum_blocks + (a.has_m ? 1 : 0), 256, 0, nvf_stream(stream)>>>(a, sum_blocks);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
