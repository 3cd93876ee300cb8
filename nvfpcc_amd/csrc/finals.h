// Deferred "final" passes.  Several reductions of a training step are two launches: workgroups write partial sums,
// then one tiny launch adds them in a fixed order (the focal terms, the bias-gradient channel sums, the weight-rate
// term).  Every launch of a dependent chain costs ~5 us on this GPU whatever it does, and nothing inside the step
// reads those results, so between nvf_finals_begin() and nvf_finals_flush() the final passes are queued instead of
// launched and flush runs all of them in ONE launch.  Same device code either way: results are identical.
#pragma once
#include "nvf_common.h"

static const int kLossMaxWG = 1024;

// LowerBound floors of the GDN re-parametrisation (gdn_3d.py:40-60): beta = max(beta_hat, bound)^2 - pedestal
#define NVF_BETA_BOUND 1.0000072759311445e-03f /* sqrt(1e-6 + 2^-36) */
#define NVF_GAMMA_BOUND 3.814697265625e-06f   /* 2^-18 */

// ---- Adam fused into the launch that writes a gradient element (NvfAdamFuse, include/nvf_hip.h): nvf_step_tail's
// arithmetic for ONE element -- an element whose gradient is not finite keeps its parameter and moments.  Returns 1 for
// such an element (the caller counts them), 0 otherwise; gp outside [g_base, g_base + n): nothing happens.
__device__ __forceinline__ int adam_fused_elem(const NvfAdamFuse& a, const float* gp, float gi) {
  const long i = gp - a.g_base;
  if (i < 0 || i >= a.n) return 0;
  if (!(fabsf(gi) <= 3.402823466e38f)) return 1;
  const float step_size = a.coef_dev ? a.coef_dev[0] : a.coef0_host, bc2_sqrt = a.coef_dev ? a.coef_dev[1] : a.coef1_host;
  const float mi = a.m_base[i] * a.beta1 + gi * (1.f - a.beta1);
  const float vi = a.v_base[i] * a.beta2 + (gi * gi) * (1.f - a.beta2);
  a.m_base[i] = mi;
  a.v_base[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
  a.p_base[i] = a.p_base[i] - step_size * (mi / denom);
  return 0;
}
__device__ __forceinline__ NvfAdamFuse adam_fuse_of(const NvfStepTail& t) {
  NvfAdamFuse a{};
  a.g_base = t.g; a.p_base = t.p; a.m_base = t.m; a.v_base = t.v; a.n = t.n; a.coef_dev = t.coef_dev;
  a.coef0_host = t.coef0_host; a.coef1_host = t.coef1_host; a.beta1 = t.beta1; a.beta2 = t.beta2; a.eps = t.eps;
  a.bad_count = t.acc ? t.acc + 6 : nullptr;
  return a;
}

// the three focal terms of the objective (main output + two heads, NVFPCC.py:166-184)
struct FocalMulti {
  const float* p[3];
  const float* gt[3];
  const float* dist[3];
  float* dp[3];
  float alpha[3], beta[3];
  long n[3];
  int nwg[3];
};

// per-channel sums of up to 12 tensors [batch, c, spatial] (the bias gradients)
struct MultiSumDesc {
  const float* x[12];
  float* out[12];
  int32_t c[12], spatial[12], chan_base[12];
  int32_t ntensors, batch, total_channels, nchunk;
};

// rate term of all quantised kernels of the decoder
struct WeightRateBatch {
  const float* kernel[8];
  float* dk[8];
  int32_t n[8];
  int32_t first_wg[9];   // workgroups [first_wg[l], first_wg[l+1]) belong to layer l
  int32_t nlayers, chunk;
};

// one focal term (utils/loss.py:61-72, 94-111) and its gradient w.r.t. p (chain_sigmoid: w.r.t. the logit)
// Contraction into FMAs is switched off: the function is inlined into several kernels and every one of them must
// produce the same bits.  g scales the gradient (1 = as is).
__device__ __forceinline__ float focal_elem(float pv, float gv, float dv, bool has_dist, float a1, float a0, float beta,
                                            int chain_sigmoid, float& dp, float g = 1.f) {
#pragma clang fp contract(off)
  const bool occ = gv != 0.f;
  const float F = occ ? pv : 1.f - pv;
  const float at = occ ? a1 : a0;
  float w = 1.f;
  if (has_dist) w = dv + (occ ? beta : 0.f);
  const float Fc = F < 1e-9f ? 1e-9f : F;   // torch.clamp(min=1e-9): same values as fmaxf, but a NaN stays a NaN
                                            // (loss.py:68, 105; it is what NVFPCC.py:199's NaN check looks for)
  const float om = 1.f - Fc;
  const float lg = logf(Fc);
  float d = 0.f;
  if (F >= 1e-9f) d = -at * w * (-2.f * om * lg + om * om / Fc);
  const float dd = g * (occ ? d : -d);
  dp = chain_sigmoid ? dd * ((1.f - pv) * pv) : dd;
  return -1.f * at * (om * om) * w * lg;
}

// one wave per loss term: lanes take the partials 64 apart (ascending), then a fixed-order wave sum
__device__ __forceinline__ void focal_multi_final_body(const FocalMulti& m, const float* __restrict__ part,
                                                       float* __restrict__ loss, int nterm, int tid) {
  const int t = tid >> 6, lane = tid & 63;
  if (t >= nterm) return;
  float s = 0.f;
  for (int g = lane; g < m.nwg[t]; g += 64) s += part[t * kLossMaxWG + g];
  s = nvf_wave_sum(s);
  if (lane == 0) loss[t] = s;
}

__device__ __forceinline__ int multi_channel_sum_final_body(const MultiSumDesc& d, const float* __restrict__ part,
                                                            int gch, const NvfAdamFuse* adam = nullptr) {
  if (gch >= d.total_channels) return 0;
  int t = 0;
  while (t + 1 < d.ntensors && gch >= d.chan_base[t + 1]) ++t;
  float s = 0.f;
  for (int g = 0; g < d.nchunk; ++g) s += part[(size_t)g * d.total_channels + gch];
  float* o = d.out[t] + (gch - d.chan_base[t]);
  *o = s;
  return adam ? adam_fused_elem(*adam, o, s) : 0;
}

// one wave: lanes take the chunk partials 64 apart (ascending), then a fixed-order wave sum
__device__ __forceinline__ int weight_rate_batch_final_body(const WeightRateBatch& b, const float* __restrict__ part,
                                                            const float* __restrict__ sigma, float* __restrict__ bits,
                                                            float* __restrict__ dsigma, float* __restrict__ dmu,
                                                            const float* __restrict__ g_dev, float g_host, int lane,
                                                            const NvfAdamFuse* adam = nullptr) {
  const float g = g_host * (g_dev ? g_dev[0] : 1.f);
  // lane i holds the partials of workgroups i, i + 64, ... (at most 512 of them): every load is in flight before the
  // first sum -- one memory round trip instead of one per layer
  constexpr int SLOTS = 8;
  const int nwg = b.first_wg[b.nlayers];
  float pb[SLOTS], ps[SLOTS], pm[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int wg = lane + 64 * i;
    const bool ok = wg < nwg;
    pb[i] = ok ? part[3 * wg] : 0.f;
    ps[i] = ok ? part[3 * wg + 1] : 0.f;
    pm[i] = ok ? part[3 * wg + 2] : 0.f;
  }
  float acc_s = 0.f, acc_m = 0.f;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) { acc_s += ps[i]; acc_m += pm[i]; }
  for (int l = 0; l < b.nlayers; ++l) {
    float tb = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int wg = lane + 64 * i;
      tb += (wg >= b.first_wg[l] && wg < b.first_wg[l + 1]) ? pb[i] : 0.f;
    }
    tb = nvf_wave_sum(tb);
    if (lane == 0) bits[l] = tb;
  }
  acc_s = nvf_wave_sum(acc_s);
  acc_m = nvf_wave_sum(acc_m);
  if (lane != 0) return 0;
  const float sraw = sigma[0];
  const float sgn = sraw > 0.f ? 1.f : (sraw < 0.f ? -1.f : 0.f);
  int bad = 0;
  if (dsigma) {
    dsigma[0] = g * acc_s * sgn;
    if (adam) bad += adam_fused_elem(*adam, dsigma, g * acc_s * sgn);
  }
  if (dmu) {
    dmu[0] = g * acc_m;
    if (adam) bad += adam_fused_elem(*adam, dmu, g * acc_m);
  }
  return bad;
}

// IGDN parameter gradients of the fused stem from its slabs (column p: p < c0 is d beta_p, the rest d gamma): fixed-order
// sum, re-parametrisation chain rule, LowerBound rule (gdn_3d.py:72-95 backward)
struct StemGdnFinal {
  const float* slab_gdn;
  const float* beta_hat;
  const float* gamma_hat;
  float* dbeta_hat;
  float* dgamma_hat;
  int32_t nslab, c0;
};

// thread p0 of `stride` takes the columns p0, p0 + stride, ... (c0 + c0^2 columns: 72 narrow, 272 wide)
__device__ __forceinline__ int stem_gdn_final_body(const StemGdnFinal& f, int p0, int stride,
                                                   const NvfAdamFuse* adam = nullptr) {
  const int ncol = f.c0 + f.c0 * f.c0;
  int bad = 0;
  for (int p = p0; p < ncol; p += stride) {
    float s = 0.f;
    for (int g = 0; g < f.nslab; ++g) s += f.slab_gdn[(size_t)g * ncol + p];
    float* o;
    float v;
    if (p < f.c0) {
      const float h = f.beta_hat[p];
      const float g = s * 2.f * fmaxf(h, NVF_BETA_BOUND);
      o = f.dbeta_hat + p;
      v = (h >= NVF_BETA_BOUND || g < 0.f) ? g : 0.f;
    } else {
      const float h = f.gamma_hat[p - f.c0];
      const float g = s * 2.f * fmaxf(h, NVF_GAMMA_BOUND);
      o = f.dgamma_hat + (p - f.c0);
      v = (h >= NVF_GAMMA_BOUND || g < 0.f) ? g : 0.f;
    }
    *o = v;
    if (adam) bad += adam_fused_elem(*adam, o, v);
  }
  return bad;
}

// metric partials of term t (rows of 6) -> out[6 t + k], k = tid % 6, in row order (the arithmetic of finalize_partials)
__device__ __forceinline__ void metrics_final_body(const float* __restrict__ part, float* __restrict__ out,
                                                   const int32_t* nwg, int nterm, int accumulate, int tid) {
  if (tid >= 6 * nterm) return;
  const int t = tid / 6, k = tid - 6 * t;
  float s = 0.f;
  for (int g0 = 0; g0 < nwg[t]; g0 += 16) {        // 16 loads in flight, added in ascending row order
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = g0 + i < nwg[t] ? part[((size_t)t * kLossMaxWG + g0 + i) * 6 + k] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[tid] = accumulate ? out[tid] + s : s;
}

// Everything one finals launch needs (at most one job of each kind)
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
  const float* m_part;   // metrics (nvf_metrics / nvf_metrics3): per term t, m_nwg[t] rows of 6 partial sums at row
  float* m_out;          // t * kLossMaxWG, summed in row order into m_out[6 t .. 6 t + 5]
  int32_t m_nwg[3], m_nterm, m_accumulate;
  int32_t f_nterm, has_f, has_s, has_r, has_g, has_m;
  const float* hb_part;  // the heads' bias gradients (nvf_heads3_loss_bwd_data_bias): head h = the sum of its logit
  float* hb_out[3];      // gradient, left as hb_n[h] per-workgroup partials at hb_part + h * kLossMaxWG
  int32_t hb_n[3], has_hb;
};

// one wave per head: lanes take the partials 64 apart (ascending), then a fixed-order wave sum (as the focal terms)
__device__ __forceinline__ int head_bias_final_body(const FinalsArgs& a, int h, int lane, const NvfAdamFuse* adam = nullptr) {
  float s = 0.f;
  for (int g = lane; g < a.hb_n[h]; g += 64) s += a.hb_part[h * kLossMaxWG + g];
  s = nvf_wave_sum(s);
  if (lane != 0) return 0;
  *a.hb_out[h] = s;
  return adam ? adam_fused_elem(*adam, a.hb_out[h], s) : 0;
}

// workgroups of a finals launch that run the bias sums (64 channels each); the first one also hosts the heads' sums
__host__ __device__ inline int finals_sum_blocks(const FinalsArgs& a) {
  const int n = a.has_s ? (a.s.total_channels + 63) / 64 : 0;
  return n > 0 ? n : (a.has_hb ? 1 : 0);
}

// The queued final passes without the optimiser (nvf_finals_flush; data-parallel steps, whose gradients are not final
// before the all-reduce).  Workgroup 0: the focal terms (one wave each); workgroup 1: the weight-rate term (wave 0) and the
// stem's IGDN parameter gradients (waves 1-2); workgroups 2 .. 2 + sum_blocks - 1: 64 bias channels each (the first one also
// the heads' bias sums, waves 1-3); with metrics queued, one more workgroup: the metric partials in row order.  Workgroups
// of >= 256 threads.
__device__ __forceinline__ void finals_plain_body(const FinalsArgs& a, int sum_blocks, int bid) {
  const int tid = threadIdx.x;
  if (a.has_hb && bid == 2 && tid >= 64 && tid < 256) head_bias_final_body(a, (tid >> 6) - 1, tid & 63);
  if (bid == 2 + sum_blocks) {
    if (a.has_m) metrics_final_body(a.m_part, a.m_out, a.m_nwg, a.m_nterm, a.m_accumulate, tid);
  } else if (bid == 0) {
    if (a.has_f) focal_multi_final_body(a.f, a.f_part, a.f_loss, a.f_nterm, tid);
  } else if (bid == 1) {
    if (a.has_r && tid < 64)
      weight_rate_batch_final_body(a.r, a.r_part, a.r_sigma, a.r_bits, a.r_dsigma, a.r_dmu, a.r_gdev, a.r_ghost, tid);
    if (a.has_g && tid >= 64 && tid < 192) stem_gdn_final_body(a.g, tid - 64, 128);
  } else if (a.has_s && tid < 64) {
    multi_channel_sum_final_body(a.s, a.s_part, (bid - 2) * 64 + tid);
  }
}

struct TailRanges { long lo[16], hi[16]; int n; };

// finals + the tail of a single-GPU training step (nvf_finals_flush_tail).  Workgroup 0 runs the passes the epoch
// statistics read -- the focal terms (waves 0-2), the weight-rate term (wave 3), then the metric counts -- and adds the
// statistics itself, so no sum crosses a workgroup; workgroup 1 the stem's IGDN parameter gradients; workgroups
// 2 .. 2 + sum_blocks - 1 the bias sums; the last one the index ranges whose gradients earlier launches wrote.  Every
// gradient element gets its Adam update from the thread that produced (or owns) it.  The last workgroup to arrive --
// all of them have read the step buffer's coefficients by then -- copies the next schedule row over the step buffer.
// bid / nblocks: this workgroup's index among, and the number of, workgroups that run this body or -- in a launch shared
// with other work -- take part in the arrival count (every workgroup of the launch must then call finals_tail_arrive).
// Workgroups may be larger than 256 threads: threads >= 256 only take part in the barriers.
__device__ __forceinline__ void finals_tail_arrive(const NvfStepTail& t, unsigned long long cur, int nblocks) {
  if (!t.sched_rows) return;
  const int tid = threadIdx.x;
  __shared__ int last;
  int64_t next_word = 0;
  if (tid < t.sched_words) next_word = t.sched_rows[cur * (unsigned long long)t.sched_words + tid];
  // (no device-scope fence: it writes the L2 back, +10 us over ~800 workgroups.  The callers' barrier before this call
  // waits for the workgroup's loads -- the step buffer's words among them -- so they precede the arrival.)
  if (tid == 0) last = atomicAdd(t.done, 1u) == (unsigned)nblocks - 1;
  __syncthreads();
  if (!last) return;
  if (tid < t.sched_words) t.sched_buf[tid] = next_word;
  for (int w = tid + blockDim.x; w < t.sched_words; w += blockDim.x)
    t.sched_buf[w] = t.sched_rows[cur * (unsigned long long)t.sched_words + w];
  if (tid == 0) { t.sched_cursor[0] = cur + 1; t.done[0] = 0u; }
}

__device__ __forceinline__ void finals_tail_body(const FinalsArgs& a, int sum_blocks, const NvfStepTail& t,
                                                 const TailRanges& rg, int bid, int nblocks) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const NvfAdamFuse ad = adam_fuse_of(t);
  // the schedule cursor is fetched now (it only moves at the very end of this launch) and the row it points at right
  // before the arrival counter is bumped: whichever workgroup turns out to be the last already holds the row when it
  // learns so -- one round trip at the end instead of three dependent ones
  const unsigned long long cur = t.sched_rows ? t.sched_cursor[0] : 0ull;   // in flight under the work below
  int bad = 0;
  if (bid == 0) {
    if (a.has_f) focal_multi_final_body(a.f, a.f_part, a.f_loss, a.f_nterm, tid);
    if (a.has_r && wave == 3)
      bad += weight_rate_batch_final_body(a.r, a.r_part, a.r_sigma, a.r_bits, a.r_dsigma, a.r_dmu, a.r_gdev, a.r_ghost,
                                          lane, &ad);
    if (a.has_m) metrics_final_body(a.m_part, a.m_out, a.m_nwg, a.m_nterm, a.m_accumulate, tid);
  } else if (bid == 1) {
    if (a.has_g && tid < 256) bad += stem_gdn_final_body(a.g, tid, 256, &ad);
  } else if (bid < 2 + sum_blocks) {
    if (a.has_s && tid < 64) bad += multi_channel_sum_final_body(a.s, a.s_part, (bid - 2) * 64 + tid, &ad);
    if (a.has_hb && bid == 2 && wave >= 1 && wave <= 3) bad += head_bias_final_body(a, wave - 1, lane, &ad);
  } else {
    for (int r = 0; r < rg.n; ++r)
      for (long i = rg.lo[r] + tid; i < rg.hi[r]; i += blockDim.x) bad += adam_fused_elem(ad, t.g + i, t.g[i]);
  }
  if (t.acc) {
    const unsigned long long any = __ballot(bad != 0);
    // a lane's count is 0 .. a few: add the lanes' counts (integer-valued floats: order-free)
    float c = (float)bad;
    c = nvf_wave_sum(c);
    if (any && lane == 0) atomicAdd(t.acc + 6, c);
  }
  __syncthreads();                                            // workgroup 0: the sums it wrote are visible to its thread 0
  if (bid == 0 && tid == 0 && t.acc) {
    int bad_terms = 0;
    for (int k = 0; k < 3; ++k) {
      const float x = t.loss_terms[k];
      t.acc[k] += x;
      bad_terms += !(fabsf(x) <= 3.402823466e38f);
    }
    const float bl = t.lbits[0] * (t.inv_npts_dev ? t.inv_npts_dev[0] : t.inv_npts_host);
    float nb = 0.f;
    for (int l = 0; l < t.nnb; ++l) nb += t.nbits[l];
    nb *= t.nbits_scale;
    t.acc[3] += bl;
    t.acc[4] += nb;
    bad_terms += !(fabsf(bl) <= 3.402823466e38f) + !(fabsf(nb) <= 3.402823466e38f);
    t.acc[5] += (float)bad_terms;
    t.acc[7] += 1.f;
    if (t.counts) {
      for (int k = 0; k < 3; ++k) {
        t.acc[8 + 2 * k] += t.counts[6 * k] / t.counts[6 * k + 1];
        t.acc[9 + 2 * k] += t.counts[6 * k + 2] / t.counts[6 * k + 3];
      }
      t.acc[14] += t.counts[4];
      t.acc[15] += t.counts[5];
    }
  }
  finals_tail_arrive(t, cur, nblocks);
}

// Queue (finals.hip), held in the caller's NvfStepCtx (step_ctx.h); ctx == nullptr means "nothing is deferred".  A push
// returns false when nothing is being deferred or a job of that kind is already waiting: the caller then launches its
// own final pass as usual.
struct NvfStepCtx;
bool nvf_finals_push_focal(NvfStepCtx* ctx, const FocalMulti& m, const float* part, float* loss, int nterm);
// queue the stem's IGDN final pass or, when nothing is being deferred, launch it on `stream` (any c0)
int nvf_finals_run_stem_gdn(NvfStepCtx* ctx, const StemGdnFinal& f, void* stream);
// queue the focal final pass or, when nothing is being deferred, launch it on `stream`
int nvf_finals_run_focal(NvfStepCtx* ctx, const FocalMulti& m, const float* part, float* loss, int nterm, void* stream);
bool nvf_finals_push_sums(NvfStepCtx* ctx, const MultiSumDesc& d, const float* part);
// queue the heads' bias sums or, when nothing is being deferred, launch them on `stream`
int nvf_finals_run_head_bias(NvfStepCtx* ctx, const float* part, float* const* outs, const int* n, void* stream);
bool nvf_finals_push_metrics(NvfStepCtx* ctx, const float* part, float* out, const int* nwg, int nterm, int accumulate);
bool nvf_finals_push_rate(NvfStepCtx* ctx, const WeightRateBatch& b, const float* part, const float* sigma, float* bits,
                          float* dsigma, float* dmu, const float* g_dev, float g_host);
