// NvfStepCtx: the host-side state of ONE training step in flight -- the queue of deferred final passes (finals.h) and
// the queued latent tail (latent_tail.h).  The caller allocates nvf_step_ctx_bytes() of host memory, initialises it with
// nvf_step_ctx_init and passes it to every entry point that can defer work; the library itself keeps no such state,
// so several engines / streams / threads can drive the library at once, each with a context of its own
// (SURVEY.md section 8(b) "Threading / streams": no hidden global state, re-entrant).
#pragma once
#include "finals.h"
#include "latent_tail.h"
#include "stem_bwd.h"

constexpr uint32_t kStepCtxMagic = 0x4E564631u;   // "NVF1"

struct NvfStepCtx {
  uint32_t magic;
  int32_t deferring;
  int32_t tail_pending;
  int32_t direct_forms;   // nonzero: every launch given this context keeps the direct (non-Winograd) arithmetic
  int32_t wg_conv2_zsplit;   // conv2's Winograd weight gradient: z steps split over this many work items (0 / 1: one)
  int32_t wg_conv1_wino;     // nonzero: conv1's weight gradient in the Winograd form as well (another summation order)
  int32_t stem_pending;      // nonzero: `stem` is queued for the next five-gradient launch (nvf_stem_bwd_queue)
  FinalsArgs args;
  LatentTail tail;
  StemBwdJob stem;
};

static inline bool nvf_ctx_ok(const NvfStepCtx* c) { return c && c->magic == kStepCtxMagic; }
