// Weight gradients of the one-channel classifier heads (conv0_cls / conv1_cls / conv2_cls: Conv3d(C -> 1, k = 3,
// padding 1), utils/network.py:735-742, 677-687; autograd's bwd-weight of :741 / :687) on the matrix cores:
//
//   dW[c][t] = sum_{b, p} x[b, c, p] * dl[b, p - t + 1]          t = (kz, ky, kx), p = (z, y, x), dl = d loss / d logit
//
// One output channel gives an implicit GEMM nothing to put in its rows -- unless the roles are swapped: the INPUT
// position is the reduction index and the 27 taps are the columns (the scatter form):
//
//   D[c][t] += sum_{k=0..3} A[c][k] * B[k][t]      A[c][k] = x[c, z, y, x0 + k]          rows = C channels (8 or 16)
//                                                   B[k][t] = dl[z - kz + 1, y - ky + 1, x0 + k - kx + 1]   (0 outside)
//
// with K = four consecutive x positions and the taps in two 16-column tiles (27 of 32 columns useful).  Per four
// input positions and C channels that is two v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chains) fed by ONE A read and two
// B reads from LDS; the VALU kernel this replaces spent ~190 instructions per 108 FMAs (heads.hip).  The x tile has no
// halo (every element is read once), the dl tile a one-voxel halo of zeros.
//
// A workgroup (4 waves) walks items = (batch element, TZ planes, TY rows, all S columns); both tiles of the NEXT item are
// in registers (buffer loads: scalar descriptor + scalar offsets + one per-thread voffset) while this item's MFMAs issue.
// Its four waves split an item's K groups; their partial D tiles are added in wave order through LDS at the end and
// leave as ONE slab of C * 27 floats per workgroup (added in a fixed order by nvf_wgrad_reduce_multi): no atomics.
// A head with 32 channels (the wide decoder's first) is two groups of 16 rows: two workgroups per slab.
#include "heads_wgrad_mfma.h"

namespace {

template <class H0, class H1, class H2, int G0>
__global__ __launch_bounds__(256) void heads3_wgrad_mfma_kernel(HeadsW3 m) {
  __shared__ __attribute__((aligned(16))) float lds[hmax3(H0::SMEM, H1::SMEM, H2::SMEM)];
  heads3_wgrad_mfma_dispatch<H0, H1, H2, G0>(m, blockIdx.x, lds);
}

template <class H0, class H1, class H2, int G0>
int launch_heads3(const float* const* dls, const float* const* xs, float* const* slabs, int batch, int max_slabs,
                  int* nslabs, hipStream_t s) {
  HeadsW3 m{};
  const int rc = heads3_wgrad_mfma_fill<H0, H1, H2, G0>(m, dls, xs, slabs, batch, max_slabs, nslabs);
  if (rc != NVF_OK) return rc;
  heads3_wgrad_mfma_kernel<H0, H1, H2, G0><<<m.n[0] + m.n[1] + m.n[2], 256, 0, s>>>(m);
  return NVF_OK;
}

}  // namespace

// matrix-core form of nvf_heads3_wgrad_partial (same contract: slabs of cs[h] * 27 floats); narrow: the (16, 8, 8)-
// channel heads, otherwise the wide decoder's (32, 16, 16)
int nvf_heads3_wgrad_mfma_launch(const float* const* dls, const float* const* xs, float* const* slabs, int narrow,
                                 int batch, int max_slabs, int* nslabs, hipStream_t s) {
  if (narrow) return launch_heads3<HeadW0, HeadW1, HeadW2, 1>(dls, xs, slabs, batch, max_slabs, nslabs, s);
  return launch_heads3<HeadWw0, HeadWw1, HeadWw2, 2>(dls, xs, slabs, batch, max_slabs, nslabs, s);
}
