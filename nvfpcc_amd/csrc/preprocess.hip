// Ground-truth occupancy / distance grids of the leaf blocks.
//
// Reference: util_get_grids.py:19-46 queries an open3d KD-tree once per voxel in a Python loop
// (917 x 32 768 = 30 M queries).  Here one workgroup owns an 8^3 sub-tile of one 32^3 block and
// scans the points of the block itself and of the occupied blocks within +-2 block steps (a voxel's
// nearest point is at most sqrt(3)*31 < 64 away because its own block holds a point), staging point
// chunks in LDS.  A neighbour block is skipped when the box-to-box lower bound already exceeds the
// worst best-so-far distance of the workgroup.  Output is the exact integer squared distance
// (coordinates are 10-bit integers), so sqrt on the host reproduces the reference's float64 values.
#include "nvf_common.h"

__global__ __launch_bounds__(512) void nearest_dist2_kernel(const int32_t* __restrict__ pts,      // [P,3] block-sorted
                                                            const int32_t* __restrict__ blk_off,  // [N+1]
                                                            const int32_t* __restrict__ origins,  // [N,3]
                                                            const int32_t* __restrict__ nb_off,   // [N+1]
                                                            const int32_t* __restrict__ nb_idx,   // candidates
                                                            int32_t* __restrict__ d2out /* [N,32,32,32] */) {
  __shared__ int sx[512], sy[512], sz[512];
  __shared__ int wmax[8];
  const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int tx0 = (tile >> 4) * 8, ty0 = ((tile >> 2) & 3) * 8, tz0 = (tile & 3) * 8;   // grid dims are (x, y, z)
  const int lx = tid >> 6, ly = (tid >> 3) & 7, lz = tid & 7;
  const int ox = origins[3 * b], oy = origins[3 * b + 1], oz = origins[3 * b + 2];
  const int vx = ox + tx0 + lx, vy = oy + ty0 + ly, vz = oz + tz0 + lz;
  int best = 0x7fffffff;
  int wg_worst = 0x7fffffff;
  for (int k = nb_off[b]; k < nb_off[b + 1]; ++k) {
    const int nb = nb_idx[k];
    // lower bound between this sub-tile's box and the neighbour block's box
    const int nx = origins[3 * nb], ny = origins[3 * nb + 1], nz = origins[3 * nb + 2];
    const int gx = max(0, max(nx - (ox + tx0 + 7), (ox + tx0) - (nx + 31)));
    const int gy = max(0, max(ny - (oy + ty0 + 7), (oy + ty0) - (ny + 31)));
    const int gz = max(0, max(nz - (oz + tz0 + 7), (oz + tz0) - (nz + 31)));
    if (gx * gx + gy * gy + gz * gz >= wg_worst) continue;     // wave-uniform decision
    const int p0 = blk_off[nb], p1 = blk_off[nb + 1];
    for (int c0 = p0; c0 < p1; c0 += 512) {
      const int n = min(512, p1 - c0);
      __syncthreads();
      if (tid < n) {
        sx[tid] = pts[3 * (c0 + tid)];
        sy[tid] = pts[3 * (c0 + tid) + 1];
        sz[tid] = pts[3 * (c0 + tid) + 2];
      }
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        const int dx = sx[j] - vx, dy = sy[j] - vy, dz = sz[j] - vz;
        best = min(best, dx * dx + dy * dy + dz * dz);
      }
    }
    // worst best-so-far of the workgroup (same value in every thread)
    int m = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    m = wmax[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = max(m, wmax[w]);
    wg_worst = m;
  }
  d2out[(((size_t)b * 32 + tx0 + lx) * 32 + ty0 + ly) * 32 + tz0 + lz] = best;
}

extern "C" int nvf_nearest_dist2(const int32_t* pts, const int32_t* blk_off, const int32_t* origins,
                                 const int32_t* nb_off, const int32_t* nb_idx, int32_t* d2out, int nblocks,
                                 void* stream) {
  if (!pts || !blk_off || !origins || !nb_off || !nb_idx || !d2out || nblocks <= 0) return NVF_EINVAL;
  nearest_dist2_kernel<<<dim3(64, nblocks), 512, 0, nvf_stream(stream)>>>(pts, blk_off, origins, nb_off, nb_idx,
                                                                          d2out);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}
