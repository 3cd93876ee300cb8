// Cook-Toom transforms on the points {0, 1, -1, 2, inf} shared by the Winograd kernels (conv_wino.hip: F(2, 4) per axis,
// backward-data; wgrad_wino.h: F(4, 2) per axis, weight gradient).  B^T depends on the points only; its rational factors
// (rows x 2, x 2, x 6, x -6, x 1) are folded into the transform of the OTHER operand (G, packed once per step) or into
// the final output transform, so the data transform is small integers.
#pragma once
#include "nvf_common.h"

// B^T d for one axis: (2, -1, -2, 1, 0 | 0, 2, 1, -1, 0 | 0, -2, 3, -1, 0 | 0, 1, 0, -1, 0 | 0, 2, -1, -2, 1): 9 operations
__device__ __forceinline__ void wino_bt(float d0, float d1, float d2, float d3, float d4, float& v0, float& v1, float& v2,
                                        float& v3, float& v4) {
  const float t13 = d1 - d3, t02 = d0 - d2, t24 = d2 - d4, t23 = d2 - d3;
  v0 = fmaf(2.f, t02, -t13);
  v1 = fmaf(2.f, d1, t23);
  v2 = fmaf(3.f, d2, fmaf(-2.f, d1, -d3));
  v3 = t13;
  v4 = fmaf(2.f, t13, -t24);
}
