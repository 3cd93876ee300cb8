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

// The same transform on the packed fp32 pipe (v_pk_add_f32 / v_pk_fma_f32: two fp32 operations per lane and
// instruction).  Measured on gfx950 (profiles/r04_mfma_valu_overlap.md): vector instructions do not hide behind fp32 MFMAs,
// every one of them is SIMD time, so the transforms are written for instruction count.  Left to itself the compiler's SLP
// pass packs arbitrary pairs and pays ~48 register moves per 5 x 5 window (115 instructions); below: 57.
typedef float wino_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ wino_f2 wino_fma2(wino_f2 a, wino_f2 b, wino_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// B^T d for TWO independent columns at once (the .x and .y halves): 9 packed operations
__device__ __forceinline__ void wino_bt2(wino_f2 d0, wino_f2 d1, wino_f2 d2, wino_f2 d3, wino_f2 d4, wino_f2& v0, wino_f2& v1,
                                         wino_f2& v2, wino_f2& v3, wino_f2& v4) {
  const wino_f2 two = {2.f, 2.f}, three = {3.f, 3.f}, mtwo = {-2.f, -2.f};
  const wino_f2 t13 = d1 - d3, t02 = d0 - d2, t24 = d2 - d4, t23 = d2 - d3;
  v0 = wino_fma2(two, t02, -t13);
  v1 = wino_fma2(two, d1, t23);
  v2 = wino_fma2(three, d2, wino_fma2(mtwo, d1, -d3));
  v3 = t13;
  v4 = wino_fma2(two, t13, -t24);
}

// B^T d along a ROW held as pairs a = (d0, d1), b = (d2, d3) and c = d4: 6 operations (one packed difference, a packed
// pair of fmas for the two middle outputs, scalar fmas for the ends)
__device__ __forceinline__ void wino_bt_row(wino_f2 a, wino_f2 b, float c, float& v0, float& v1, float& v2, float& v3,
                                            float& v4) {
  const wino_f2 t = a - b;                                                    // (d0 - d2, d1 - d3)
  const wino_f2 w = wino_fma2(b.xx, wino_f2{1.f, 3.f}, -b.yy);                // (d2 - d3, 3 d2 - d3)
  const wino_f2 m = wino_fma2(a.yy, wino_f2{2.f, -2.f}, w);                   // (2 d1 + d2 - d3, -2 d1 + 3 d2 - d3)
  v0 = fmaf(2.f, t.x, -t.y);
  v1 = m.x;
  v2 = m.y;
  v3 = t.y;
  v4 = fmaf(2.f, t.y, c - b.x);
}
