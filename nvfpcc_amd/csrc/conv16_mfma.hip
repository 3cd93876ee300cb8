// Matrix-core gather convolutions for layers with 16 (or 32) OUTPUT channels -- the wide decoder
// (chanstr 16,32,16,16: F.conv3d network.py:687 forward and backward-data; backward-data of the stride-2
// transposed convolutions, network.py:621).
//
//   y[b,co,o] = act(bias[co] + sum_{ci,k} x[b,ci, S*o - pad + k] * w[ci][k][co]) (+ addend) (* (mask > 0))
//
// With 16 output channels the 16 rows of v_mfma_f32_16x16x4_f32 (an exact fp32 fmaf chain at the fp32 vector
// rate) are the output channels themselves -- no pairing of outputs, every MFMA lane does useful work:
//
//   D[co][col] += sum_{k=0..3} A[co][k] * B[k][col]      k = four consecutive input channels of one tap,
//                                                        col = 16 outputs (a CTY x CTX patch of one plane)
//
// A workgroup computes OZ planes x (NCY x NCX) column tiles; each wave owns R of those OZ*NCY*NCX tiles.  Four input
// channels of the haloed input tile are staged per step into one of two LDS buffers by LDS-DMA (the next step's loads
// are in flight during this step's MFMAs); B fragments are one conflict-free ds_read_b32 per MFMA (row / channel strides
// found at compile time so that the 32 lanes of a read group hit 32 banks); A fragments are read straight from the
// pre-packed weights in global memory (L2), one kz-slice ahead of their use, and each feeds R MFMAs.
// Per output the accumulation order is fixed -- (channel group, kz, ky, kx) -- independent of batch and tiling.
#include "nvf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct G16Dims {
  int din, hin, win, dout, hout, wout, pad, act, cout, tiles_x, tiles_y, tiles_z;
};

// 32 lanes of one ds_read_b32 group: kq in {0,1} (or {2,3}: same pattern shifted by 2 CS), j = 0..15
constexpr bool g16_conflict_free(int rs, int cs, int s, int cty, int ctx) {
  bool used[32] = {};
  for (int kq = 0; kq < 2; ++kq)
    for (int j = 0; j < 16; ++j) {
      const int a = kq * cs + (j / ctx) * s * rs + (j % ctx) * s;
      const int bank = ((a % 32) + 32) % 32;
      if (used[bank]) return false;
      used[bank] = true;
    }
  (void)cty;
  return true;
}

template <int CIN_, int K_, int S_, int OZ_, int NCY_, int NCX_, int CTY_, int CTX_, int NW_>
struct G16 {
  static constexpr int CIN = CIN_, K = K_, S = S_, OZ = OZ_, NCY = NCY_, NCX = NCX_, CTY = CTY_, CTX = CTX_, NW = NW_;
  static_assert(CTY * CTX == 16 && CIN % 4 == 0, "a column tile has 16 outputs; K runs over groups of four channels");
  static constexpr int K3 = K * K * K, NG = CIN / 4, NT = NW * 64;
  static constexpr int OY = NCY * CTY, OX = NCX * CTX;
  static constexpr int NSEG = OZ * NCY * NCX;
  static constexpr int R = (NSEG + NW - 1) / NW;           // tiles per wave (an uneven split leaves the last wave short:
  static constexpr bool EVEN = NSEG % NW == 0;             //  its missing tiles repeat tile NSEG - 1 and are not stored)
  static constexpr bool ZUNI = EVEN && (NCY * NCX) % R == 0;   // every wave's R tiles lie in one output plane
  static constexpr int IZ = (OZ - 1) * S + K, IY = (OY - 1) * S + K, IX = (OX - 1) * S + K;
  static constexpr int find_rs() {
    for (int rs = IX; rs < IX + 64; ++rs)
      for (int cs = IZ * IY * rs; cs < IZ * IY * rs + 64; ++cs)
        if (g16_conflict_free(rs, cs, S, CTY, CTX)) return rs;
    return -1;
  }
  static constexpr int RS = find_rs();
  static_assert(RS > 0, "no conflict-free row stride");
  static constexpr int PS = IY * RS;
  static constexpr int find_cs() {
    for (int cs = IZ * PS; cs < IZ * PS + 64; ++cs)
      if (g16_conflict_free(RS, cs, S, CTY, CTX)) return cs;
    return -1;
  }
  static constexpr int CS = find_cs();
  static constexpr int BUF = 4 * CS;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two tile buffers in LDS");
  static constexpr int NIT = (IZ * PS + NT - 1) / NT;      // LDS-DMA instructions per wave per staged channel
};

template <class C>
__global__ __launch_bounds__(C::NT) void conv_g16_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       const float* __restrict__ addend,
                                                       const float* __restrict__ mask, G16Dims d) {
  constexpr int K = C::K, S = C::S, R = C::R, RS = C::RS, PS = C::PS, CS = C::CS, NG = C::NG, K3 = C::K3, NT = C::NT;
  constexpr int IZ = C::IZ, IX = C::IX, NIT = C::NIT;
  __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  // work item: (batch element, output-channel group of 16, tile)
  const int ntile = d.tiles_x * d.tiles_y * d.tiles_z, ncog = (d.cout + 15) >> 4;
  int wi = blockIdx.x;
  const int tile = wi % ntile; wi /= ntile;
  const int cog = wi % ncog, b = wi / ncog;
  const int ox0 = (tile % d.tiles_x) * C::OX, oy0 = ((tile / d.tiles_x) % d.tiles_y) * C::OY,
            oz0 = (tile / (d.tiles_x * d.tiles_y)) * C::OZ;
  const int gz0 = oz0 * S - d.pad, gy0 = oy0 * S - d.pad, gx0 = ox0 * S - d.pad;   // first input element of the tile
  const size_t vol = (size_t)d.din * d.hin * d.win;
  const float* xb = x + (size_t)b * C::CIN * vol;
  const float* wg = wp + (size_t)cog * NG * K3 * 64 + lane;

  // per-wave tiles: s = wave * R + r -> (z, cy, cx); LDS word of this lane's first tap
  int base[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = min(wave * R + r, C::NSEG - 1);
    const int cx = s % C::NCX, cy = (s / C::NCX) % C::NCY, z = s / (C::NCX * C::NCY);
    base[r] = kq * CS + z * S * PS + (cy * C::CTY + j / C::CTX) * S * RS + (cx * C::CTX + j % C::CTX) * S;
  }

  // staging by LDS-DMA (global_load_lds_dword: no VGPR destination): wave-instruction i of channel c fills the 64
  // consecutive LDS words w = (i NW + wave) 64 + lane of that channel's image.  Which element of the tile a lane
  // fetches, and whether it lies inside the tensor, is the same for every channel and chunk of this workgroup: the
  // offsets are computed once, the words outside the tensor (and the row padding) are zeroed once in both buffers and
  // never touched again -- an instruction then costs a scalar base address and one DMA issue.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  unsigned soff[NIT];
  bool sok[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int w = (i * C::NW + wave) * 64 + lane;
    const int zz = w / PS, rem = w - zz * PS, yy = rem / RS, xx = rem - yy * RS;
    const int gx = gx0 + xx, gy = gy0 + yy, gz = gz0 + zz;
    const bool live = w < IZ * PS;
    sok[i] = live && xx < IX && gx >= 0 && gx < d.win && gy >= 0 && gy < d.hin && gz >= 0 && gz < d.din;
    soff[i] = sok[i] ? (unsigned)((zz * d.hin + yy) * d.win + xx) * 4u : 0u;
    if (live && !sok[i]) {
#pragma unroll
      for (int c = 0; c < 8; ++c) lds[c * CS + w] = 0.f;          // 2 buffers x 4 channels (BUF = 4 CS)
    }
  }
  const float* xorg = nvf_uniform_ptr(xb + ((ptrdiff_t)gz0 * d.hin + gy0) * d.win + gx0);     // tile origin (may lie in the padding)
  auto stage_chunk = [&](int g, int buf) {
    const float* xc = xorg + (size_t)(4 * g) * vol;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned cbase = lds0 + (unsigned)(buf * C::BUF + c * CS) * 4u;
      const float* src = nvf_uniform_ptr(xc + (size_t)c * vol);
#pragma unroll
      for (int i = 0; i < NIT; ++i)
        if (sok[i]) nvf_glds_row(src, soff[i], cbase + (unsigned)((i * C::NW + wave) * 64) * 4u);
    }
  };

  // first input plane (tensor coordinates) read by this wave's tiles, when they share one output plane
  const int zplane = gz0 + ((wave * R) / (C::NCX * C::NCY)) * S;
  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ac[K * K], an[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    ac[i] = wg[(size_t)i * 64];
    an[i] = 0.f;
  }
  stage_chunk(0, 0);

#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of chunk g has landed
    __syncthreads();                                  // ... everyone's; the other buffer is free
    if (g + 1 < NG) stage_chunk(g + 1, (g + 1) & 1);  // in flight during this chunk's MFMAs
    const float* ldsb = lds + (g & 1) * C::BUF;
    // B fragments are fetched one tap ahead of the MFMAs that use them (2 R registers), and the scheduling barriers
    // keep the compiler from hoisting whole slices of LDS reads (which cost hundreds of registers and spills)
    float bc[R], bn[R];
#pragma unroll
    for (int r = 0; r < R; ++r) bc[r] = ldsb[base[r]];
#pragma unroll
    for (int kz = 0; kz < K; ++kz) {
      // the next kz-slice of A fragments (of this group, or the first one of the next group)
      const int nslice = g * K + kz + 1;
      if (nslice < NG * K) {
#pragma unroll
        for (int i = 0; i < K * K; ++i) an[i] = wg[((size_t)nslice * K * K + i) * 64];
      }
      // backward-data runs over a zero-padded gradient: when all tiles of this wave lie in ONE output plane, a kz whose
      // input plane is padding multiplies zeros only -- skip its K*K*R MFMAs (acc + 0 = acc: same values)
      const bool skip = C::ZUNI && (zplane + kz < 0 || zplane + kz >= d.din);
      if (skip) {
        if (kz + 1 < K) {
#pragma unroll
          for (int r = 0; r < R; ++r) bc[r] = ldsb[base[r] + (kz + 1) * PS];
        }
      } else
#pragma unroll
      for (int t = 0; t < K * K; ++t) {
        const int tn = kz * K * K + t + 1;            // next tap of this chunk
        if (tn < K3) {
          const int nz = tn / (K * K), ny = (tn / K) % K, nx = tn % K;
#pragma unroll
          for (int r = 0; r < R; ++r) bn[r] = ldsb[base[r] + nz * PS + ny * RS + nx];
        }
        const float a = ac[t];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) bc[r] = bn[r];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < K * K; ++i) ac[i] = an[i];
    }
  }

  // epilogue: lane holds rows co = 4 kq + r4 of column j of each of its tiles
  const size_t ovol = (size_t)d.dout * d.hout * d.wout;
  const int co0 = cog * 16 + 4 * kq;
  float bv4[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) bv4[r4] = co0 + r4 < d.cout ? bias[co0 + r4] : 0.f;
  }
  constexpr int TILE = 16 * C::OZ * C::OY * C::OX;      // the workgroup's outputs as [channel][z][y][x]
  if constexpr (C::CTX < 16 && TILE <= 2 * C::BUF) {
    // Patch-shaped column tiles (4 x 4, 2 x 8): written directly, every store / mask read of a wave touches 16-byte
    // pieces of 16 different lines (s_memrealtime stamps: 21-29 us of the 88 us a conv2 backward-data workgroup takes).
    // The tile goes through the (now idle) staging buffers instead and leaves as whole rows, lanes along x.
    __syncthreads();                                     // every wave is done with its B fragments
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = wave * R + r;
      if (!C::EVEN && s >= C::NSEG) continue;
      const int cx = s % C::NCX, cy = (s / C::NCX) % C::NCY, z = s / (C::NCX * C::NCY);
      const int ty = cy * C::CTY + j / C::CTX, tx = cx * C::CTX + j % C::CTX;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        lds[(((4 * kq + r4) * C::OZ + z) * C::OY + ty) * C::OX + tx] = nvf_act(acc[r][r4] + bv4[r4], d.act);
    }
    __syncthreads();
    const size_t obase = (size_t)b * d.cout * ovol;
#pragma unroll 4
    for (int e = tid; e < TILE; e += NT) {
      const int tx = e % C::OX, t1 = e / C::OX, ty = t1 % C::OY, t2 = t1 / C::OY, z = t2 % C::OZ, ch = t2 / C::OZ;
      const int oz = oz0 + z, oy = oy0 + ty, ox = ox0 + tx, co = cog * 16 + ch;
      if (oz >= d.dout || oy >= d.hout || ox >= d.wout || co >= d.cout) continue;
      const size_t oo = obase + (size_t)co * ovol + ((size_t)oz * d.hout + oy) * d.wout + ox;
      float v = lds[e];
      if (addend) v += addend[oo];
      if (mask) v = mask[oo] > 0.f ? v : 0.f;
      y[oo] = v;
    }
  } else {
    // (a tile row is CTX consecutive x: every (kq, r4, patch row) writes one CTX * 4-byte segment)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = wave * R + r;
      if (!C::EVEN && s >= C::NSEG) continue;
      const int cx = s % C::NCX, cy = (s / C::NCX) % C::NCY, z = s / (C::NCX * C::NCY);
      const int oz = oz0 + z, oy = oy0 + cy * C::CTY + j / C::CTX, ox = ox0 + cx * C::CTX + j % C::CTX;
      if (oz >= d.dout || oy >= d.hout || ox >= d.wout) continue;
      const size_t o = ((size_t)b * d.cout + co0) * ovol + ((size_t)oz * d.hout + oy) * d.wout + ox;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (co0 + r4 >= d.cout) continue;            // rows past an 8-channel output (zero weights) are not stored
        const size_t oo = o + (size_t)r4 * ovol;
        float v = nvf_act(acc[r][r4] + bv4[r4], d.act);
        if (addend) v += addend[oo];
        if (mask) v = mask[oo] > 0.f ? v : 0.f;
        y[oo] = v;
      }
    }
  }
}

__global__ void pack_g16_kernel(const float* __restrict__ gw, float* __restrict__ wp, int cin, int cout, int k3) {
  const int total = ((cout + 15) / 16) * (cin / 4) * k3 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int lane = r % 64; r /= 64;
    const int tap = r % k3; r /= k3;
    const int g = r % (cin / 4), cog = r / (cin / 4);
    const int ci = 4 * g + (lane >> 4), co = cog * 16 + (lane & 15);
    wp[idx] = co < cout ? gw[((size_t)ci * k3 + tap) * cout + co] : 0.f;      // 8 outputs: rows 8..15 are zero
  }
}

template <class C>
int launch_g16(const float* x, const float* wp, const float* bias, float* y, const float* addend, const float* mask,
               int batch, G16Dims d, hipStream_t s) {
  d.tiles_x = (d.wout + C::OX - 1) / C::OX;
  d.tiles_y = (d.hout + C::OY - 1) / C::OY;
  d.tiles_z = (d.dout + C::OZ - 1) / C::OZ;
  const long grid = (long)d.tiles_x * d.tiles_y * d.tiles_z * ((d.cout + 15) / 16) * batch;
  conv_g16_mfma<C><<<(unsigned)grid, C::NT, 0, s>>>(x, wp, bias, y, addend, mask, d);
  return NVF_OK;
}

}  // namespace

// A fragments of a gather-form weight gw [cin][k^3][cout] (cout a multiple of 16, or 8: rows 8..15 zero): wp[cog][g][tap][lane],
// lane = (ci & 3) * 16 + (co & 15)
extern "C" size_t nvf_pack_g16_mfma_floats(int cin, int cout, int k) {
  return (size_t)((cout + 15) / 16) * (cin / 4) * k * k * k * 64;
}

extern "C" int nvf_pack_g16_mfma(const float* gather_w, int cin, int cout, int k, float* wp, void* stream) {
  if (!gather_w || !wp || cin <= 0 || cin % 4 || cout <= 0 || (cout % 16 && cout != 8) || k <= 0) return NVF_EINVAL;
  const int total = (int)nvf_pack_g16_mfma_floats(cin, cout, k);
  pack_g16_kernel<<<(total + 255) / 256, 256, 0, nvf_stream(stream)>>>(gather_w, wp, cin, cout, k * k * k);
  NVF_LAUNCH_CHECK();
  return NVF_OK;
}

// Same contract as nvf_conv3d_gather (stride 1 or 2) for cout in {16, 32}; wp = nvf_pack_g16_mfma of the packed
// gather weight (w_fwd for a forward pass, w_bwd for a backward-data pass).  NVF_EINVAL = no instantiation for
// this shape (the caller then uses nvf_conv3d_gather).
extern "C" int nvf_conv3d_g16_mfma(const float* x, const float* wp, const float* bias, float* y, const float* addend,
                                   const float* mask, int batch, int cin, int cout, int k, int stride, int pad,
                                   int din, int hin, int win, int dout, int hout, int wout, int act, int variant,
                                   void* stream) {
  if (!x || !wp || !y || batch <= 0 || cout <= 0 || (cout % 16 && cout != 8)) return NVF_EINVAL;
  G16Dims d{din, hin, win, dout, hout, wout, pad, act, cout, 0, 0, 0};
  hipStream_t s = nvf_stream(stream);
  int rc = 1;
#define NVF_G16(VAR, CI, KS, ST, WLO, WHI, OZ, NCY, NCX, CTY, CTX, NW)                                       \
  if (rc == 1 && variant == VAR && cin == CI && k == KS && stride == ST && wout >= WLO && wout <= WHI)       \
    rc = launch_g16<G16<CI, KS, ST, OZ, NCY, NCX, CTY, CTX, NW>>(x, wp, bias, y, addend, mask, batch, d, s);
  // (tile choices: tools/g16_sweep.py at batch 16)
  NVF_G16(0, 16, 4, 1, 21, 32, 8, 4, 2, 1, 16, 8)     // conv2 forward: 8 planes x 4 rows x 32 (158 us = 109 TF in the step)
  NVF_G16(0, 16, 4, 1, 33, 40, 7, 1, 9, 4, 4, 8)      // conv2 backward-data (35^3): 7 planes of 1 x 9 patches of 4 x 4, the 63 patches over 8 waves
  NVF_G16(0, 16, 4, 1, 9, 16, 2, 8, 1, 1, 16, 8)      // conv1 forward: 2 planes x 8 rows x 16 (26 us)
  NVF_G16(0, 16, 4, 1, 17, 20, 2, 2, 5, 4, 4, 4)      // conv1 backward-data (19^3): 2 planes x 8 rows x 20 on four waves (58 us; one plane of 3 x 5 patches on five waves: 88)
  NVF_G16(0, 16, 5, 2, 9, 16, 2, 8, 1, 1, 16, 4)      // up2 backward-data (35^3 -> 16^3) (58 us)
  NVF_G16(0, 16, 5, 2, 5, 8, 1, 4, 1, 2, 8, 4)        // up1 backward-data (19^3 -> 8^3, 32 output channels)
  NVF_G16(0, 32, 5, 2, 3, 4, 4, 1, 1, 4, 4, 4)        // conv0 backward-data (8^3 -> 4^3, padding 2)
  NVF_G16(0, 16, 5, 2, 1, 2, 2, 1, 1, 4, 4, 2)        // up0 backward-data (4^3 -> 2^3, padding 2; 8 output channels)
  // tuning alternatives
  NVF_G16(2, 16, 4, 1, 21, 32, 4, 8, 2, 1, 16, 4)
  NVF_G16(3, 16, 4, 1, 21, 32, 2, 8, 2, 1, 16, 4)
  NVF_G16(4, 16, 4, 1, 21, 32, 2, 8, 2, 1, 16, 8)
  NVF_G16(5, 16, 4, 1, 33, 40, 4, 4, 5, 2, 8, 8)
  NVF_G16(6, 16, 4, 1, 33, 40, 4, 3, 9, 4, 4, 12)
  NVF_G16(7, 16, 4, 1, 33, 40, 2, 3, 9, 4, 4, 6)
  NVF_G16(8, 16, 4, 1, 33, 40, 5, 1, 9, 4, 4, 5)
  NVF_G16(3, 16, 4, 1, 17, 20, 1, 5, 5, 4, 4, 5)
  NVF_G16(4, 16, 4, 1, 17, 20, 1, 3, 5, 4, 4, 3)
  NVF_G16(5, 16, 4, 1, 17, 20, 2, 3, 5, 4, 4, 6)
  NVF_G16(5, 16, 4, 1, 21, 32, 4, 4, 2, 1, 16, 4)
  NVF_G16(6, 16, 4, 1, 21, 32, 4, 4, 2, 1, 16, 8)
  NVF_G16(7, 16, 4, 1, 21, 32, 4, 8, 2, 1, 16, 8)
  NVF_G16(2, 32, 5, 2, 3, 4, 1, 1, 1, 4, 4, 1)
  NVF_G16(2, 16, 4, 1, 33, 40, 4, 4, 5, 2, 8, 4)
  NVF_G16(3, 16, 4, 1, 33, 40, 2, 4, 5, 2, 8, 4)
  NVF_G16(4, 16, 4, 1, 33, 40, 2, 4, 5, 2, 8, 8)
  NVF_G16(2, 16, 4, 1, 9, 16, 4, 8, 1, 1, 16, 4)
  NVF_G16(3, 16, 4, 1, 9, 16, 2, 8, 1, 1, 16, 4)
  NVF_G16(2, 16, 4, 1, 17, 20, 2, 5, 5, 4, 4, 5)
  NVF_G16(2, 16, 5, 2, 9, 16, 2, 4, 1, 1, 16, 4)
  NVF_G16(3, 16, 5, 2, 9, 16, 1, 4, 1, 1, 16, 4)
  NVF_G16(2, 16, 5, 2, 5, 8, 2, 4, 1, 2, 8, 4)
  NVF_G16(9, 16, 4, 1, 33, 40, 4, 1, 9, 4, 4, 4)      // conv2 bwd: four planes, one per wave, two workgroups per CU
  NVF_G16(14, 16, 4, 1, 33, 40, 7, 1, 9, 4, 4, 7)
  NVF_G16(14, 16, 4, 1, 21, 32, 2, 4, 2, 1, 16, 4)     // (160 us in the step)
  NVF_G16(14, 16, 4, 1, 17, 20, 1, 3, 5, 4, 4, 5)
  NVF_G16(10, 16, 4, 1, 33, 40, 3, 1, 9, 4, 4, 3)
  NVF_G16(12, 16, 4, 1, 33, 40, 2, 1, 9, 4, 4, 2)
  NVF_G16(9, 16, 4, 1, 17, 20, 1, 5, 5, 4, 4, 4)      // conv1 bwd: 25 patches of a plane over 4 / 8 waves
  NVF_G16(10, 16, 4, 1, 17, 20, 1, 5, 5, 4, 4, 8)
  NVF_G16(11, 16, 4, 1, 17, 20, 2, 5, 5, 4, 4, 8)
  NVF_G16(12, 16, 4, 1, 17, 20, 1, 2, 5, 4, 4, 2)     // 8 rows x 20
  NVF_G16(9, 16, 4, 1, 21, 32, 4, 4, 2, 1, 16, 4)     // conv2 fwd: R = 8 on four waves, two workgroups per CU
#undef NVF_G16
  if (rc == 1) return NVF_EINVAL;
  NVF_LAUNCH_CHECK();
  return rc;
}
