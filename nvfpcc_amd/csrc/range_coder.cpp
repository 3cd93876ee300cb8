// Range coder for the quantised latents (host code, C ABI in include/nvf_codec.h).
// Algorithm: the reference's module_arithmeticcoding.cpp (64-bit state arithmetic coder with a Gaussian
// frequency model computed on the fly), restated as a library.  Stream-compatible with the reference.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/nvf_codec.h"

namespace {
typedef unsigned __int128 u128;

const int kBits = 64;                       // state width (module_arithmeticcoding.cpp:12)
const u128 kFull = (u128)1 << kBits;        // 2^64
const u128 kMask = kFull - 1;
const u128 kTop = kFull >> 1;               // top bit
const u128 kSecond = kTop >> 1;
const u128 kMinRange = (kFull >> 2) + 2;
const int kScale = 10000000;                // mul_factor (:124)
const int kSymbols = 1025;                  // (:124)
const int kTotal = kScale + 1025;           // (:131)

inline float drop_mantissa_bits(float v, int level) {
  uint32_t bits;
  std::memcpy(&bits, &v, 4);
  bits &= ~(((uint32_t)1 << level) - 1u);   // (:96-113)
  float r;
  std::memcpy(&r, &bits, 4);
  return r;
}

struct Model {
  float mu, sigma;
  Model(float m, float s, int lm, int ls) : mu(drop_mantissa_bits(m, lm)), sigma(drop_mantissa_bits(s, ls)) {}
  // cumulative count below symbol s: (int)(floorf(c * 1e7) + s), c = (float) Phi((s - .5 - mu) / sigma)   (:155-160)
  inline int64_t low(int64_t s) const { return edge(s - 1, s); }
  inline int64_t high(int64_t s) const { return edge(s, s + 1); }   // (:162-167)
 private:
  inline int64_t edge(int64_t upto, int64_t add) const {
    const float tiny = 1e-10f;
    const double z = ((double)upto + 0.5 - (double)mu) / ((double)(float)(sigma + tiny) * std::sqrt(2.0));
    const float c = (float)(0.5 * (1.0 + std::erf(z)));
    const float scaled = std::floor(c * (float)kScale);        // float product, float floor
    return (int64_t)(int)(scaled + (float)add);                // float add, then (int), as in the reference
  }
};

struct BitWriter {
  uint8_t* out; int64_t cap, pos; int cur, fill; bool overflow;
  BitWriter(uint8_t* o, int64_t c) : out(o), cap(c), pos(0), cur(0), fill(0), overflow(false) {}
  inline void put(int b) {
    cur = (cur << 1) | b;
    if (++fill == 8) {
      if (pos < cap) out[pos] = (uint8_t)cur; else overflow = true;
      ++pos; cur = 0; fill = 0;
    }
  }
};

struct BitReader {
  const uint8_t* in; int64_t n, pos; int cur, left;
  BitReader(const uint8_t* p, int64_t nb) : in(p), n(nb), pos(0), cur(0), left(0) {}
  inline int get() {
    if (left == 0) {
      if (pos >= n) return 0;              // past the end: zeros (:358-362)
      cur = in[pos++]; left = 8;
    }
    --left;
    return (cur >> left) & 1;
  }
};

struct Coder {
  u128 low, high;
  Coder() : low(0), high(kMask) {}
};
}  // namespace

extern "C" int nvf_codec_version(void) { return 100; }

extern "C" int64_t nvf_ac_encode(const int16_t* symbols, const float* mu, const float* sigma, int64_t n,
                                 int level_mu, int level_sigma, uint8_t* out, int64_t out_cap) {
  if (!symbols || !mu || !sigma || !out || n < 0) return -1;
  BitWriter bw(out, out_cap);
  Coder c;
  int64_t pending = 0;                       // underflow bits (:262-271)
  for (int64_t i = 0; i <= n; ++i) {
    const bool term = i == n;                // terminator: symbol 512 under N(255, 1) (:394-398)
    const Model m(term ? 255.f : mu[i], term ? 1.f : sigma[i], level_mu, level_sigma);
    const int64_t s = term ? 512 : symbols[i];
    if (s < 0 || s >= kSymbols) return -1;
    const int64_t lo = m.low(s), hi = m.high(s);
    if (lo == hi) return -1;                 // zero frequency
    const u128 range = c.high - c.low + 1;
    if (range < kMinRange || range > kFull) return -1;
    const u128 nl = c.low + (u128)lo * range / (u128)kTotal;
    const u128 nh = c.low + (u128)hi * range / (u128)kTotal - 1;
    c.low = nl; c.high = nh;
    while (((c.low ^ c.high) & kTop) == 0) {           // matching top bit: emit it plus pending underflow bits
      const int bit = (int)(c.low >> (kBits - 1));
      bw.put(bit);
      for (; pending > 0; --pending) bw.put(bit ^ 1);
      c.low = (c.low << 1) & kMask;
      c.high = ((c.high << 1) & kMask) | 1;
    }
    while ((c.low & ~c.high & kSecond) != 0) {         // underflow (:233-237)
      ++pending;
      c.low = (c.low << 1) & (kMask >> 1);
      c.high = ((c.high << 1) & (kMask >> 1)) | kTop | 1;
    }
  }
  bw.put(1);                                            // finish() (:257-259); no flush of the partial byte
  if (bw.overflow) return -2;
  return bw.pos;
}

extern "C" int nvf_ac_decode(const uint8_t* stream, int64_t nbytes, const float* mu, const float* sigma, int64_t n,
                             int level_mu, int level_sigma, int16_t* symbols_out) {
  if ((!stream && nbytes > 0) || !mu || !sigma || !symbols_out || n < 0) return -1;
  BitReader br(stream, nbytes);
  Coder c;
  u128 code = 0;
  for (int i = 0; i < kBits; ++i) code = (code << 1) | (u128)br.get();
  for (int64_t i = 0; i < n; ++i) {
    const Model m(mu[i], sigma[i], level_mu, level_sigma);
    const u128 range = c.high - c.low + 1;
    const u128 offset = code - c.low;
    // exact integer form of the scaled value; the reference estimates it in double (:296) and asserts the
    // result (:327-339) -- wherever the reference does not abort, both give the same symbol
    const u128 value = ((offset + 1) * (u128)kTotal - 1) / range;
    if (value >= (u128)kTotal) return -1;
    int64_t a = 0, b = kSymbols;
    while (b - a > 1) {                                 // largest symbol with low(symbol) <= value (:313-323)
      const int64_t mid = (a + b) >> 1;
      if ((u128)m.low(mid) > value) b = mid; else a = mid;
    }
    const int64_t s = a;
    const int64_t lo = m.low(s), hi = m.high(s);
    if (!((u128)lo * range / (u128)kTotal <= offset && offset < (u128)hi * range / (u128)kTotal)) return -1;
    c.high = c.low + (u128)hi * range / (u128)kTotal - 1;
    c.low = c.low + (u128)lo * range / (u128)kTotal;
    while (((c.low ^ c.high) & kTop) == 0) {
      code = ((code << 1) & kMask) | (u128)br.get();
      c.low = (c.low << 1) & kMask;
      c.high = ((c.high << 1) & kMask) | 1;
    }
    while ((c.low & ~c.high & kSecond) != 0) {
      code = (code & kTop) | ((code << 1) & (kMask >> 1)) | (u128)br.get();
      c.low = (c.low << 1) & (kMask >> 1);
      c.high = ((c.high << 1) & (kMask >> 1)) | kTop | 1;
    }
    if (!(c.low <= code && code <= c.high)) return -1;
    symbols_out[i] = (int16_t)s;
  }
  return 0;
}
