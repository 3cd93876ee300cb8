/*
 * nvf_codec.h -- C ABI of libnvf_codec.so: the host-side entropy coder of the latent codes.
 *
 * Replaces the reference's stand-alone executable `module_arithmeticcoding` that NVFPCC.py drives through
 * subprocess pipes (NVFPCC.py:459-467, 601-604; module_arithmeticcoding.cpp:368-432) by two in-process calls.
 * The coded stream is bit-identical to the reference's for the same (symbols, mu, sigma, level1, level2):
 *   - 64-bit low/high range coder held in 128-bit integers (module_arithmeticcoding.cpp:11-18, 189-238);
 *   - per-symbol frequency table from a Gaussian CDF evaluated in double, rounded to float, scaled by 1e7 in
 *     float (:115-167), total = 10 001 025, 1025 symbols; mu / sigma first lose their `level` lowest
 *     mantissa bits (:96-113);
 *   - after the last symbol a terminator (symbol 512 under N(255, 1)) and one 1 bit are coded (:394-399);
 *     the final partial byte is NOT flushed (the reference never calls BitOutputStream::close()), so the
 *     stream holds floor(bits / 8) bytes.
 * Everything is plain host memory; the functions are re-entrant.
 */
#ifndef NVF_CODEC_H
#define NVF_CODEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Encodes n symbols (each in [0, 1023]).  Returns the number of bytes written to `out`, or -1 if a symbol is
 * out of range / has zero frequency, or -2 if out_cap is too small (8 * n + 16 bytes always suffice). */
int64_t nvf_ac_encode(const int16_t* symbols, const float* mu, const float* sigma, int64_t n, int level_mu,
                      int level_sigma, uint8_t* out, int64_t out_cap);

/* Decodes n symbols from `stream` (bits past the end read as 0, as in the reference).  Returns 0, or -1 when
 * the stream is inconsistent with the model. */
int nvf_ac_decode(const uint8_t* stream, int64_t nbytes, const float* mu, const float* sigma, int64_t n,
                  int level_mu, int level_sigma, int16_t* symbols_out);

int nvf_codec_version(void);

#ifdef __cplusplus
}
#endif
#endif
