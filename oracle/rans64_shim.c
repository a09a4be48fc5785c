/* TEST INFRASTRUCTURE ONLY (oracle/): a thin shim that exposes the reference's vendored rANS state machine
 * -- /root/reference/third_party/ryg_rans/rans64.h, public domain, included from where it lies via -I -- to ctypes,
 * so that tests can check csrc/rans.cpp word for word against the very header the reference's coder is built on.
 * Built by oracle/build_ref.sh into oracle/_ref/librans64_ref.so (git-ignored, travels to the GPU box).
 * Nothing under image-compression-for-machine_amd/ links or loads this. */
#include <stdint.h>
#include <string.h>
#include "rans64.h"

/* encode n symbols given as (start, freq) pairs out of 2^scale_bits: symbols are pushed in reverse order, the stream
 * is written backwards into a scratch buffer (rans64.h:70-72 NOTE) and copied to out; returns the number of words */
int64_t ref_rans64_encode(const uint32_t* start, const uint32_t* freq, int64_t n, uint32_t scale_bits, uint32_t* out,
                          int64_t cap_words) {
  if (cap_words < n + 2) return -1;
  uint32_t* end = out + cap_words;
  uint32_t* ptr = end;
  Rans64State r;
  Rans64EncInit(&r);
  for (int64_t i = n - 1; i >= 0; --i) Rans64EncPut(&r, &ptr, start[i], freq[i], scale_bits);
  Rans64EncFlush(&r, &ptr);
  const int64_t words = end - ptr;
  memmove(out, ptr, (size_t)words * 4);
  return words;
}

/* decode n symbols of ONE cumulative table cdf[0..m] (cdf[m] = 2^scale_bits) */
int ref_rans64_decode(const uint32_t* stream, int64_t words, const uint32_t* cdf, int m, uint32_t scale_bits,
                      int32_t* out, int64_t n) {
  uint32_t* ptr = (uint32_t*)stream;
  Rans64State r;
  if (words < 2) return 1;
  Rans64DecInit(&r, &ptr);
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t cum = Rans64DecGet(&r, scale_bits);
    int s = 0;
    while (s + 1 < m && cdf[s + 1] <= cum) ++s;
    Rans64DecAdvance(&r, &ptr, cdf[s], cdf[s + 1] - cdf[s], scale_bits);
    out[i] = s;
  }
  return (ptr - (uint32_t*)stream) <= words ? 0 : 2;
}
