// Entropy-coding back end of update() / compress() / decompress(): quantised CDF tables and the interleaved
// 64-bit rANS stream.  HOST code (the reference codes on the host too: compressai/entropy_models/
// entropy_models.py:172-290 hands Python lists to the `compressai.ans` / `compressai._CXX` extensions).
//
// What this replaces.  The reference tree ships those two extensions only as cp38 binaries (never loaded here); their
// sources (CompressAI 1.1.6dev0, compressai/cpp_exts/{rans,ops}) are absent from /root/reference.  The arithmetic
// core they wrap IS in the tree: third_party/ryg_rans/rans64.h (public domain, F. Giesen).  This file restates
//   * the rans64.h state machine (Rans64EncPut / EncFlush / DecInit / DecGet / DecAdvance; rans64.h:62-142) --
//     pinned bit for bit against that very header by tests/test_rans_codec.py (oracle/_ref builds a shim from it);
//   * CompressAI's published interface conventions on top of it: 16-bit precision, symbols shifted by the table's
//     offset, values outside the table escaped through the last ("overflow") bin and written as 4-bit bypass groups
//     (count in unary-ish nibbles, then the zig-zagged magnitude), one stream per call, symbols pushed in reverse;
//   * pmf_to_quantized_cdf: round(p * 2^precision), renormalise to 2^precision, cumulate, then steal counts from the
//     least frequent symbol with more than one count until no symbol has zero width.
// Parity status: round-trip identity and the rans64.h pin are tested; equality with the reference's own byte
// streams is "parity unpinned"  because the reference's coder binaries may not be run.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/icm_hip.h"

namespace {

constexpr uint32_t kPrecision = 16;        // entropy_coder_precision (entropy_models.py:78-81)
constexpr uint32_t kBypassBits = 4;        // escape payload granularity
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;
constexpr uint64_t kRansL = 1ull << 31;    // lower bound of the normalisation interval (rans64.h:58)

// ---- rANS state machine (word-wise renormalisation, 63-bit state) -----------------------------------------------
struct Enc {
  uint64_t x = kRansL;                     // Rans64EncInit
  std::vector<uint32_t> words;             // emitted renormalisation words, in emission order (stream is reversed)
  // x = C(s, x) for a symbol with cumulative start `start`, width `freq`, out of 2^bits   (rans64.h:73-90)
  inline void put(uint32_t start, uint32_t freq, uint32_t bits) {
    const uint64_t x_max = ((kRansL >> bits) << 32) * freq;
    if (x >= x_max) {
      words.push_back((uint32_t)x);
      x >>= 32;
    }
    x = ((x / freq) << bits) + (x % freq) + start;
  }
  // raw bits: a symbol of width 1 out of 2^nbits whose start is the value (uniform distribution)
  inline void put_bits(uint32_t val, uint32_t nbits) {
    const uint32_t freq = 1u << (kPrecision - nbits);
    const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
    if (x >= x_max) {
      words.push_back((uint32_t)x);
      x >>= 32;
    }
    x = (x << nbits) | val;
  }
};

struct Dec {
  const uint32_t* p = nullptr;
  const uint32_t* end = nullptr;
  uint64_t x = 0;
  bool bad = false;
  inline uint32_t next() {
    if (p >= end) { bad = true; return 0; }
    return *p++;
  }
  void init(const uint32_t* s, const uint32_t* e) {   // Rans64DecInit (rans64.h:102-110)
    p = s; end = e; bad = false;
    const uint64_t lo = next(), hi = next();
    x = lo | (hi << 32);
  }
  inline uint32_t peek(uint32_t bits) const { return (uint32_t)(x & ((1u << bits) - 1)); }   // Rans64DecGet
  inline void advance(uint32_t start, uint32_t freq, uint32_t bits) {                           // Rans64DecAdvance
    const uint64_t mask = (1ull << bits) - 1;
    x = freq * (x >> bits) + (x & mask) - start;
    if (x < kRansL) x = (x << 32) | next();
  }
  inline uint32_t get_bits(uint32_t nbits) {
    const uint32_t v = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < kRansL) x = (x << 32) | next();
    return v;
  }
};

struct Tables {
  const int32_t* cdfs;
  int stride;
  const int32_t* sizes;
  const int32_t* offsets;
  int n;
  bool ok() const { return cdfs && sizes && offsets && n > 0 && stride >= 2; }
  bool ok(int idx) const { return idx >= 0 && idx < n && sizes[idx] >= 2 && sizes[idx] <= stride; }
};

struct Item {
  uint16_t start, width;
  bool raw;
};

// symbols -> (start, width) items in FORWARD order; the stream is then produced by pushing them in reverse
int plan_items(const int32_t* symbols, const int32_t* indexes, int64_t n, const Tables& T, std::vector<Item>& items) {
  items.reserve(items.size() + (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const int idx = indexes[i];
    if (!T.ok(idx)) return ICM_ERR_ARG;
    const int32_t* cdf = T.cdfs + (int64_t)idx * T.stride;
    const int32_t overflow = T.sizes[idx] - 2;       // index of the escape bin
    int64_t v = (int64_t)symbols[i] - T.offsets[idx];
    uint64_t raw = 0;                                // 64-bit: |symbol| up to 2^31 needs 33 bits after the zig-zag
    if (v < 0) {
      raw = (uint64_t)(-2 * v - 1);                  // odd  = below the table
      v = overflow;
    } else if (v >= overflow) {
      raw = (uint64_t)(2 * (v - overflow));          // even = above the table (0 = exactly the last regular bin + 1)
      v = overflow;
    }
    const int32_t lo = cdf[v], hi = cdf[v + 1];
    if (hi <= lo || lo < 0 || hi > (1 << kPrecision)) return ICM_ERR_ARG;   // zero-width symbol: table not usable
    items.push_back({(uint16_t)lo, (uint16_t)(hi - lo), false});
    if (v == overflow) {
      uint32_t groups = 0;
      while ((raw >> (groups * kBypassBits)) != 0) ++groups;
      uint32_t c = groups;
      while (c >= kBypassMax) {                      // group count: nibbles of 15 until the remainder is < 15
        items.push_back({(uint16_t)kBypassMax, 1, true});
        c -= kBypassMax;
      }
      items.push_back({(uint16_t)c, 1, true});
      for (uint32_t j = 0; j < groups; ++j)
        items.push_back({(uint16_t)((raw >> (j * kBypassBits)) & kBypassMax), 1, true});
    }
  }
  return ICM_OK;
}

int64_t emit(const std::vector<Item>& items, uint8_t* out, int64_t cap) {
  Enc e;
  e.words.reserve(items.size() / 2 + 4);
  for (size_t k = items.size(); k-- > 0;) {
    const Item& it = items[k];
    if (it.raw) e.put_bits(it.start, kBypassBits);
    else e.put(it.start, it.width, kPrecision);
  }
  // Rans64EncFlush (rans64.h:93-100): the final state heads the stream, low word first; the renormalisation words
  // follow in reverse emission order (the encoder writes backwards, the decoder reads forwards)
  const int64_t nwords = (int64_t)e.words.size() + 2;
  if (out) {
    if (cap < nwords * 4) return -1;
    uint32_t* w = reinterpret_cast<uint32_t*>(out);
    uint32_t tmp;
    tmp = (uint32_t)e.x; std::memcpy(w, &tmp, 4);
    tmp = (uint32_t)(e.x >> 32); std::memcpy(w + 1, &tmp, 4);
    for (int64_t k = 0; k < (int64_t)e.words.size(); ++k)
      std::memcpy(w + 2 + k, &e.words[e.words.size() - 1 - k], 4);
  }
  return nwords * 4;
}

int decode_run(Dec& d, const int32_t* indexes, int64_t n, const Tables& T, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const int idx = indexes[i];
    if (!T.ok(idx)) return ICM_ERR_ARG;
    const int32_t* cdf = T.cdfs + (int64_t)idx * T.stride;
    const int32_t size = T.sizes[idx], overflow = size - 2;
    const uint32_t cum = d.peek(kPrecision);
    // first table entry above the cumulative value (tables are short: binary search over [0, size))
    const int32_t* it = std::upper_bound(cdf, cdf + size, (int32_t)cum);
    int32_t s = (int32_t)(it - cdf) - 1;
    if (s < 0 || s > overflow) return ICM_ERR_ARG;
    d.advance((uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s]), kPrecision);
    int32_t v = s;
    if (s == overflow) {
      uint32_t c = d.get_bits(kBypassBits), groups = c;
      while (c == kBypassMax) {
        c = d.get_bits(kBypassBits);
        groups += c;
        if (groups > 16 || d.bad) return ICM_ERR_ARG;   // 16 nibbles = 64 payload bits
      }
      if (groups > 16) return ICM_ERR_ARG;
      uint64_t raw = 0;
      for (uint32_t j = 0; j < groups; ++j) raw |= (uint64_t)d.get_bits(kBypassBits) << (j * kBypassBits);
      const int64_t half = (int64_t)(raw >> 1);
      const int64_t vv = (raw & 1) ? -half - 1 : half + overflow;
      if (vv < INT32_MIN || vv > INT32_MAX) return ICM_ERR_ARG;
      v = (int32_t)vv;
    }
    if (d.bad) return ICM_ERR_ARG;                   // ran past the end of the stream: truncated / corrupt input
    out[i] = v + T.offsets[idx];
  }
  return ICM_OK;
}

struct Decoder {
  std::vector<uint32_t> words;
  Dec d;
};

}  // namespace

extern "C" {

int icm_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf) {
  if (!pmf || !cdf || n < 1 || precision < 1 || precision > 16) return ICM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!(pmf[i] >= 0.0f) || !std::isfinite(pmf[i])) return ICM_ERR_ARG;   // domain error in the reference
  const uint32_t one = 1u << precision;
  std::vector<uint32_t> c((size_t)n + 1);
  c[0] = 0;
  uint64_t total = 0;
  for (int i = 0; i < n; ++i) {
    c[i + 1] = (uint32_t)std::round(pmf[i] * (float)one);
    total += c[i + 1];
  }
  if (total == 0) return ICM_ERR_ARG;
  uint32_t run = 0;
  for (int i = 0; i <= n; ++i) {                     // renormalise each count to the 2^precision budget, cumulate
    run += (uint32_t)(((uint64_t)one * c[i]) / total);
    c[i] = run;
  }
  c[n] = one;
  for (int i = 0; i < n; ++i) {
    if (c[i] != c[i + 1]) continue;
    // symbol i has zero width: take one count from the narrowest symbol that can spare one
    uint32_t best = ~0u;
    int donor = -1;
    for (int j = 0; j < n; ++j) {
      const uint32_t f = c[j + 1] - c[j];
      if (f > 1 && f < best) { best = f; donor = j; }
    }
    if (donor < 0) return ICM_ERR_ARG;               // more symbols than counts
    if (donor < i) for (int j = donor + 1; j <= i; ++j) --c[j];
    else for (int j = i + 1; j <= donor; ++j) ++c[j];
  }
  for (int i = 0; i <= n; ++i) cdf[i] = (int32_t)c[i];
  return ICM_OK;
}

int64_t icm_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                     int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdf,
                                     uint8_t* out, int64_t out_capacity) {
  const Tables T{cdfs, cdf_stride, cdf_sizes, offsets, ncdf};
  if (n < 0 || (n > 0 && (!symbols || !indexes)) || !T.ok()) return -1;
  std::vector<Item> items;
  try {
    if (plan_items(symbols, indexes, n, T, items) != ICM_OK) return -1;
    return emit(items, out, out_capacity);
  } catch (const std::bad_alloc&) {
    return -1;
  }
}

int icm_rans_decode_with_indexes(const uint8_t* stream, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                 int ncdf, int32_t* out) {
  void* h = icm_rans_decoder_create(stream, nbytes);
  if (!h) return ICM_ERR_ARG;
  const int rc = icm_rans_decoder_decode(h, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, ncdf, out);
  icm_rans_decoder_destroy(h);
  return rc;
}

void* icm_rans_decoder_create(const uint8_t* stream, int64_t nbytes) {
  if (!stream || nbytes < 8 || (nbytes & 3)) return nullptr;
  Decoder* D = new (std::nothrow) Decoder();
  if (!D) return nullptr;
  try {
    D->words.resize((size_t)(nbytes / 4));
  } catch (const std::bad_alloc&) {
    delete D;
    return nullptr;
  }
  std::memcpy(D->words.data(), stream, (size_t)nbytes);
  D->d.init(D->words.data(), D->words.data() + D->words.size());
  return D;
}

int icm_rans_decoder_decode(void* handle, const int32_t* indexes, int64_t n, const int32_t* cdfs, int cdf_stride,
                            const int32_t* cdf_sizes, const int32_t* offsets, int ncdf, int32_t* out) {
  const Tables T{cdfs, cdf_stride, cdf_sizes, offsets, ncdf};
  if (!handle || n < 0 || (n > 0 && (!indexes || !out)) || !T.ok()) return ICM_ERR_ARG;
  return decode_run(static_cast<Decoder*>(handle)->d, indexes, n, T, out);
}

void icm_rans_decoder_destroy(void* handle) { delete static_cast<Decoder*>(handle); }

}  // extern "C"
