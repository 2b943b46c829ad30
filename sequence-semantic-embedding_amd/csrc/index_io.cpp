// targetEncodingIndex.tsv text <-> numbers, host side.
//
// The reference writes every component of every target encoding with Python's str(np.float32)
// (sse_index.py:93-95: ",".join([str(n) for n in vec])) and reads it back with float()
// (sse_evaluator.py:87, sse_demo.py:87) -- after the GPU encode these two loops ARE the run time of
// sse_index / Evaluator.__init__.  These routines produce / consume byte-identical text:
//   format: numpy's scalar str() = shortest decimal that round-trips the float32 (Dragon4 "unique"),
//           positional when 1e-4 <= |x| < 1e16 (at least one digit after the point), else scientific
//           with a two-digit exponent; the shortest digits come from std::to_chars (Ryu);
//   parse:  correctly rounded decimal -> float64, as Python's float().
// Rows are independent: both directions split the rows over std::threads.
#include <mutex>
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/sse_hip.h"

namespace {

// one float32 -> text at p (needs <= 24 bytes); returns the new end
char *format_f32(float x, char *p) {
  if (std::isnan(x)) {
    memcpy(p, "nan", 3);
    return p + 3;
  }
  if (std::isinf(x)) {
    if (x < 0) *p++ = '-';
    memcpy(p, "inf", 3);
    return p + 3;
  }
  const double ax = std::fabs((double)x);
  if (x == 0.0f) {
    if (std::signbit(x)) *p++ = '-';
    memcpy(p, "0.0", 3);
    return p + 3;
  }
  char tmp[32];
  const auto r = std::to_chars(tmp, tmp + sizeof tmp - 1, x, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX
  *r.ptr = 0;
  if (!(ax >= 1e-4 && ax < 1e16)) {
    const size_t n = (size_t)(r.ptr - tmp);
    memcpy(p, tmp, n);
    return p + n;
  }
  const char *s = tmp;
  if (*s == '-') *p++ = *s++;
  char digits[16];
  int nd = 0;
  for (; s < r.ptr && *s != 'e'; ++s)
    if (*s != '.') digits[nd++] = *s;
  const int e10 = atoi(s + 1);  // value = d.ddd * 10^e10
  if (e10 >= 0) {
    for (int i = 0; i <= e10; ++i) *p++ = (i < nd) ? digits[i] : '0';
    *p++ = '.';
    if (nd > e10 + 1) {
      for (int i = e10 + 1; i < nd; ++i) *p++ = digits[i];
    } else {
      *p++ = '0';
    }
  } else {
    *p++ = '0';
    *p++ = '.';
    for (int i = 0; i < -e10 - 1; ++i) *p++ = '0';
    for (int i = 0; i < nd; ++i) *p++ = digits[i];
  }
  return p;
}

int n_threads(int64_t rows) {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (n > 32) n = 32;
  if ((int64_t)n > rows) n = (int)std::max<int64_t>(rows, 1);
  return n;
}

template <class F>
void parallel_rows(int64_t rows, F f) {
  const int nt = n_threads(rows / 64);
  if (nt <= 1) {
    f(0, rows);
    return;
  }
  std::vector<std::thread> th;
  const int64_t per = (rows + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t a = t * per, b = std::min(rows, a + per);
    if (a < b) th.emplace_back(f, a, b);
  }
  for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

int64_t sse_format_rows_stride(int32_t S) { return (int64_t)S * 24; }  // longest: -9999999800000000.0 (19) + separator

int sse_format_rows_f32(const float *rows, int64_t n_rows, int32_t S, char *out, int64_t *lengths) {
  if (!rows || !out || !lengths || n_rows < 0 || S < 1) return 1;
  const int64_t stride = sse_format_rows_stride(S);
  parallel_rows(n_rows, [=](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r) {
      char *p = out + r * stride, *p0 = p;
      const float *v = rows + r * S;
      for (int32_t i = 0; i < S; ++i) {
        if (i) *p++ = ',';
        p = format_f32(v[i], p);
      }
      lengths[r] = p - p0;
    }
  });
  return 0;
}

int sse_parse_rows_f64(const char *text, const int64_t *offsets, int64_t n_rows, int32_t S, double *out, int64_t *bad_row) {
  if (!text || !offsets || !out || n_rows < 0 || S < 1) return 1;
  std::vector<int64_t> bad((size_t)std::max<int64_t>(n_rows, 1), -1);
  int64_t first_bad = -1;
  parallel_rows(n_rows, [&, text, offsets, out](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r) {
      const char *p = text + offsets[r], *end = text + offsets[r + 1];
      while (end > p && (end[-1] == '\n' || end[-1] == '\r' || end[-1] == ' ')) --end;
      double *o = out + r * S;
      int32_t i = 0;
      bool ok = true;
      while (p <= end && i < S) {
        const char *q = (const char *)memchr(p, ',', (size_t)(end - p));
        if (!q) q = end;
        while (p < q && *p == ' ') ++p;
        const char *qe = q;
        while (qe > p && qe[-1] == ' ') --qe;
        if (p < qe && *p == '+') ++p;  // float() accepts a leading '+', from_chars does not
        auto res = std::from_chars(p, qe, o[i]);
        if (res.ec == std::errc::result_out_of_range && res.ptr == qe && qe - p < 400) {
          char tok[400];  // float('1e400') = inf, float('1e-400') = 0.0: strtod gives Python's answers
          memcpy(tok, p, (size_t)(qe - p));
          tok[qe - p] = 0;
          o[i] = strtod(tok, nullptr);
          res.ec = std::errc();
        }
        if (res.ec != std::errc() || res.ptr != qe) {
          ok = false;
          break;
        }
        ++i;
        p = q + 1;
        if (q == end) break;
      }
      if (!ok || i != S || p <= end) bad[(size_t)r] = r;  // wrong count (too few or trailing fields) or malformed number
    }
  });
  for (int64_t r = 0; r < n_rows; ++r)
    if (bad[(size_t)r] >= 0) {
      first_bad = r;
      break;
    }
  if (bad_row) *bad_row = first_bad;
  return first_bad >= 0 ? 2 : 0;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a byte range, slicing-by-8: the checksum of LevelDB table
// blocks and of tensor bytes in TensorFlow V2 checkpoints (tf_checkpoint.py verifies both; a pure-Python loop takes
// seconds per embedding table).  `seed` chains calls (0 for a fresh sum).
uint32_t sse_crc32c(const void *data, int64_t n, uint32_t seed) {
  static uint32_t tab[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
  });
  const unsigned char *p = (const unsigned char *)data;
  uint32_t crc = ~seed;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^ tab[3][hi & 0xFF] ^
          tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) crc = tab[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

}  // extern "C"
