// Jump-ahead tables for MT19937 (host side): lets the sampler's word generator (sampler.hip) produce one
// torch-compatible mt19937 stream with MANY workgroups instead of one serial recurrence.
//
// The raw stream r[t] (r[0..623] = the engine's array, r[t] = r[t-227] ^ twist(r[t-624], r[t-623])) is
// linear over GF(2): every bit of r[t], t >= 1, obeys the recurrence whose characteristic polynomial
// phi (degree 19937) is that of the generator.  With g_J = x^J mod phi,
//       r[1 + J + w]  =  XOR_{i : bit i of g_J set}  r[1 + i + w]          for all w >= 0,
// so the 624-word window that starts segment k of the stream (J = k * kMtSeg) is an XOR of windows of the
// first 19937 + 624 values -- data-parallel -- and all segments can then be generated concurrently.
// phi is found once by Berlekamp-Massey on one output bit (no constants to trust), the g_J by repeated
// multiplication; both are verified by construction in tests (the generated words must equal torch's CPU
// stream bit for bit).  Tables are immutable after creation and cached per device.
#include "common.h"

#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace pyg_hip {
namespace {

typedef std::vector<uint64_t> Poly;  // little-endian coefficient bits
constexpr int kDeg = 19937;
constexpr int kWords = 312;  // 19968 bits

inline int get_bit(const Poly& p, int i) { return (int)((p[(size_t)i >> 6] >> (i & 63)) & 1); }
inline void flip_bit(Poly& p, int i) { p[(size_t)i >> 6] ^= 1ull << (i & 63); }

inline uint32_t twist(uint32_t u, uint32_t v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

// p ^= q << shift (p must be long enough)
inline void xor_shifted(Poly& p, const Poly& q, int qwords, int shift) {
  const int ws = shift >> 6, bs = shift & 63;
  for (int w = 0; w < qwords; ++w) {
    p[(size_t)(w + ws)] ^= q[(size_t)w] << bs;
    if (bs) p[(size_t)(w + ws + 1)] ^= q[(size_t)w] >> (64 - bs);
  }
}

// Berlekamp-Massey over GF(2): connection polynomial C (C[0] = 1) of the bit sequence s; returns L
int berlekamp_massey(const std::vector<uint8_t>& s, Poly& c_out) {
  const int n_bits = (int)s.size();
  const int nw = (n_bits + 64) / 64 + 1;
  Poly c((size_t)nw, 0), b((size_t)nw, 0), t, r((size_t)nw, 0);  // r[i] = s[n - i]
  c[0] = 1;
  b[0] = 1;
  int len = 0, m = 1;
  for (int n = 0; n < n_bits; ++n) {
    for (int w = nw - 1; w > 0; --w) r[(size_t)w] = (r[(size_t)w] << 1) | (r[(size_t)w - 1] >> 63);
    r[0] = (r[0] << 1) | s[(size_t)n];
    uint64_t acc = 0;
    const int lw = len / 64 + 1;
    for (int w = 0; w < lw && w < nw; ++w) acc ^= c[(size_t)w] & r[(size_t)w];
    if (!__builtin_parityll(acc)) {
      ++m;
      continue;
    }
    if (2 * len <= n) {
      t = c;
      Poly cc = c;
      cc.resize((size_t)nw + 2, 0);
      xor_shifted(cc, b, nw - (m >> 6) - 1, m);
      cc.resize((size_t)nw);
      c = cc;
      len = n + 1 - len;
      b = t;
      m = 1;
    } else {
      Poly cc = c;
      cc.resize((size_t)nw + 2, 0);
      xor_shifted(cc, b, nw - (m >> 6) - 1, m);
      cc.resize((size_t)nw);
      c = cc;
      ++m;
    }
  }
  c_out = c;
  return len;
}

struct Tables {
  bool ready = false;
  bool failed = false;
  Poly phi_low;            // phi = x^19937 + sum_j phi_low[j] x^j
  Poly g1;                 // x^kMtSeg mod phi
  std::vector<Poly> g;     // g[k] = x^(k * kMtSeg) mod phi, grown on demand (g[0] unused)
  std::vector<Poly> gs;    // strided: gs[b] = x^((1 + b * kMtStride) * kMtSeg) mod phi (long rounds)
  struct DevList {
    uint16_t* idx = nullptr;
    int count = 0;
    int max_span = 0;  // largest (last - first + 624) over the `parts` equal shares of the list
  };
  std::vector<std::vector<DevList>> dev;  // [device][k]
  std::mutex mu;
};

Tables& tables() {
  static Tables t;
  return t;
}

Poly mulmod(const Poly& sparse, const Poly& dense, const Poly& phi_low) {
  Poly t((size_t)2 * kWords + 2, 0);
  for (int i = 0; i < kDeg; ++i)
    if (get_bit(sparse, i)) xor_shifted(t, dense, kWords, i);
  for (int d = 2 * kDeg - 2; d >= kDeg; --d) {
    if (!((t[(size_t)d >> 6] >> (d & 63)) & 1)) continue;
    t[(size_t)d >> 6] ^= 1ull << (d & 63);
    xor_shifted(t, phi_low, kWords, d - kDeg);
  }
  t.resize((size_t)kWords);
  return t;
}

int init_tables(Tables& t) {
  // one output bit of an arbitrary non-zero stream, from t = 1 on
  const int n_bits = 2 * kDeg + 64;
  std::vector<uint32_t> r((size_t)n_bits + 700);
  r[0] = 5489u;
  for (int i = 1; i < 624; ++i) r[(size_t)i] = 1812433253u * (r[(size_t)i - 1] ^ (r[(size_t)i - 1] >> 30)) + (uint32_t)i;
  for (size_t k = 624; k < r.size(); ++k) r[k] = r[k - 227] ^ twist(r[k - 624], r[k - 623]);
  std::vector<uint8_t> s((size_t)n_bits);
  for (int n = 0; n < n_bits; ++n) s[(size_t)n] = (uint8_t)(r[(size_t)n + 1] & 1u);
  Poly c;
  const int len = berlekamp_massey(s, c);
  if (len != kDeg) return fail(PYG_HIP_ERR_RUNTIME, "mt19937 jump tables: linear complexity %d, expected %d", len, kDeg);
  t.phi_low.assign((size_t)kWords, 0);
  for (int j = 0; j < kDeg; ++j)
    if (get_bit(c, len - j)) flip_bit(t.phi_low, j);
  // g1 = x^kMtSeg mod phi by square-and-multiply
  Poly result((size_t)kWords, 0), base((size_t)kWords, 0);
  result[0] = 1;
  base[0] = 2;
  for (long long e = kMtSeg; e;) {
    if (e & 1) result = mulmod(result, base, t.phi_low);
    e >>= 1;
    if (e) base = mulmod(base, base, t.phi_low);
  }
  t.g1 = result;
  t.g.assign(2, Poly());
  t.g[1] = t.g1;
  // self-check of the window identity on an unrelated stream
  std::vector<uint32_t> q((size_t)kMtSeg + 2000);
  q[0] = 12345u;
  for (int i = 1; i < 624; ++i) q[(size_t)i] = 1812433253u * (q[(size_t)i - 1] ^ (q[(size_t)i - 1] >> 30)) + (uint32_t)i;
  for (size_t k = 624; k < q.size(); ++k) q[k] = q[k - 227] ^ twist(q[k - 624], q[k - 623]);
  for (int w = 0; w < 624; w += 89) {
    uint32_t acc = 0;
    for (int i = 0; i < kDeg; ++i)
      if (get_bit(t.g1, i)) acc ^= q[(size_t)(1 + i + w)];
    if (acc != q[(size_t)(1 + kMtSeg + w)]) return fail(PYG_HIP_ERR_RUNTIME, "mt19937 jump tables: self-check failed");
  }
  t.ready = true;
  return PYG_HIP_OK;
}

}  // namespace

// Device-resident list of the set coefficient positions of g_k = x^(k * kMtSeg) mod phi (1 <= k < kMtMaxSeg, or
// k = 1 + b * kMtStride for the long rounds).
int mt_jump_list(int k, int parts, const uint16_t** idx_dev, int* count, int* max_span) {
  Tables& t = tables();
  std::lock_guard<std::mutex> lock(t.mu);
  if (t.failed) return fail(PYG_HIP_ERR_RUNTIME, "mt19937 jump tables unavailable");
  if (!t.ready) {
    int rc = init_tables(t);
    if (rc != PYG_HIP_OK) {
      t.failed = true;
      return rc;
    }
  }
  PYG_HIP_REQUIRE(k >= 1 && k <= 1 + (kMtMaxSeg - 2) * kMtStride, "mt19937 jump: segment %d out of range", k);
  // k < kMtMaxSeg: the unit chain g[k] = g1 * g[k-1]; larger k (long rounds): only k = 1 + b * kMtStride, through the
  // chain gs[b] = g[kMtStride] * gs[b-1] (one multiplication per list either way, each list built on first use)
  const Poly* poly = nullptr;
  if (k < kMtMaxSeg) {
    while ((int)t.g.size() <= k) t.g.push_back(mulmod(t.g1, t.g.back(), t.phi_low));
    poly = &t.g[(size_t)k];
  } else {
    PYG_HIP_REQUIRE((k - 1) % kMtStride == 0, "mt19937 jump: segment %d is not on the long-round grid", k);
    const int b = (k - 1) / kMtStride;
    while ((int)t.g.size() <= kMtStride) t.g.push_back(mulmod(t.g1, t.g.back(), t.phi_low));
    if (t.gs.empty()) t.gs.push_back(t.g[1]);
    while ((int)t.gs.size() <= b) t.gs.push_back(mulmod(t.g[(size_t)kMtStride], t.gs.back(), t.phi_low));
    poly = &t.gs[(size_t)b];
  }
  int dev = 0;
  PYG_HIP_CHECK(hipGetDevice(&dev));
  if ((int)t.dev.size() <= dev) t.dev.resize((size_t)dev + 1);
  if ((int)t.dev[(size_t)dev].size() <= k) t.dev[(size_t)dev].resize((size_t)k + 1);
  Tables::DevList& dl = t.dev[(size_t)dev][(size_t)k];
  if (!dl.idx) {
    std::vector<uint16_t> host;
    for (int i = 0; i < kDeg; ++i)
      if (get_bit(*poly, i)) host.push_back((uint16_t)i);
    uint16_t* p = nullptr;
    PYG_HIP_CHECK(hipMalloc(&p, sizeof(uint16_t) * std::max<size_t>(host.size(), 1)));
    // immutable table, uploaded once per device; synchronous on purpose (first use only)
    PYG_HIP_CHECK(hipMemcpy(p, host.data(), sizeof(uint16_t) * host.size(), hipMemcpyHostToDevice));
    dl.idx = p;
    dl.count = (int)host.size();
    const int n = dl.count;
    for (int part = 0; part < parts; ++part) {
      const int j0 = (int)((int64_t)part * n / parts), j1 = (int)((int64_t)(part + 1) * n / parts);
      if (j0 < j1) dl.max_span = std::max(dl.max_span, (int)host[(size_t)j1 - 1] - (int)host[(size_t)j0] + 624);
    }
  }
  *idx_dev = dl.idx;
  *count = dl.count;
  *max_span = dl.max_span;
  return PYG_HIP_OK;
}

}  // namespace pyg_hip
