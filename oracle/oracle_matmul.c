/*
 * oracle_matmul.c -- CPU restatement of pyg-lib's segment_matmul / grouped_matmul.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pyg_lib_amd/ may import, link or call this file; it
 * is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Follows (paths relative to the pyg-lib v0.9.0 tree):
 *   pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:410-439  segment_matmul_kernel: sizes = ptr[1:]-ptr[:-1]
 *       (pyg_lib/csrc/utils/convert.cpp:6-9), out = new_empty({N, M}),
 *       out[ptr[b]:ptr[b+1]] = input[ptr[b]:ptr[b+1]] @ other[b]  (at::matmul_out per segment,
 *       :195-201,428-434).
 *   pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:281-312  grouped_matmul_kernel: outs[i] = inputs[i] @ others[i].
 *   pyg_lib/ops/__init__.py:169-171                 bias: out[ptr[i]:ptr[i+1]] += bias[i] (in dtype T).
 *
 * The GEMM arithmetic itself lives in libtorch (at::matmul -> oneDNN/MKL), a third-party
 * dependency outside the reference tree, so only tolerance-level parity is defined for it
 * (test/ops/test_matmul.py:14-45 compares against `@` with atol 1e-6).  The reference CPU kernel
 * cannot be built here (needs the un-vendored parallel-hashmap and a cmake-generated config.h),
 * so this oracle is pinned by tests/golden/matmul_*.npz, generated with torch's own `@` on CPU by
 * tests/golden/make_matmul_golden.py -- exactly the expectation the reference's tests encode.
 *
 * Numerics: products are accumulated in double and rounded ONCE to the output type (fp32, or
 * bf16/fp16 round-to-nearest-even), i.e. the correctly rounded result that both the reference
 * (fp32 accumulate, one rounding) and the HIP kernel (fp32 MFMA accumulate, one rounding)
 * approximate.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf16_to_f32(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  uint32_t lsb = (u >> 16) & 1u;
  u += 0x7fffu + lsb;
  return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else {
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_f16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf (incl. >= 65520) */
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (a < 0x33000001u) return (uint16_t)sign; /* underflow to zero */
  int32_t exp = (int32_t)(a >> 23) - 127 + 15;
  uint32_t man = a & 0x7fffffu;
  if (exp <= 0) {
    man |= 0x800000u;
    int shift = 14 - exp;
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;
  return (uint16_t)(sign | half);
}

/* dtype codes mirror include/pyg_hip.h (kept in sync by tests/test_abi.py). */
enum { O_F32 = 0, O_F64 = 1, O_F16 = 2, O_BF16 = 3 };

static double load_elem(int dtype, const void* p, int64_t i) {
  switch (dtype) {
    case O_F32: return ((const float*)p)[i];
    case O_F64: return ((const double*)p)[i];
    case O_F16: return f16_to_f32(((const uint16_t*)p)[i]);
    default: return bf16_to_f32(((const uint16_t*)p)[i]);
  }
}

static void store_elem(int dtype, void* p, int64_t i, double v) {
  switch (dtype) {
    case O_F32: ((float*)p)[i] = (float)v; break;
    case O_F64: ((double*)p)[i] = v; break;
    case O_F16: ((uint16_t*)p)[i] = f32_to_f16((float)v); break;
    default: ((uint16_t*)p)[i] = f32_to_bf16((float)v); break;
  }
}

/* out[rows, M] = x[rows, K] @ w[K, M] (+ bias[M]); row-major; double accumulation. */
static void gemm_rows(int dtype, const void* x, const void* w, const void* bias, void* out,
                      int64_t rows, int64_t K, int64_t M) {
  double* wd = (double*)malloc(sizeof(double) * (size_t)(K * M > 0 ? K * M : 1));
  for (int64_t i = 0; i < K * M; ++i) wd[i] = load_elem(dtype, w, i);
#pragma omp parallel
  {
    double* acc = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
#pragma omp for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
      for (int64_t m = 0; m < M; ++m) acc[m] = 0.0;
      for (int64_t k = 0; k < K; ++k) {
        const double xv = load_elem(dtype, x, r * K + k);
        const double* wr = wd + k * M;
        for (int64_t m = 0; m < M; ++m) acc[m] += xv * wr[m];
      }
      for (int64_t m = 0; m < M; ++m) {
        if (bias) {
          /* reference: `out` is materialised in T, then `out += bias` in T */
          store_elem(dtype, out, r * M + m, acc[m]);
          double o = load_elem(dtype, out, r * M + m) + load_elem(dtype, bias, m);
          store_elem(dtype, out, r * M + m, o);
        } else {
          store_elem(dtype, out, r * M + m, acc[m]);
        }
      }
    }
    free(acc);
  }
  free(wd);
}

static size_t esize(int dtype) { return dtype == O_F64 ? 8 : (dtype == O_F32 ? 4 : 2); }

/* Returns 0, or -1 if ptr is not a valid segmentation of [0, N] in the reference's sense
 * (split_with_sizes would throw, matmul_kernel.cpp:428). */
int oracle_segment_matmul(int dtype, const void* input, const int64_t* ptr, const void* other,
                          const void* bias, void* out, int64_t N, int64_t K, int64_t M,
                          int64_t B) {
  const size_t es = esize(dtype);
  for (int64_t b = 0; b < B; ++b) {
    if (ptr[b + 1] < ptr[b] || ptr[b] < 0 || ptr[b + 1] > N) return -1;
  }
  for (int64_t b = 0; b < B; ++b) {
    const int64_t r0 = ptr[b], rows = ptr[b + 1] - ptr[b];
    gemm_rows(dtype, (const char*)input + (size_t)(r0 * K) * es,
              (const char*)other + (size_t)(b * K * M) * es,
              bias ? (const char*)bias + (size_t)(b * M) * es : NULL,
              (char*)out + (size_t)(r0 * M) * es, rows, K, M);
  }
  return 0;
}

/* One group: out = input[rows, k] @ other[k, m]. */
int oracle_matmul(int dtype, const void* input, const void* other, void* out, int64_t rows,
                  int64_t K, int64_t M) {
  gemm_rows(dtype, input, other, NULL, out, rows, K, M);
  return 0;
}
