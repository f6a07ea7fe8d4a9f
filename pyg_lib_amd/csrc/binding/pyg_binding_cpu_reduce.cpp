// libpyg.so, CPU kernels of scatter_* / segment_*_coo / gather_coo / segment_*_csr / gather_csr / softmax_csr: see
// cpu_reduce.h.  Every loop keeps the reference's order of operations per output element (sequential over the reduced
// axis, opmath accumulators, one rounding where the reference rounds), so the work can be split over leading slices
// AND column blocks without changing a bit: columns never interact.
#include "cpu_reduce.h"

#include <ATen/OpMathType.h>
#include <ATen/Parallel.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pyg_hip.h"

namespace pyg_amd {
namespace cpu {
namespace {

constexpr int64_t kColBlock = 32;  // columns per task

// tasks = B x ceil(K / kColBlock); fn(b, k0, k1)
template <typename F>
void for_slices_and_column_blocks(int64_t B, int64_t K, int64_t work_per_column, const F& fn) {
  const int64_t cb = (K + kColBlock - 1) / kColBlock;
  const int64_t tasks = B * cb;
  const int64_t grain = std::max<int64_t>(1, at::internal::GRAIN_SIZE / std::max<int64_t>(work_per_column * kColBlock, 1));
  at::parallel_for(0, tasks, grain, [&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t b = t / cb, c = t % cb;
      fn(b, c * kColBlock, std::min(K, (c + 1) * kColBlock));
    }
  });
}

inline void check_index(int64_t i, int64_t N) {
  TORCH_CHECK(i >= 0 && i < N, "index ", i, " is out of bounds for dimension of size ", N);
}

template <typename scalar_t>
void scatter_typed(int op, const scalar_t* src, const int64_t* index, int64_t isb, int64_t ise, int64_t isk,
                   scalar_t* out, int64_t* arg, int64_t B, int64_t E, int64_t K, int64_t N, bool coo) {
  using opmath_t = at::opmath_type<scalar_t>;
  for_slices_and_column_blocks(B, K, E, [&](int64_t b, int64_t k0, int64_t k1) {
    const scalar_t* s = src + b * E * K;
    scalar_t* o = out + b * N * K;
    int64_t* a = arg ? arg + b * N * K : nullptr;
    const int64_t* ix = index + b * isb;
    if (op == PYG_REDUCE_SUM && coo) {
      // sorted index: accumulate runs of equal indices in opmath, seeded from `out` (segment_coo_kernel.cpp:104-166)
      // (a COO index never varies along k -- it is broadcast there --, so column k0's entries stand for the block)
      if (E == 0) return;
      ix += k0 * isk;
      opmath_t acc[kColBlock];
      int64_t cur = ix[0];
      check_index(cur, N);
      for (int64_t k = k0; k < k1; ++k) acc[k - k0] = static_cast<opmath_t>(o[cur * K + k]);
      for (int64_t e = 0; e < E; ++e) {
        for (int64_t k = k0; k < k1; ++k) acc[k - k0] += static_cast<opmath_t>(s[e * K + k]);
        const bool last = e == E - 1;
        const int64_t next = last ? cur : ix[(e + 1) * ise];
        if (last || next != cur) {
          for (int64_t k = k0; k < k1; ++k) o[cur * K + k] = static_cast<scalar_t>(acc[k - k0]);
          if (!last) {
            cur = next;
            check_index(cur, N);
            for (int64_t k = k0; k < k1; ++k) acc[k - k0] = static_cast<opmath_t>(o[cur * K + k]);
          }
        }
      }
      return;
    }
    for (int64_t e = 0; e < E; ++e) {
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t i = ix[e * ise + k * isk];
        check_index(i, N);
        scalar_t* slot = o + i * K + k;
        const scalar_t v = s[e * K + k];
        switch (op) {
          case PYG_REDUCE_SUM:  // scatter_kernel.cpp:113-128: one rounding per element
            *slot = static_cast<scalar_t>(static_cast<opmath_t>(*slot) + static_cast<opmath_t>(v));
            break;
          case PYG_REDUCE_MUL:  // scatter_kernel.cpp:203-222
            *slot = static_cast<scalar_t>(static_cast<opmath_t>(*slot) * static_cast<opmath_t>(v));
            break;
          case PYG_REDUCE_MIN:  // strict <: first match wins (scatter_kernel.cpp:333-355)
            if (v < *slot) {
              *slot = v;
              a[i * K + k] = e;
            }
            break;
          default:  // PYG_REDUCE_MAX
            if (v > *slot) {
              *slot = v;
              a[i * K + k] = e;
            }
            break;
        }
      }
    }
  });
}

template <typename scalar_t>
void segment_csr_typed(int op, const scalar_t* src, const int64_t* indptr, int64_t stride, scalar_t* out, int64_t* arg,
                       int64_t leading, int64_t rows, int64_t E, int64_t K) {
  using opmath_t = at::opmath_type<scalar_t>;
  const int64_t N = leading * rows;
  const int64_t cb = (K + kColBlock - 1) / kColBlock;
  const int64_t avg_row = rows > 0 ? std::max<int64_t>(E / rows, 1) : 1;
  const int64_t grain = std::max<int64_t>(1, at::internal::GRAIN_SIZE / (avg_row * kColBlock));
  at::parallel_for(0, N * cb, grain, [&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t n = t / cb, c = t % cb;
      const int64_t k0 = c * kColBlock, k1 = std::min(K, k0 + kColBlock);
      const int64_t slice = n / rows, row = n % rows;
      const int64_t* ip = indptr + slice * stride + row;
      const int64_t rs = ip[0], re = ip[1];
      TORCH_CHECK(rs >= 0 && re <= E && rs <= re, "indptr must be non-decreasing within [0, ", E, "]");
      const scalar_t* s = src + slice * E * K;
      scalar_t* o = out + n * K;
      if (op == 0 || op == 1) {  // sum (seeded from out) / mean (overwrites): segment_csr_kernel.cpp:104-140, 226-262
        opmath_t acc[kColBlock];
        for (int64_t k = k0; k < k1; ++k) acc[k - k0] = op == 0 ? static_cast<opmath_t>(o[k]) : static_cast<opmath_t>(0);
        for (int64_t e = rs; e < re; ++e)
          for (int64_t k = k0; k < k1; ++k) acc[k - k0] += static_cast<opmath_t>(s[e * K + k]);
        if (op == 0) {
          for (int64_t k = k0; k < k1; ++k) o[k] = static_cast<scalar_t>(acc[k - k0]);
        } else {
          const opmath_t denom = static_cast<opmath_t>(re - rs > 0 ? re - rs : 1);
          for (int64_t k = k0; k < k1; ++k) o[k] = static_cast<scalar_t>(acc[k - k0] / denom);
        }
      } else {  // min / max with first-match arg (segment_csr_kernel.cpp:354-396)
        int64_t* a = arg + n * K;
        for (int64_t e = rs; e < re; ++e)
          for (int64_t k = k0; k < k1; ++k) {
            const scalar_t v = s[e * K + k];
            if (op == 2 ? v < o[k] : v > o[k]) {
              o[k] = v;
              a[k] = e;
            }
          }
      }
    }
  });
}

template <typename scalar_t>
void softmax_typed(const scalar_t* src, const int64_t* ptr, scalar_t* out, int64_t outer, int64_t D, int64_t inner,
                   int64_t groups) {
  // softmax_kernel.cpp:55-155: per group [beg, end) of D and per head (o, q): max, exp and running sum in source order,
  // divide; a group of one position is 1.  All arithmetic in scalar_t, as the reference's.
  at::parallel_for(0, groups * outer, 1, [&](int64_t t0, int64_t t1) {
    std::vector<scalar_t> mx(inner), sm(inner);
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t g = t / outer, o = t % outer;
      const int64_t beg = ptr[g], end = ptr[g + 1];
      TORCH_CHECK(beg >= 0 && end <= D && beg <= end, "ptr must be non-decreasing within [0, ", D, "]");
      const scalar_t* s = src + o * D * inner;
      scalar_t* y = out + o * D * inner;
      if (end - beg == 1) {
        std::fill(y + beg * inner, y + end * inner, static_cast<scalar_t>(1));
        continue;
      }
      std::fill(mx.begin(), mx.end(), std::numeric_limits<scalar_t>::lowest());
      std::fill(sm.begin(), sm.end(), static_cast<scalar_t>(0));
      for (int64_t p = beg; p < end; ++p)
        for (int64_t q = 0; q < inner; ++q) mx[q] = std::max(mx[q], s[p * inner + q]);
      for (int64_t p = beg; p < end; ++p)
        for (int64_t q = 0; q < inner; ++q) {
          const scalar_t v = std::exp(s[p * inner + q] - mx[q]);
          sm[q] += v;
          y[p * inner + q] = v;
        }
      for (int64_t p = beg; p < end; ++p)
        for (int64_t q = 0; q < inner; ++q) y[p * inner + q] /= sm[q];
    }
  });
}

template <typename scalar_t>
void softmax_backward_typed(const scalar_t* out, const scalar_t* og, const int64_t* ptr, scalar_t* ig, int64_t outer,
                            int64_t D, int64_t inner, int64_t groups) {
  // softmax_kernel.cpp:157-233
  at::parallel_for(0, groups * outer, 1, [&](int64_t t0, int64_t t1) {
    std::vector<scalar_t> sm(inner);
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t g = t / outer, o = t % outer;
      const int64_t beg = ptr[g], end = ptr[g + 1];
      TORCH_CHECK(beg >= 0 && end <= D && beg <= end, "ptr must be non-decreasing within [0, ", D, "]");
      const scalar_t* y = out + o * D * inner;
      const scalar_t* dy = og + o * D * inner;
      scalar_t* dx = ig + o * D * inner;
      std::fill(sm.begin(), sm.end(), static_cast<scalar_t>(0));
      for (int64_t p = beg; p < end; ++p)
        for (int64_t q = 0; q < inner; ++q) sm[q] += y[p * inner + q] * dy[p * inner + q];
      for (int64_t p = beg; p < end; ++p)
        for (int64_t q = 0; q < inner; ++q) dx[p * inner + q] = y[p * inner + q] * (dy[p * inner + q] - sm[q]);
    }
  });
}

}  // namespace

void scatter(int op, const at::Tensor& src_c, const int64_t* index, int64_t isb, int64_t ise, int64_t isk, at::Tensor& out,
             int64_t* arg, int64_t B, int64_t E, int64_t K, int64_t N, bool coo) {
  AT_DISPATCH_ALL_TYPES_AND2(at::ScalarType::Half, at::ScalarType::BFloat16, src_c.scalar_type(), "scatter_cpu", [&] {
    scatter_typed<scalar_t>(op, src_c.data_ptr<scalar_t>(), index, isb, ise, isk, out.data_ptr<scalar_t>(), arg, B, E, K, N,
                            coo);
  });
}

void fill_identity(int op, at::Tensor& out) {
  AT_DISPATCH_ALL_TYPES_AND2(at::ScalarType::Half, at::ScalarType::BFloat16, out.scalar_type(), "fill_identity_cpu", [&] {
    out.fill_(op == PYG_REDUCE_MIN ? std::numeric_limits<scalar_t>::max() : std::numeric_limits<scalar_t>::lowest());
  });
}

void gather_coo(const at::Tensor& src_c, const int64_t* index, at::Tensor& out, int64_t B, int64_t E, int64_t K, int64_t N) {
  // segment_coo_kernel.cpp:716-738: out[b, e, :] = src[b, index[b, e], :]
  const int64_t row_bytes = K * (int64_t)src_c.element_size();
  const char* s = static_cast<const char*>(src_c.data_ptr());
  char* o = static_cast<char*>(out.data_ptr());
  const int64_t grain = std::max<int64_t>(1, at::internal::GRAIN_SIZE / std::max<int64_t>(K, 1));
  at::parallel_for(0, B * E, grain, [&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t b = t / E;
      const int64_t i = index[t];
      check_index(i, N);
      std::memcpy(o + t * row_bytes, s + (b * N + i) * row_bytes, (size_t)row_bytes);
    }
  });
}

void segment_csr(int op, const at::Tensor& src_c, const int64_t* indptr, int64_t stride, at::Tensor& out, int64_t* arg,
                 int64_t leading, int64_t rows, int64_t E, int64_t K) {
  if (leading * rows == 0 || K == 0) return;
  AT_DISPATCH_ALL_TYPES_AND2(at::ScalarType::Half, at::ScalarType::BFloat16, src_c.scalar_type(), "segment_csr_cpu", [&] {
    segment_csr_typed<scalar_t>(op, src_c.data_ptr<scalar_t>(), indptr, stride, out.data_ptr<scalar_t>(), arg, leading, rows, E,
                                K);
  });
}

void gather_csr(const at::Tensor& src_c, const int64_t* indptr, int64_t stride, at::Tensor& out, int64_t leading,
                int64_t rows, int64_t E, int64_t K) {
  // segment_csr_kernel.cpp:612-645: out[slice, e, :] = src[slice, row, :] for e in [indptr[row], indptr[row + 1])
  const int64_t row_bytes = K * (int64_t)src_c.element_size();
  const char* s = static_cast<const char*>(src_c.data_ptr());
  char* o = static_cast<char*>(out.data_ptr());
  const int64_t avg_row = rows > 0 ? std::max<int64_t>(E / rows, 1) : 1;
  const int64_t grain = std::max<int64_t>(1, at::internal::GRAIN_SIZE / std::max<int64_t>(avg_row * K, 1));
  at::parallel_for(0, leading * rows, grain, [&](int64_t n0, int64_t n1) {
    for (int64_t n = n0; n < n1; ++n) {
      const int64_t slice = n / rows, row = n % rows;
      const int64_t* ip = indptr + slice * stride + row;
      TORCH_CHECK(ip[0] >= 0 && ip[1] <= E && ip[0] <= ip[1], "indptr must be non-decreasing within [0, ", E, "]");
      for (int64_t e = ip[0]; e < ip[1]; ++e)
        std::memcpy(o + (slice * E + e) * row_bytes, s + n * row_bytes, (size_t)row_bytes);
    }
  });
}

void softmax_csr(const at::Tensor& src, const int64_t* ptr, at::Tensor& out, int64_t outer, int64_t D, int64_t inner,
                 int64_t groups) {
  AT_DISPATCH_FLOATING_TYPES(src.scalar_type(), "softmax_csr_forward_kernel_impl", [&] {
    softmax_typed<scalar_t>(src.data_ptr<scalar_t>(), ptr, out.data_ptr<scalar_t>(), outer, D, inner, groups);
  });
}

void softmax_csr_backward(const at::Tensor& out, const at::Tensor& out_grad, const int64_t* ptr, at::Tensor& in_grad,
                          int64_t outer, int64_t D, int64_t inner, int64_t groups) {
  AT_DISPATCH_FLOATING_TYPES(out.scalar_type(), "softmax_csr_backward_kernel_impl", [&] {
    softmax_backward_typed<scalar_t>(out.data_ptr<scalar_t>(), out_grad.data_ptr<scalar_t>(), ptr,
                                     in_grad.data_ptr<scalar_t>(), outer, D, inner, groups);
  });
}

}  // namespace cpu
}  // namespace pyg_amd
