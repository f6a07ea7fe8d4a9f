// CPU kernels of the scatter / segment_coo / segment_csr / softmax_csr families (dispatch key `CPU`).
//
// Same calling convention as the C-ABI entry points the device path uses (pyg_hip_scatter, pyg_hip_gather_coo,
// pyg_hip_segment_csr, pyg_hip_gather_csr, pyg_hip_softmax_csr(_backward); include/pyg_hip.h), so the operator fronts
// in pyg_binding_reduce.cpp / pyg_binding_csr.cpp serve both devices: shape checks, broadcasting, output allocation
// and autograd are shared, only the last call differs.  Written for this build from the reference's CPU algorithms
// (pyg_lib/csrc/ops/cpu/scatter_kernel.cpp:29-511, segment_coo_kernel.cpp:31-746, segment_csr_kernel.cpp:31-650,
// softmax_kernel.cpp:55-233): same accumulation order and opmath, hence bit-identical results (tests/test_cpu_key.py
// checks them against outputs recorded from the real reference build).  Nothing under oracle/ is used.
#pragma once

#include <ATen/ATen.h>

namespace pyg_amd {
namespace cpu {

// op: PYG_REDUCE_SUM / MUL / MIN / MAX.  `index` is addressed as index[b * isb + e * ise + k * isk] (broadcast views are
// read in place).  `coo` != 0: the index is sorted along e -- sums accumulate runs of equal indices in opmath, seeded
// from `out` (segment_coo_kernel.cpp:104-166).  min / max: `out` holds the running state (identity or the caller's
// values), `arg` the sentinel E; strict compare, first match wins.
void scatter(int op, const at::Tensor& src_c, const int64_t* index, int64_t isb, int64_t ise, int64_t isk, at::Tensor& out,
             int64_t* arg, int64_t B, int64_t E, int64_t K, int64_t N, bool coo);

// out.fill_(numeric_limits::max() / lowest()) for min / max
void fill_identity(int op, at::Tensor& out);

void gather_coo(const at::Tensor& src_c, const int64_t* index, at::Tensor& out, int64_t B, int64_t E, int64_t K, int64_t N);

// op: 0 sum, 1 mean, 2 min, 3 max.  indptr: [leading, rows + 1] with `stride` elements between slices (0 = shared).
void segment_csr(int op, const at::Tensor& src_c, const int64_t* indptr, int64_t stride, at::Tensor& out, int64_t* arg,
                 int64_t leading, int64_t rows, int64_t E, int64_t K);

void gather_csr(const at::Tensor& src_c, const int64_t* indptr, int64_t stride, at::Tensor& out, int64_t leading,
                int64_t rows, int64_t E, int64_t K);

// src / out viewed as [outer, D, inner]; ptr [groups + 1] cuts D
void softmax_csr(const at::Tensor& src, const int64_t* ptr, at::Tensor& out, int64_t outer, int64_t D, int64_t inner,
                 int64_t groups);
void softmax_csr_backward(const at::Tensor& out, const at::Tensor& out_grad, const int64_t* ptr, at::Tensor& in_grad,
                          int64_t outer, int64_t D, int64_t inner, int64_t groups);

}  // namespace cpu
}  // namespace pyg_amd
