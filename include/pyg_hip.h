/*
 * pyg_hip.h -- C-ABI of the MI355X-native (gfx950) pyg-lib hot path.
 *
 * This is the drop-in boundary: a torch-free shared library (libpyg_hip.so) with plain
 * pointers, sizes and a hipStream_t.  The thin torch binding (libpyg.so, sources in
 * pyg_lib_amd/csrc/binding/) registers the reference's `pyg::*` operator schemas and calls
 * nothing but these entry points; INTEGRATION.md shows the binding a pyg-lib maintainer
 * would add.  Every entry point cites the reference interface it replaces
 * (paths relative to the pyg-lib source tree, v0.9.0).
 *
 * Conventions
 *  - every function returns PYG_HIP_OK (0) or a negative pyg_hip_status; no exception crosses
 *    this boundary.  pyg_hip_last_error() returns a thread-local message for the last failure
 *    (the binding turns it into TORCH_CHECK -> RuntimeError, the reference's error convention:
 *    pyg_lib/csrc/ops/matmul.cpp:14-32,49-55).
 *  - all data pointers are DEVICE pointers unless the name ends in `_host`.
 *  - inputs are borrowed and never written; outputs are caller-allocated, or allocated through
 *    the caller's allocator callback when their size is data dependent (sampler).
 *  - `stream` is a hipStream_t passed as void* so that C callers need no HIP headers.  All work
 *    is enqueued on it; only the sampler synchronises it (data-dependent output sizes).
 */
#ifndef PYG_HIP_H_
#define PYG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYG_HIP_API __attribute__((visibility("default")))

typedef enum {
  PYG_HIP_OK = 0,
  PYG_HIP_ERR_INVALID = -1,     /* argument check failed (reference: TORCH_CHECK)          */
  PYG_HIP_ERR_UNSUPPORTED = -2, /* valid in the reference, not implemented on device yet     */
  PYG_HIP_ERR_RUNTIME = -3,     /* HIP runtime / launch failure                              */
  PYG_HIP_ERR_WORKSPACE = -4    /* workspace too small                                       */
} pyg_hip_status;

/* Arithmetic type of a buffer (AT_DISPATCH_ALL_TYPES_AND2(Half, BFloat16),
 * pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:419-421). */
typedef enum {
  PYG_F32 = 0,
  PYG_F64 = 1,
  PYG_F16 = 2,
  PYG_BF16 = 3,
  PYG_I8 = 4,
  PYG_U8 = 5,
  PYG_I16 = 6,
  PYG_I32 = 7,
  PYG_I64 = 8
} pyg_dtype;

/* ---- library ----------------------------------------------------------------------------- */

/* Version of THIS interface: bumped whenever a signature or the meaning of an argument changes, so that a caller built
 * against an older header can tell (pyg_hip_abi_version() != the PYG_HIP_ABI_VERSION it was compiled with).
 *   8: round 6 -- pyg_hip_segment_csr_ws / pyg_hip_gather_csr_ws / pyg_hip_csr_hub_workspace_size (scratch for hub rows);
 *      pyg_hip_scatter uses the dead parts of its workspace for the same purpose (no change for its callers).
 *   7: round 6 -- pyg_hip_rgcn_relation::scatter_rows (rows of the relation's destination segment).
 *   6: round 5 -- PYG_HIP_RGCN_GROUPED + pyg_hip_rgcn_grouped_workspace_size (atomic-free fused layer), PYG_HIP_SCATTER_DETERMINISTIC.
 *   5: round 5 -- pyg_hip_hetero_neighbor_sample_batched, PYG_HIP_SCATTER_CAS / PYG_HIP_RGCN_* flag bits (`checked` of
 *      pyg_hip_rgcn_fused became a bit field), pyg_hip_set_float_atomic_mode, pyg_hip_atomic_selftest,
 *      pyg_hip_sampler_table_cache_release; the weight-gradient workspace holds partial slabs instead of an fp32 image.
 *   4: round 4 -- `flags` in front of `stream` in pyg_hip_segment_matmul / pyg_hip_grouped_matmul, `index_sorted` of
 *      pyg_hip_scatter became a bit field, pyg_hip_matmul_set_schedule / _set_f32_split removed, fp32 default = IEEE MFMAs. */
#define PYG_HIP_ABI_VERSION 8
PYG_HIP_API int pyg_hip_abi_version(void);
/* Replaces pyg::cuda_version (pyg_lib/csrc/library.cpp:19-29): returns the HIP runtime version
 * the library was built against (HIP_VERSION), never -1. */
PYG_HIP_API int64_t pyg_hip_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
PYG_HIP_API const char* pyg_hip_last_error(void);
/* Name of the offload architecture the kernels were compiled for ("gfx950"). */
PYG_HIP_API const char* pyg_hip_arch(void);

/*
 * Floating-point accumulation through atomics.  The weight gradients, the large scatter / COO sums and the CSR family are
 * atomic-free and bit-reproducible (as the reference's sequential CPU loops, ops/cpu/scatter_kernel.cpp:29-127,
 * ops/autograd/matmul_kernel.cpp:92-107).  What still adds through atomics -- small or element-wise indexed
 * scatter sums, float64 sums, pyg_hip_rgcn_fused -- uses the hardware's floating-point atomic adds by default
 * (global_atomic_add_f32 / _f64 / _pk_add_bf16 / _pk_add_f16) and can be switched to compare-and-swap loops on the containing
 * word: per call (PYG_HIP_SCATTER_CAS, PYG_HIP_RGCN_CAS) or as the process-wide default below (0 = hardware adds,
 * 1 = CAS loops; initial value from the environment, PYG_HIP_FLOAT_ATOMICS=hw|cas).  Same sums up to the order the adds
 * land in; the CAS form is 1.5 - 3x slower on contended rows.  Returns the previous default.
 */
PYG_HIP_API int pyg_hip_set_float_atomic_mode(int mode);
/* What the calling process's LAST launch of an atomically accumulating kernel was (thread-local text buffer): operator,
 * accumulator address / bytes / hipPointerGetAttributes, who cleared it and how, stream, flavour.  For failure reports. */
PYG_HIP_API const char* pyg_hip_last_accumulate_info(void);
/*
 * In-process health check of "clear an accumulator, add to it from all over the chip, read it back" on the CALLER'S memory
 * and stream: 5 add flavours (hw f32 / packed bf16 / f64, CAS f32, int32) x 3 ways of clearing (hipMemsetAsync, fill kernel
 * with plain stores, fill kernel with write-through stores) x 2 readbacks (copy kernel + D2H, D2H), `rounds` times each,
 * the accumulator scribbled with NaN patterns before every round.  `scratch`: >= 64 KiB of 16-byte aligned device memory
 * (up to 16 MiB are used).  Writes a text report (one line per failing variant, classified: NaN = the clear never
 * arrived, low = updates lost / clear late / stale read, high = doubled) and returns the number of failing variants
 * (0 = healthy) or a negative pyg_hip_status.  Synchronises `stream`.
 */
PYG_HIP_API int pyg_hip_atomic_selftest(void* scratch, size_t scratch_bytes, int rounds, char* report, size_t report_cap,
                                        void* stream);

/* ---- segment_matmul / grouped_matmul ------------------------------------------------------ */

/*
 * Per-call mode of pyg_hip_segment_matmul / pyg_hip_grouped_matmul (`flags` argument; 0 = defaults).  The mode is an
 * argument, not process state: calls in flight on different threads (autograd workers next to the main thread) never
 * see each other's choice -- the reference keeps its launch state in unlocked process globals
 * (ops/cuda/matmul_kernel.cu:19,118-119); this build does not.
 *
 * Bits 0-3: tile schedule / kernel family (PYG_HIP_MM_SCHED_*); every choice gives the same bits per output element
 * for the 16-bit types (same k order), they differ in speed only.
 *   AUTO        (default) the ticket schedule once every CU has several tiles to sweep, the item ring for many short
 *               relations (16-bit: fewer than 4096 rows per relation on average; fp32 split-bf16: fewer than 512),
 *               contiguous ranges below that;
 *   CONTIGUOUS  one contiguous tile range per workgroup (mfma_rows_lds_kernel): up to 6.1 TB/s on C2 when the
 *               allocator happened to place input and output favourably, 5.0 TB/s otherwise;
 *   CYCLIC      banded cyclic (mfma_rows_cyc_kernel): 6.2 - 6.3 TB/s on favourably placed buffers, 5.4 - 5.6 otherwise;
 *   TICKET      tiles drawn in address order from per-XCD counters, W in registers (mfma_rows_ticket_kernel):
 *               6.1 - 6.2 TB/s on either placement;
 *   GENERAL     (measurement only) every bf16 / f16 / f32 call through the general-shape MFMA kernel (matmul_gen.hip),
 *               also the shapes that have a specialised kernel;
 *   NAIVE       (measurement only) every call through the one-thread-per-output kernel;
 *   RING        the item-ring kernels (matmul_ring.hip): W slices in registers, X tiles and W chunks through one
 *               LDS-DMA ring: 5.0 - 5.4 TB/s at any segment length, where the ticket kernel falls to 3.0 at 256 rows.
 * 16-bit K = M = 256 has three kernels: AUTO / RING = W in registers + LDS-DMA item ring (mfma_rows_k256_regw_kernel),
 * CONTIGUOUS = W in LDS, 32 rows per wave (mfma_rows_wide256_kernel), CYCLIC / TICKET = W in LDS, 64 rows per wave
 * (mfma_rows_wide256r2_kernel).  The reference has no counterpart (its CUTLASS problem visitor is fixed,
 * ops/cuda/matmul_kernel.cu:121-287).
 *
 * Bit 8, PYG_HIP_MM_F32_SPLIT: arithmetic of the fp32 K = 128, M % 128 == 0 kernels.
 *   clear (default)  v_mfma_f32_32x32x2_f32 (mfma_rows_f32_pipe_kernel): IEEE fp32 products and sums, Inf / NaN / the
 *      whole fp32 range behave as in the reference's CPU kernel; bound by the fp32 matrix rate (157 TFLOP/s).
 *   set   split-bf16: every fp32 operand is split, round-to-nearest, into three bf16 terms (8 + 8 + 8 significant
 *      bits; the split is exact to 2^-27 relative) and the six leading cross products run on v_mfma_f32_32x32x16_bf16
 *      with fp32 accumulation.  Dropped terms: 2^-26 |x||w| per product at most, unbiased -- below the rounding unit
 *      of an fp32 multiply-add; 2.7x less matrix time: the kernel is HBM-bound.  Special values: a NaN operand gives
 *      NaN as it must; a +-Inf operand, or a finite one beyond the largest bf16 (|v| > 3.3895e38), gives NaN in every
 *      output it feeds (first term Inf, residual Inf - Inf) where the exact kernel gives +-Inf / a finite product;
 *      third terms of operands below ~2^-100 fall into the bf16 denormals the matrix unit flushes (those products keep
 *      16 instead of 24 bits).  This is the reduced-guarantee mode in the sense of the reference's TF32 switch: the
 *      torch binding sets the bit only when at::globalContext().float32MatmulPrecision() != HIGHEST, the rule of
 *      ops/cuda/matmul_kernel.cu:158-165 (torch's default is HIGHEST, i.e. the bit is clear unless the user called
 *      torch.set_float32_matmul_precision('high' | 'medium')).  Unlike TF32 the mode keeps full fp32 accuracy on
 *      finite data inside 2^-100 ... 2^127 (relative Frobenius error ~1e-7 against float64, as the exact kernel).
 * Any other bit set: PYG_HIP_ERR_INVALID.
 */
#define PYG_HIP_MM_SCHED_AUTO 0
#define PYG_HIP_MM_SCHED_CONTIGUOUS 1
#define PYG_HIP_MM_SCHED_CYCLIC 2
#define PYG_HIP_MM_SCHED_TICKET 3
#define PYG_HIP_MM_SCHED_GENERAL 4
#define PYG_HIP_MM_SCHED_NAIVE 5
#define PYG_HIP_MM_SCHED_RING 6
#define PYG_HIP_MM_SCHED_MASK 0xf
#define PYG_HIP_MM_F32_SPLIT 0x100

/* Workspace (device bytes) needed by pyg_hip_segment_matmul / pyg_hip_grouped_matmul for
 * `num_groups` segments/groups. */
PYG_HIP_API size_t pyg_hip_matmul_workspace_size(int64_t num_groups);

/*
 * out[ptr[b]:ptr[b+1]] = input[ptr[b]:ptr[b+1]] @ other[b]            for b in [0, B)
 * Replaces pyg::segment_matmul (schema pyg_lib/csrc/ops/matmul.cpp:66-67; CPU kernel
 * pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:410-439; CUDA kernel
 * pyg_lib/csrc/ops/cuda/matmul_kernel.cu:304-319).
 *   input  [N, K] row-major, other [B, K, M] row-major, out [N, M] row-major, all `dtype`.
 *   ptr    B+1 int64 boundaries; on device if ptr_on_device != 0, else on the host (the
 *          reference's preferred placement, pyg_lib/ops/__init__.py:160-161).  Unlike the
 *          reference no host synchronisation happens in either case.
 *   bias   optional [B, M] (may be NULL): fused epilogue for the Python-side loop
 *          pyg_lib/ops/__init__.py:169-171.
 *   flags  PYG_HIP_MM_* above (0 = automatic schedule, exact fp32).
 *   Rows outside [ptr[0], ptr[B]) are left untouched (the reference leaves them
 *   uninitialised, matmul_kernel.cpp:416).
 */
PYG_HIP_API int pyg_hip_segment_matmul(int dtype, const void* input, const int64_t* ptr,
                                       int ptr_on_device, const void* other, const void* bias,
                                       void* out, int64_t N, int64_t K, int64_t M, int64_t B,
                                       void* workspace, size_t workspace_bytes, int flags, void* stream);

/* One group of a grouped matmul: out[rows, m] = input[rows, k] @ other[k, m].
 * `other_trans` != 0 means `other` is stored [m, k] row-major (a transposed view, as produced
 * by the backward pass pyg_lib/ops/__init__.py:84,91), read in place. */
typedef struct {
  const void* input;
  const void* other;
  void* out;
  int64_t rows;
  int32_t k;
  int32_t m;
  int32_t other_trans;
  int32_t reserved;
} pyg_hip_group;

/*
 * outs[i] = inputs[i] @ others[i] for i in [0, G).
 * Replaces pyg::grouped_matmul (schema pyg_lib/csrc/ops/matmul.cpp:64-65; CPU kernel
 * pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:281-312; CUDA kernel
 * pyg_lib/csrc/ops/cuda/matmul_kernel.cu:289-302).  `groups_host` is a HOST array of G
 * descriptors (copied asynchronously into the workspace).
 */
PYG_HIP_API int pyg_hip_grouped_matmul(int dtype, const pyg_hip_group* groups_host, int64_t G,
                                       void* workspace, size_t workspace_bytes, int flags, void* stream);

/* Name of the kernel variant the last matmul call on this thread dispatched to
 * ("mfma_bf16_k128_m128", "naive", ...): lets tests assert that the MFMA path ran. */
PYG_HIP_API const char* pyg_hip_matmul_last_variant(void);

/*
 * Weight gradient of segment_matmul:  grad_other[b] = input[ptr[b]:ptr[b+1]]^T @ grad_out[ptr[b]:ptr[b+1]]
 * (input [N, K], grad_out [N, M], grad_other [B, K, M]; fp32 accumulation, one rounding).  Replaces the
 * per-relation loop of SegmentMatmul::backward (ops/autograd/matmul_kernel.cpp:92-107: B x
 * at::matmul(input_i^T, grad_out_i) + at::stack) with one persistent launch.  fp32 / bf16 / fp16, ANY K and M and any
 * element-aligned operands: K in {64, 128, 256} with M % 64 == 0 and 16-byte aligned operands run the shape-specialised
 * kernels of matmul_dw.hip, everything else the general-shape kernel (matmul_dw_gen.hip: 128 x 128 output blocks, tails
 * zero-filled in LDS, widest vector loads the alignment allows).  fp32 multiplies on v_mfma_f32_32x32x2_f32 (IEEE fp32).
 * Other dtypes return PYG_HIP_ERR_UNSUPPORTED (the caller keeps the reference formula).  `workspace`:
 * pyg_hip_segment_matmul_dw_workspace_size(B, K, M) bytes of device scratch (tile plan + fp32 accumulators).  Never
 * synchronises.
 */
PYG_HIP_API size_t pyg_hip_segment_matmul_dw_workspace_size(int64_t B, int64_t K, int64_t M);
/* Grouped form (the others_grad of GroupedMatmul.backward, pyg_lib/ops/__init__.py:88-94): for every
 * group i, out_i = input_i^T @ other_i with input_i [rows_i, k_i] and other_i [rows_i, m_i] row-major, PER-GROUP k_i,
 * m_i (`out` / `other_trans` of pyg_hip_group are ignored); out_pool receives the [k_i, m_i] results back to back in
 * group order (out_i starts at element sum_{j < i} k_j m_j; uniform shapes: one [G, k, m] block).  Workspace:
 * pyg_hip_grouped_matmul_dw_workspace_size(groups, G). */
PYG_HIP_API size_t pyg_hip_grouped_matmul_dw_workspace_size(const pyg_hip_group* groups, int64_t G);
/* Diagnostic: calls served by the shape-specialised / the general-shape weight-gradient kernels since the library was
 * loaded (process wide).  Either pointer may be NULL. */
PYG_HIP_API void pyg_hip_matmul_dw_counters(int64_t* specialised, int64_t* general);
PYG_HIP_API int pyg_hip_grouped_matmul_dw(int dtype, const pyg_hip_group* groups, int64_t G, void* out_pool,
                                          void* workspace, size_t workspace_bytes, void* stream);
PYG_HIP_API int pyg_hip_segment_matmul_dw(int dtype, const void* input, const int64_t* ptr, int ptr_on_device,
                                          const void* grad_out, void* grad_other, int64_t N, int64_t K,
                                          int64_t M, int64_t B, void* workspace, size_t workspace_bytes,
                                          void* stream);

/* ---- fused relational graph convolution (BASELINE config C5) -------------- */

/* One relation of pyg_hip_rgcn_fused: its sampled edges e in [0, num_edges) read row
 * gather_index[e] + gather_offset of x and add their message to row scatter_index[e] + scatter_offset of out.
 * The index vectors are the per-relation `col` / `row` outputs of hetero_neighbor_sample, used in place. */
typedef struct {
  const int64_t* gather_index;  /* device */
  const int64_t* scatter_index; /* device; runs of equal values are summed before they touch memory */
  int64_t num_edges;
  int64_t gather_offset;
  int64_t scatter_offset;
  const void* weight;           /* device, [K, M] row-major, 16-byte aligned */
  /* Optional second indirection ("true fusion": the per-batch feature matrix x is never materialised).  With `x`
   * non-NULL this relation gathers from its OWN table x [x_rows, K] (e.g. the global feature table of its source
   * node type) at row gather_map[gather_index[e]] (gather_map = that type's sampled node ids, gather_map_len
   * entries), or gather_index[e] + gather_offset if gather_map is NULL.  All zero: the call's x, as before. */
  const void* x;
  const int64_t* gather_map;
  int64_t x_rows;
  int64_t gather_map_len;
  /* Rows of `out` this relation may write: [scatter_offset, scatter_offset + scatter_rows) -- the row count of its
   * destination node type (the `dim_size` of the reference's reductions, pyg_lib/csrc/ops/scatter.cpp:156-160).  0: up to
   * num_out_rows.  PYG_HIP_RGCN_GROUPED validates scatter_index against it (error 2) and sizes its row-start workspace by
   * it (sum of scatter_rows over the relations instead of R x num_out_rows); the atomic kernel validates against
   * num_out_rows only. */
  int64_t scatter_rows;
} pyg_hip_rgcn_relation;

PYG_HIP_API size_t pyg_hip_rgcn_fused_workspace_size(int64_t num_relations, int64_t num_edges);

/*
 * out[scatter_index_r[e] + scatter_offset_r] += x[gather_index_r[e] + gather_offset_r] @ weight_r  for all r, e.
 * One launch replacing the chain gather_coo -> segment_matmul -> scatter_sum of the reference ops
 * (pyg_lib/csrc/ops/cuda/segment_coo_kernel.cu:1316-1360, ops/cuda/matmul_kernel.cu:304-319,
 * ops/cuda/scatter_kernel.cu:56-71): the gathered rows and the messages never exist in HBM.
 *   x [num_x_rows, K], out [num_out_rows, M] (ACCUMULATED into: zero it for a plain aggregation), both `dtype`
 *   (PYG_BF16 / PYG_F16) row-major; K = M = 128 (other shapes: PYG_HIP_ERR_UNSUPPORTED, the caller keeps the
 *   three-op chain).  Messages are rounded to `dtype` once (as the chain does), runs of equal destination are
 *   summed in fp32 and added with packed 16-bit atomics.  `relations` is a host array.  `checked` is a bit field
 *   (PYG_HIP_RGCN_*): PYG_HIP_RGCN_CAS makes the packed adds compare-and-swap loops (see pyg_hip_set_float_atomic_mode).
 *   Never synchronises -- unless PYG_HIP_RGCN_CHECKED is set: then every gather / scatter index is validated against num_x_rows (x_rows, gather_map_len)
 *   / num_out_rows on the device, offenders are redirected to row 0, and the call waits for the stream and returns
 *   PYG_HIP_ERR_INVALID if there was one (unchecked, a bad index is an out-of-bounds read / an atomic into foreign memory).
 */
#define PYG_HIP_RGCN_CHECKED 1
#define PYG_HIP_RGCN_CAS 2
#define PYG_HIP_RGCN_DEFERRED 4
/* PYG_HIP_RGCN_DEFERRED (ignored with _CHECKED): the same validation WITHOUT the synchronisation -- offenders are
 * redirected to row 0, so a stale node id is never an out-of-bounds access, and the kernel leaves 1 (gather) / 2
 * (scatter) / 3 (not grouped, PYG_HIP_RGCN_GROUPED) in a pinned word of the device.  The next pyg_hip_rgcn_fused call with this flag on that device fails with
 * PYG_HIP_ERR_INVALID naming the earlier call; pyg_hip_rgcn_pending_error() returns and clears the word (meaningful once
 * the stream has been synchronised).  This is what the torch binding passes by default (PYG_HIP_RGCN_CHECK=1: _CHECKED,
 * =0: no validation). */
PYG_HIP_API int pyg_hip_rgcn_pending_error(void);
/* PYG_HIP_RGCN_GROUPED: the caller promises that every relation's scatter_index is NONDECREASING (edges grouped by
 * destination -- what the neighbour samplers emit: `row` of every edge type, csc = false).  Then no atomics are needed:
 * a small launch finds every destination's first edge, and an owner-computes kernel (32 rows of `out` per workgroup)
 * sums every row's source features in fp32 in edge order, multiplies the 32 sums of a relation with its weight in one
 * MFMA tile, accumulates the relations of a row in fp32 and WRITES every row of `out` once (rows without edges: zeros).
 *   - `out` is OVERWRITTEN, not accumulated into (do not zero it); it must be 16-byte aligned;
 *   - K and M may be any multiples of 8 up to 256 (the feature rows are walked per 128-feature slice, W travels through
 *     LDS in 128 x 128 chunks; K, M in {128, 256} have pipelined instances, the others one instance with run-time row sizes
 *     and masked lanes); dtype may also be PYG_F32 with K, M multiples of 4 up to 128 (fp32 sums; 128 x 128: fp32
 *     MFMAs, otherwise FMAs);
 *   - the same bits on every run; rounding: the per-relation feature sum and the result are each rounded once;
 *   - the workspace is pyg_hip_rgcn_grouped_workspace_size() bytes (4 bytes per row of every relation's destination
 *     segment -- scatter_rows, or the rows of `out` at and behind its scatter_offset if that is 0: row starts, touched
 *     only where edges arrive);
 *   - fewer than 2^31 rows of `out` and edges per relation (PYG_HIP_ERR_UNSUPPORTED otherwise);
 *   - with _CHECKED / _DEFERRED the promise is verified on the device: a descent in a scatter_index is error 3
 *     (PYG_HIP_ERR_INVALID "not grouped"; the result of such a call is unspecified but every access stays in bounds),
 *     an out-of-range scatter index drops its edge (error 2), a gather index is redirected to row 0 (error 1). */
#define PYG_HIP_RGCN_GROUPED 8
PYG_HIP_API size_t pyg_hip_rgcn_grouped_workspace_size(const pyg_hip_rgcn_relation* relations, int64_t num_relations,
                                                       int64_t num_out_rows);
PYG_HIP_API int pyg_hip_rgcn_fused(int dtype, const void* x, int64_t num_x_rows, const pyg_hip_rgcn_relation* relations,
                                   int64_t num_relations, void* out, int64_t num_out_rows, int64_t K, int64_t M,
                                   int checked, void* workspace, size_t workspace_bytes, void* stream);

/* ---- neighbor_sample / hetero_neighbor_sample ---------------------------------------------- */

/* Host services the sampler needs from its caller (the torch binding supplies the PyTorch
 * caching allocator and the global CPU generator, so device memory and torch.manual_seed()
 * behave exactly as for the reference operator). */
/* State of an at::mt19937 engine (ATen/core/MT19937RNGEngine.h: state_[624], left_, next_). */
typedef struct {
  uint32_t state[624];
  int32_t left;
  uint32_t next;
} pyg_hip_mt19937;

typedef struct {
  void* user;
  /* Device allocation on the stream the sampler runs on; NULL on failure. */
  void* (*alloc)(void* user, size_t bytes);
  /* Release a block obtained from `alloc` (stream-ordered with the sampler's stream). */
  void (*free)(void* user, void* ptr);
  /* Fill num_blocks x 128 host int64 words exactly like that many consecutive RandintEngine
   * prefetches (pyg_lib/csrc/random/cpu/rand_engine.h:79-91): the first prefetch of a call is
   * at::randint(INT64_MIN, INT64_MAX, {128}) (first != 0), later ones are in-place
   * random_(INT64_MIN, INT64_MAX) refills.  Both draw the same serial mt19937 stream, so one
   * random_ over num_blocks*128 elements is equivalent. */
  void (*rng_blocks)(void* user, int64_t* words_host, int64_t num_blocks, int first);
  /* Optional fast path: the state of the CPU generator's mt19937 engine (in/out, host memory).  If
   * non-NULL the words are generated ON THE DEVICE by continuing this engine exactly as random_
   * would (two 32-bit outputs per word, high half first, % (2^64-1) + INT64_MIN), rng_blocks is not
   * called, and the advanced state is written back before the call returns. */
  pyg_hip_mt19937* mt19937;
} pyg_hip_sampler_host;

/* One CSR relation of a (heterogeneous) graph.  src_type / dst_type index `node_types` in the
 * order the reference's `edge_types` tuples name them (roles swap when csc, neighbor_kernel.cpp
 * :715-716). */
typedef struct {
  const int64_t* rowptr; /* device, num_rows + 1 */
  int64_t num_rows;
  const int64_t* col;    /* device */
  int64_t num_cols;
  int32_t src_type;
  int32_t dst_type;
  const int64_t* num_neighbors_host; /* L fan-outs, host */
  const int64_t* edge_time;          /* device, per edge, or NULL (edge-level temporal sampling) */
  /* Biased sampling (neighbor_kernel.cpp:39-56,245-285; hetero :732-745): per-edge weights on the device, or
   * NULL.  edge_weight_dtype is PYG_F32 or PYG_F64 (the dtype decides how many generator outputs a draw
   * takes, see pyg_hip_hetero_neighbor_sample).  Needs host->mt19937. */
  const void* edge_weight;
  int32_t edge_weight_dtype;
  /* != 0: rowptr and col point at int32 arrays (declared int64_t* for source compatibility) and are read in
   * place; the reference's int32 instantiation (neighbor_kernel.cpp:893,930).  Seeds, times and all outputs stay
   * int64 on this interface (they are a few thousand to a million elements; the binding converts them). */
  int32_t index_is32;
} pyg_hip_relation;

/* Seeds of one node type, in seed_dict iteration order. */
typedef struct {
  int32_t node_type;
  int32_t reserved;
  const int64_t* seed; /* device */
  int64_t num_seed;
  const int64_t* seed_time; /* device, per seed, or NULL (then node_time[seed] is used) */
} pyg_hip_seed_set;

/* Results.  All pointers come from host->alloc and are owned by the caller afterwards (blocks
 * may be larger than the element counts given here).
 *   per node type t:  node_id[t] -> num_nodes[t] ids ([n] int64, or [n, 2] (batch, node) pairs
 *                     when disjoint), nodes_per_hop_host[t*(L+1) ..]
 *   per relation e:   row[e], col[e], edge_id[e] (NULL unless return_edge_id) -> num_edges[e],
 *                     edges_per_hop_host[e*L ..]
 * The `*_host` arrays and the pointer arrays are caller-provided host memory. */
typedef struct {
  int64_t** node_id;
  int64_t* num_nodes;
  int64_t* nodes_per_hop_host;
  int64_t** row;
  int64_t** col;
  int64_t** edge_id;
  int64_t* num_edges;
  int64_t* edges_per_hop_host;
  int64_t rng_blocks; /* 128-word prefetches consumed (incl. the constructor's) */
} pyg_hip_sample_result;

/*
 * Multi-hop neighbour sampling with first-occurrence-ordered node relabelling, bit-exact with
 * the reference's single-threaded CPU kernel (pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp
 * :518-841; homogeneous :332-514 is the 1-type / 1-relation case; schemas
 * pyg_lib/csrc/sampler/neighbor.cpp:130-147).  The reference has no device sampler at all.
 *
 * flags: csc, replace, disjoint, return_edge_id as in the schema.  `directed` must be true
 * (the reference errors otherwise, neighbor_kernel.cpp:501).
 * Temporal sampling (node_temporal_sample / edge_temporal_sample, neighbor_kernel.cpp:74-144;
 * requires disjoint): node_time_by_type is a host array of num_node_types device pointers (entries
 * or the array itself may be NULL), relation.edge_time takes precedence; temporal_last selects the
 * "last" strategy.  A neighbourhood that is not time-sorted fails with the reference's message.
 * Biased sampling (relation.edge_weight; _biased_sample, neighbor_kernel.cpp:245-285): a row with more
 * neighbours than the fan-out draws one uniform number per neighbour STRAIGHT from the generator (one
 * 32-bit output per float32 weight, 24 bits kept; two per float64 weight, 53 bits kept -- Tensor.uniform_),
 * key = log(u) / weight, and takes the `count` largest keys in Tensor.topk order (ties as libstdc++'s
 * partial_sort / nth_element + sort leave them, ATen/native/TopKImpl.h).  The device path reproduces that
 * stream and that order; `log` is the correctly rounded logarithm where libtorch calls MKL's (<1 ulp, closed
 * source: 15,372 of the 2^24 possible float32 arguments round differently), so a selection can differ from
 * the reference's only if two keys of one row lie within one ulp of each other.  Weighted and unweighted
 * relations may be mixed (the engine's later 128-word blocks then lie behind the weighted relations' draws in
 * the generator stream, as in the reference).  With replace != 0 the reference calls
 * at::multinomial(weight, count, true): for count > 1 that is a sequential cumulative sum in the weights' type,
 * a division by the sum, and one 53-bit double per sample located by binary search -- reproduced exactly
 * (PYG_HIP_ERR_INVALID "invalid multinomial distribution" for rows at::multinomial rejects); for count == 1
 * at::multinomial takes argmax(weight / exponential_()) instead, where libtorch 2.10 draws one 53-bit double
 * per neighbour and evaluates -log1p(-u) -- reproduced too (first of equal maxima).  Temporal arguments and a
 * missing host->mt19937 fail with PYG_HIP_ERR_UNSUPPORTED.
 * Synchronises `stream` (output sizes are data dependent).
 */
/* Driver the calling thread's last neighbor / hetero sampler call ran: "fused" (bounded fan-outs <= 1024: 2 - 3 launches
 * per hop -- scans of up to 256 tiles are one launch, PYG_HIP_SAMPLER_ONEPASS=0: always a reduce + apply pair --,
 * csrc/hip/sampler_fused.h; since round 5 also with rows of degree >= 2^16, whose 32-bit draws the chain's transition
 * tables carry, and with fan-outs above 64, sampled one wave per node), "queued" (round 2's chain:
 * PYG_HIP_SAMPLER_FUSED=0 or more than 3 relations expanding one node type; fan-outs <= 64), "synchronising"
 * (unbounded / > 1024 fan-outs, weighted relations, more wide draws than the speculated random words allow, or
 * PYG_HIP_SAMPLER_SYNC_MODE=1).  Diagnostics only; every driver returns the same bits. */
PYG_HIP_API const char* pyg_hip_sampler_last_mode(void);
/* Direct-address node tables of the fused chain are kept between calls (per device and node count; blocks come from
 * host->alloc and are never freed): a call's values carry an epoch in their upper bits, so what an earlier call left
 * behind reads as "empty" and the 19.6 MB clear of a products-sized table (one launch, 7 % of a batch's traffic; four
 * launches for the MAG-shaped C5 graph) happens once per 2^20 - 1 calls instead of every call.  Diagnostics / tests:
 * `limit` = epochs per clear (0: the default 2^20 - 1); returns the number of tables cached for the current device.
 * PYG_HIP_SAMPLER_TABLE_CACHE=0 disables the cache (every call clears a fresh table, as before round 4). */
PYG_HIP_API int pyg_hip_sampler_table_cache(int64_t limit);
/* Frees the cached tables of the current device that no call is using, through host->free (the allocator they came
 * from); returns how many stay (busy ones).  The cache holds at most 8 tables of <= 128 MiB per device; an idle table
 * of another node count is evicted when a new size needs room. */
PYG_HIP_API int pyg_hip_sampler_table_cache_release(const pyg_hip_sampler_host* host);
/*
 * The random-word stream kept between calls.  A data loader samples batch after batch on ONE generator
 * (benchmark/sampler/neighbor.py:101-121 seeds once per run): the engine a call hands back through host->mt19937 is the
 * engine the next call presents, and the words the library generated beyond a call's own consumption are exactly the next
 * call's words.  The fused chain keeps them per device (buffer of <= 64 MiB from host->alloc, the handed-back engine) and
 * adopts them when the presented engine equals the kept one bit for bit: then no generation launch and no cross-stream
 * wait lies in front of any hop, and the next round is generated in the background, two calls' worth ahead.  Any other
 * engine (reseeded, used elsewhere in between) misses and the call starts cold -- same bits either way, the generator
 * ends where the reference's engine leaves it.  PYG_HIP_SAMPLER_RNG_CARRY=0 disables it;
 * pyg_hip_sampler_table_cache_release also frees the idle stream.  Counters since process start: calls that adopted a
 * kept stream / calls that looked for one and started cold.
 */
PYG_HIP_API int pyg_hip_sampler_rng_carry_stats(int64_t* adopted, int64_t* cold);

PYG_HIP_API int pyg_hip_hetero_neighbor_sample(int num_node_types, int num_relations,
                                               const pyg_hip_relation* relations_host,
                                               int num_seed_sets,
                                               const pyg_hip_seed_set* seeds_host,
                                               const int64_t* const* node_time_by_type,
                                               int temporal_last, int L, int csc, int replace,
                                               int disjoint, int return_edge_id,
                                               const pyg_hip_sampler_host* host,
                                               pyg_hip_sample_result* result, void* stream);

/*
 * K independent sampler calls on one graph at once (the epoch loop of the reference's benchmark,
 * benchmark/sampler/neighbor.py:101-121, handed over as a whole; semantics per batch: sampler/cpu/neighbor_kernel.cpp
 * :332-514 / :518-841).  Batch b = its own seed sets, its own `host` (allocator context + generator state: every batch
 * continues ITS OWN mt19937 stream -- torch.manual_seed(s_b) per batch is the reference benchmark's protocol) and its own
 * result; everything else is shared.  Results are bit for bit those of pyg_hip_hetero_neighbor_sample on each batch alone.
 * Batches that name different `stream`s are driven by different host threads of a persistent pool inside the library and
 * overlap on the device (a single batch is a chain of ~12 small dependent launches that cannot fill 256 CUs); batches on the
 * same stream run one after the other.  host->alloc of a batch must allocate for THAT batch's stream.  `stream` = the
 * caller's stream: work queued on it before the call is ordered in front of every batch; all batches are complete (their
 * streams synchronised) when the call returns.  Returns the first failing batch's status; `status` / `error` / `mode`
 * (pyg_hip_sampler_last_mode of that batch) are filled per batch.
 */
typedef struct {
  int num_seed_sets;
  const pyg_hip_seed_set* seeds_host;
  const pyg_hip_sampler_host* host;
  pyg_hip_sample_result* result;
  void* stream;
  int status;         /* out */
  const char* mode;   /* out */
  char error[256];    /* out */
} pyg_hip_sample_batch;
PYG_HIP_API int pyg_hip_hetero_neighbor_sample_batched(int num_node_types, int num_relations,
                                                       const pyg_hip_relation* relations_host,
                                                       const int64_t* const* node_time_by_type, int temporal_last,
                                                       int L, int csc, int replace, int disjoint, int return_edge_id,
                                                       int num_batches, pyg_hip_sample_batch* batches, void* stream);

/*
 * The float32 logarithm biased sampling evaluates (key = log(u) / weight), element-wise over device arrays.
 * Exposed so that it can be pinned on every argument Tensor.uniform_ can produce (k * 2^-24).
 */
PYG_HIP_API int pyg_hip_biased_log_f32(const float* in, float* out, int64_t n, void* stream);

/*
 * One-hop sampling WITHOUT relabelling for PyG's distributed sampler.
 * Replaces pyg::dist_neighbor_sample (schema pyg_lib/csrc/sampler/neighbor.cpp:148-153; CPU kernel
 * sampler/cpu/neighbor_kernel.cpp:957-978, `distributed` flag :296-303,386-388,446-447).
 *   node_id   -> S + E ids (seeds first, then every sampled destination, duplicates kept;
 *                [S+E, 2] (batch, node) pairs when disjoint), from host->alloc
 *   edge_id   -> E sampled edge ids, from host->alloc
 *   cumsum_host  caller array of S + 1 entries: [S, size after seed 0, size after seed 1, ...]
 * Random words, temporal arguments, biased sampling (edge_weight: device pointer or NULL, edge_weight_dtype
 * PYG_F32 / PYG_F64) and error behaviour as in pyg_hip_hetero_neighbor_sample.  index_is32 != 0: rowptr and col point at
 * int32 arrays and are read in place (the reference's int32 instantiation, neighbor_kernel.cpp:893); seeds, times and the
 * outputs stay int64 on this interface.
 */
PYG_HIP_API int pyg_hip_dist_neighbor_sample(const int64_t* rowptr, const int64_t* col,
                                             const int64_t* seed, int64_t num_seed,
                                             int64_t num_neighbors, const int64_t* node_time,
                                             const int64_t* edge_time, const int64_t* seed_time,
                                             const void* edge_weight, int edge_weight_dtype,
                                             int temporal_last, int replace, int disjoint, int index_is32,
                                             const pyg_hip_sampler_host* host, int64_t** node_id,
                                             int64_t** edge_id, int64_t* num_edges,
                                             int64_t* cumsum_host, void* stream);

/*
 * Building blocks of pyg::hetero_relabel_neighborhood (schema sampler/dist_relabel.cpp:77-83; CPU
 * sampler/cpu/dist_relabel_kernel.cpp:96-262): there every node type has ONE Mapper and consumes its
 * `sampled_nodes_with_duplicates` list strictly in order, so the local id of list position j is a per-type
 * quantity (pyg_hip_relabel_nodes), and an edge type's rows / cols are segments of source indices
 * (pyg_hip_expand_rows) / of the destination type's local ids.
 *   pyg_hip_relabel_nodes: local_out[j] = Mapper id of sampled[j] after the seeds (ids of first occurrences;
 *     disjoint: keys (batch, node), the seeds carry batch ids seed_batch0, seed_batch0 + 1, ..., all batch ids
 *     < num_batches).  Workspace: pyg_hip_relabel_workspace_size(num_seed, num_sampled).
 *   pyg_hip_expand_rows: row_out[j] = i for count_prefix[i] <= j < count_prefix[i + 1] (device prefix of
 *     num_src + 1 entries), j < total.
 */
PYG_HIP_API int pyg_hip_relabel_nodes(const int64_t* seed, int64_t num_seed, int64_t seed_batch0,
                                      int64_t num_batches, const int64_t* sampled, int64_t num_sampled,
                                      const int64_t* batch, int disjoint, int64_t* local_out,
                                      void* workspace, size_t workspace_bytes, void* stream);
PYG_HIP_API int pyg_hip_expand_rows(const int64_t* count_prefix, int64_t num_src, int64_t total,
                                    int64_t* row_out, void* stream);

/*
 * Distributed-sampling helpers (homogeneous forms).
 *
 * pyg_hip_relabel_neighborhood replaces pyg::relabel_neighborhood (schema sampler/dist_relabel.cpp:71-76; CPU
 * sampler/cpu/dist_relabel_kernel.cpp:30-94): local ids in insertion order -- the seeds first (a duplicate
 * seed keeps the id of its first occurrence; disjoint: key (i, seed[i])), then the externally sampled
 * sequence `sampled` (disjoint: key (batch[j], sampled[j])).  col_out[j] = id of sampled[j]; row_out[j] = the
 * source node i with count_prefix[i] <= j < count_prefix[i + 1] (count_prefix: device array of num_src + 1
 * offsets, the running sum of num_sampled_neighbors_per_node).  The csc swap is the caller's.
 * `workspace`: pyg_hip_relabel_workspace_size(num_seed, num_sampled) bytes.
 *
 * pyg_hip_segment_concat is the device part of pyg::merge_sampler_outputs (schema
 * sampler/dist_merge_outputs.cpp:51-55; CPU sampler/cpu/dist_merge_outputs_kernel.cpp:17-138): out = the
 * segments bases[part[j]][begin[j] ...) of length dst_off[j+1] - dst_off[j], concatenated in j order; with
 * `fill`, out[i] = fill[j] for the positions of segment j instead (the batch vector).  All arrays are device
 * arrays; `bases` is a device array of device pointers.
 */
PYG_HIP_API size_t pyg_hip_relabel_workspace_size(int64_t num_seed, int64_t num_sampled);
PYG_HIP_API int pyg_hip_relabel_neighborhood(const int64_t* seed, int64_t num_seed, const int64_t* sampled,
                                             int64_t num_sampled, const int64_t* count_prefix, int64_t num_src,
                                             const int64_t* batch, int disjoint, int64_t* row_out, int64_t* col_out,
                                             void* workspace, size_t workspace_bytes, void* stream);
PYG_HIP_API int pyg_hip_segment_concat(const int64_t* const* bases, const int64_t* part, const int64_t* begin,
                                       const int64_t* dst_off, int64_t n, const int64_t* fill, int64_t* out,
                                       int64_t total, void* stream);

/* ---- index_sort ---------------------------------------------------------------------------- */

PYG_HIP_API size_t pyg_hip_index_sort_workspace_size(int dtype, int64_t n);

/*
 * Ascending STABLE sort of `n` integer keys: keys_out = sorted keys (same dtype), index_out =
 * int64 permutation, bit-identical to torch.sort(stable=True).
 * Replaces pyg::index_sort (schema pyg_lib/csrc/ops/index_sort.cpp:25-28; CPU kernel
 * pyg_lib/csrc/ops/cpu/index_sort_kernel.cpp:14-59 + ops/cpu/radix_sort.h:58-198; the reference
 * has no device kernel, pyg_lib/ops/__init__.py:319-320 falls back to torch.sort).
 *   dtype      PYG_U8 / PYG_I8 / PYG_I16 / PYG_I32 / PYG_I64 (anything else: "Input should contain
 *              integral values.", index_sort_kernel.cpp:55-56)
 *   max_value  with has_max != 0: an upper bound of the (non-negative) keys, only sets the number of
 *              8-bit passes (radix_sort.h:170-176).  With has_max == 0 the extremes are reduced on
 *              the device and read back (one stream synchronisation, like the reference's
 *              input.max().item()); negative keys are then sorted correctly as well.
 */
PYG_HIP_API int pyg_hip_index_sort(int dtype, const void* keys, int64_t n, int64_t max_value,
                                   int has_max, void* keys_out, int64_t* index_out, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---- scatter / segment_coo / gather_coo ---------------------------------------------------- */

typedef enum {
  PYG_REDUCE_SUM = 0,
  PYG_REDUCE_MUL = 1,
  PYG_REDUCE_MIN = 2,
  PYG_REDUCE_MAX = 3
} pyg_reduce;

/*
 * out[b, index(b, e, k), k]  (op)=  src[b, e, k]      over the (B, E, K) view of `src`
 * (layout of pyg_lib/csrc/ops/cpu/scatter_kernel.cpp:16-24; `out` is [B, N, K]).
 * Replaces pyg::scatter_{sum,mul,min,max} (schemas pyg_lib/csrc/ops/scatter.cpp:156-172; CPU
 * kernels ops/cpu/scatter_kernel.cpp:29-511; CUDA ops/cuda/scatter_kernel.cu:56-651) and, with a
 * sorted [B, E] index, pyg::segment_{sum,min,max}_coo (schemas ops/segment_coo.cpp:150-165; CPU
 * ops/cpu/segment_coo_kernel.cpp:31-651; CUDA ops/cuda/segment_coo_kernel.cu:73-1301).
 *   index     int64, element (b, e, k) at index[b*stride_b + e*stride_e + k*stride_k]; a stride of
 *             0 broadcasts (1-D index: (0, 1, 0); COO index [B, E]: (E, 1, 0); fully expanded:
 *             (E*K, K, 1)).
 *   out       running state, updated in place (`out=` contract, ops/scatter.h:11-113): the caller
 *             pre-fills it (zeros / ones / pyg_hip_fill_reduce_identity for fresh outputs).
 *   MIN/MAX   arg_out [B, N, K] int64 is filled here: source position of the first element that
 *             produced the final value, or the sentinel E.  out_init = NULL means `out` was filled
 *             with the reduce identity: empty buckets are then reset to 0
 *             (scatter_kernel.cpp:351-360).  Otherwise out_init points to a copy of the caller's
 *             initial `out` and untouched buckets keep their value.
 *   index_sorted  bit 0 (PYG_HIP_SCATTER_SORTED) promises an ascending index along e (the COO contract): runs are
 *             then reduced without atomics.  Bit 1 (PYG_HIP_SCATTER_FRESH_SUM, SUM only): `out` is a fresh,
 *             UNINITIALISED output -- the sorted path then writes every slot without reading or clearing it (2 x N x K
 *             bytes less traffic), every other path clears it first.
 *   workspace optional scratch of pyg_hip_scatter_workspace_size(B, E, N) bytes; with it, sums and min / max with an
 *             index broadcast along k run atomic-free: buckets become CSR rows (directly for a sorted index, after
 *             a stable index sort for one large unsorted index vector with rows of >= 64 bytes) reduced in SOURCE
 *             order -- deterministic, fp32 accumulation with one rounding per output, every output row written
 *             once; min / max: no CAS loops, no second arg pass, same exact values and first-match arg.  Without it
 *             (and for small or element-wise indexed inputs, float64 sums) the atomic kernels run.
 *             Bit 2 (PYG_HIP_SCATTER_CAS): the atomic kernels add floats / doubles / packed 16-bit pairs through
 *             compare-and-swap loops instead of the hardware's floating-point atomic adds.
 *             Bit 3 (PYG_HIP_SCATTER_DETERMINISTIC, floating SUM): no float atomics at all -- an unsorted index vector
 *             broadcast along k (one vector, B == 1) takes the stable-sort + CSR-row path WHATEVER its size, row width
 *             or floating type (needs the workspace): sums in source order, the same bits in every run, like the
 *             reference's sequential CPU loop (ops/cpu/scatter_kernel.cpp:29-127).  Layouts without an atomic-free
 *             kernel (element-wise indices, B > 1 unsorted) and floating MUL return PYG_HIP_ERR_UNSUPPORTED with the
 *             bit set.  The torch binding sets it when torch.are_deterministic_algorithms_enabled().
 */
#define PYG_HIP_SCATTER_SORTED 1
#define PYG_HIP_SCATTER_FRESH_SUM 2
#define PYG_HIP_SCATTER_CAS 4
#define PYG_HIP_SCATTER_DETERMINISTIC 8
PYG_HIP_API size_t pyg_hip_scatter_workspace_size(int64_t B, int64_t E, int64_t N);
PYG_HIP_API int pyg_hip_scatter(int op, int dtype, const void* src, const int64_t* index,
                                int64_t index_stride_b, int64_t index_stride_e,
                                int64_t index_stride_k, void* out, int64_t* arg_out,
                                const void* out_init, int64_t B, int64_t E, int64_t K, int64_t N,
                                int index_sorted, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Fill `n` elements with numeric_limits<T>::max() (MIN) / lowest() (MAX): the start state of a
 * fresh min/max output (scatter_kernel.cpp:296-300). */
PYG_HIP_API int pyg_hip_fill_reduce_identity(int op, int dtype, void* out, int64_t n, void* stream);

/*
 * out[b, e, k] = src[b, index[b, e], k]     src [B, N, K], index [B, E] contiguous, out [B, E, K].
 * Replaces pyg::gather_coo (schema ops/segment_coo.cpp:164-165; CPU
 * ops/cpu/segment_coo_kernel.cpp:666-746; CUDA ops/cuda/segment_coo_kernel.cu:1316-1444).
 */
PYG_HIP_API int pyg_hip_gather_coo(int dtype, const void* src, const int64_t* index, void* out,
                                   int64_t B, int64_t E, int64_t K, int64_t N, void* stream);

/* ---- measurement hooks (bench.py) --------------------------------------------------------- */

/*
 * CSR reductions.  src viewed as [leading, E, K]; indptr holds rows + 1 ascending offsets per slice,
 * `indptr_slice_stride` elements apart (0: one indptr shared by all slices, read in place);
 * out / arg_out [leading, rows, K].  Replaces pyg::segment_{sum,mean,min,max}_csr (schemas
 * ops/segment_csr.cpp:153-172; CPU ops/cpu/segment_csr_kernel.cpp:32-536; CUDA
 * ops/cuda/segment_csr_kernel.cu).
 *   op 0 sum : every row adds src[slice, indptr[r] .. indptr[r+1]) to the CURRENT contents of its out
 *              slot (the caller zero-fills a fresh output; a caller-supplied `out` accumulates)
 *   op 1 mean: out = row sum / max(row length, 1), previous contents ignored; floating dtypes only
 *   op 2 min / 3 max: strict < / >, first match; fresh == 0: the running state starts from the current contents
 *              of `out`; fresh != 0: from numeric_limits max() / lowest() WITHOUT reading `out` (it need not be
 *              pre-filled; ABI <= 7 read it and wanted pyg_hip_fill_reduce_identity first -- still harmless), and rows
 *              without a contribution are reset to 0.  arg_out receives the winning source position or the
 *              sentinel E for EVERY slot (it need not be pre-filled either).
 * Rows are reduced in source order in the reference's opmath, so results are bit-identical to the CPU
 * kernel for every dtype unless rows are long and few (then lanes split a row) or longer than 512
 * positions per lane (4096 for rows narrower than 64 bytes: hub rows, see pyg_hip_segment_csr_ws); floating
 * sums of such rows differ by rounding, min/max/arg stay exact, and every run gives the same bits.
 */
PYG_HIP_API int pyg_hip_segment_csr(int op, int dtype, const void* src, const int64_t* indptr,
                                    int64_t indptr_slice_stride, void* out, int64_t* arg_out, int fresh,
                                    int64_t leading, int64_t rows, int64_t E, int64_t K, void* stream);

/*
 * The same with scratch for HUB rows.  The row kernels give a row to 1 / 8 / 64 lanes; a row of more than 512 positions per
 * lane (a power-law graph's popular destination) is skipped there.  Without scratch a second launch gives every such row to
 * ONE workgroup (~9 GB/s per row: a row holding 2.5 % of 8 M positions of 256 bytes then takes 6 ms of a 0.5 ms call).
 * With `workspace` (pyg_hip_csr_hub_workspace_size() bytes of device memory, contents irrelevant, not kept) the skipped rows
 * are registered there in chunks of 2048 positions, a second launch deals the chunks to all workgroups, and a third
 * combines the chunks' partial results of a row in chunk order: no float atomics, the same bits on
 * every run; sums of hub rows differ from the sequential order by rounding (like the lane-split rows), min / max / arg
 * stay exact.  A smaller workspace is legal: the chunk length doubles until the partial results fit, too small means "without".
 * pyg_hip_csr_hub_workspace_size: op 0 ... 3 as above, 4 = gather_csr; 0 when no row can be a hub (leading * E <= 512).
 */
PYG_HIP_API size_t pyg_hip_csr_hub_workspace_size(int op, int dtype, int64_t leading, int64_t E, int64_t K);
PYG_HIP_API int pyg_hip_segment_csr_ws(int op, int dtype, const void* src, const int64_t* indptr,
                                       int64_t indptr_slice_stride, void* out, int64_t* arg_out, int fresh,
                                       int64_t leading, int64_t rows, int64_t E, int64_t K, void* workspace,
                                       size_t workspace_bytes, void* stream);

/*
 * out[slice, e, :] = src[slice, r, :] for every position e of row r; positions covered by no row keep
 * their contents.  src [leading, rows, K], out [leading, E, K].  Replaces pyg::gather_csr (schema
 * ops/segment_csr.cpp:170-172; CPU ops/cpu/segment_csr_kernel.cpp:551-648).
 */
PYG_HIP_API int pyg_hip_gather_csr(int dtype, const void* src, const int64_t* indptr,
                                   int64_t indptr_slice_stride, void* out, int64_t leading, int64_t rows,
                                   int64_t E, int64_t K, void* stream);
/* ... with scratch for hub rows (pyg_hip_csr_hub_workspace_size(4, ...)): their positions are written by all workgroups. */
PYG_HIP_API int pyg_hip_gather_csr_ws(int dtype, const void* src, const int64_t* indptr,
                                      int64_t indptr_slice_stride, void* out, int64_t leading, int64_t rows,
                                      int64_t E, int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Softmax over the groups ptr[g] .. ptr[g+1] along the middle axis of src [outer, D, inner], per
 * (group, outer, inner) head; `out` must be zero-filled by the caller (positions outside every group
 * stay 0).  float32 / float64.  Replaces pyg::softmax_csr / pyg::softmax_csr_backward (schemas
 * ops/softmax.cpp:46-53; CPU only in the reference: ops/cpu/softmax_kernel.cpp:58-222).
 */
PYG_HIP_API int pyg_hip_softmax_csr(int dtype, const void* src, const int64_t* ptr, void* out, int64_t outer,
                                    int64_t D, int64_t inner, int64_t groups, void* stream);
PYG_HIP_API int pyg_hip_softmax_csr_backward(int dtype, const void* out, const void* out_grad,
                                             const int64_t* ptr, void* in_grad, int64_t outer, int64_t D,
                                             int64_t inner, int64_t groups, void* stream);

/*
 * Device hash map key -> first position: the kernels behind torch.classes.pyg.CUDAHashMap
 * (pyg_lib/csrc/classes/cuda/hash_map.cu:36-100, a cuco::static_map wrapper in the reference).
 * The table is caller-owned: `slots` = pyg_hip_hash_map_slots(n, load_factor) entries (a power of two)
 * of table_keys (uint64) and table_vals (int64).  Keys are int16 / int32 / int64; INT64_MIN is the empty
 * sentinel, as in the reference.  `build` leaves the number of distinct keys in *distinct_dev (device
 * memory); duplicate keys map to their first position.  `get` writes the position or -1 per query.
 */
PYG_HIP_API int64_t pyg_hip_hash_map_slots(int64_t n, double load_factor);
PYG_HIP_API int pyg_hip_hash_map_build(int key_dtype, const void* keys, int64_t n, uint64_t* table_keys,
                                       int64_t* table_vals, int64_t slots, int64_t* distinct_dev, void* stream);
PYG_HIP_API int pyg_hip_hash_map_get(int key_dtype, const void* query, int64_t m, const uint64_t* table_keys,
                                     const int64_t* table_vals, int64_t slots, int64_t* out, void* stream);

/* When enabled (per calling thread), every dominant-kernel launch is bracketed by a pair of HIP
 * events recorded on the stream the kernel is launched on.  pyg_hip_profile_collect waits for the
 * recorded launches, writes up to `capacity` durations (milliseconds, launch order) and returns
 * how many launches were recorded since the last collect. */
PYG_HIP_API void pyg_hip_profile_enable(int on);
PYG_HIP_API int pyg_hip_profile_collect(float* ms_out, int capacity);

/* Hand-written device-to-device streaming copy of `bytes` bytes (16-byte aligned buffers), the yardstick
 * bench.py prints as roofline.achievable: mode 0 = fine-grained non-persistent sweep (the best copy found on
 * this hardware), 1 = persistent with one contiguous range per workgroup, 2 = persistent cyclic -- the two
 * segment_matmul tile schedules without the arithmetic.  Measurement support; no operator calls it. */
PYG_HIP_API int pyg_hip_stream_copy(const void* src, void* dst, size_t bytes, int mode, void* stream);

/* The shader clock the chip actually runs at while other streams load it: one wave on `stream` watches the shader-clock
 * counter and the constant 100 MHz counter for `milliseconds` and leaves {shader cycles, 100 MHz ticks} in out2 (device
 * memory, 2 x uint64): MHz = out2[0] / out2[1] * 100.  The fp32 MFMA peak of the data sheet (157 TFLOP/s) is quoted at
 * 2.4 GHz; under a sustained MFMA + HBM load the chip clocks lower, and bench.py prices the fp32 kernel against both. */
PYG_HIP_API int pyg_hip_clock_probe(uint64_t* out2, double milliseconds, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* PYG_HIP_H_ */
