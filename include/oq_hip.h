/*
 * oq_hip.h -- C ABI of the MI355X (gfx950) implementation of onnx_quantize's numeric hot path.
 *
 * This is the drop-in boundary.  The reference is pure Python/NumPy and has no
 * FFI of its own; every entry point below replaces one NumPy function of the reference, cited as
 * file:line relative to /root/reference/src/onnx_quantize.  The binding a reference maintainer would
 * add is a ctypes stub -- shown in INTEGRATION.md and implemented in onnx_quantize_amd/hip/_lib.py.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / HIP types.  `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream).  All data pointers are DEVICE pointers unless marked HOST.
 *   - every function is asynchronous on `stream` and never allocates: the caller owns inputs, outputs
 *     and the workspace (size from the matching *_workspace_bytes query).  Inputs are never mutated
 *     unless the parameter is documented in/out.
 *   - return value: 0 (OQ_OK) or a negative oq_status; oq_last_error() returns a thread-local,
 *     human-readable description of the last failure on the calling thread.
 *   - weights are fp32, row-major [K, N] = (in_channels, out_channels) with leading dimension ldw
 *     (elements); groups run along K for a fixed output channel, exactly like
 *     core/_algorithms/utils.py:6-26.
 *   - 4-bit results are returned one value per byte (int8/uint8 containers, like the reference's
 *     in-memory ml_dtypes arrays) unless a packed layout is requested.
 *   - bounds, checked before any arithmetic on an argument (a call outside them returns a negative
 *     status; it never wraps a product, divides by a wrapped one or truncates a grid): every extent of
 *     an operand (rows, columns, tokens, leading dimensions) is in [1, 2^31 - 1]; an operand has at
 *     most 2^40 elements; flat element counts with one thread per element (oq_qparams_*,
 *     oq_quantize_bias_f32, the packers) at most 2^38; lists and batches at most 65535 entries; the
 *     width K of a Hessian / factor at most 2^17; running sample counts at most 2^52.  The matching
 *     *_workspace_bytes query returns 0 (or its 256-byte floor) for a request outside the bounds.
 */
#ifndef OQ_HIP_H
#define OQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OQ_ABI_VERSION 2   /* 2: the Hessian method is an argument of every call that uses it (no process-wide setter); q_layout in oq_gptq_loop_f32 */

typedef enum {
    OQ_OK = 0,
    OQ_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, bad enum ...            */
    OQ_ERR_UNSUPPORTED = -2,      /* valid request this build has no kernel for           */
    OQ_ERR_WORKSPACE = -3,        /* workspace missing or too small                       */
    OQ_ERR_LAUNCH = -4,           /* HIP reported an error (text in oq_last_error())      */
    OQ_ERR_NOT_SPD = -5           /* Cholesky hit a non-positive pivot (see oq_gptq_factor) */
} oq_status;

/* core/_dtypes.py:33-41 (QuantType); ranges core/_dtypes.py:8-30 */
typedef enum { OQ_INT4 = 0, OQ_UINT4 = 1, OQ_INT8 = 2, OQ_UINT8 = 3, OQ_INT32 = 4, OQ_UINT32 = 5 } oq_qtype;
/* core/_qconfig.py:31-36 (QuantizationStrategy) */
typedef enum { OQ_TENSOR = 0, OQ_CHANNEL = 1, OQ_GROUP = 2 } oq_strategy;
/* layout of the integer output of oq_rtn_quantize_f32 */
typedef enum {
    OQ_LAYOUT_KN = 0,         /* [K, N], one value per byte: what _rtn_quantize returns (rtn.py:106-109)     */
    OQ_LAYOUT_NBITS = 1,      /* MatMulNBits blob [N, K/g, g*bits/8] (qrules/_common.py:65-87), 4/8-bit group */
    OQ_LAYOUT_KN_PACKED4 = 2  /* [K, N/2] bytes: core/_pack.py:8-22 applied to the [K, N] result of a 4-bit type (flat
                                 order, even index in the low nibble, signed values as two's-complement nibbles) -- what
                                 the reference serialises for int4 / uint4 initializers; written by the kernel epilogue.
                                 4-bit types, group strategy with K % g == 0 and g <= 256, N % 4 == 0 (else
                                 OQ_ERR_UNSUPPORTED) */
} oq_layout;
/* error-feedback indexing of the GPTQ loop */
typedef enum {
    OQ_GPTQ_PARITY = 0,            /* gptq.py:199,:208 as written: column of the UPPER factor below the diagonal (zeros) */
    OQ_GPTQ_CORRECTED = 1,         /* row of the upper factor right of the diagonal (what GPTQ intends); opt-in          */
    OQ_GPTQ_CORRECTED_COLUMNS = 2  /* the same bytes as 1 from the one-column-per-lane kernel (the fallback of 1 for group
                                      sizes that are not a multiple of 16); selectable so that the two kernels can be held
                                      against each other without an environment switch                                  */
} oq_gptq_mode;

int32_t oq_abi_version(void);
const char* oq_last_error(void);
const char* oq_status_string(int32_t status);
/* name of the device code object the library was built for ("gfx950") */
const char* oq_target_arch(void);

/* T1: core/_dtypes.py:61-70 QuantType.qrange(is_symmetric, reduce_range).  HOST pointers. */
int32_t oq_qrange(int32_t qtype, int32_t symmetric, int32_t reduce_range, int64_t* qmin, int64_t* qmax);

/* ---------------------------------------------------------------------------------------------
 * A1  core/_algorithms/rtn.py:54-109  _rtn_quantize  (= L1 utils.py:6 + R1 utils.py:42 + Q1 utils.py:242
 *     [+ M1 utils.py:140 when mse] + K1 utils.py:72 + L2 utils.py:29, fused).
 *
 *   W          [K, N] fp32, leading dimension ldw >= N
 *   group_size GROUP only: > 0 (clamped to K), or -1 (= K).  K % group_size must be 0 (the reference's
 *              product path guarantees it, qrules/_common.py:13-29) or N*K % group_size == 0 (groups
 *              then straddle columns exactly like W.T.reshape(-1, g)).
 *   q_out      OQ_LAYOUT_KN: K*N bytes.  OQ_LAYOUT_KN_PACKED4: K*N/2 bytes (4-byte aligned).  OQ_LAYOUT_NBITS: N*(K/g)*(g*bits/8) bytes (group strategy, 4/8-bit types,
 *              K % g == 0, g % 16 == 0, 16-byte aligned; g <= 256 in the fused kernels, larger g -- group_size -1
 *              = the whole column is MatMulNBits-eligible, qrules/_common.py:32-62 -- needs g % 128 == 0 and
 *              N % 4 == 0; anything else is OQ_ERR_UNSUPPORTED, never a silent fallback).  Signed types are
 *              stored as two's-complement nibbles / bytes.
 *   scale_out  fp32: 1 (tensor) | N (channel) | N*K/g (group; entry n*(K/g)+kg)
 *   zp_out     same count, 1 byte each (the weight container dtype)
 * ------------------------------------------------------------------------------------------- */
size_t oq_rtn_workspace_bytes(int64_t K, int64_t N, int32_t strategy, int64_t group_size, int32_t mse);
int32_t oq_rtn_quantize_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype,
                            int32_t strategy, int64_t group_size, int32_t symmetric,
                            int32_t reduce_range, float clip_ratio, int32_t mse, void* q_out,
                            float* scale_out, void* zp_out, int32_t layout, void* workspace,
                            size_t workspace_bytes, void* stream);

/* A1 with caller-kept STATE for the strategies whose range spans several workgroups (per channel, per tensor, groups taller than
 *     256 rows: rtn_resident.hip).  Those kernels hand work out by tickets and complete ranges through counters and keys in
 *     device memory that must start from zero; oq_rtn_quantize_f32 clears them in its workspace with a launch of its own (a
 *     tenth of the call on a 4096 x 11008 matrix).  Here the caller provides `state` (oq_rtn_state_bytes, 16-byte aligned):
 *     ZERO before the first call that uses it, and left zero by every call (the last workgroup to leave cleans up), so no clear
 *     launch runs.  One state buffer serves any sequence of calls of any shapes (each call needs its own size) on ONE stream; two
 *     calls that may run concurrently need two buffers.  state == NULL or too small: exactly oq_rtn_quantize_f32.  A call
 *     that fails on the device leaves the state undefined.
 *     Round 6: the fused group kernel uses the state too -- layouts OQ_LAYOUT_KN and OQ_LAYOUT_KN_PACKED4, 128-row groups, at most 64
 *     k-groups: the n-major parameters are staged in it as self-validating words and transposed INSIDE the launch by blocks appended to
 *     the grid (no second launch: 4096 x 11008 bytes 45.0 -> 42.5 us, packed nibbles 43.3-44.3 -> 40.0-40.6; with write-through stores and K1 by row groups 40.4 / 37.9); oq_rtn_state_bytes says
 *     how much it needs for a shape (0: the call keeps no state).  Those appended blocks wait for the main blocks of their own launch,
 *     so such calls are ordered like the ticketed ones (below).
 *     Concurrency of the ticketed kernels (both entry points): inside a call a workgroup may wait for other workgroups of the
 *     SAME launch, which is safe on its own (see rtn_resident.hip); two such launches running at the same time on one device
 *     would compete for the CUs their waiting workgroups hold and could stop each other for good.  The library therefore
 *     orders them itself.  While every such call of a device comes from one stream, the stream orders them.  The first call from
 *     a second stream blocks the HOST once (a device synchronisation: the library keeps no handle of a caller's stream, so a
 *     stream that was destroyed in between cannot hurt); from then on every such call makes its stream wait, on the device, for
 *     an event recorded behind the previous one, on the calling stream itself (2.5-3.5 us per call).  So these
 *     kernels never overlap whatever LIVE streams and threads they come from.  A stream that is being CAPTURED into a graph
 *     never runs them: such a call takes the three-launch path (W read twice, no tickets; it needs the workspace
 *     oq_rtn_workspace_bytes states), because replays of a graph are ordered against nothing the library can see.  What the library cannot order are kernels of ANOTHER PROCESS on
 *     the same GPU: do not run two processes that issue these calls on one device.  The fused group kernels
 *     (group_size <= 256) never wait and have no such restriction. */
size_t oq_rtn_state_bytes(int64_t K, int64_t N, int32_t strategy, int64_t group_size);
int32_t oq_rtn_quantize_stateful_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype,
                                     int32_t strategy, int64_t group_size, int32_t symmetric,
                                     int32_t reduce_range, float clip_ratio, int32_t mse, void* q_out,
                                     float* scale_out, void* zp_out, int32_t layout, void* workspace,
                                     size_t workspace_bytes, void* state, size_t state_bytes, void* stream);

/* A1 over a strided batch of equally shaped matrices (stacked projection weights, experts, ...): matrix b is
 *     W + b * w_stride (fp32 elements); outputs are packed back to back (q: K*N bytes resp. the blob size per
 *     matrix; scale / zp: N*K/g entries per matrix).  Group strategy, K % group_size == 0, group_size <= 256.
 *     One launch covers all matrices, so the tail of one matrix overlaps the head of the next. */
size_t oq_rtn_batched_workspace_bytes(int64_t batch, int64_t K, int64_t N, int64_t group_size);
int32_t oq_rtn_quantize_batched_f32(const float* W, int64_t batch, int64_t w_stride, int64_t K, int64_t N,
                                    int64_t ldw, int32_t qtype, int64_t group_size, int32_t symmetric,
                                    int32_t reduce_range, float clip_ratio, void* q_out, float* scale_out,
                                    void* zp_out, int32_t layout, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* A1 over a LIST of equally shaped matrices that live anywhere in device memory (the MatMul weights of a model: the
 *     reference walks them node by node, qrules/_common.py:126-142): entry i of the table holds the four device pointers
 *     of matrix i -- W [K, N] fp32 with leading dimension ldw, q_out (K*N bytes resp. the blob), scale_out / zp_out
 *     (N*K/g entries).  About 1.6e8 parameters share a launch (blockIdx.y = entry; 9 matrices of 4096 x 4096, 3 of
 *     4096 x 11008, hundreds of gemma-sized ones): rounds of waves merge across matrices and small matrices are not
 *     launch-bound (measured per shape in rtn.hip).  Per matrix the result is bit-identical to oq_rtn_quantize_f32.
 *     `table_host` and `table_device` are the same `count` entries in host and device memory (the host copy is what the
 *     alignment checks read; the kernels read the device copy, which may be NULL when count == 1).  Group strategy, K % group_size == 0, group_size <= 256;
 *     every W and q_out 16-byte aligned.  Workspace: oq_rtn_batched_workspace_bytes(count, K, N, group_size). */
typedef struct {
    const float* W;
    void* q_out;
    float* scale_out;
    void* zp_out;
} oq_rtn_ptrs;
int32_t oq_rtn_quantize_ptrs_f32(const oq_rtn_ptrs* table_host, const oq_rtn_ptrs* table_device, int64_t count, int64_t K,
                                 int64_t N, int64_t ldw, int32_t qtype, int64_t group_size, int32_t symmetric,
                                 int32_t reduce_range, float clip_ratio, int32_t layout, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* Q2  core/_algorithms/utils.py:302-348  _compute_qparams_from_array on [K, N] weights (no integer
 *     output): same arguments and outputs as above minus q_out. */
int32_t oq_rtn_qparams_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype,
                           int32_t strategy, int64_t group_size, int32_t symmetric,
                           int32_t reduce_range, float clip_ratio, int32_t mse, float* scale_out,
                           void* zp_out, void* workspace, size_t workspace_bytes, void* stream);

/* Q1  core/_algorithms/utils.py:242-299  _compute_qparams on `count` (rmin, rmax) pairs.
 *     zp_out is written as int32 (caller narrows to the zero-point dtype). */
int32_t oq_qparams_f32(const float* rmin, const float* rmax, int64_t count, int32_t qtype,
                       int32_t symmetric, int32_t reduce_range, float* scale_out, int32_t* zp_out,
                       void* stream);

/* Same for float64 ranges (the reference's arithmetic follows the input dtype, utils.py:260-262; its own
 * known-answer tests feed float64): scale is computed in double and written as fp32. */
int32_t oq_qparams_f64(const double* rmin, const double* rmax, int64_t count, int32_t qtype,
                       int32_t symmetric, int32_t reduce_range, float* scale_out, int32_t* zp_out,
                       void* stream);

/* R1  core/_algorithms/utils.py:60-61 with axis=1: per-row min and max of a row-major [R, C] array (the
 *     reference's preprocessed layout, one row per (scale, zp)).  Raw extrema: no clip ratio, no zero. */
int32_t oq_minmax_rows_f32(const float* x, int64_t R, int64_t C, int64_t ldx, float* min_out,
                           float* max_out, void* stream);

/* K1  core/_algorithms/utils.py:72-79  _quantize_array_from_qparams, and
 * K2  core/_algorithms/utils.py:102-137 _dequantize_array, on a row-major [R, C] array.
 *     The (scale, zp) entry used by element (r, c) is  (r / row_div) * row_stride + c * col_stride :
 *       per tensor            row_div=1 row_stride=0 col_stride=0
 *       per row  ([R] params) row_div=1 row_stride=1 col_stride=0   (the reference's preprocessed layout)
 *       per column ([C])      row_div=1 row_stride=0 col_stride=1   (channel params on [K, N])
 *       groups of g rows      row_div=g row_stride=1 col_stride=R/g (group params on [K, N])
 *     zp is int32 per entry; q is 1 byte per value (4/8-bit types) or int32 (32-bit types). */
int32_t oq_quantize_f32(const float* x, int64_t R, int64_t C, int64_t ldx, const float* scale,
                        const int32_t* zp, int64_t row_div, int64_t row_stride, int64_t col_stride,
                        int32_t qtype, int32_t symmetric, int32_t reduce_range, void* q_out, void* stream);
int32_t oq_dequantize_f32(const void* q, int64_t R, int64_t C, int32_t qtype, const float* scale,
                          const int32_t* zp, int64_t row_div, int64_t row_stride, int64_t col_stride,
                          float* x_out, int64_t ldo, void* stream);

/* K2 with FLOAT zero points (core/_algorithms/hqq.py:77-78 keeps zero points in the scale dtype; utils.py:130-132 then
 *     subtracts them as they are): (f32(q) - zp) * scale, same addressing as oq_dequantize_f32.  4- / 8-bit containers. */
int32_t oq_dequantize_fzp_f32(const void* q, int64_t R, int64_t C, int32_t qtype, const float* scale,
                              const float* zp, int64_t row_div, int64_t row_stride, int64_t col_stride,
                              float* x_out, int64_t ldo, void* stream);

/* A3  core/_algorithms/rtn.py:112-138  _quantize_bias: bias_scale = w_scale * x_scale (fp32),
 *     q = clip(int32(rint(bias / bias_scale))).  w_scale has 1 or n entries. */
int32_t oq_quantize_bias_f32(const float* bias, int64_t n, const float* w_scale, int64_t n_w_scale,
                             float x_scale, int32_t* q_out, float* bias_scale_out, void* stream);

/* A1 with strategy = tensor for n weight tensors in three launches (a model of many small MatMul weights makes a
 *     loop over oq_rtn_quantize_f32 launch-bound).  `desc` is a DEVICE array of n oq_rtn_tensor_desc: w = `count`
 *     contiguous fp32 values, q_out = `count` bytes (4-bit values one per byte, signed ones as two's complement),
 *     scale_out / zp_out = one entry each.  4- and 8-bit types only.  Same bits as oq_rtn_quantize_f32 per tensor. */
typedef struct {
    const float* w;
    int64_t count;
    void* q_out;
    float* scale_out;
    void* zp_out;
} oq_rtn_tensor_desc;
size_t oq_rtn_tensor_many_workspace_bytes(int64_t n);
int32_t oq_rtn_tensor_many_f32(const void* desc, int64_t n, int32_t qtype, int32_t symmetric,
                               int32_t reduce_range, float clip_ratio, void* workspace,
                               size_t workspace_bytes, void* stream);

/* C1 for a whole calibration batch: n tensors in ONE launch pair (calibrate.py:264-266 calls collect once per
 *     tensor and batch; dozens of 10-40 MB tensors per batch make that loop launch-bound).  `desc` is a DEVICE
 *     array of n oq_minmax_desc {x, count, state}; every count > 0; each state as in oq_minmax_collect_f32. */
typedef struct {
    const float* x;
    int64_t count;
    float* state;
} oq_minmax_desc;
size_t oq_minmax_many_workspace_bytes(int64_t n);
int32_t oq_minmax_collect_many_f32(const void* desc, int64_t n, double momentum, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C1  core/_calibration/minmax.py:40-64  MinMaxCalibrator.collect for one activation tensor:
 *     cur = (min(x), max(x)); first sight: state = cur; momentum > 0: state = m*state + (1-m)*cur
 *     (fp32, the two products rounded separately); else running min / max.
 *   state  DEVICE float[4] in/out: {min, max, seen (0.0 = nothing collected yet), unused}
 *   momentum is the Python float of the reference; the kernel uses T(momentum) and T(1 - momentum)
 *   exactly as NumPy's weak-scalar promotion does.  The whole update happens on the device; nothing
 *   is copied back.
 * ------------------------------------------------------------------------------------------- */
size_t oq_minmax_workspace_bytes(int64_t count);
int32_t oq_minmax_collect_f32(const float* x, int64_t count, float* state, double momentum,
                              void* workspace, size_t workspace_bytes, void* stream);
int32_t oq_minmax_collect_f64(const double* x, int64_t count, double* state, double momentum,
                              void* workspace, size_t workspace_bytes, void* stream);

/* S1  pre_passes/smooth_quant.py:62-74: out[c] = max_r |x[r, c]| for a row-major [R, C] array
 *     (activations flattened to [T, K]; for a weight's per-in-channel absmax pass transposed=1 to
 *     get out[r] = max_c |x[r, c]|). */
size_t oq_absmax_workspace_bytes(int64_t R, int64_t C, int32_t transposed);
int32_t oq_absmax_f32(const float* x, int64_t R, int64_t C, int64_t ldx, int32_t transposed,
                      float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * G1  core/_algorithms/gptq.py:246-260  _accumulate_hessian:
 *       H *= n_seen / (n_seen + n_add);  H += (2 / (n_seen + n_add)) * X^T X      (fp32, MFMA)
 *     X [T, K] row-major (ldx), H [K, K] row-major, in/out, full symmetric matrix is maintained.
 *     n_add is the LEADING dimension of the activation before flattening (gptq.py:247), i.e. the
 *     number of samples, not T.  workspace (oq_hessian_workspace_bytes; optional for OQ_HESSIAN_F32, where
 *     it may be NULL): the bf16 operand pieces of the split methods (6 B per element of X) and partial-sum
 *     slabs that let the T dimension be split over more workgroups (deterministic two-stage sum).
 *     `method` (an argument of every entry point that multiplies on the matrix cores -- there is no process-wide setting and
 *     the library reads no environment variable for it, so two threads may use different methods):
 *       OQ_HESSIAN_F32     v_mfma_f32_32x32x2_f32 on the operands scaled by sqrt(2/n) as gptq.py:257 does;
 *       OQ_HESSIAN_BF16X6  every fp32 element split EXACTLY into three bf16 pieces (x = hi + mid + lo), the six piece
 *                          products down to 2^-16 |x y| on v_mfma_f32_32x32x16_bf16, fp32 accumulation; what is left
 *                          out is <= 2^-23 |x y| per product (one fp32 rounding); 2 / n applied to the sum;
 *       OQ_HESSIAN_BF16X9  all nine piece products (no product rounding at all);
 *       OQ_HESSIAN_F16X3   two fp16 pieces of every element after scaling the call's X by a power of two (max |x| in
 *                          [2^14, 2^15)): x = hi + lo to 22 bits (fewer below 2^-17 max |x|, where fp16 runs out of
 *                          exponent; a sample whose fp32 square is still positive never vanishes entirely, so a channel
 *                          is dead here exactly when it is dead for gptq.py:284), products hi.hi + hi.lo + lo.hi on
 *                          v_mfma_f32_16x16x32_f16, 2 / (n s^2) applied to the sum: half the matrix-core work of
 *                          BF16X6, measured error against float64 not larger (8e-7 vs 2e-6 of max |H|, K = 11008);
 *       OQ_HESSIAN_AUTO    F16X3 for K >= 1024 (its block tile is 256 x 256; K < 2048 also needs T >= 2048) when the
 *                          workspace holds the pieces,
 *                          else F32.  An explicit split method with too small a workspace is OQ_ERR_WORKSPACE.
 *     The reference's own H goes through sgemm in BLAS order: parity is to a tolerance for every method.
 * ------------------------------------------------------------------------------------------- */
enum { OQ_HESSIAN_AUTO = 0, OQ_HESSIAN_F32 = 1, OQ_HESSIAN_BF16X6 = 2, OQ_HESSIAN_BF16X9 = 3, OQ_HESSIAN_F16X3 = 4 };
size_t oq_hessian_workspace_bytes(int64_t T, int64_t K);
int32_t oq_hessian_accumulate_f32(const float* X, int64_t T, int64_t K, int64_t ldx, int64_t n_seen,
                                  int64_t n_add, float* H, int32_t method, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* G1 for a LIST of inputs in one launch chain (core/_calibration/calibrate.py:292-305 walks the GPTQ nodes and hands
 *     `_accumulate_hessian`, gptq.py:246-260, one input each; a calibration batch of gemma-3-270m taps 72 of them, 5120 rows
 *     by 640 / 1024 / 2048 columns: 6 to 36 tiles of the 256 x 256 product kernel each, five launches per tensor).  Every
 *     item is one oq_hessian_accumulate_f32 call: H <- H n_seen / (n_seen + n_add) + 2 / (n_seen + n_add) X^T X.  All items
 *     run the fp16-piece method (OQ_HESSIAN_F16X3 above) whatever their size, each with a scale of its own, the product
 *     writing H directly: the same <= 1e-5 max |H| bound; NOT bit-identical to the per-tensor call, which may slice T or
 *     pick the fp32 kernel for small K; an item's rows are summed in ONE fp32 chain, so very long items (beyond ~32 768 rows)
 *     are better served by the per-tensor call.  A caller that wants another method calls oq_hessian_accumulate_f32 per item.
 *     `items_host` / `items_device`: the same `count` items in host and in device memory (the host copy sizes the
 *     launches, the kernels read the device copy).  Workspace: oq_hessian_many_workspace_bytes(items_host, count) =
 *     the fp16 pieces of all items (4 B per element of the zero-padded inputs) + a table. */
typedef struct {
    const float* X;   /* [T, K] fp32 rows, leading dimension ldx */
    float* H;         /* [K, K] fp32, contiguous; read only when n_seen > 0 */
    int64_t T, K, ldx;
    int64_t n_seen;   /* samples (leading-dimension entries, gptq.py:247) already in H */
    int64_t n_add;    /* samples this X adds */
    int64_t reserved; /* 0 */
} oq_hessian_item;
size_t oq_hessian_many_workspace_bytes(const oq_hessian_item* items_host, int64_t count);
int32_t oq_hessian_accumulate_many_f32(const oq_hessian_item* items_host, const oq_hessian_item* items_device,
                                       int64_t count, void* workspace, size_t workspace_bytes, void* stream);

/* G1 in two halves, for callers that own more than one stream.  With the fp16-piece method a batch costs one HBM-bound
 *     preparation (max |x|, the power-of-two scale, the two fp16 pieces: X read twice, 4 B / element written) and one
 *     matrix-core bound product; issued on two streams, the preparation of batch i + 1 runs beside the product of batch i.
 *       oq_hessian_prepare_f32               X -> `pieces` (256-byte aligned, oq_hessian_pieces_bytes(T, K)); n_total = the
 *                                            sample count AFTER this batch (n_seen + n_add: gptq.py:284's dead-channel test
 *                                            sees the same 2 / n as the fp32 path)
 *       oq_hessian_accumulate_prepared_f32   `pieces` -> H, exactly what oq_hessian_accumulate_f32 with OQ_HESSIAN_F16X3 does
 *                                            behind its own preparation: the two calls in sequence give the same bits.
 *     `slabs` (oq_hessian_slab_bytes(K), may be NULL): partial sums of the T-slices. */
size_t oq_hessian_pieces_bytes(int64_t T, int64_t K);
size_t oq_hessian_slab_bytes(int64_t K);
int32_t oq_hessian_prepare_f32(const float* X, int64_t T, int64_t K, int64_t ldx, int64_t n_total, void* pieces,
                               size_t pieces_bytes, void* stream);
int32_t oq_hessian_accumulate_prepared_f32(const void* pieces, int64_t T, int64_t K, int64_t n_seen, int64_t n_add,
                                           float* H, void* slabs, size_t slab_bytes, void* stream);

/* N1  calibrate.py:244-251 runs the augmented model in an onnxruntime session; on the device the large products of that
 *     run (activations [M, Kd] times a constant weight [Kd, N]) can take the Hessian's route through the fp16 matrix
 *     cores: both operands split into two fp16 pieces under a power-of-two scale (22 significand bits), three products,
 *     fp32 accumulate.
 *       oq_matmul_pieces_bytes(Kd, cols)   size of the piece buffer of one operand with `cols` rows / columns (0: too large)
 *       oq_matmul_prepare_f32              operand -> pieces (256-byte aligned).  contraction_is_fast_axis != 0: the source is
 *                                          [cols, Kd] row-major (activations X [M, Kd]); 0: [Kd, cols] row-major (a weight
 *                                          W [Kd, N], prepared once).  ldx: leading dimension of the source.  per_row_scales != 0
 *                                          (fast-axis sources only): one power-of-two scale per source row instead of one for the
 *                                          whole operand -- rows of very different magnitude each keep their 22 bits.
 *       oq_matmul_pieces_f32               C [M, N] (ldc) = beta C + alpha A B from the two piece buffers; a_per_row_scales: how
 *                                          A was prepared. */
size_t oq_matmul_pieces_bytes(int64_t Kd, int64_t cols);
int32_t oq_matmul_prepare_f32(const float* X, int64_t Kd, int64_t cols, int64_t ldx, int32_t contraction_is_fast_axis,
                              int32_t per_row_scales, void* pieces, size_t pieces_bytes, void* stream);
int32_t oq_matmul_pieces_f32(const void* pieces_a, const void* pieces_b, int64_t M, int64_t N, int64_t Kd, float alpha,
                             float beta, float* C, int64_t ldc, int32_t a_per_row_scales, void* stream);

/* G3 prologue  gptq.py:118-127: dead = diag(H) == 0 -> H[d,d] = 1, W[d,:] = 0 (both in place);
 *     when actorder: perm_out = argsort(diag(H)) reversed (ties: larger index first) and W, H are
 *     permuted in place (Wtmp/Htmp-free: uses the workspace).  perm_out may be NULL otherwise. */
size_t oq_gptq_prepare_workspace_bytes(int64_t K, int64_t N, int32_t actorder);
int32_t oq_gptq_prepare_f32(float* W, int64_t K, int64_t N, float* H, int32_t actorder,
                            int32_t* perm_out, void* workspace, size_t workspace_bytes, void* stream);

/* G2  gptq.py:134-150: H += percdamp*mean(diag H) on the diagonal, then the UPPER factor U with
 *     inv(H) = U^T U, written to U_out [K, K] (strictly-lower part zeroed).  H is read only.
 *     info (DEVICE int32): 0 = ok; > 0 = first non-positive pivot (1-based), in which case U_out = I
 *     (the reference's LinAlgError fallback, gptq.py:143-150).  The call itself still returns OQ_OK.
 *     Arithmetic: fp32 throughout, except that the large products (trailing updates of 512-row panels on squares of
 *     >= 1024 columns, inverse levels of >= 2048-row blocks) take their operands as two fp16 pieces (22 bits, fp32
 *     accumulate) like the Hessian's OQ_HESSIAN_F16X3 -- measured no less accurate against float64 than fp32 products
 *      --; method = OQ_HESSIAN_F32 keeps them on the fp32 kernel as well (any other value: fp16 pieces). */
size_t oq_gptq_factor_workspace_bytes(int64_t K);
int32_t oq_gptq_factor_f32(const float* H, int64_t K, float percdamp, float* U_out, int32_t* info,
                           int32_t method, void* workspace, size_t workspace_bytes, void* stream);
/*     The same for `count` matrices of one width in lock-step (the inputs of a model that share K: the reference
 *     factors each node's H by itself, gptq.py:134-150 once per `_gptq` call).  Matrix i is H + i * h_stride, its
 *     factor U_out + i * u_stride (strides in floats, >= K * K), its status info[i].  The chain of a factorisation is
 *     ceil(K / 128) strictly sequential diagonal blocks; every launch of that chain here carries all `count` matrices,
 *     so a batch costs the latency of ONE chain plus the matrix-core work of `count`.  Per matrix the operations and
 *     their order are those of oq_gptq_factor_f32: the results are bit-identical.
 *     fix_dead != 0: a zero diagonal entry of H counts as 1 (gptq.py:119-120; for an H that did not pass
 *     oq_gptq_prepare_f32 -- the weights' dead rows remain the caller's business, gptq.py:121). */
size_t oq_gptq_factor_batched_workspace_bytes(int64_t K, int64_t count);
int32_t oq_gptq_factor_batched_f32(const float* H, int64_t K, int64_t h_stride, int64_t count, float percdamp,
                                   int32_t fix_dead, float* U_out, int64_t u_stride, int32_t* info,
                                   int32_t method, void* workspace, size_t workspace_bytes, void* stream);

/* G3  gptq.py:153-216: the block / row loop.  W [K, N] is the working copy (after oq_gptq_prepare_f32); it
 *     receives the lazy batch updates of gptq.py:208 (OQ_GPTQ_CORRECTED only: in OQ_GPTQ_PARITY the update term
 *     is structurally zero) and is otherwise left as is.
 *     group_size: the reference's loop value; > 0 re-derives per-column (scale, zp) from rows
 *                 [r, r+group_size) of W whenever r % group_size == 0 (gptq.py:168-184); <= 0: never.
 *     init_scale/init_zp: the parameters computed before the loop (gptq.py:104-116), init_count = 1 (tensor) or N.
 *     block_size: gptq.py:153, any value > 0.  A launch walks <= 128 rows; a taller block is a chain of such launches on a
 *                 working copy of its rows (the reference's W1) that alone receives the in-block updates, while group
 *                 parameters are read from W as it stood when the block began -- the reference's semantics.  OQ_GPTQ_CORRECTED
 *                 with block_size <= 128 additionally defers the update of everything behind a SUPER-block of up to 512 rows
 *                 (a multiple of block_size and group_size) to one product with Kd = 512: the same sums, grouped differently.
 *     method: OQ_HESSIAN_F32 keeps the deferred updates on the fp32 MFMA kernel, any other value runs the large ones on
 *                 the fp16-piece GEMM (22-bit operands, fp32 accumulate).
 *     Outputs: q_int_out [K, N] one value per byte (q_layout = OQ_LAYOUT_KN) or [K, N/2] nibble pairs in core/_pack.py:8-22
 *     order (OQ_LAYOUT_KN_PACKED4: 4-bit types, N even; written by the loop kernels themselves); q_deq_out [K, N] fp32 (the dequantized rows the final qparams
 *     are re-derived from, gptq.py:219-231); used_scale/used_zp [ceil(K/group_size), N] = the parameters actually
 *     applied per (k-group, column) (NULL to skip; only written when group_size > 0).
 *     mse != 0 with group_size > 0: the per-group parameters come from the MSE search (utils.py:140-239, channel strategy
 *     on rows [r, r+group_size) of W) run beforehand for every group that starts in the block; the rows are then walked
 *     by the sequential kernel in both modes. */
size_t oq_gptq_loop_workspace_bytes(int64_t K, int64_t N, int64_t block_size);
int32_t oq_gptq_loop_f32(float* W, int64_t K, int64_t N, const float* U, int32_t qtype, int64_t group_size,
                         int32_t symmetric, int32_t reduce_range, float clip_ratio, int32_t mse,
                         int64_t block_size, int32_t mode, int32_t method, const float* init_scale,
                         const int32_t* init_zp, int64_t init_count, void* q_int_out, int32_t q_layout,
                         float* q_deq_out, float* used_scale, int32_t* used_zp, void* workspace,
                         size_t workspace_bytes, void* stream);

/* N2  core/_algorithms/hqq.py:106-213 (`_optimize_zero_point` + the final quantize of `_hqq_quantize`): uint4,
 *     asymmetric, group strategy, float zero points.  W [K, N] fp32; scale / zero_point_in / zero_point_out
 *     [N * K/g] fp32 in the rtn.py:98-109 order (entry n * K/g + kg); the initial scale and zero points come
 *     from oq_rtn_qparams_f32 (hqq.py:181-192; zero points converted to fp32).  beta / kappa are float64 like
 *     the reference's Python floats (beta *= kappa per round).  q_out: OQ_LAYOUT_KN [K, N] one value per byte, or
 *     OQ_LAYOUT_NBITS the MatMulNBits blob [N, K/g, g/2] HQQ is meant for (qrules/_common.py:65-87; float zero
 *     points stay unpacked [N, K/g], :96-99); NULL: zero points only.  rounds_out (device int32, may be NULL) receives the number of rounds evaluated before the
 *     early stop.  Every round, the mean-error reduction and the best / early-stop decision run on the
 *     device; nothing synchronises with the host.  K % group_size != 0 is OQ_ERR_UNSUPPORTED.
 *     Route: group sizes 16 / 32 / 64 / 128 with 1 .. 32 rounds walk ALL rounds of a row out of registers in one pass over
 *     W (a row's zero-point trajectory depends on the global error only through WHICH round is kept, hqq.py:131-140) and
 *     replay the decisions afterwards: 3 launches; anything else, or per_round_launches != 0, runs one launch pair per round
 *     (W re-read every round).  The two routes give the same bits. */
size_t oq_hqq_workspace_bytes(int64_t K, int64_t N, int64_t group_size);
int32_t oq_hqq_optimize_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t group_size,
                            int32_t reduce_range, const float* scale, const float* zero_point_in,
                            double lp_norm, double beta, double kappa, int32_t iters, int32_t early_stop,
                            int32_t per_round_launches, void* q_out, int32_t layout, float* zero_point_out,
                            int32_t* rounds_out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* N2  pre_passes/awq.py:47-72, 114-184: AWQ's scale search for one layer, device resident.  X [T, K] calibration rows
 *     (leading dimension ldx), W [K, N].  For the n_grid candidates i: s_i = clip(mean|x|^r / weight_scale^(1-r), 1e-4),
 *     r = i / n_grid, normalised by sqrt(max * min); loss_i = mean((X W - X W^_i)^2) with W^_i = dequant(RTN(W * s_i)) / s_i
 *     (rtn.py:54-109 with the given type / strategy / group size).  The loss is evaluated as || X (W - W^_i) ||^2 / (T N) by
 *     ONE product per candidate on the matrix cores (fp16 pieces of fp32 operands, fp32 accumulate) whose epilogue squares
 *     and sums: no [T, N] product is written.  Outputs (device): scales_out [n_grid, K], losses_out [n_grid], best_out =
 *     the first minimum (awq.py:178).  4- / 8-bit types; group strategy needs group_size >= 4 dividing K (16, 32, 64, 128 run the fused kernel).
 *     Workspace (256-byte aligned): oq_awq_workspace_bytes(T, K, N). */
size_t oq_awq_workspace_bytes(int64_t T, int64_t K, int64_t N);
int32_t oq_awq_scale_search_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw,
                                int32_t qtype, int32_t strategy, int64_t group_size, int32_t symmetric,
                                int32_t reduce_range, int32_t n_grid, float* scales_out, float* losses_out,
                                int32_t* best_out, void* workspace, size_t workspace_bytes, void* stream);
/* N2  pre_passes/awq.py:207-259: the clip search: losses_out[i] (i = 0..9) = mean((X W - X dequant(RTN(W, clip_ratio =
 *     1 - i / 100)))^2), best_out = the first minimum.  Same machinery and workspace as the scale search. */
int32_t oq_awq_clip_search_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw,
                               int32_t qtype, int32_t strategy, int64_t group_size, int32_t symmetric,
                               int32_t reduce_range, float* losses_out, int32_t* best_out, void* workspace,
                               size_t workspace_bytes, void* stream);
/* N2  pre_passes/smooth_quant.py:62-74, 104-113: scale_out[k] = max(max_t |x[t,k]|, 1e-5)^alpha /
 *     (max_n |w[k,n]| + 1e-9)^(1 - alpha). */
/* N2 from streamed statistics: a calibration walk that consumes its batches as they come does not hold X.  The searches need
 *     of it only  act_sum [K] = sum_t |x[t, k]|  (oq_abs_sum_cols_f32, a running sum over batches; awq.py:47-50 divides by T)  and
 *     G [K, K] = (2 / T) X^T X  (oq_hessian_accumulate_f32 with n counting ROWS) -- the loss is <D, G D> / (2 N), the form long
 *     calibration sets take anyway.  Same outputs as the searches above.  The in-place rescale of the reference
 *     (`node.meta["input"] /= scale`, awq.py:191) becomes act_sum / scale and G / (scale scale^T) on the caller's side. */
size_t oq_awq_stats_workspace_bytes(int64_t K, int64_t N);
int32_t oq_awq_scale_search_stats_f32(const float* act_sum, const float* G, int64_t T, int64_t K, const float* W, int64_t N, int64_t ldw,
                                      int32_t qtype, int32_t strategy, int64_t group_size, int32_t symmetric, int32_t reduce_range,
                                      int32_t n_grid, float* scales_out, float* losses_out, int32_t* best_out, void* workspace,
                                      size_t workspace_bytes, void* stream);
int32_t oq_awq_clip_search_stats_f32(const float* G, int64_t T, int64_t K, const float* W, int64_t N, int64_t ldw, int32_t qtype,
                                     int32_t strategy, int64_t group_size, int32_t symmetric, int32_t reduce_range, float* losses_out,
                                     int32_t* best_out, void* workspace, size_t workspace_bytes, void* stream);
size_t oq_abs_sum_cols_workspace_bytes(int64_t K);
int32_t oq_abs_sum_cols_f32(const float* X, int64_t T, int64_t K, int64_t ldx, float* sum_inout, int32_t accumulate, void* workspace,
                            size_t workspace_bytes, void* stream);

/* (b)  seam: `node.meta["input"]` is one array per value name, shared by its consumers (calibrate.py:301-307); everything `_gptq`
 *     derives from it alone (gptq.py:118-150, :246-260) is kept for the next consumer, keyed by CONTENT: out[0] = a 64-bit
 *     fingerprint of `nbytes` bytes at `data` (device memory, 16-byte aligned; `out` device memory, 8-byte aligned), one pass
 *     at the HBM rate.  Sum modulo 2^64 of a bijective mix of every 16-byte word chained with its index, plus the length:
 *     any edit of the bytes changes it except with probability 2^-64. */
int32_t oq_fingerprint64(const void* data, int64_t nbytes, uint64_t* out, void* stream);

size_t oq_smooth_quant_workspace_bytes(int64_t K);
int32_t oq_smooth_quant_scale_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N,
                                  int64_t ldw, float alpha, float* scale_out, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* N3  qrules/_common.py:65-123: MatMulNBits zero-point packing [N, ceil(K/g / 2)] (pad nibble 0x8)
 *     from the per-group zero points [N*K/g] (1 byte each).  4-bit only. */
int32_t oq_pack_zero_points_u4(const uint8_t* zp, int64_t N, int64_t blocks, uint8_t* out, void* stream);
/* N3  qrules/_common.py:72-87: the MatMulNBits B blob [N, K/g, g*bits/8] from a [K, N] array of 4- / 8-bit values stored
 *     one per byte (what `_rtn_quantize` / `_gptq_quantize` / `_hqq_quantize` return): even k in the low nibble.  The RTN
 *     and HQQ entry points write the blob directly with OQ_LAYOUT_NBITS; this is for integers that already exist. */
int32_t oq_pack_matmul_nbits(const void* q, int64_t K, int64_t N, int64_t group_size, int32_t bits, uint8_t* out,
                             void* stream);
/* N3  core/_pack.py:8-22: flat nibble packing (element 2j -> low nibble of byte j) of `count` 4-bit
 *     values stored one per byte; out has (count+1)/2 bytes. */
int32_t oq_pack_nibbles(const void* values, int64_t count, uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OQ_HIP_H */
