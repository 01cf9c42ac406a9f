// fp32 "TN" GEMM on the gfx950 matrix cores, shared by the GPTQ kernels:
//
//     C[m, n] = beta * C[m, n] + alpha * sum_k (sa * At[k, m]) * (sb * B[k, n])
//
// Both operands are stored k-major (row k holds all m / all n), which is exactly the layout of
//   * the Hessian  H += (2/n) X^T X            (At = B = X [T, K],          gptq.py:246-260)
//   * the Cholesky trailing update A -= P^T P  (At = B = panel^T [nb, K])
//   * the GPTQ lazy batch update W -= U_rows^T Err (At = rows of U, B = Err [nb, N], gptq.py:208)
// so no operand is ever transposed in memory.  v_mfma_f32_32x32x2_f32 takes ONE fp32 VGPR per operand
// per lane -- lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] -- i.e. two
// consecutive k-rows of 32 contiguous floats: a k-major LDS tile is read conflict-free with
// ds_read_b32 and needs no swizzle.  Accumulation is exact fp32 (an fmaf chain), like the sgemm the
// reference calls; only the summation order differs.
#pragma once

#include "oq_common.hpp"

namespace oq {

struct GemmTN {
    const float* At;  // [Kd, M], leading dimension lda
    const float* B;   // [Kd, N], leading dimension ldb
    float* C;         // [M, N], leading dimension ldc
    int64_t M, N, Kd, lda, ldb, ldc;
    float alpha, beta;
    float sa, sb;     // operand pre-scales applied on load (fp32 rounding each, like `sqrt(2/n) * inp`)
    int32_t upper_only;  // compute only tiles with tile_n >= tile_m (symmetric result, At == B)
    int32_t mirror;      // with upper_only: also store C[n, m] for off-diagonal tiles
    // strided batch (blockIdx.z): problem z uses At + z * stride_a, B + z * stride_b, C + z * stride_c (elements)
    int64_t batch = 1, stride_a = 0, stride_b = 0, stride_c = 0;
    // second batch level (blockIdx.y, not with T-slice slabs): `outer` independent matrices, each with its own strided batch
    int64_t outer = 1, outer_a = 0, outer_b = 0, outer_c = 0, outer_ct = 0;
    // optional second, transposed copy of the result: Ct[n * ldct + m] = C[m, n] (+ z * stride_ct)
    float* Ct = nullptr;
    int64_t ldct = 0, stride_ct = 0;
    // triangular operands: skip the k-range that only multiplies structural zeros
    //   k_from_n: B[k][n] == 0 for k < n (B lower triangular)  -> a tile starts at k = its first column
    //   k_to_m:   At[k][m] == 0 for k > m (At upper triangular) -> a tile stops behind its last row
    int32_t k_from_n = 0, k_to_m = 0;
};

int32_t launch_gemm_tn(const GemmTN& g, hipStream_t s);

// Upper-triangle tiles of an ntiles x ntiles grid enumerated by 8 x 8 SUPER-TILES (row-major over the upper triangle
// of super-tiles, tiles row-major inside one).  With XCD-contiguous ids (oq_common.hpp::xcd_remap) the tiles resident
// on one XCD then share 8 A-panels and 8 B-panels per k-stage through that XCD's L2, instead of 1 + 64 panels for 64
// neighbours of one tile row -- a quarter of the L2 fill traffic.  Speed only; every tile is visited once.
__device__ __forceinline__ void upper_tile_of(int rem, int ntiles, int& tile_m, int& tile_n) {
    const int ns = (ntiles + 7) >> 3;
    int R = 0, C = 0, nr = 0, nc = 0;
    bool found = false;
    for (R = 0; R < ns && !found; ++R) {
        nr = min(8, ntiles - 8 * R);
        for (C = R; C < ns; ++C) {
            nc = min(8, ntiles - 8 * C);
            const int cnt = C == R ? nr * (nr + 1) / 2 : nr * nc;
            if (rem < cnt) { found = true; break; }
            rem -= cnt;
        }
        if (found) break;
    }
    int r, c;
    if (C == R) {
        r = 0;
        while (rem >= nr - r) { rem -= nr - r; ++r; }
        c = r + rem;
    } else {
        r = rem / nc;
        c = rem - r * nc;
    }
    tile_m = 8 * R + r;
    tile_n = 8 * C + c;
}

// Sums the `splits` partial results slab[z][K][K] (valid where the GEMM's `tile` x `tile` block tiles lie on or above the
// diagonal) in slice order, applies alpha / beta and writes both triangles of C.
// `post_scale` (device, optional): alpha is multiplied by post_scale[1] (the fp16-piece kernel's exact 1 / s^2).
int32_t launch_syrk_reduce(const float* slab, int splits, int64_t K, float alpha, float beta, float* C, int tile, hipStream_t s,
                           const float* post_scale = nullptr);

// The same update on the bf16 matrix cores with fp32-exact operands (syrk_bf16x3.hip): every fp32 element is split
// into three bf16 pieces whose sum is the element exactly; `terms` = 6 (piece products down to 2^-16, the dropped
// ones are below fp32 rounding of a product), 9 (all of them) or 3 (two fp16 pieces of the scaled element, three products).  workspace = the pieces (syrk_bf16x3_pieces_bytes)
// followed by optional T-slice slabs of K x K floats.
size_t syrk_bf16x3_pieces_bytes(int64_t T, int64_t K);
int32_t launch_syrk_bf16x3(const float* X, int64_t T, int64_t K, int64_t ldx, float alpha, float beta, float* C,
                           void* workspace, size_t workspace_bytes, int terms, hipStream_t s);

int32_t syrk_pieces_phases(const float* X, int64_t T, int64_t K, int64_t ldx, float alpha, float beta, float* C, unsigned char* base, float* slab,
                           size_t slab_bytes, int terms, int phases, hipStream_t s);   // phases: 1 = X -> pieces, 2 = pieces -> C, 3 = both

// Many Hessian updates in one launch chain (syrk_bf16x3.hip, section 2c).  items: int64 {X, H, T, K, ldx, n_seen, n_add, 0} each.
size_t syrk_f16x3_many_workspace_bytes(const int64_t* items_host, int64_t count);
int32_t launch_syrk_f16x3_many(const int64_t* items_host, const int64_t* items_device, int64_t count, void* workspace, size_t workspace_bytes,
                               hipStream_t s);

// The blocked Cholesky's deferred trailing update on the same grouped kernels: for m < count,
//   P_m[pend:, pend:] -= Lt_m[O:pend, pend:K]^T Lt_m[O:pend, pend:K]   (both triangles), matrices `ms` floats apart.
// workspace: syrk_f16x3_factor_update_bytes(columns = K - pend, rows = pend - O, count).
size_t syrk_f16x3_factor_update_bytes(int64_t K, int64_t kd_max, int64_t count);
int32_t launch_syrk_f16x3_factor_update(const float* Lt, float* P, int64_t ms, int64_t count, int64_t K, int64_t O, int64_t pend, void* workspace,
                                        size_t workspace_bytes, hipStream_t s);

// One level of the factor's recursive-doubling inverse on the fp16-piece kernels (syrk_bf16x3.hip, section 3b): for every pair
// p < pairs (first block at (first + p) * 2b, b rows; second b2 <= b rows) of every matrix m < count:  X21 = -X22 (L21 X11),
// Y12 = X21^T.  S: scratch of (first + pairs) * b * b floats per matrix.  workspace: inverse_level_f16x3_bytes(b, b2, pairs * count).
size_t inverse_level_f16x3_bytes(int64_t b, int64_t b2, int64_t problems);
int32_t launch_inverse_level_f16x3(const float* Lt, float* X, float* Y, float* S, int64_t ms, int64_t count, int64_t K, int64_t b, int64_t first,
                                   int64_t pairs, int64_t b2, void* workspace, size_t workspace_bytes, hipStream_t s);

// Two-operand GEMM on fp16 pieces (syrk_bf16x3.hip, section 3): C = beta C + alpha A^T B for k-major A [Kd, M], B [Kd, N].
//   make_f16x2_pieces      absmax -> power-of-two scale -> two fp16 pieces of every element, zero-padded to 32 contraction rows
//                          and 256 columns; `pieces` (256-byte aligned, gemm_f16x3_pieces_bytes(Kd, cols)) holds the scale
//                          header and the pieces.  contraction_is_fast_axis: the source is [cols, Kd] row-major (A = X^T).
//   launch_gemm_f16x3      exactly one of C (store: alpha / beta like launch_gemm_tn) and loss_partial (one float per block
//                          = gemm_f16x3_tiles(M, N): the sum of squares of that block's part of A^T B, nothing else written).
//                          hi_pieces_only (loss form only): the product of the first pieces alone -- 11-bit operands, a third of
//                          the matrix work; what the AWQ / clip searches' losses need (awq.hip).
//                          dot_with_c: C [M, N] is READ and loss_partial gets, per block, the sum of (A^T B) o C over its part
//                          (the quadratic form <D, G D> of the searches' Gram route); with b_first_piece_only two products
//                          instead of three: A with both pieces against B's first.
size_t gemm_f16x3_pieces_bytes(int64_t Kd, int64_t cols);
int32_t make_f16x2_pieces(const float* X, int64_t Kd, int64_t cols, int64_t ldx, bool contraction_is_fast_axis, void* pieces, hipStream_t s,
                          bool wide_range = false,    // wide_range: a scale over the whole fp32 exponent range (plain GEMM operands; 1 / s^2 unset)
                          float* row_scales = nullptr);   // [2 cols] (fast-axis sources only): one power-of-two scale per source row instead of one per operand
// For a producer that writes the FIRST (hi) fp16 pieces of a row-major [Kd, cols] operand itself instead of handing a
// fp32 matrix to make_f16x2_pieces* (awq.hip: the quantize-residual kernel, round 5).  The eight fp16 of rows 8c .. 8c + 7 of
// column n (k ascending, each fl16(x * s), round to nearest even) are the 16-byte vector
//     P[(2 c) * gemm_f16x3_padded_cols(cols) + n],   P = pieces + gemm_f16x3_header_bytes();
// columns from `cols` to the padded width and chunks from ceil(Kd / 8) to gemm_f16x3_chunks(Kd) must hold zeros.  The
// producer also writes the header's first three floats [s, 1 / s^2, 1 / s] with s a power of two that maps an upper bound of
// |x| below 2^15 (any bound works: fp16 keeps its 11 bits over 29 binades below that).
size_t gemm_f16x3_header_bytes();
int64_t gemm_f16x3_padded_cols(int64_t cols);
int64_t gemm_f16x3_chunks(int64_t Kd);
int32_t make_f16x2_pieces_from_partials(const float* X, int64_t Kd, int64_t cols, int64_t ldx, const float* absmax_partials, int npart, void* pieces,
                                        hipStream_t s, bool first_pieces_only = false);   // true: the lo plane is left untouched (consumers that read first pieces only)   // [Kd, cols] row-major source whose max |x| is already folded into `npart` device partials
int32_t launch_gemm_f16x3(const void* pieces_a, const void* pieces_b, int64_t M, int64_t N, int64_t Kd, float alpha, float beta, float* C,
                          int64_t ldc, float* loss_partial, hipStream_t s, bool hi_pieces_only = false, bool dot_with_c = false,
                          bool b_first_piece_only = false, const float* row_unscale_a = nullptr);   // row_unscale_a [M]: A was split with per-row scales
int64_t gemm_f16x3_tiles(int64_t M, int64_t N);

}  // namespace oq
