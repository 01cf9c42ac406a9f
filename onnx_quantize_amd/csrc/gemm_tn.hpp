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
    // optional second, transposed copy of the result: Ct[n * ldct + m] = C[m, n] (+ z * stride_ct)
    float* Ct = nullptr;
    int64_t ldct = 0, stride_ct = 0;
    // triangular operands: skip the k-range that only multiplies structural zeros
    //   k_from_n: B[k][n] == 0 for k < n (B lower triangular)  -> a tile starts at k = its first column
    //   k_to_m:   At[k][m] == 0 for k > m (At upper triangular) -> a tile stops behind its last row
    int32_t k_from_n = 0, k_to_m = 0;
};

int32_t launch_gemm_tn(const GemmTN& g, hipStream_t s);

// Symmetric rank-k update C = beta*C + alpha*(s*X)^T (s*X) for tall X [T, K] (the GPTQ Hessian).  When the
// upper-triangular tile count cannot fill the chip (K = 4096: 528 tiles for 512 block slots -> a 2-round
// tail), the T dimension is split into `splits` slices whose partial tiles go to `slab`
// ([splits][K][K] fp32) and a second kernel sums them in a fixed order (deterministic), applies beta and
// writes both triangles with coalesced stores.  slab may be null (splits forced to 1).
size_t syrk_slab_bytes(int64_t T, int64_t K);
int32_t launch_syrk_tn(const float* X, int64_t T, int64_t K, int64_t ldx, float scale_x, float alpha, float beta, float* C,
                       void* slab, size_t slab_bytes, hipStream_t s);

}  // namespace oq
