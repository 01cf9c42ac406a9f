// G3 prologue (gptq.py:118-127) and G2 (gptq.py:134-150) for gfx950.
//
// The reference computes  L = chol(H);  Li = inv(L);  U = chol(Li^T Li)^T  -- two Cholesky
// factorisations, a general inverse and a GEMM, 9.3 s of LAPACK at K = 4096 -- to obtain the upper
// factor U of H^-1 = U^T U.  The same (unique, positive-diagonal) U is R^-1 where H = R R^T with R upper
// triangular, and R is the ordinary lower Cholesky factor of the index-reversed matrix:
//     P = J H J,  P = L' L'^T,  R = J L' J,  U = R^-1 = J L'^-1 J.
// So this file does ONE blocked Cholesky and ONE blocked triangular inverse, both with all O(K^3) work in
// the MFMA TN GEMM of gemm_tn.hip (panels are kept transposed, k-major, so every product is a TN GEMM),
// and 128 x 128 diagonal blocks factored / inverted inside one workgroup's LDS.  Results agree with the
// reference to fp32 rounding (different, shorter, operation sequence), not bit for bit.
#include "gemm_tn.hpp"

namespace oq {

constexpr int kNB = 128;   // block size of the factorisation
constexpr int kLd = kNB + 1;  // LDS row pitch (bank-conflict-free column walks)

// ------------------------------------------------------------------------------------ prologue
__global__ void dead_channels_kernel(float* W, int64_t K, int64_t N, float* H) {
    const int64_t k = blockIdx.x;
    if (H[k * K + k] != 0.0f) return;            // gptq.py:119 dead = diag(H) == 0
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) W[k * N + n] = 0.0f;   // :121
    __syncthreads();
    if (threadIdx.x == 0) H[k * K + k] = 1.0f;  // :120
}

// perm = argsort(diag(H))[::-1] (gptq.py:125).  Equal keys: the larger index comes first (a stable
// ascending sort, reversed); NumPy's introsort leaves the order of ties unspecified.
__global__ void rank_desc_kernel(const float* H, int64_t K, int32_t* perm) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const float di = H[i * K + i];
    int32_t rank = 0;
    for (int64_t j = 0; j < K; ++j) {
        const float dj = H[j * K + j];
        rank += (dj > di) || (dj == di && j > i);
    }
    perm[rank] = static_cast<int32_t>(i);
}

__global__ void gather_rows_kernel(const float* src, int64_t cols, const int32_t* perm, float* dst) {
    const int64_t r = blockIdx.x;
    const float* s = src + static_cast<int64_t>(perm[r]) * cols;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) dst[r * cols + c] = s[c];
}

__global__ void gather_sym_kernel(const float* H, int64_t K, const int32_t* perm, float* dst) {
    const int64_t r = blockIdx.x;
    const float* s = H + static_cast<int64_t>(perm[r]) * K;
    for (int64_t c = threadIdx.x; c < K; c += blockDim.x) dst[r * K + c] = s[perm[c]];
}

// ------------------------------------------------------------------------------------ factor
__global__ void reverse_copy_kernel(const float* src, int64_t K, float* dst) {
    const int64_t r = blockIdx.x;
    for (int64_t c = threadIdx.x; c < K; c += blockDim.x) dst[r * K + c] = src[(K - 1 - r) * K + (K - 1 - c)];
}

// gptq.py:135-138: damp = percdamp * mean(diag(H)); H[diag] += damp.  One block.
__global__ __launch_bounds__(1024) void damp_kernel(float* P, int64_t K, float percdamp) {
    __shared__ float s_part[16];
    __shared__ float s_damp;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) acc += P[i * K + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t += s_part[w];
        s_damp = percdamp * (t / static_cast<float>(K));
    }
    __syncthreads();
    const float d = s_damp;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) P[i * K + i] += d;
}

// Diagonal block kb: Cholesky of the n x n block (n <= 128) in LDS, then its triangular inverse.
//   Lt   [K, K]: Lt[k][i] = L[i][k]  (upper triangular = L^T), diag block written here
//   Dinv [nb][128][128]: Dinv[kb][k][c] = inv(L_kk)[c][k]  (transposed, zero above the diagonal of the inverse)
//   info: first non-positive pivot (1-based, in reversed index space), 0 if none so far
__global__ __launch_bounds__(256) void chol_diag_kernel(const float* P, int64_t K, int64_t kb, float* Lt, float* Dinv,
                                                        int32_t* info) {
    extern __shared__ float lds[];
    float (*A)[kLd] = reinterpret_cast<float (*)[kLd]>(lds);
    float (*M)[kLd] = reinterpret_cast<float (*)[kLd]>(lds + kNB * kLd);
    const int64_t o = kb * kNB;
    const int n = static_cast<int>((K - o) < kNB ? (K - o) : kNB);
    const int t = threadIdx.x;
    for (int idx = t; idx < n * n; idx += blockDim.x) {
        const int r = idx / n, c = idx - r * n;
        A[r][c] = P[(o + r) * K + o + c];
    }
    __syncthreads();
    // Left-looking (dot-product) Cholesky: column j is finished from the already final columns 0..j-1,
    //   s_i = A[i][j] - sum_{k<j} L[i][k] * L[j][k],   L[j][j] = sqrt(s_j),   L[i][j] = s_i / L[j][j]  (i > j).
    // One thread per row (row walks are conflict-free with the 129-float pitch, row j is a broadcast);
    // two barriers per column and no trailing-matrix sweep.
    for (int j = 0; j < n; ++j) {
        float s = 0.f;
        const int i = t;
        if (i >= j && i < n) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int k = 0;
            for (; k + 3 < j; k += 4) {
                s0 = fmaf(A[i][k], A[j][k], s0);
                s1 = fmaf(A[i][k + 1], A[j][k + 1], s1);
                s2 = fmaf(A[i][k + 2], A[j][k + 2], s2);
                s3 = fmaf(A[i][k + 3], A[j][k + 3], s3);
            }
            for (; k < j; ++k) s0 = fmaf(A[i][k], A[j][k], s0);
            s = A[i][j] - ((s0 + s1) + (s2 + s3));
            A[i][j] = s;
        }
        __syncthreads();
        float ajj = A[j][j];
        if (!(ajj > 0.0f)) {  // also catches NaN: LAPACK spotrf's "leading minor not positive definite"
            if (t == 0 && *info == 0) *info = static_cast<int32_t>(o + j + 1);
            ajj = 1.0f;
        }
        const float d = sqrtf(ajj);
        __syncthreads();  // everyone has read the pivot before it is overwritten
        if (i > j && i < n) A[i][j] = s / d;
        if (i == j) A[j][j] = d;
        // column j is only read by later columns' dot products, which start after the next barrier
    }
    __syncthreads();
    // inverse of the lower factor, one column per thread (forward substitution)
    for (int idx = t; idx < n * n; idx += blockDim.x) M[idx / n][idx % n] = 0.0f;
    __syncthreads();
    if (t < n) {
        const int c = t;
        M[c][c] = 1.0f / A[c][c];
        for (int i = c + 1; i < n; ++i) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four chains: the dot product is LDS-latency bound otherwise
            int k = c;
            for (; k + 3 < i; k += 4) {
                s0 = fmaf(A[i][k], M[k][c], s0);
                s1 = fmaf(A[i][k + 1], M[k + 1][c], s1);
                s2 = fmaf(A[i][k + 2], M[k + 2][c], s2);
                s3 = fmaf(A[i][k + 3], M[k + 3][c], s3);
            }
            for (; k < i; ++k) s0 = fmaf(A[i][k], M[k][c], s0);
            M[i][c] = -((s0 + s1) + (s2 + s3)) / A[i][i];
        }
    }
    __syncthreads();
    for (int idx = t; idx < kNB * kNB; idx += blockDim.x) {
        const int k = idx / kNB, c = idx - k * kNB;
        // L^T diag block (zero below the diagonal of Lt) and the transposed inverse
        if (k < n && c < n) Lt[(o + k) * K + o + c] = (c >= k) ? A[c][k] : 0.0f;
        Dinv[(kb * kNB + k) * kNB + c] = (k < n && c < n && c >= k) ? M[c][k] : 0.0f;
    }
}

// X diag block (lower triangular inverse block) from Dinv: X[o+c][o+k] = Dinv[kb][k][c]
__global__ void place_diag_inverse_kernel(const float* Dinv, int64_t K, int64_t kb, float* X) {
    const int64_t o = kb * kNB;
    const int n = static_cast<int>((K - o) < kNB ? (K - o) : kNB);
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int c = idx / n, k = idx - c * n;
        X[(o + c) * K + o + k] = Dinv[(kb * kNB + k) * kNB + c];
    }
}

// U[a][b] = X[K-1-a][K-1-b] when the factorisation succeeded, identity otherwise (gptq.py:143-150).
__global__ void finish_factor_kernel(const float* X, int64_t K, const int32_t* info, float* U) {
    const int64_t r = blockIdx.x;
    const bool ok = *info == 0;
    for (int64_t c = threadIdx.x; c < K; c += blockDim.x) {
        float v;
        if (ok) v = (c >= r) ? X[(K - 1 - r) * K + (K - 1 - c)] : 0.0f;
        else v = (c == r) ? 1.0f : 0.0f;
        U[r * K + c] = v;
    }
}

static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace oq

extern "C" {

using namespace oq;

size_t oq_gptq_prepare_workspace_bytes(int64_t K, int64_t N, int32_t actorder) {
    if (!actorder || K <= 0 || N <= 0) return 256;
    return align256(static_cast<size_t>(K) * N * 4) + align256(static_cast<size_t>(K) * K * 4) + 256;
}

int32_t oq_gptq_prepare_f32(float* W, int64_t K, int64_t N, float* H, int32_t actorder, int32_t* perm_out, void* workspace,
                            size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(W && H && K > 0 && N > 0, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_prepare_f32: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(dead_channels_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N, H);
    int32_t st = check_launch("dead_channels_kernel");
    if (st != OQ_OK || !actorder) return st;
    OQ_REQUIRE(perm_out, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_prepare_f32: actorder needs perm_out");
    const size_t need = oq_gptq_prepare_workspace_bytes(K, N, 1);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_gptq_prepare_f32: workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
    float* Wt = static_cast<float*>(workspace);
    float* Ht = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256(static_cast<size_t>(K) * N * 4));
    hipLaunchKernelGGL(rank_desc_kernel, dim3(static_cast<uint32_t>(ceil_div(K, 256))), dim3(256), 0, s, H, K, perm_out);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, N, perm_out, Wt);
    hipLaunchKernelGGL(gather_sym_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, H, K, perm_out, Ht);
    st = check_launch("actorder gather");
    if (st != OQ_OK) return st;
    if (hipMemcpyAsync(W, Wt, static_cast<size_t>(K) * N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(H, Ht, static_cast<size_t>(K) * K * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(OQ_ERR_LAUNCH, "oq_gptq_prepare_f32: device copy failed");
    return OQ_OK;
}

size_t oq_gptq_factor_workspace_bytes(int64_t K) {
    if (K <= 0) return 0;
    const size_t kk = align256(static_cast<size_t>(K) * K * 4);
    const int64_t nb = ceil_div(K, kNB);
    // P (reversed, damped, factored in place), Lt, X, Dinv, S (one block row)
    return 3 * kk + align256(static_cast<size_t>(nb) * kNB * kNB * 4) + align256(static_cast<size_t>(kNB) * K * 4) + 256;
}

int32_t oq_gptq_factor_f32(const float* H, int64_t K, float percdamp, float* U_out, int32_t* info, void* workspace,
                           size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(H && U_out && info && K > 0, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_factor_f32: bad argument");
    const size_t need = oq_gptq_factor_workspace_bytes(K);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_gptq_factor_f32: workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
    hipStream_t s = as_stream(stream);
    const size_t kk = align256(static_cast<size_t>(K) * K * 4);
    const int64_t nb = ceil_div(K, kNB);
    char* base = static_cast<char*>(workspace);
    float* P = reinterpret_cast<float*>(base);
    float* Lt = reinterpret_cast<float*>(base + kk);
    float* X = reinterpret_cast<float*>(base + 2 * kk);
    float* Dinv = reinterpret_cast<float*>(base + 3 * kk);
    float* S = reinterpret_cast<float*>(base + 3 * kk + align256(static_cast<size_t>(nb) * kNB * kNB * 4));

    if (hipMemsetAsync(info, 0, sizeof(int32_t), s) != hipSuccess || hipMemsetAsync(X, 0, static_cast<size_t>(K) * K * 4, s) != hipSuccess)
        return fail(OQ_ERR_LAUNCH, "oq_gptq_factor_f32: memset failed");
    hipLaunchKernelGGL(reverse_copy_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, H, K, P);
    hipLaunchKernelGGL(damp_kernel, dim3(1), dim3(1024), 0, s, P, K, percdamp);
    int32_t st = check_launch("reverse/damp");
    if (st != OQ_OK) return st;

    const size_t diag_lds = 2 * kNB * kLd * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(diag_lds)) != hipSuccess)
            return fail(OQ_ERR_LAUNCH, "oq_gptq_factor_f32: cannot reserve %zu bytes of LDS", diag_lds);
        attr_set = true;
    }

    // ---- blocked right-looking Cholesky of P (both triangles of the trailing matrix are kept current)
    for (int64_t kb = 0; kb < nb; ++kb) {
        const int64_t o = kb * kNB;
        const int64_t n = (K - o) < kNB ? (K - o) : kNB;
        const int64_t rest = K - o - n;  // rows / columns behind this block
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), diag_lds, s, P, K, kb, Lt, Dinv, info);
        st = check_launch("chol_diag_kernel");
        if (st != OQ_OK) return st;
        if (rest <= 0) break;
        // panel (kept transposed): Lt[o+c][i] = sum_k inv(L_kk)[c][k] * P[o+k][i],  i behind the block
        GemmTN pg;
        pg.At = Dinv + kb * kNB * kNB; pg.lda = kNB; pg.M = n;
        pg.B = P + o * K + o + n; pg.ldb = K; pg.N = rest;
        pg.C = Lt + o * K + o + n; pg.ldc = K;
        pg.Kd = n; pg.alpha = 1.0f; pg.beta = 0.0f; pg.sa = 1.0f; pg.sb = 1.0f; pg.upper_only = 0; pg.mirror = 0;
        st = launch_gemm_tn(pg, s);
        if (st != OQ_OK) return st;
        // trailing update: P[i][j] -= sum_c Lt[o+c][i] * Lt[o+c][j]
        GemmTN tg;
        tg.At = Lt + o * K + o + n; tg.lda = K; tg.M = rest;
        tg.B = tg.At; tg.ldb = K; tg.N = rest;
        tg.C = P + (o + n) * K + o + n; tg.ldc = K;
        tg.Kd = n; tg.alpha = -1.0f; tg.beta = 1.0f; tg.sa = 1.0f; tg.sb = 1.0f; tg.upper_only = 1; tg.mirror = 1;
        st = launch_gemm_tn(tg, s);
        if (st != OQ_OK) return st;
    }

    // ---- X = L'^-1 by block rows:  X_ii = inv(L_ii),  X_i,<i = -inv(L_ii) * (L_i,<i * X_<i,<i)
    for (int64_t ib = 0; ib < nb; ++ib) {
        const int64_t o = ib * kNB;
        const int64_t n = (K - o) < kNB ? (K - o) : kNB;
        hipLaunchKernelGGL(place_diag_inverse_kernel, dim3(1), dim3(256), 0, s, Dinv, K, ib, X);
        st = check_launch("place_diag_inverse_kernel");
        if (st != OQ_OK) return st;
        if (ib == 0) continue;
        GemmTN sg;  // S[r][j] = sum_k Lt[k][o+r] * X[k][j],  k, j < o
        sg.At = Lt + o; sg.lda = K; sg.M = n;
        sg.B = X; sg.ldb = K; sg.N = o;
        sg.C = S; sg.ldc = o;
        sg.Kd = o; sg.alpha = 1.0f; sg.beta = 0.0f; sg.sa = 1.0f; sg.sb = 1.0f; sg.upper_only = 0; sg.mirror = 0;
        st = launch_gemm_tn(sg, s);
        if (st != OQ_OK) return st;
        GemmTN xg;  // X[o+r][j] = -sum_c inv(L_ii)[r][c] * S[c][j]
        xg.At = Dinv + ib * kNB * kNB; xg.lda = kNB; xg.M = n;
        xg.B = S; xg.ldb = o; xg.N = o;
        xg.C = X + o * K; xg.ldc = K;
        xg.Kd = n; xg.alpha = -1.0f; xg.beta = 0.0f; xg.sa = 1.0f; xg.sb = 1.0f; xg.upper_only = 0; xg.mirror = 0;
        st = launch_gemm_tn(xg, s);
        if (st != OQ_OK) return st;
    }
    hipLaunchKernelGGL(finish_factor_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, X, K, info, U_out);
    return check_launch("finish_factor_kernel");
}

}  // extern "C"
