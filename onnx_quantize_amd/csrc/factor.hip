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
// blockIdx.y = matrix of the batch.  fix_dead: a zero diagonal entry becomes 1 on the way (gptq.py:119-120, for callers
// that did not run oq_gptq_prepare_f32 on this H).
__global__ void reverse_copy_kernel(const float* src, int64_t K, float* dst, int64_t src_stride, int64_t dst_stride, int fix_dead) {
    const int64_t r = blockIdx.x;
    src += static_cast<int64_t>(blockIdx.y) * src_stride;
    dst += static_cast<int64_t>(blockIdx.y) * dst_stride;
    for (int64_t c = threadIdx.x; c < K; c += blockDim.x) {
        float v = src[(K - 1 - r) * K + (K - 1 - c)];
        if (fix_dead && c == r && v == 0.0f) v = 1.0f;
        dst[r * K + c] = v;
    }
}

// gptq.py:135-138: damp = percdamp * mean(diag(H)); H[diag] += damp.  One block.
__global__ __launch_bounds__(1024) void damp_kernel(float* P, int64_t K, float percdamp, int64_t stride) {
    __shared__ float s_part[16];
    __shared__ float s_damp;
    P += static_cast<int64_t>(blockIdx.x) * stride;
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

// ---- 128 x 128 diagonal block in one workgroup ------------------------------------------------------------
// Recursive blocking by 32: a 32 x 32 sub-block is factored and inverted by ONE wave out of registers (lane i
// holds row i; the rows / columns another lane needs arrive by v_readlane, no LDS round trip, no barrier); the
// panel below it, the trailing update and the off-diagonal blocks of the inverse are dense LDS products spread
// over all 512 threads (2 x 2 outputs per thread).  The sequential chain is 4 x (two 32-step register loops)
// instead of 128 barrier-separated columns plus a 127-step substitution per thread.
constexpr int kSB = 32;
constexpr int kDiagThreads = 512;   // 8 waves: the dense LDS phases are latency-bound with one wave per SIMD (1024 threads cap the VGPRs at 128 and spill the register loops: same time)

__device__ __forceinline__ float lane_value(float v, int src_lane) {   // src_lane is a compile-time constant below
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// In-place Cholesky of the 32 x 32 block whose row `lane & 31` is a[0..31]; on return a[k], k <= row, = L[row][k].
// Returns 0 or the 1-based index of the first non-positive pivot (which is replaced by 1 like chol_diag of old).
__device__ __forceinline__ int chol32_regs(float (&a)[kSB], int row) {
    int bad = 0;
#pragma unroll
    for (int j = 0; j < kSB; ++j) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const float ljk = lane_value(a[k], j);
            if (k & 1) s1 = fmaf(a[k], ljk, s1);
            else s0 = fmaf(a[k], ljk, s0);
        }
        const float s = a[j] - (s0 + s1);
        float piv = lane_value(s, j);
        if (!(piv > 0.0f)) {   // also NaN: LAPACK spotrf's "leading minor not positive definite"
            if (bad == 0) bad = j + 1;
            piv = 1.0f;
        }
        const float d = sqrtf(piv);
        a[j] = row == j ? d : s / d;   // rows above j hold garbage in column j: the upper triangle is never read
    }
    return bad;
}

// m[i] = inv(L)[i][col] for the lane's column `col`, from the rows of L held by the lanes (a[] as left by chol32_regs).
__device__ __forceinline__ void inv32_regs(const float (&a)[kSB], float (&m)[kSB], int col) {
#pragma unroll
    for (int i = 0; i < kSB; ++i) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const float lik = lane_value(a[k], i);
            if (k & 1) s1 = fmaf(lik, m[k], s1);   // m[k] == 0 above the column's diagonal entry
            else s0 = fmaf(lik, m[k], s0);
        }
        const float lii = lane_value(a[i], i);
        m[i] = i >= col ? ((i == col ? 1.0f : 0.0f) - (s0 + s1)) / lii : 0.0f;
    }
}

// out(r, c) = sum_{k < kd} fa(r, k) * fb(k, c) for r < mr, c < nc (mr, nc even), 2 x 2 outputs per thread and step.
template <class FA, class FB, class FS>
__device__ __forceinline__ void lds_product(int mr, int nc, int kd, FA fa, FB fb, FS store) {
    const int tc = nc >> 1;
    const int tiles = (mr >> 1) * tc;
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) {
        const int r = (t / tc) * 2, c = (t % tc) * 2;
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 4
        for (int k = 0; k < kd; ++k) {
            const float p0 = fa(r, k), p1 = fa(r + 1, k), q0 = fb(k, c), q1 = fb(k, c + 1);
            a00 = fmaf(p0, q0, a00); a01 = fmaf(p0, q1, a01);
            a10 = fmaf(p1, q0, a10); a11 = fmaf(p1, q1, a11);
        }
        store(r, c, a00); store(r, c + 1, a01); store(r + 1, c, a10); store(r + 1, c + 1, a11);
    }
}

// Diagonal block kb: Cholesky of the n x n block (n <= 128; padded with the identity to 128) and its inverse.
//   Lt   [K, K]: Lt[k][i] = L[i][k]  (upper triangular = L^T), diag block written here
//   Dinv [nb][128][128]: Dinv[kb][k][c] = inv(L_kk)[c][k]  (transposed, zero above the diagonal of the inverse)
//   info: first non-positive pivot (1-based, in reversed index space), 0 if none so far
//   blockIdx.x = matrix of the batch (P, Lt advance by `stride`, Dinv by `dinv_stride` floats, info by one)
__global__ __launch_bounds__(kDiagThreads) void chol_diag_kernel(const float* P, int64_t K, int64_t kb, float* Lt, float* Dinv,
                                                        int32_t* info, int64_t stride, int64_t dinv_stride) {
    extern __shared__ float lds[];
    P += static_cast<int64_t>(blockIdx.x) * stride;
    Lt += static_cast<int64_t>(blockIdx.x) * stride;
    Dinv += static_cast<int64_t>(blockIdx.x) * dinv_stride;
    info += blockIdx.x;
    float (*A)[kLd] = reinterpret_cast<float (*)[kLd]>(lds);                    // the block; lower triangle becomes L
    float (*M)[kLd] = reinterpret_cast<float (*)[kLd]>(lds + kNB * kLd);        // inv(L), lower triangular
    float (*S)[kLd] = reinterpret_cast<float (*)[kLd]>(lds + 2 * kNB * kLd);    // [32][129] scratch of the inverse
    const int64_t o = kb * kNB;
    const int n = static_cast<int>((K - o) < kNB ? (K - o) : kNB);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    // the block arrives with 16 loads in flight per thread (a load -> LDS store chain per element would pay the
    // memory latency 64 times); clamped coordinates, identity padding applied on the way into LDS
    for (int b0 = 0; b0 < kNB * kNB; b0 += 16 * kDiagThreads) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = b0 + u * kDiagThreads + t;
            const int r = idx / kNB, c = idx - r * kNB;
            x[u] = P[(o + (r < n ? r : n - 1)) * K + o + (c < n ? c : n - 1)];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = b0 + u * kDiagThreads + t;
            const int r = idx / kNB, c = idx - r * kNB;
            A[r][c] = (r < n && c < n) ? x[u] : (r == c ? 1.0f : 0.0f);
            M[r][c] = 0.0f;
        }
    }
    __syncthreads();

    for (int jb = 0; jb < kNB / kSB; ++jb) {
        const int so = jb * kSB;
        if (wave == 0) {   // 32 x 32 sub-block: factor + inverse out of registers
            const int i = lane & 31;
            float a[kSB], m[kSB];
#pragma unroll
            for (int k = 0; k < kSB; ++k) a[k] = A[so + i][so + k];
            const int bad = chol32_regs(a, i);
            inv32_regs(a, m, i);
            if (lane < kSB) {
#pragma unroll
                for (int k = 0; k < kSB; ++k) {
                    if (k <= i) A[so + i][so + k] = a[k];
                    M[so + k][so + i] = m[k];          // column i of the inverse (zeros above the diagonal)
                }
            }
            if (lane == 0 && bad != 0 && so + bad <= n && *info == 0) *info = static_cast<int32_t>(o + so + bad);
        }
        __syncthreads();
        const int below = kNB - so - kSB;   // rows under the sub-block
        if (below > 0) {
            const int ro = so + kSB;
            // panel: L21 = A21 * inv(L11)^T, staged in the still unused part of M under the sub-block (A21 is an input
            // of every output of its row)
            lds_product(below, kSB, kSB,
                        [&](int r, int k) { return A[ro + r][so + k]; },
                        [&](int k, int c) { return M[so + c][so + k]; },
                        [&](int r, int c, float v) { M[ro + r][so + c] = v; });
            __syncthreads();
            for (int idx = t; idx < below * kSB; idx += blockDim.x) {
                const int r = idx / kSB, c = idx - r * kSB;
                A[ro + r][so + c] = M[ro + r][so + c];
            }
            __syncthreads();
            // trailing update (full square: both triangles stay current): A22 -= L21 * L21^T
            lds_product(below, below, kSB,
                        [&](int r, int k) { return A[ro + r][so + k]; },
                        [&](int k, int c) { return A[ro + c][so + k]; },
                        [&](int r, int c, float v) { A[ro + r][ro + c] -= v; });
            __syncthreads();
        }
    }
    // off-diagonal blocks of the inverse by block rows:  X_i,<i = -inv(L_ii) * (L_i,<i * X_<i,<i)
    for (int ib = 1; ib < kNB / kSB; ++ib) {
        const int ro = ib * kSB;
        lds_product(kSB, ro, ro,
                    [&](int r, int k) { return A[ro + r][k]; },
                    [&](int k, int c) { return k >= c ? M[k][c] : 0.0f; },   // staging leftovers sit only below the diagonal blocks
                    [&](int r, int c, float v) { S[r][c] = v; });
        __syncthreads();
        lds_product(kSB, ro, kSB,
                    [&](int r, int k) { return M[ro + r][ro + k]; },
                    [&](int k, int c) { return S[k][c]; },
                    [&](int r, int c, float v) { M[ro + r][c] = -v; });
        __syncthreads();
    }
    for (int idx = t; idx < kNB * kNB; idx += blockDim.x) {
        const int k = idx / kNB, c = idx - k * kNB;
        // L^T diag block (zero below the diagonal of Lt) and the transposed inverse
        if (k < n && c < n) Lt[(o + k) * K + o + c] = (c >= k) ? A[c][k] : 0.0f;
        Dinv[(kb * kNB + k) * kNB + c] = (k < n && c < n && c >= k) ? M[c][k] : 0.0f;
    }
}

// Diagonal blocks of X = L'^-1 (lower) and of Y = X^T (upper) from Dinv[kb][k][c] = inv(L_kk)[c][k]; one block per kb.
__global__ void place_diag_inverse_kernel(const float* Dinv, int64_t K, float* X, float* Y, int64_t stride, int64_t dinv_stride) {
    const int64_t kb = blockIdx.x, o = kb * kNB;
    Dinv += static_cast<int64_t>(blockIdx.y) * dinv_stride;
    X += static_cast<int64_t>(blockIdx.y) * stride;
    Y += static_cast<int64_t>(blockIdx.y) * stride;
    const int n = static_cast<int>((K - o) < kNB ? (K - o) : kNB);
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int k = idx / n, c = idx - k * n;
        const float v = Dinv[(kb * kNB + k) * kNB + c];
        Y[(o + k) * K + o + c] = v;
        X[(o + c) * K + o + k] = v;
    }
}

// U[a][b] = X[K-1-a][K-1-b] when the factorisation succeeded, identity otherwise (gptq.py:143-150).
__global__ void finish_factor_kernel(const float* X, int64_t K, const int32_t* info, float* U, int64_t x_stride, int64_t u_stride) {
    const int64_t r = blockIdx.x;
    X += static_cast<int64_t>(blockIdx.y) * x_stride;
    U += static_cast<int64_t>(blockIdx.y) * u_stride;
    const bool ok = info[blockIdx.y] == 0;
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
    if (!actorder || !matrix_ok(K, N, N) || K > kMaxHessianWidth) return 256;
    return align256(static_cast<size_t>(K) * N * 4) + align256(static_cast<size_t>(K) * K * 4) + 256;
}

int32_t oq_gptq_prepare_f32(float* W, int64_t K, int64_t N, float* H, int32_t actorder, int32_t* perm_out, void* workspace,
                            size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(W && H && matrix_ok(K, N, N) && K <= kMaxHessianWidth, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_prepare_f32: bad argument");
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

// Workspace of a batch: [P x count][Lt x count][X x count][Y x count][Dinv x count] -- P is reversed, damped and factored in
// place (later the S scratch of the inverse), Y = X^T, Dinv the inverted diagonal blocks.  One region per role, so the
// matrices of a role are a strided batch (stride kk) and X, Y are cleared by one memset.
static size_t factor_matrix_bytes(int64_t K) { return align256(static_cast<size_t>(K) * K * 4); }
static size_t factor_dinv_bytes(int64_t K) { return align256(static_cast<size_t>(ceil_div(K, kNB)) * kNB * kNB * 4); }

// levels of the recursive-doubling inverse from this block size on run on the fp16-piece kernels (section 3b of syrk_bf16x3.hip)
constexpr int64_t kPieceInverseMin = 2048;

// room for the operand pieces of the largest such level (0 when no level qualifies)
static size_t inverse_pieces_bytes(int64_t K, int64_t count) {
    size_t most = 0;
    for (int64_t b = kNB; b < K; b *= 2) {
        if (b < kPieceInverseMin) continue;
        const int64_t npairs = (K - b + 2 * b - 1) / (2 * b), full = K / (2 * b);
        if (full > 0) most = std::max(most, inverse_level_f16x3_bytes(b, b, full * count));
        if (npairs > full) most = std::max(most, inverse_level_f16x3_bytes(b, K - (full * 2 * b + b), (npairs - full) * count));
    }
    return most ? most + 256 : 0;
}

size_t oq_gptq_factor_batched_workspace_bytes(int64_t K, int64_t count) {
    if (K <= 0 || K > kMaxHessianWidth || count <= 0 || count > 65535) return 0;
    return static_cast<size_t>(count) * (4 * factor_matrix_bytes(K) + factor_dinv_bytes(K)) + inverse_pieces_bytes(K, count) + 256;
}

size_t oq_gptq_factor_workspace_bytes(int64_t K) { return oq_gptq_factor_batched_workspace_bytes(K, 1); }

// `count` independent K x K matrices in lock-step: every launch of the chain (diagonal block, panel, strip, deferred
// square, the levels of the inverse) carries all of them -- blockIdx of the small kernels, the outer batch of the TN
// GEMM -- so a batch costs the LATENCY of one chain (nb sequential diagonal blocks) and the GEMM work of `count`.
static int32_t factor_batched(const float* H, int64_t K, int64_t h_stride, int64_t count, float percdamp, float* U_out, int64_t u_stride,
                              int32_t* info, int32_t fix_dead, int32_t method, void* workspace, size_t workspace_bytes, hipStream_t s,
                              const char* who) {
    OQ_REQUIRE(method >= OQ_HESSIAN_AUTO && method <= OQ_HESSIAN_F16X3, OQ_ERR_INVALID_ARGUMENT, "%s: unknown method %d", who, method);
    OQ_REQUIRE(H && U_out && info && K > 0 && K <= kMaxHessianWidth && count > 0 && count <= 65535, OQ_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
    OQ_REQUIRE(count == 1 || (h_stride >= K * K && u_stride >= K * K), OQ_ERR_INVALID_ARGUMENT, "%s: matrices of the batch overlap", who);
    const size_t need = oq_gptq_factor_batched_workspace_bytes(K, count);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "%s: workspace of %zu bytes needed, %zu given", who, need, workspace_bytes);
    const size_t kk = factor_matrix_bytes(K);
    const int64_t ms = static_cast<int64_t>(kk / 4);                      // floats between the matrices of one role
    const int64_t ds = static_cast<int64_t>(factor_dinv_bytes(K) / 4);
    const int64_t nb = ceil_div(K, kNB);
    const uint32_t cnt = static_cast<uint32_t>(count);
    char* base = static_cast<char*>(workspace);
    float* P = reinterpret_cast<float*>(base);
    float* Lt = reinterpret_cast<float*>(base + count * kk);
    float* X = reinterpret_cast<float*>(base + 2 * count * kk);
    float* Y = reinterpret_cast<float*>(base + 3 * count * kk);
    float* Dinv = reinterpret_cast<float*>(base + 4 * count * kk);
    unsigned char* inv_pieces = reinterpret_cast<unsigned char*>(base + 4 * count * kk + static_cast<size_t>(count) * factor_dinv_bytes(K));
    const size_t inv_pieces_bytes = inverse_pieces_bytes(K, count);
    float* S = P;   // P is dead once the Cholesky loop has finished

    if (hipMemsetAsync(info, 0, sizeof(int32_t) * count, s) != hipSuccess) return fail(OQ_ERR_LAUNCH, "%s: memset failed", who);
    hipLaunchKernelGGL(reverse_copy_kernel, dim3(static_cast<uint32_t>(K), cnt), dim3(256), 0, s, H, K, P, h_stride, ms, fix_dead);
    hipLaunchKernelGGL(damp_kernel, dim3(cnt), dim3(1024), 0, s, P, K, percdamp, ms);
    int32_t st = check_launch("reverse/damp");
    if (st != OQ_OK) return st;

    const size_t diag_lds = (2 * kNB + kSB) * kLd * sizeof(float);
    // once per call, not once per process: the attribute belongs to the current device's copy of the kernel
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(diag_lds)) != hipSuccess)
        return fail(OQ_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS", who, diag_lds);
    auto outer = [&](GemmTN& g, int64_t a, int64_t b, int64_t c, int64_t ct) {
        g.outer = count; g.outer_a = a; g.outer_b = b; g.outer_c = c; g.outer_ct = ct;
    };

    // ---- blocked right-looking Cholesky of P, two levels: inside an outer panel of kOuter rows the 128-row steps
    // update only the panel's own rows (a strip of <= 384 rows x all columns behind); the square behind the panel gets
    // ONE update per outer panel with Kd = kOuter instead of four with Kd = 128 (each a read-modify-write of the whole
    // trailing matrix through 4 stages of MFMA work per tile: 27 TFLOP/s on K = 11008).
    constexpr int64_t kOuter = 4 * kNB;
    constexpr int64_t kPieceUpdateMin = 1024;      // trailing squares narrower than this stay on the fp32 kernel
    // OQ_HESSIAN_F32 keeps the whole GPTQ path on fp32 operands (the reference's arithmetic class); the X region must hold the pieces
    const bool pieces_ok = method != OQ_HESSIAN_F32 &&
                           syrk_f16x3_factor_update_bytes(K, kOuter, count) <= static_cast<size_t>(count) * kk;
    for (int64_t O = 0; O < K; O += kOuter) {
        const int64_t pend = O + kOuter < K ? O + kOuter : K;
        for (int64_t o = O; o < pend; o += kNB) {
            const int64_t kb = o / kNB;
            const int64_t n = (K - o) < kNB ? (K - o) : kNB;
            const int64_t rest = K - o - n;  // columns behind this block
            hipLaunchKernelGGL(chol_diag_kernel, dim3(cnt), dim3(kDiagThreads), diag_lds, s, P, K, kb, Lt, Dinv, info, ms, ds);
            st = check_launch("chol_diag_kernel");
            if (st != OQ_OK) return st;
            if (rest <= 0) break;
            // panel (kept transposed): Lt[o+c][i] = sum_k inv(L_kk)[c][k] * P[o+k][i],  i behind the block
            GemmTN pg;
            pg.At = Dinv + kb * kNB * kNB; pg.lda = kNB; pg.M = n;
            pg.B = P + o * K + o + n; pg.ldb = K; pg.N = rest;
            pg.C = Lt + o * K + o + n; pg.ldc = K;
            pg.Kd = n; pg.alpha = 1.0f; pg.beta = 0.0f; pg.sa = 1.0f; pg.sb = 1.0f; pg.upper_only = 0; pg.mirror = 0;
            outer(pg, ds, ms, ms, 0);
            st = launch_gemm_tn(pg, s);
            if (st != OQ_OK) return st;
            // strip update: rows of the outer panel behind this block x all columns behind it
            const int64_t strip = pend - (o + n);
            if (strip > 0) {
                GemmTN tg;
                tg.At = Lt + o * K + o + n; tg.lda = K; tg.M = strip;
                tg.B = Lt + o * K + o + n; tg.ldb = K; tg.N = rest;
                tg.C = P + (o + n) * K + o + n; tg.ldc = K;
                tg.Kd = n; tg.alpha = -1.0f; tg.beta = 1.0f; tg.sa = 1.0f; tg.sb = 1.0f; tg.upper_only = 0; tg.mirror = 0;
                outer(tg, ms, ms, ms, 0);
                st = launch_gemm_tn(tg, s);
                if (st != OQ_OK) return st;
            }
        }
        const int64_t rest2 = K - pend;
        if (rest2 >= kPieceUpdateMin && pieces_ok) {
            // the same update on the fp16 matrix cores (two fp16 pieces per operand element, three products: 22-bit operands,
            // fp32 accumulate -- scripts/lab_factor_accuracy.py (lab_factor_precision.py until the pruning of round 4): the factor's error against float64 does not move, 5.8e-8 vs
            // 6.0e-8 at K = 11008): the panel Lt[O:pend, pend:] is a 512-row "batch of activations", the update a Hessian
            // accumulation with alpha = -1, beta = 1 into a block of P.  The X region (not needed before the inverse) holds
            // the pieces.  fp32 kernel: 83 TFLOP/s on the first update of K = 11008 (a read-modify-write of 3.5 GB).
            st = launch_syrk_f16x3_factor_update(Lt, P, ms, count, K, O, pend, X, static_cast<size_t>(count) * kk, s);
            if (st != OQ_OK) return st;
        } else if (rest2 > 0) {
            // deferred update of the square behind the outer panel (both triangles stay current):
            // P[i][j] -= sum_{c in panel} Lt[c][i] * Lt[c][j]
            GemmTN tg;
            tg.At = Lt + O * K + pend; tg.lda = K; tg.M = rest2;
            tg.B = tg.At; tg.ldb = K; tg.N = rest2;
            tg.C = P + pend * K + pend; tg.ldc = K;
            tg.Kd = pend - O; tg.alpha = -1.0f; tg.beta = 1.0f; tg.sa = 1.0f; tg.sb = 1.0f; tg.upper_only = 1; tg.mirror = 1;
            outer(tg, ms, ms, ms, 0);
            st = launch_gemm_tn(tg, s);
            if (st != OQ_OK) return st;
        }
    }

    // ---- X = L'^-1 by recursive doubling.  With L = [[L11, 0], [L21, L22]]:  X21 = -X22 * (L21 * X11).  At level b
    // every pair of neighbouring b x b diagonal blocks is one problem of a strided batch (the diagonal stride is
    // 2b * (K + 1)), so a level is two launches with K / 2b * (b / 128)^2 tiles each instead of one 128-row block
    // row (a single tile row) at a time.  Both X and Y = X^T are kept: the TN GEMM wants its left operand k-major,
    // i.e. X22 transposed; the second GEMM writes its result to both.
    if (hipMemsetAsync(X, 0, 2 * count * kk, s) != hipSuccess) return fail(OQ_ERR_LAUNCH, "%s: memset failed", who);   // X and Y start from zero
    hipLaunchKernelGGL(place_diag_inverse_kernel, dim3(static_cast<uint32_t>(nb), cnt), dim3(256), 0, s, Dinv, K, X, Y, ms, ds);
    st = check_launch("place_diag_inverse_kernel");
    if (st != OQ_OK) return st;
    for (int64_t b = kNB; b < K; b *= 2) {
        const int64_t npairs = (K - b + 2 * b - 1) / (2 * b);       // pairs whose second block is not empty
        const int64_t full = (K / (2 * b));                          // pairs with a full second block
        for (int part = 0; part < 2; ++part) {                       // 0: the full pairs (one batch), 1: the ragged last pair
            const int64_t first = part == 0 ? 0 : full;
            const int64_t pairs = part == 0 ? full : npairs - full;
            if (pairs <= 0) continue;
            const int64_t o1 = first * 2 * b, o2 = o1 + b;
            const int64_t b2 = part == 0 ? b : K - o2;               // rows of the second block
            if (b >= kPieceInverseMin && pieces_ok && inv_pieces_bytes > 0) {
                // the two products of a level with 22-bit operands on the fp16 matrix cores; the fp32 kernel runs the top
                // level of K = 11008 at 138 TFLOP/s (88 % of its peak) -- only another instruction makes it faster
                st = launch_inverse_level_f16x3(Lt, X, Y, S, ms, count, K, b, first, pairs, b2, inv_pieces, inv_pieces_bytes, s);
                if (st != OQ_OK) return st;
                continue;
            }
            GemmTN sg;  // S[r][j] = sum_k L21[r][k] * X11[k][j] = sum_k Lt[o1+k][o2+r] * X[o1+k][o1+j]
            sg.At = Lt + o1 * K + o2; sg.lda = K; sg.M = b2;
            sg.B = X + o1 * K + o1; sg.ldb = K; sg.N = b;
            sg.C = S + first * b * b; sg.ldc = b;
            sg.Kd = b; sg.alpha = 1.0f; sg.beta = 0.0f; sg.sa = 1.0f; sg.sb = 1.0f; sg.upper_only = 0; sg.mirror = 0;
            sg.k_from_n = 1;   // X11 is lower triangular
            sg.batch = pairs; sg.stride_a = 2 * b * (K + 1); sg.stride_b = 2 * b * (K + 1); sg.stride_c = b * b;
            outer(sg, ms, ms, ms, 0);
            st = launch_gemm_tn(sg, s);
            if (st != OQ_OK) return st;
            GemmTN xg;  // X21[r][j] = -sum_c X22[r][c] * S[c][j] = -sum_c Y[o2+c][o2+r] * S[c][j];  Y12 = X21^T
            xg.At = Y + o2 * K + o2; xg.lda = K; xg.M = b2;
            xg.B = S + first * b * b; xg.ldb = b; xg.N = b;
            xg.C = X + o2 * K + o1; xg.ldc = K;
            xg.Ct = Y + o1 * K + o2; xg.ldct = K;
            xg.Kd = b2; xg.alpha = -1.0f; xg.beta = 0.0f; xg.sa = 1.0f; xg.sb = 1.0f; xg.upper_only = 0; xg.mirror = 0;
            xg.k_to_m = 1;     // X22^T is upper triangular
            xg.batch = pairs; xg.stride_a = 2 * b * (K + 1); xg.stride_b = b * b; xg.stride_c = 2 * b * (K + 1);
            xg.stride_ct = 2 * b * (K + 1);
            outer(xg, ms, ms, ms, ms);
            st = launch_gemm_tn(xg, s);
            if (st != OQ_OK) return st;
        }
    }
    hipLaunchKernelGGL(finish_factor_kernel, dim3(static_cast<uint32_t>(K), cnt), dim3(256), 0, s, X, K, info, U_out, ms, u_stride);
    return check_launch("finish_factor_kernel");
}

int32_t oq_gptq_factor_f32(const float* H, int64_t K, float percdamp, float* U_out, int32_t* info, int32_t method, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return factor_batched(H, K, 0, 1, percdamp, U_out, 0, info, 0, method, workspace, workspace_bytes, as_stream(stream), "oq_gptq_factor_f32");
}

int32_t oq_gptq_factor_batched_f32(const float* H, int64_t K, int64_t h_stride, int64_t count, float percdamp, int32_t fix_dead,
                                   float* U_out, int64_t u_stride, int32_t* info, int32_t method, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    return factor_batched(H, K, h_stride, count, percdamp, U_out, u_stride, info, fix_dead, method, workspace, workspace_bytes, as_stream(stream),
                          "oq_gptq_factor_batched_f32");
}

}  // extern "C"
