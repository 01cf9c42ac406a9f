// fp32 TN GEMM on v_mfma_f32_32x32x2_f32 (see gemm_tn.hpp) and G1, the GPTQ Hessian accumulate.
#include "gemm_tn.hpp"

#include <cmath>

namespace oq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBT = 32;  // block tile and k-rows per LDS stage
constexpr int kGemmThreads = 256;              // 4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles

// One stage of one operand: kBT rows x 128 floats = 1024 float4, 4 per thread.
struct StageRegs {
    float4 v[4];
};

__device__ __forceinline__ void stage_load(StageRegs& r, const float* __restrict__ P, int64_t ld, int64_t k0, int64_t c0,
                                           int64_t Kd, int64_t cols, float scale, bool vec_ok) {
    const int t = threadIdx.x;
    const int c4 = (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t k = k0 + (t >> 5) + p * 8;
        const int64_t c = c0 + c4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < Kd) {
            const float* src = P + k * ld + c;
            if (vec_ok && c + 3 < cols) {
                x = *reinterpret_cast<const float4*>(src);
            } else {
                if (c < cols) x.x = src[0];
                if (c + 1 < cols) x.y = src[1];
                if (c + 2 < cols) x.z = src[2];
                if (c + 3 < cols) x.w = src[3];
            }
        }
        if (scale != 1.0f) { x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale; }
        r.v[p] = x;
    }
}

__device__ __forceinline__ void stage_store(const StageRegs& r, float (*tile)[kBM]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(&tile[(t >> 5) + p * 8][(t & 31) * 4]) = r.v[p];
}

__global__ __launch_bounds__(kGemmThreads) void gemm_tn_kernel(const GemmTN g, const bool vec_a, const bool vec_b) {
    const int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (g.upper_only && tile_n < tile_m) return;
    __shared__ float sA[2][kBT][kBM];
    __shared__ float sB[2][kBT][kBN];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wave position inside the 2 x 2 grid
    const int64_t m0 = static_cast<int64_t>(tile_m) * kBM, n0 = static_cast<int64_t>(tile_n) * kBN;
    const bool same = g.upper_only && (g.At == g.B) && tile_m == tile_n;  // diagonal tile of a SYRK: one operand

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t nstages = (g.Kd + kBT - 1) / kBT;
    StageRegs ra, rb;
    stage_load(ra, g.At, g.lda, 0, m0, g.Kd, g.M, g.sa, vec_a);
    if (!same) stage_load(rb, g.B, g.ldb, 0, n0, g.Kd, g.N, g.sb, vec_b);
    stage_store(ra, sA[0]);
    if (!same) stage_store(rb, sB[0]);
    __syncthreads();

    const int kl = lane >> 5, cl = lane & 31;
    for (int64_t s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nstages;
        if (more) {
            stage_load(ra, g.At, g.lda, (s + 1) * kBT, m0, g.Kd, g.M, g.sa, vec_a);
            if (!same) stage_load(rb, g.B, g.ldb, (s + 1) * kBT, n0, g.Kd, g.N, g.sb, vec_b);
        }
        float (*tA)[kBM] = sA[buf];
        float (*tB)[kBN] = same ? sA[buf] : sB[buf];
#pragma unroll
        for (int kk = 0; kk < kBT; kk += 2) {
            const float a0 = tA[kk + kl][wm * 64 + cl];
            const float a1 = tA[kk + kl][wm * 64 + 32 + cl];
            const float b0 = tB[kk + kl][wn * 64 + cl];
            const float b1 = tB[kk + kl][wn * 64 + 32 + cl];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            stage_store(ra, sA[buf ^ 1]);
            if (!same) stage_store(rb, sB[buf ^ 1]);
        }
        __syncthreads();
    }

    // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
    const bool mirror = g.mirror && g.upper_only && tile_m != tile_n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int64_t col = n0 + wn * 64 + j * 32 + cl;
                if (row < g.M && col < g.N) {
                    float v = g.alpha * acc[i][j][e];
                    if (g.beta != 0.0f) v = g.beta * g.C[row * g.ldc + col] + v;
                    g.C[row * g.ldc + col] = v;
                    if (mirror) g.C[col * g.ldc + row] = v;
                }
            }
}

int32_t launch_gemm_tn(const GemmTN& g, hipStream_t s) {
    OQ_REQUIRE(g.At && g.B && g.C && g.M > 0 && g.N > 0 && g.Kd >= 0, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad argument");
    OQ_REQUIRE(g.lda >= g.M && g.ldb >= g.N && g.ldc >= g.N, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad leading dimension");
    const bool vec_a = (g.lda % 4 == 0) && (reinterpret_cast<uintptr_t>(g.At) & 15u) == 0;
    const bool vec_b = (g.ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(g.B) & 15u) == 0;
    const dim3 grid(static_cast<uint32_t>(ceil_div(g.N, kBN)), static_cast<uint32_t>(ceil_div(g.M, kBM)));
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(kGemmThreads), 0, s, g, vec_a, vec_b);
    return check_launch("gemm_tn_kernel");
}

}  // namespace oq

extern "C" {

using namespace oq;

// G1  gptq.py:246-260.
int32_t oq_hessian_accumulate_f32(const float* X, int64_t T, int64_t K, int64_t ldx, int64_t n_seen, int64_t n_add, float* H,
                                  void* stream) {
    OQ_REQUIRE(X && H && T > 0 && K > 0 && ldx >= K, OQ_ERR_INVALID_ARGUMENT, "oq_hessian_accumulate_f32: bad argument");
    OQ_REQUIRE(n_seen >= 0 && n_add > 0, OQ_ERR_INVALID_ARGUMENT, "oq_hessian_accumulate_f32: bad sample counts %lld + %lld",
               (long long)n_seen, (long long)n_add);
    const int64_t n_total = n_seen + n_add;
    GemmTN g;
    g.At = X; g.B = X; g.C = H;
    g.M = K; g.N = K; g.Kd = T; g.lda = ldx; g.ldb = ldx; g.ldc = K;
    g.alpha = 1.0f;
    // gptq.py:254  H *= num_samples / (num_samples + num_added): a Python float applied to an fp32 array
    g.beta = static_cast<float>(static_cast<double>(n_seen) / static_cast<double>(n_total));
    // gptq.py:257  inp = math.sqrt(2 / num_samples) * inp  (double evaluated, weak scalar -> fp32 multiply)
    const float sx = static_cast<float>(std::sqrt(2.0 / static_cast<double>(n_total)));
    g.sa = sx; g.sb = sx;
    g.upper_only = 1;
    g.mirror = 1;
    if (n_seen == 0) {
        // beta == 0 but H must still be defined for the kernel's "beta != 0" shortcut: the first call of the
        // reference starts from zeros (gptq.py:304), so nothing is read.
        g.beta = 0.0f;
    }
    return launch_gemm_tn(g, as_stream(stream));
}

}  // extern "C"
