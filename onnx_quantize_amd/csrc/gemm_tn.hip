// fp32 TN GEMM on v_mfma_f32_32x32x2_f32 (see gemm_tn.hpp) and G1, the GPTQ Hessian accumulate.
#include "gemm_tn.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace oq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBT = 32;  // block tile and k-rows per LDS stage
constexpr int kGemmThreads = 256;              // 4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles

// One stage of one operand: kBT rows x 128 floats = 1024 float4, 4 per thread.
// Loads are UNCONDITIONAL (clamped coordinates) and their results are not touched until stage_store: a
// predicated or immediately-consumed load puts exec branches and vmcnt(0) waits in front of the MFMA loop
// (cdna_hip_programming.md section 5, trap (c)) and the whole global latency is exposed every stage.
struct StageRegs {
    float4 v[4];
    uint32_t ok;   // 4 bits per load: which of the 4 floats are inside the matrix
};

// Piece p (0..3) of a stage: rows k0 + (t >> 5) + 8 p, four floats per thread.  Loads are UNCONDITIONAL (clamped
// coordinates, branch-free): a predicated load puts an exec branch and a vmcnt(0) in front of the MFMA stream
// (cdna_hip_programming.md section 5, trap (c)).  The validity bits of the four floats go to bits 4p..4p+3 of r.ok.
template <bool VEC>
__device__ __forceinline__ void load_piece(StageRegs& r, int p, const float* __restrict__ P, int64_t ld, int64_t k0, int64_t c0,
                                           int64_t Kd, int64_t cols) {
    const int t = threadIdx.x;
    const int64_t c = c0 + (t & 31) * 4;
    const int64_t k = k0 + (t >> 5) + p * 8;
    const int64_t kc = k < Kd ? k : Kd - 1;
    uint32_t ok;
    if constexpr (VEC) {   // cols % 4 == 0, ld % 4 == 0, 16-byte aligned base
        const int64_t cc = c < cols ? c : cols - 4;
        r.v[p] = *reinterpret_cast<const float4*>(P + kc * ld + cc);
        ok = (k < Kd && c < cols) ? 0xfu : 0u;
    } else {
        const float* row = P + kc * ld;
        const int64_t last = cols - 1;
        r.v[p].x = row[c < cols ? c : last];
        r.v[p].y = row[c + 1 < cols ? c + 1 : last];
        r.v[p].z = row[c + 2 < cols ? c + 2 : last];
        r.v[p].w = row[c + 3 < cols ? c + 3 : last];
        ok = k < Kd ? ((c < cols ? 1u : 0u) | (c + 1 < cols ? 2u : 0u) | (c + 2 < cols ? 4u : 0u) | (c + 3 < cols ? 8u : 0u)) : 0u;
    }
    r.ok = (r.ok & ~(0xfu << (4 * p))) | (ok << (4 * p));
}

// Zero-fill of the out-of-range elements and the operand pre-scale happen on the way into LDS.
__device__ __forceinline__ void store_piece(const StageRegs& r, int p, float (*tile)[kBM], float scale) {
    const int t = threadIdx.x;
    float4 x = r.v[p];
    // pins the first use of the loaded registers HERE: without it instruction selection hoists the scale multiply to
    // right behind the load (with a vmcnt(0) in front of it) and every inlined load stalls the MFMA stream
    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
    const uint32_t m = r.ok >> (4 * p);
    x.x = (m & 1u) ? x.x * scale : 0.f; x.y = (m & 2u) ? x.y * scale : 0.f;
    x.z = (m & 4u) ? x.z * scale : 0.f; x.w = (m & 8u) ? x.w * scale : 0.f;
    *reinterpret_cast<float4*>(&tile[(t >> 5) + p * 8][(t & 31) * 4]) = x;
}

__device__ __forceinline__ void store_piece_plain(const StageRegs& r, int p, float (*tile)[kBM], float scale) {
    const int t = threadIdx.x;
    float4 x = r.v[p];
    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));   // see store_piece
    x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;             // x * 1.0f == x bit for bit
    *reinterpret_cast<float4*>(&tile[(t >> 5) + p * 8][(t & 31) * 4]) = x;
}

template <bool VEC>
__device__ __forceinline__ void stage_load(StageRegs& r, const float* __restrict__ P, int64_t ld, int64_t k0, int64_t c0,
                                           int64_t Kd, int64_t cols) {
    r.ok = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) load_piece<VEC>(r, p, P, ld, k0, c0, Kd, cols);
}

__device__ __forceinline__ void stage_store(const StageRegs& r, float (*tile)[kBM], float scale) {
#pragma unroll
    for (int p = 0; p < 4; ++p) store_piece(r, p, tile, scale);
}

// Partial-slab mode (slab != null): blockIdx.y = T-slice; the raw accumulator tile is stored to
// slab[slice][m][n] and alpha / beta / mirror are left to syrk_reduce_kernel.
template <bool VA, bool VB>
__global__ __launch_bounds__(kGemmThreads) void gemm_tn_kernel(const GemmTN g_in, float* slab, const int64_t k_per_slice, const int ntiles_n) {
    GemmTN g = g_in;
    if (g.batch > 1) {
        const int64_t z = blockIdx.z;
        g.At += z * g.stride_a; g.B += z * g.stride_b; g.C += z * g.stride_c;
        if (g.Ct) g.Ct += z * g.stride_ct;
    }
    if (g.outer > 1) {
        const int64_t y = blockIdx.y;
        g.At += y * g.outer_a; g.B += y * g.outer_b; g.C += y * g.outer_c;
        if (g.Ct) g.Ct += y * g.outer_ct;
    }
    int tile_m, tile_n;
    if (g.upper_only) {
        // Upper-triangle tiles enumerated by 8 x 8 SUPER-TILES (row-major over the upper triangle of super-tiles, tiles
        // row-major inside one) and XCD-contiguous ids (oq_common.hpp::xcd_remap): the ~64 tiles resident on one XCD
        // then share 8 A-panels and 8 B-panels per k-stage through that XCD's L2, instead of 1 + 64 panels for 64
        // neighbours of one tile row -- a quarter of the L2 fill traffic.  Speed only; every tile is visited once.
        upper_tile_of(static_cast<int>(xcd_remap(blockIdx.x, gridDim.x)), ntiles_n, tile_m, tile_n);
    } else {
        tile_m = blockIdx.x / ntiles_n;
        tile_n = blockIdx.x - tile_m * ntiles_n;
    }
    int64_t k_begin = slab ? static_cast<int64_t>(blockIdx.y) * k_per_slice : 0;
    int64_t k_end = slab ? (k_begin + k_per_slice < g.Kd ? k_begin + k_per_slice : g.Kd) : g.Kd;
    __shared__ float sA[2][kBT][kBM];
    __shared__ float sB[2][kBT][kBN];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wave position inside the 2 x 2 grid
    const int64_t m0 = static_cast<int64_t>(tile_m) * kBM, n0 = static_cast<int64_t>(tile_n) * kBN;
    if (g.k_from_n && n0 > k_begin) k_begin = n0 < k_end ? n0 : k_end;          // kBN is a multiple of kBT
    if (g.k_to_m && m0 + kBM < k_end) k_end = m0 + kBM > k_begin ? m0 + kBM : k_begin;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t nstages = (k_end - k_begin + kBT - 1) / kBT;
    StageRegs ra, rb;
    stage_load<VA>(ra, g.At, g.lda, k_begin, m0, k_end, g.M);
    stage_load<VB>(rb, g.B, g.ldb, k_begin, n0, k_end, g.N);
    stage_store(ra, sA[0], g.sa);
    stage_store(rb, sB[0], g.sb);
    __syncthreads();

    const int kl = lane >> 5, cl = lane & 31;
    const uint32_t off_a = static_cast<uint32_t>((threadIdx.x >> 5) * g.lda + (threadIdx.x & 31) * 4) * 4u;
    const uint32_t off_b = static_cast<uint32_t>((threadIdx.x >> 5) * g.ldb + (threadIdx.x & 31) * 4) * 4u;
    const char* base_a = reinterpret_cast<const char*>(g.At + k_begin * g.lda + m0);
    const char* base_b = reinterpret_cast<const char*>(g.B + k_begin * g.ldb + n0);
    const int64_t stage_a = kBT * g.lda * 4, stage_b = kBT * g.ldb * 4, rows8_a = 8 * g.lda * 4, rows8_b = 8 * g.ldb * 4;
    // One stage = 16 groups of 4 MFMAs (k-pairs 0, 2, ..., 30), two operand register sets alternating.  A wave issues
    // in order and an MFMA occupies the matrix pipe for 64 cycles, so whatever sits BETWEEN two MFMAs in program
    // order issues for free while the first one executes, and whatever sits in a lump outside the MFMA stream leaves
    // the pipe idle (measured with loads / LDS refill in lumps around the stream: 62 % pipe use with one wave per
    // SIMD, 73 % with two -- the two waves of a SIMD reach their lumps together).  sched_barrier(0) pins the order:
    //   after the 4th MFMA of a group    refill its operand set (consumed two groups later: lgkmcnt(2) suffices)
    //   NEXT, groups 0-2, slots 1-3      the 8 clamped global loads of the next stage, one piece per slot
    //   NEXT, groups 12-15, slots 1-2    mask + scale + ds_write of the 8 pieces into the other LDS buffer
    // The stream itself is branch-free (a branch inside it makes the waitcnt pass guard the inlined loads with
    // vmcnt(0)); the last stage of a tile runs the variant without fillers.  The diagonal tiles of a SYRK load their
    // (identical) second operand like every other tile.
    auto stage_body = [&](auto next_tag, auto plain_tag, int64_t s) {
        constexpr bool NEXT = decltype(next_tag)::value;
        constexpr bool PLAIN = decltype(plain_tag)::value;   // every element of the next stage lies inside both operands
        const int buf = s & 1;
        float (*tA)[kBM] = sA[buf];
        float (*tB)[kBN] = sB[buf];
        auto rd = [&](int k, float& x0, float& x1, float& y0, float& y1) {
            // a wave owns rows / columns {w*32 .. w*32+31} and {64 + w*32 ..} of the block tile: its two operand halves
            // sit 256 B apart in LDS, so each pair is ONE ds_read2st64_b32 with immediate offsets for every k-pair
            x0 = tA[k + kl][wm * 32 + cl];
            x1 = tA[k + kl][wm * 32 + 64 + cl];
            y0 = tB[k + kl][wn * 32 + cl];
            y1 = tB[k + kl][wn * 32 + 64 + cl];
        };
        float xa[2][2], yb[2][2];   // [set][half]
        rd(0, xa[0][0], xa[0][1], yb[0][0], yb[0][1]);
        rd(2, xa[1][0], xa[1][1], yb[1][0], yb[1][1]);
        const int64_t kn = k_begin + (s + 1) * kBT;
        auto filler = [&](int grp, int slot) {
            if constexpr (NEXT) {
                if (grp < 3) {
                    const int q = grp * 3 + (slot - 1);
                    if constexpr (PLAIN) {   // scalar stage base + constant 32-bit thread offset: no address arithmetic, no masks
                        if (q < 4) ra.v[q] = *reinterpret_cast<const float4*>(base_a + (s + 1) * stage_a + q * rows8_a + off_a);
                        else if (q < 8) rb.v[q - 4] = *reinterpret_cast<const float4*>(base_b + (s + 1) * stage_b + (q - 4) * rows8_b + off_b);
                    } else {
                        if (q < 4) load_piece<VA>(ra, q, g.At, g.lda, kn, m0, k_end, g.M);
                        else if (q < 8) load_piece<VB>(rb, q - 4, g.B, g.ldb, kn, n0, k_end, g.N);
                    }
                } else if (grp >= 12 && slot <= 2) {
                    const int q = (grp - 12) * 2 + (slot - 1);   // 0..7
                    if constexpr (PLAIN) {
                        if (q < 4) store_piece_plain(ra, q, sA[buf ^ 1], g.sa);
                        else store_piece_plain(rb, q - 4, sB[buf ^ 1], g.sb);
                    } else {
                        if (q < 4) store_piece(ra, q, sA[buf ^ 1], g.sa);
                        else store_piece(rb, q - 4, sB[buf ^ 1], g.sb);
                    }
                }
            }
        };
#pragma unroll
        for (int grp = 0; grp < kBT / 2; ++grp) {
            const int set = grp & 1;
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[set][0], yb[set][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler(grp, 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[set][0], yb[set][1], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler(grp, 2);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[set][1], yb[set][0], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler(grp, 3);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[set][1], yb[set][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * grp + 4 < kBT) rd(2 * grp + 4, xa[set][0], xa[set][1], yb[set][0], yb[set][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // interior tile of vector-aligned operands: all stages but a ragged last one take the mask-free fillers (one uniform
    // branch per tile selects the stream; the stages of one tile never mix streams except at that last stage)
    const bool interior = VA && VB && m0 + kBM <= g.M && n0 + kBN <= g.N && g.lda < (1 << 22) && g.ldb < (1 << 22);
    const int64_t nplain = interior ? (k_end - k_begin) / kBT - 1 : 0;   // stages s whose NEXT stage is full: s + 2 <= full stages
    int64_t s = 0;
    for (; s < nplain && s + 1 < nstages; ++s) {
        stage_body(std::true_type{}, std::true_type{}, s);
        __syncthreads();
    }
    for (; s + 1 < nstages; ++s) {
        stage_body(std::true_type{}, std::false_type{}, s);
        __syncthreads();
    }
    if (nstages > 0) stage_body(std::false_type{}, std::false_type{}, nstages - 1);

    // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
    if (slab != nullptr) {
        float* out = slab + static_cast<int64_t>(blockIdx.y) * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = m0 + wm * 32 + i * 64 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                    const int64_t col = n0 + wn * 32 + j * 64 + cl;
                    if (row < g.M && col < g.N) out[row * g.N + col] = acc[i][j][e];
                }
        return;
    }
    const bool mirror = g.mirror && g.upper_only && tile_m != tile_n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm * 32 + i * 64 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int64_t col = n0 + wn * 32 + j * 64 + cl;
                if (row < g.M && col < g.N) {
                    float v = g.alpha * acc[i][j][e];
                    if (g.beta != 0.0f) v = g.beta * g.C[row * g.ldc + col] + v;
                    g.C[row * g.ldc + col] = v;
                    if (mirror) g.C[col * g.ldc + row] = v;
                    if (g.Ct) g.Ct[col * g.ldct + row] = v;
                }
            }
}

static void launch_variant(bool va, bool vb, dim3 grid, hipStream_t s, const GemmTN& g, float* slab, int64_t per, int tn) {
    if (va && vb) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, dim3(kGemmThreads), 0, s, g, slab, per, tn);
    else if (va) hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, dim3(kGemmThreads), 0, s, g, slab, per, tn);
    else if (vb) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, dim3(kGemmThreads), 0, s, g, slab, per, tn);
    else hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, dim3(kGemmThreads), 0, s, g, slab, per, tn);
}

int32_t launch_gemm_tn(const GemmTN& g, hipStream_t s) {
    OQ_REQUIRE(g.At && g.B && g.C && g.M > 0 && g.N > 0 && g.Kd >= 0, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad argument");
    OQ_REQUIRE(g.lda >= g.M && g.ldb >= g.N && g.ldc >= g.N, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad leading dimension");
    const bool vec_a = (g.lda % 4 == 0) && (g.M % 4 == 0) && (reinterpret_cast<uintptr_t>(g.At) & 15u) == 0;
    const bool vec_b = (g.ldb % 4 == 0) && (g.N % 4 == 0) && (reinterpret_cast<uintptr_t>(g.B) & 15u) == 0;
    const int tn = static_cast<int>(ceil_div(g.N, kBN)), tm = static_cast<int>(ceil_div(g.M, kBM));
    const uint32_t ntiles = g.upper_only ? static_cast<uint32_t>(tn) * (tn + 1) / 2 : static_cast<uint32_t>(tn) * tm;
    OQ_REQUIRE(g.batch >= 1 && g.batch <= 65535, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad batch %lld", (long long)g.batch);
    OQ_REQUIRE(g.outer >= 1 && g.outer <= 65535, OQ_ERR_INVALID_ARGUMENT, "gemm_tn: bad outer batch %lld", (long long)g.outer);
    const bool vec_sa = (g.batch == 1 || g.stride_a % 4 == 0) && (g.outer == 1 || g.outer_a % 4 == 0);
    const bool vec_sb = (g.batch == 1 || g.stride_b % 4 == 0) && (g.outer == 1 || g.outer_b % 4 == 0);
    launch_variant(vec_a && vec_sa, vec_b && vec_sb, dim3(ntiles, static_cast<uint32_t>(g.outer), static_cast<uint32_t>(g.batch)), s, g, nullptr, 0,
                   tn);
    return check_launch("gemm_tn_kernel");
}

// Sum the T-slices of one 64 x 64 tile in slice order, apply alpha / beta, and write C[m][n] and (for
// off-diagonal tiles) C[n][m]; the transposed copy goes through LDS so both stores are row-contiguous.
__global__ __launch_bounds__(256) void syrk_reduce_kernel(const float* slab, int splits, int64_t K, float alpha_in, float beta, float* C, int tile,
                                                          const float* post_scale) {
    __shared__ float t[64][65];
    const float alpha = post_scale ? alpha_in * post_scale[1] : alpha_in;   // fp16-piece GEMM: 1 / s^2, a power of two
    const int64_t m0 = static_cast<int64_t>(blockIdx.y) * 64, n0 = static_cast<int64_t>(blockIdx.x) * 64;
    if (n0 < m0) return;                                        // below the diagonal: written as the mirror of block (n0, m0)
    (void)tile;                                                 // every 64-block on or above the diagonal lies in a computed GEMM tile
    const bool diag = n0 == m0;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int64_t row = m0 + r, col = n0 + tx;
        float v = 0.f;
        if (row < K && col < K && (!diag || tx >= r)) {
            for (int z = 0; z < splits; ++z) v += slab[(static_cast<int64_t>(z) * K + row) * K + col];
            v = alpha * v;
            if (beta != 0.0f) v = beta * C[row * K + col] + v;
            C[row * K + col] = v;
        }
        t[r][tx] = v;
    }
    __syncthreads();
    // the lower triangle is the mirror of the upper one bit for bit (a GEMM that computes both halves of a diagonal
    // tile may have summed them in different orders)
    for (int r = ty; r < 64; r += 4) {
        const int64_t row = n0 + r, col = m0 + tx;               // transposed position
        if (row < K && col < K && (!diag || tx < r)) C[row * K + col] = t[tx][r];
    }
}

int32_t launch_syrk_reduce(const float* slab, int splits, int64_t K, float alpha, float beta, float* C, int tile, hipStream_t s,
                           const float* post_scale) {
    const uint32_t t64 = static_cast<uint32_t>(ceil_div(K, 64));
    hipLaunchKernelGGL(syrk_reduce_kernel, dim3(t64, t64), dim3(256), 0, s, slab, splits, K, alpha, beta, C, tile, post_scale);
    return check_launch("syrk_reduce_kernel");
}

size_t syrk_slab_bytes(int64_t T, int64_t K) {
    (void)T;
    if (K <= 0) return 0;
    return static_cast<size_t>(16) * K * K * sizeof(float);     // up to 16 slices
}

int32_t launch_syrk_tn(const float* X, int64_t T, int64_t K, int64_t ldx, float scale_x, float alpha, float beta, float* C,
                       void* slab, size_t slab_bytes, hipStream_t s) {
    GemmTN g;
    g.At = X; g.B = X; g.C = C;
    g.M = K; g.N = K; g.Kd = T; g.lda = ldx; g.ldb = ldx; g.ldc = K;
    g.alpha = alpha; g.beta = beta; g.sa = scale_x; g.sb = scale_x;
    g.upper_only = 1; g.mirror = 1;
    const int tn = static_cast<int>(ceil_div(K, kBN));
    const int64_t tiles = static_cast<int64_t>(tn) * (tn + 1) / 2;
    // choose the number of T-slices: enough blocks for ~8 rounds of the 512 block slots, slices of >= 256 rows
    int splits = 1;
    if (slab != nullptr && tiles < 8 * 512) {
        splits = static_cast<int>((8 * 512 + tiles - 1) / tiles);
        const int64_t max_by_t = T / 256 > 0 ? T / 256 : 1;
        if (splits > max_by_t) splits = static_cast<int>(max_by_t);
        if (splits > 16) splits = 16;
        while (splits > 1 && static_cast<size_t>(splits) * K * K * sizeof(float) > slab_bytes) --splits;
    }
    if (splits <= 1) return launch_gemm_tn(g, s);
    const bool vec = (ldx % 4 == 0) && (K % 4 == 0) && (reinterpret_cast<uintptr_t>(X) & 15u) == 0;
    int64_t per = ceil_div(T, splits);
    per = ceil_div(per, kBT) * kBT;
    splits = static_cast<int>(ceil_div(T, per));
    launch_variant(vec, vec, dim3(static_cast<uint32_t>(tiles), static_cast<uint32_t>(splits)), s, g, static_cast<float*>(slab), per, tn);
    int32_t st = check_launch("gemm_tn_kernel(split)");
    if (st != OQ_OK) return st;
    return launch_syrk_reduce(static_cast<const float*>(slab), splits, K, alpha, beta, C, kBM, s);
}

}  // namespace oq

extern "C" {

using namespace oq;

// G1  gptq.py:246-260.
// slab slices the workspace query budgets for: 16 for K <= 8192 (1 GB at 4096), 4 above (1.9 GB at 11008)
static size_t hessian_slab_budget(int64_t K) { return (K <= 0 || K > kMaxHessianWidth) ? 0 : static_cast<size_t>(K <= 8192 ? 16 : 4) * K * K * sizeof(float); }

size_t oq_hessian_workspace_bytes(int64_t T, int64_t K) {
    if (!matrix_ok(T, K, K) || K > kMaxHessianWidth) return 256;
    const size_t f32 = K <= 8192 ? syrk_slab_bytes(T, K) : 0;
    const size_t split = syrk_bf16x3_pieces_bytes(T, K) + hessian_slab_budget(K);
    return (f32 > split ? f32 : split) + 512;
}

int32_t oq_hessian_accumulate_f32(const float* X, int64_t T, int64_t K, int64_t ldx, int64_t n_seen, int64_t n_add, float* H,
                                  int32_t method, void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(X && H && T > 0 && K > 0 && ldx >= K, OQ_ERR_INVALID_ARGUMENT, "oq_hessian_accumulate_f32: bad argument");
    OQ_REQUIRE(matrix_ok(T, K, ldx) && K <= kMaxHessianWidth, OQ_ERR_UNSUPPORTED, "oq_hessian_accumulate_f32: operand too large (T=%lld K=%lld ldx=%lld)",
               (long long)T, (long long)K, (long long)ldx);
    OQ_REQUIRE(method >= OQ_HESSIAN_AUTO && method <= OQ_HESSIAN_F16X3, OQ_ERR_INVALID_ARGUMENT, "oq_hessian_accumulate_f32: unknown method %d", method);
    OQ_REQUIRE(n_seen >= 0 && n_add > 0 && n_seen <= kMaxSamples && n_add <= kMaxSamples, OQ_ERR_INVALID_ARGUMENT,
               "oq_hessian_accumulate_f32: bad sample counts %lld + %lld", (long long)n_seen, (long long)n_add);
    const int64_t n_total = n_seen + n_add;
    // gptq.py:254  H *= num_samples / (num_samples + num_added): a Python float applied to an fp32 array.
    // The first call of the reference starts from zeros (gptq.py:304): beta = 0, nothing is read.
    const float beta = n_seen == 0 ? 0.0f : static_cast<float>(static_cast<double>(n_seen) / static_cast<double>(n_total));
    // auto: the split-operand kernels where their 256-wide tiles are worth it and the caller's workspace holds the
    // pieces; small problems stay on the fp32 MFMA
    if (method == OQ_HESSIAN_AUTO)
        method = (K >= 1024 && (T >= 2048 || K >= 2048) && workspace != nullptr && workspace_bytes >= syrk_bf16x3_pieces_bytes(T, K) + 256)
                     ? OQ_HESSIAN_F16X3 : OQ_HESSIAN_F32;   // measured cross-over: K = 1024-1536 with ~1000 rows is faster on the fp32 kernel
    if (method != OQ_HESSIAN_F32) {
        // gptq.py:257 scales the operand by sqrt(2 / n); here the factor 2 / n goes onto the sum (one rounding per
        // element of H instead of one per element of X)
        const float alpha = static_cast<float>(2.0 / static_cast<double>(n_total));
        return launch_syrk_bf16x3(X, T, K, ldx, alpha, beta, H, workspace, workspace_bytes,
                                  method == OQ_HESSIAN_BF16X9 ? 9 : (method == OQ_HESSIAN_F16X3 ? 3 : 6), as_stream(stream));
    }
    // gptq.py:257  inp = math.sqrt(2 / num_samples) * inp  (double evaluated, weak scalar -> fp32 multiply)
    const float sx = static_cast<float>(std::sqrt(2.0 / static_cast<double>(n_total)));
    return launch_syrk_tn(X, T, K, ldx, sx, 1.0f, beta, H, workspace, workspace_bytes, as_stream(stream));
}

// G1 for the tensors of one calibration batch (calibrate.py:292-305 hands `_accumulate_hessian` one input per node): one
// launch chain for all of them, fp16-piece method.
size_t oq_hessian_many_workspace_bytes(const oq_hessian_item* items_host, int64_t count) {
    return syrk_f16x3_many_workspace_bytes(reinterpret_cast<const int64_t*>(items_host), count);
}

int32_t oq_hessian_accumulate_many_f32(const oq_hessian_item* items_host, const oq_hessian_item* items_device, int64_t count, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    static_assert(sizeof(oq_hessian_item) == 64, "eight 8-byte fields");
    return launch_syrk_f16x3_many(reinterpret_cast<const int64_t*>(items_host), reinterpret_cast<const int64_t*>(items_device), count, workspace,
                                  workspace_bytes, as_stream(stream));
}

// G1 in two halves (fp16-piece method): the HBM-bound preparation of a batch and its matrix-core bound product, so that a
// caller can run the preparation of batch i + 1 on a side stream while the product of batch i occupies the matrix cores.
size_t oq_hessian_pieces_bytes(int64_t T, int64_t K) { return (!matrix_ok(T, K, K) || K > kMaxHessianWidth) ? 0 : syrk_bf16x3_pieces_bytes(T, K) + 256; }
size_t oq_hessian_slab_bytes(int64_t K) { return hessian_slab_budget(K) + 256; }

int32_t oq_hessian_prepare_f32(const float* X, int64_t T, int64_t K, int64_t ldx, int64_t n_total, void* pieces, size_t pieces_bytes, void* stream) {
    OQ_REQUIRE(X && pieces && matrix_ok(T, K, ldx) && K <= kMaxHessianWidth && n_total > 0 && n_total <= 2 * kMaxSamples, OQ_ERR_INVALID_ARGUMENT,
               "oq_hessian_prepare_f32: bad argument");
    OQ_REQUIRE((reinterpret_cast<uintptr_t>(pieces) & 255u) == 0 && pieces_bytes >= syrk_bf16x3_pieces_bytes(T, K), OQ_ERR_WORKSPACE,
               "oq_hessian_prepare_f32: 256-byte aligned buffer of %zu bytes needed, %zu given", syrk_bf16x3_pieces_bytes(T, K), pieces_bytes);
    const float alpha = static_cast<float>(2.0 / static_cast<double>(n_total));
    return syrk_pieces_phases(X, T, K, ldx, alpha, 0.0f, nullptr, static_cast<unsigned char*>(pieces), nullptr, 0, 3, 1, as_stream(stream));
}

int32_t oq_hessian_accumulate_prepared_f32(const void* pieces, int64_t T, int64_t K, int64_t n_seen, int64_t n_add, float* H, void* slabs,
                                           size_t slab_bytes, void* stream) {
    OQ_REQUIRE(pieces && H && matrix_ok(T, K, K) && K <= kMaxHessianWidth && n_seen >= 0 && n_add > 0 && n_seen <= kMaxSamples && n_add <= kMaxSamples &&
                   (reinterpret_cast<uintptr_t>(pieces) & 255u) == 0,
               OQ_ERR_INVALID_ARGUMENT, "oq_hessian_accumulate_prepared_f32: bad argument");
    const int64_t n_total = n_seen + n_add;
    const float beta = n_seen == 0 ? 0.0f : static_cast<float>(static_cast<double>(n_seen) / static_cast<double>(n_total));   // gptq.py:254
    const float alpha = static_cast<float>(2.0 / static_cast<double>(n_total));
    unsigned char* sl = static_cast<unsigned char*>(slabs);
    if (sl != nullptr) {
        const size_t pad = (256 - (reinterpret_cast<uintptr_t>(sl) & 255u)) & 255u;
        sl = slab_bytes > pad ? sl + pad : nullptr;
        slab_bytes = sl ? slab_bytes - pad : 0;
    }
    return syrk_pieces_phases(nullptr, T, K, K, alpha, beta, H, const_cast<unsigned char*>(static_cast<const unsigned char*>(pieces)),
                              reinterpret_cast<float*>(sl), slab_bytes, 3, 2, as_stream(stream));
}

// N1  the calibration walk's large products (calibrate.py:244-251 runs them in onnxruntime): Y = X W on the fp16 matrix cores
// with two-piece operands (22 significand bits, fp32 accumulate) -- the Hessian's arithmetic class, ~5 x an fp32 GEMM.
size_t oq_matmul_pieces_bytes(int64_t Kd, int64_t cols) {
    if (!extent_ok(Kd) || !extent_ok(cols) || !count_ok(Kd * cols, kMaxElements) || Kd > 8 * 65535 * 4) return 0;
    return gemm_f16x3_pieces_bytes(Kd, cols) + ((static_cast<size_t>(cols) * 8 + 255) / 256) * 256;   // + room for one scale pair per row
}

// the per-row scales of an operand live behind its pieces: [cols] s_t, [cols] 1 / s_t
static float* matmul_row_scales(void* pieces, int64_t Kd, int64_t cols) {
    return reinterpret_cast<float*>(static_cast<unsigned char*>(pieces) + gemm_f16x3_pieces_bytes(Kd, cols));
}

int32_t oq_matmul_prepare_f32(const float* X, int64_t Kd, int64_t cols, int64_t ldx, int32_t contraction_is_fast_axis, int32_t per_row_scales,
                              void* pieces, size_t pieces_bytes, void* stream) {
    OQ_REQUIRE(X && pieces && extent_ok(Kd) && extent_ok(cols) && count_ok(Kd * cols, kMaxElements) && extent_ok(ldx) &&
               ldx >= (contraction_is_fast_axis ? Kd : cols), OQ_ERR_INVALID_ARGUMENT, "oq_matmul_prepare_f32: bad argument");
    OQ_REQUIRE(!per_row_scales || contraction_is_fast_axis, OQ_ERR_INVALID_ARGUMENT,
               "oq_matmul_prepare_f32: per-row scales are for sources with the contraction on the fast axis (activations [M, Kd])");
    const size_t need = oq_matmul_pieces_bytes(Kd, cols);
    OQ_REQUIRE(need != 0, OQ_ERR_UNSUPPORTED, "oq_matmul_prepare_f32: contraction of %lld too long", (long long)Kd);
    OQ_REQUIRE(pieces_bytes >= need && (reinterpret_cast<uintptr_t>(pieces) & 255u) == 0, OQ_ERR_WORKSPACE,
               "oq_matmul_prepare_f32: a 256-byte aligned buffer of %zu bytes is needed, %zu given", need, pieces_bytes);
    return make_f16x2_pieces(X, Kd, cols, ldx, contraction_is_fast_axis != 0, pieces, as_stream(stream), true,
                             per_row_scales ? matmul_row_scales(pieces, Kd, cols) : nullptr);
}

int32_t oq_matmul_pieces_f32(const void* pieces_a, const void* pieces_b, int64_t M, int64_t N, int64_t Kd, float alpha, float beta, float* C,
                             int64_t ldc, int32_t a_per_row_scales, void* stream) {
    OQ_REQUIRE(pieces_a && pieces_b && C && matrix_ok(M, N, ldc) && extent_ok(Kd) && oq_matmul_pieces_bytes(Kd, M) != 0 &&
               oq_matmul_pieces_bytes(Kd, N) != 0 && (reinterpret_cast<uintptr_t>(pieces_a) & 255u) == 0 &&
               (reinterpret_cast<uintptr_t>(pieces_b) & 255u) == 0, OQ_ERR_INVALID_ARGUMENT, "oq_matmul_pieces_f32: bad argument");
    const float* unscale = a_per_row_scales ? matmul_row_scales(const_cast<void*>(pieces_a), Kd, M) + M : nullptr;
    return launch_gemm_f16x3(pieces_a, pieces_b, M, N, Kd, alpha, beta, C, ldc, nullptr, as_stream(stream), false, false, false, unscale);
}

}  // extern "C"
