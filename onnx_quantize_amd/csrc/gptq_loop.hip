// G3: the GPTQ block / row loop (gptq.py:153-216) for gfx950.
//
// Output columns are independent given U.  Sequential mode (CORRECTED, or PARITY with mse): gptq_block_kernel
// below, 64 columns per workgroup, 32-row sub-blocks out of registers; after each 128-row block the lazy batch
// update of the rows below (gptq.py:208) is one MFMA TN GEMM over the whole chip.  PARITY without mse needs no
// sequence at all: gptq_parity_kernel.
//
// Error-feedback indexing (oq_gptq_mode):
//   PARITY     coefficient of row j for the error of row i = U[i1+j][i1+i]   (gptq.py:199 as written: the
//              column of the UPPER factor below the diagonal, i.e. exact zeros -> the integers equal RTN
//              with per-group parameters; SURVEY.md finding 1)
//   CORRECTED  U[i1+i][i1+j]  (row of the upper factor right of the diagonal: what GPTQ intends)
// The batch update uses U[i2:, i1:i2] (PARITY: an all-zero block, the GEMM is skipped) or U[i1:i2, i2:]^T.
#include "gemm_tn.hpp"

namespace oq {

constexpr int kLoopMaxRows = 128;

struct LoopArgs {
    float* W;          // [K, N] working copy
    const float* U;    // [K, K]
    int64_t K, N;
    int64_t i1, count; // rows [i1, i1 + count) of this block
    int64_t g;         // loop group size (<= 0: none)
    QGrid grid;
    int32_t mode;
    const float* init_scale;
    const int32_t* init_zp;
    int64_t init_count;
    uint8_t* q_int;    // [K, N]
    float* q_deq;      // [K, N]
    float* used_scale; // [ceil(K/g), N] or null
    int32_t* used_zp;
    float* err;        // [count, N]
    float* carry_scale;  // [N] parameters in force at the end of the previous block (groups may span blocks)
    int32_t* carry_zp;
    const float* pre_scale;   // mse: [groups starting in this block][N] parameters found by the MSE search, else null
    const uint8_t* pre_zp;
    int32_t zp_signed;
    int64_t pre_first_group;  // index (row / g) of the first group that starts in this block
};

// One workgroup = 64 columns x one block of <= 128 rows, 4 waves.  The block is walked in sub-blocks of 32 rows:
//   * wave 0 holds the sub-block's rows of its 64 columns in REGISTERS (one column per lane) and runs the 32
//     sequential steps out of them -- quantize, error, update of the later rows of the sub-block -- with the
//     coefficients of U arriving as wave-uniform LDS broadcasts;
//   * then all four waves apply the sub-block's 32 errors to the block's remaining rows in LDS (each row receives
//     its updates in the reference's order i = 0, 1, ... and with the reference's roundings: product, then
//     subtraction, gptq.py:198-200 -- only later in time, which no value can observe).
// The old formulation (one lane per column, every row step sweeping all later rows through LDS) spent ~10 k cycles
// per row step and ran on N / 256 workgroups.
constexpr int kSubRows = 32;
constexpr int kLoopColsV2 = 64;

__global__ __launch_bounds__(256) void gptq_block_kernel(const LoopArgs a) {
    __shared__ float tile[kLoopMaxRows][kLoopColsV2];   // working copy W1 of the block (gptq.py:157)
    __shared__ float coef[kSubRows][kLoopMaxRows];      // coefficient of the error of sub-block row i for block row j
    __shared__ float err_s[kSubRows][kLoopColsV2];
    __shared__ float gp_scale[kSubRows][kLoopColsV2];   // parameters of the groups that start inside the current sub-block
    __shared__ int32_t gp_zp[kSubRows][kLoopColsV2];
    __shared__ float red_mn[4][kLoopColsV2], red_mx[4][kLoopColsV2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kLoopColsV2 + lane;
    const bool live = c < a.N;
    const int64_t cc = live ? c : a.N - 1;  // clamped column for loads
    const int count = static_cast<int>(a.count);

    // 8 loads in flight per lane (a load -> LDS store chain per row would pay the memory latency 32 times)
    for (int i0 = wave; i0 < count; i0 += 32) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 4 * u;
            x[u] = a.W[(a.i1 + (i < count ? i : count - 1)) * a.N + cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 4 * u < count) tile[i0 + 4 * u][lane] = x[u];
    }

    float scale = 1.0f;
    int32_t zp = 0;
    if (wave == 0) {
        if (a.i1 == 0) {  // gptq.py:104-116
            const int64_t pi = a.init_count == 1 ? 0 : cc;
            scale = a.init_scale[pi];
            zp = a.init_zp[pi];
        } else {
            scale = a.carry_scale[cc];
            zp = a.carry_zp[cc];
        }
    }
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;

    for (int s0 = 0; s0 < count; s0 += kSubRows) {
        const int ns = count - s0 < kSubRows ? count - s0 : kSubRows;   // rows of this sub-block
        // gptq.py:199 as written (PARITY): column i1+i of U below the diagonal -> U[(i1+j)*K + i1+i] (exact zeros);
        // CORRECTED: row i1+i right of the diagonal -> U[(i1+i)*K + i1+j].  The diagonal entry is the divisor.
        {   // 32 x 128 coefficients, 16 per thread, all loads in flight before the first LDS store
            constexpr int kPer = kSubRows * kLoopMaxRows / 256;
            float cv[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int idx = threadIdx.x + u * 256;
                const int i = idx / kLoopMaxRows, j = idx - i * kLoopMaxRows;
                const int ic = i < ns ? i : ns - 1, jc = j < count ? j : count - 1;   // clamped loads, masked stores
                const int64_t ri = a.i1 + s0 + ic, rj = a.i1 + jc;
                cv[u] = (a.mode == OQ_GPTQ_PARITY && jc != s0 + ic) ? a.U[rj * a.K + ri] : a.U[ri * a.K + rj];
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int idx = threadIdx.x + u * 256;
                const int i = idx / kLoopMaxRows, j = idx - i * kLoopMaxRows;
                if (i < ns) coef[i][j] = j < count ? cv[u] : 0.0f;
            }
        }
        // gptq.py:168-184: parameters of every group that starts in this sub-block, from rows [row, row + g) of the
        // GLOBAL working matrix (not of the block copy), channel strategy -- or the ones an MSE search left.  All
        // four waves fold the rows (8 loads in flight per lane); uniform control flow.
        // position of the sub-block's first row inside its group: ONE 64-bit division per sub-block instead of two per
        // row step (a runtime 64-bit `%` / `/` is ~100 instructions, and the row steps are the critical chain)
        const int64_t grp0 = a.g > 0 ? (a.i1 + s0) / a.g : 0;
        const int64_t rem0 = a.g > 0 ? (a.i1 + s0) - grp0 * a.g : 1;
        if (a.g > 0) {
            int slot = 0;
            int64_t rem = rem0, grp = grp0;
            for (int i = 0; i < ns; ++i, ++rem) {
                const int64_t row = a.i1 + s0 + i;
                if (rem == a.g) { rem = 0; ++grp; }
                if (rem != 0) continue;
                if (a.pre_scale != nullptr) {   // mse=True: searched beforehand on the same rows (utils.py:140-239)
                    if (wave == 0) {
                        const int64_t o = (grp - a.pre_first_group) * a.N + cc;
                        gp_scale[slot][lane] = a.pre_scale[o];
                        gp_zp[slot][lane] = a.zp_signed ? static_cast<int32_t>(static_cast<int8_t>(a.pre_zp[o])) : static_cast<int32_t>(a.pre_zp[o]);
                    }
                } else {
                    const int64_t rend = row + a.g < a.K ? row + a.g : a.K;
                    float mn = INFINITY, mx = -INFINITY;
                    for (int64_t r = row + wave * 8; r < rend; r += 32) {
                        float x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = a.W[(r + u < rend ? r + u : rend - 1) * a.N + cc];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { mn = nmin(mn, x[u]); mx = nmax(mx, x[u]); }
                    }
                    red_mn[wave][lane] = mn;
                    red_mx[wave][lane] = mx;
                    __syncthreads();
                    if (wave == 0) {
                        mn = nmin(nmin(red_mn[0][lane], red_mn[1][lane]), nmin(red_mn[2][lane], red_mn[3][lane]));
                        mx = nmax(nmax(red_mx[0][lane], red_mx[1][lane]), nmax(red_mx[2][lane], red_mx[3][lane]));
                        const QParam p = qparam_from_minmax(mn, mx, a.grid);
                        gp_scale[slot][lane] = p.scale;
                        gp_zp[slot][lane] = p.zp;
                    }
                    __syncthreads();
                }
                ++slot;
            }
        }
        __syncthreads();
        if (wave == 0) {
            float w[kSubRows];
            int slot = 0;
            int64_t rem = rem0, grp = grp0;
#pragma unroll
            for (int i = 0; i < kSubRows; ++i) w[i] = tile[(s0 + i < count) ? s0 + i : count - 1][lane];
#pragma unroll
            for (int i = 0; i < kSubRows; ++i, ++rem) {
                if (i < ns) {   // uniform
                    const int64_t row = a.i1 + s0 + i;
                    if (rem == a.g) { rem = 0; ++grp; }
                    if (rem == 0) {   // a.g <= 0: rem0 = 1 and `rem == a.g` never holds, so rem only grows
                        scale = gp_scale[slot][lane];
                        zp = gp_zp[slot][lane];
                        ++slot;
                        if (live && a.used_scale != nullptr) {
                            a.used_scale[grp * a.N + c] = scale;
                            a.used_zp[grp * a.N + c] = zp;
                        }
                    }
                    const int32_t qi = quantize_one(w[i], scale, zp, qmin, qmax);   // gptq.py:186-188
                    const float q = dequantize_one(qi, scale, zp);                    // :189
                    const float e = (w[i] - q) / coef[i][s0 + i];                     // :164, :197
                    if (live) {
                        a.q_int[row * a.N + c] = static_cast<uint8_t>(qi);
                        a.q_deq[row * a.N + c] = q;
                        a.err[(s0 + i) * a.N + c] = e;
                    }
                    err_s[i][lane] = e;
                    // gptq.py:198-200  W1[i:, :] -= outer(Hinv1[i:, i], err1): one rounding for the product (the K = 1
                    // matmul of the reference), one for the subtraction (no FMA)
#pragma unroll
                    for (int j = i + 1; j < kSubRows; ++j) w[j] = w[j] - coef[i][s0 + j] * e;
                }
            }
        }
        __syncthreads();
        // the same updates for the block's later rows, four waves, rows interleaved
        if (s0 + kSubRows < count) {
            float e[kSubRows];
#pragma unroll
            for (int i = 0; i < kSubRows; ++i) e[i] = err_s[i][lane];
            for (int r = s0 + kSubRows + wave; r < count; r += 4) {
                float t = tile[r][lane];
#pragma unroll
                for (int i = 0; i < kSubRows; ++i) t = t - coef[i][r] * e[i];
                tile[r][lane] = t;
            }
        }
        __syncthreads();
    }
    if (wave == 0 && live) {
        a.carry_scale[c] = scale;
        a.carry_zp[c] = zp;
    }
}

// PARITY mode without the sequential kernel.  As written in the reference (gptq.py:199, :208) the error of a row
// reaches no other row (the coefficients are the structural zeros below the diagonal of the upper factor), so
// every row is quantized from the untouched working matrix with the parameters of its group: the parameters of
// group kg come from rows [kg*g, min(kg*g + g, K)) of W (gptq.py:168-184), or, without a loop group, are the
// initial per-channel / per-tensor ones (:104-116) for every row.  One thread = one column of one row band.
__global__ __launch_bounds__(256) void gptq_parity_kernel(const LoopArgs a, int64_t band_rows) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const bool live = c < a.N;
    const int64_t cc = live ? c : a.N - 1;
    const int64_t band = blockIdx.y;
    const int64_t r0 = band * band_rows;
    const int64_t r1 = r0 + band_rows < a.K ? r0 + band_rows : a.K;
    float scale;
    int32_t zp;
    if (a.g > 0) {
        float mn = INFINITY, mx = -INFINITY;
        int64_t r = r0;
        for (; r + 3 < r1; r += 4) {   // four loads in flight per lane
            const float x0 = a.W[r * a.N + cc], x1 = a.W[(r + 1) * a.N + cc], x2 = a.W[(r + 2) * a.N + cc], x3 = a.W[(r + 3) * a.N + cc];
            mn = nmin(nmin(mn, x0), nmin(x1, nmin(x2, x3)));
            mx = nmax(nmax(mx, x0), nmax(x1, nmax(x2, x3)));
        }
        for (; r < r1; ++r) {
            const float x = a.W[r * a.N + cc];
            mn = nmin(mn, x);
            mx = nmax(mx, x);
        }
        const QParam p = qparam_from_minmax(mn, mx, a.grid);
        scale = p.scale;
        zp = p.zp;
        if (live && a.used_scale != nullptr) {
            a.used_scale[band * a.N + c] = scale;
            a.used_zp[band * a.N + c] = zp;
        }
    } else {
        const int64_t pi = a.init_count == 1 ? 0 : cc;
        scale = a.init_scale[pi];
        zp = a.init_zp[pi];
    }
    if (!live) return;
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    for (int64_t r = r0; r < r1; ++r) {
        const int32_t qi = quantize_one(a.W[r * a.N + c], scale, zp, qmin, qmax);   // gptq.py:186-188
        a.q_int[r * a.N + c] = static_cast<uint8_t>(qi);
        a.q_deq[r * a.N + c] = dequantize_one(qi, scale, zp);                       // :189
    }
}

int32_t rtn_impl(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy, int64_t group_size,
                 int32_t symmetric, int32_t reduce_range, float clip_ratio, int32_t mse, void* q_out, float* scale_out, void* zp_out,
                 int32_t layout, void* workspace, size_t workspace_bytes, void* stream, bool emit_q);

}  // namespace oq

extern "C" {

using namespace oq;

size_t oq_gptq_loop_workspace_bytes(int64_t K, int64_t N, int64_t block_size) {
    if (K <= 0 || N <= 0) return 0;
    (void)block_size;
    // Err [128, N] + carried (scale, zp) [N] + (mse) per-group parameters of one block [128, N] x (4 + 1) B
    // + the MSE search's own workspace for one [group, N] slice
    return static_cast<size_t>(kLoopMaxRows) * N * 4 + static_cast<size_t>(N) * 8 + static_cast<size_t>(kLoopMaxRows) * N * 5 +
           oq_rtn_workspace_bytes(K, N, OQ_CHANNEL, -1, 1) + 2048;
}

int32_t oq_gptq_loop_f32(float* W, int64_t K, int64_t N, const float* U, int32_t qtype, int64_t group_size, int32_t symmetric,
                         int32_t reduce_range, float clip_ratio, int32_t mse, int64_t block_size, int32_t mode,
                         const float* init_scale, const int32_t* init_zp, int64_t init_count, void* q_int_out,
                         float* q_deq_out, float* used_scale, int32_t* used_zp, void* workspace, size_t workspace_bytes,
                         void* stream) {
    OQ_REQUIRE(W && U && init_scale && init_zp && q_int_out && q_deq_out && K > 0 && N > 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_gptq_loop_f32: bad argument");
    OQ_REQUIRE(init_count == 1 || init_count == N, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: init_count must be 1 or N");
    OQ_REQUIRE(block_size > 0, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: block_size must be positive");
    OQ_REQUIRE(mode == OQ_GPTQ_PARITY || mode == OQ_GPTQ_CORRECTED, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: bad mode %d", mode);
    OQ_REQUIRE(clip_ratio > 0.0f && clip_ratio <= 1.0f, OQ_ERR_INVALID_ARGUMENT, "clip_ratio must be in (0.0, 1.0], got %g", clip_ratio);
    QGrid grid;
    int32_t st = make_grid(qtype, symmetric, reduce_range, clip_ratio, &grid);
    if (st != OQ_OK) return st;
    const size_t need = oq_gptq_loop_workspace_bytes(K, N, block_size);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_gptq_loop_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    hipStream_t s = as_stream(stream);

    LoopArgs a;
    a.W = W; a.U = U; a.K = K; a.N = N; a.g = group_size > 0 ? group_size : 0; a.grid = grid; a.mode = mode;
    a.init_scale = init_scale; a.init_zp = init_zp; a.init_count = init_count;
    a.q_int = static_cast<uint8_t*>(q_int_out); a.q_deq = q_deq_out;
    a.used_scale = (group_size > 0) ? used_scale : nullptr;
    a.used_zp = used_zp;
    if (a.used_scale != nullptr && used_zp == nullptr) return fail(OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: used_zp missing");
    a.err = static_cast<float*>(workspace);
    a.carry_scale = a.err + static_cast<size_t>(kLoopMaxRows) * N;
    a.carry_zp = reinterpret_cast<int32_t*>(a.carry_scale + N);

    float* pre_scale = reinterpret_cast<float*>(a.carry_zp + N);
    uint8_t* pre_zp = reinterpret_cast<uint8_t*>(pre_scale + static_cast<size_t>(kLoopMaxRows) * N);
    char* mse_ws = reinterpret_cast<char*>(pre_zp) + static_cast<size_t>(kLoopMaxRows) * N;
    mse_ws += (256 - reinterpret_cast<uintptr_t>(mse_ws) % 256) % 256;
    const size_t mse_ws_bytes = static_cast<size_t>(static_cast<char*>(workspace) + workspace_bytes - mse_ws);
    a.pre_scale = nullptr; a.pre_zp = nullptr; a.pre_first_group = 0;
    a.zp_signed = (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0;
    if (mode == OQ_GPTQ_PARITY && !(mse && a.g > 0) && ceil_div(K, a.g > 0 ? a.g : 128) <= 65535) {
        // no row depends on another one: one elementwise launch instead of K sequential steps
        const int64_t band_rows = a.g > 0 ? a.g : 128;
        a.i1 = 0; a.count = 0;
        hipLaunchKernelGGL(gptq_parity_kernel, dim3(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(ceil_div(K, band_rows))),
                           dim3(256), 0, s, a, band_rows);
        return check_launch("gptq_parity_kernel");
    }
    const int64_t bs = block_size < kLoopMaxRows ? block_size : kLoopMaxRows;
    const uint32_t nblk = static_cast<uint32_t>(ceil_div(N, kLoopColsV2));
    for (int64_t i1 = 0; i1 < K; i1 += bs) {
        const int64_t count = (K - i1) < bs ? (K - i1) : bs;
        a.i1 = i1; a.count = count;
        if (mse && a.g > 0) {
            // gptq.py:168-184 with mse=True: the MSE search (channel strategy) on rows [r, r+g) of the working
            // matrix for every group that starts inside this block, before the sequential kernel runs.
            const int64_t first = (i1 + a.g - 1) / a.g;   // first group index with start row >= i1
            int64_t slot = 0;
            for (int64_t gi = first; gi * a.g < i1 + count; ++gi, ++slot) {
                const int64_t r0 = gi * a.g;
                const int64_t rows = (r0 + a.g <= K) ? a.g : K - r0;
                st = rtn_impl(W + r0 * N, rows, N, N, qtype, OQ_CHANNEL, -1, symmetric, reduce_range, clip_ratio, 1, nullptr,
                              pre_scale + slot * N, pre_zp + slot * N, OQ_LAYOUT_KN, mse_ws, mse_ws_bytes, stream, false);
                if (st != OQ_OK) return st;
            }
            a.pre_scale = pre_scale; a.pre_zp = pre_zp; a.pre_first_group = first;
        }
        hipLaunchKernelGGL(gptq_block_kernel, dim3(nblk), dim3(256), 0, s, a);
        st = check_launch("gptq_block_kernel");
        if (st != OQ_OK) return st;
        const int64_t i2 = i1 + count;
        if (mode == OQ_GPTQ_CORRECTED && i2 < K) {
            // gptq.py:208 with the intended operand: W[i2:, :] -= U[i1:i2, i2:]^T @ Err1
            GemmTN g;
            g.At = U + i1 * K + i2; g.lda = K; g.M = K - i2;
            g.B = a.err; g.ldb = N; g.N = N;
            g.C = W + i2 * N; g.ldc = N;
            g.Kd = count; g.alpha = -1.0f; g.beta = 1.0f; g.sa = 1.0f; g.sb = 1.0f; g.upper_only = 0; g.mirror = 0;
            st = launch_gemm_tn(g, s);
            if (st != OQ_OK) return st;
        }
    }
    return OQ_OK;
}

}  // extern "C"
