// N2: the numeric cores of the AWQ and SmoothQuant pre-passes for gfx950, device resident from the activations to the
// winning grid point (reference: pre_passes/awq.py:47-72, 114-184, 207-259; pre_passes/smooth_quant.py:62-74, 104-113).
//
// The reference evaluates, for each of 20 scale candidates (10 clip ratios), loss = mean((X W - X W^)^2) with two NumPy
// matmuls.  Here the loss is || X (W - W^) ||_F^2 / (T N): ONE product per candidate, on the matrix cores with fp16 pieces
// (gemm_f16x3_kernel, 22-bit operands, fp32 accumulate), whose epilogue squares and sums the accumulators -- the [T, N]
// product is never written.  The pieces of X are made once per layer; per candidate the passes over [K, N] memory are
//   scale rows -> oq RTN kernels (rtn.hip) -> awq_diff_kernel (D = W - dequant / s, absmax partials) -> split -> GEMM.
// Round 2 ran the same composition through torch elementwise kernels and rocBLAS (23.4 ms per 4096^3 layer, bound by 20
// fp32 GEMMs of 137 GFLOP).
#include "gemm_tn.hpp"
#include "row_params.hpp"

namespace oq {

int32_t rtn_impl(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy, int64_t group_size, int32_t symmetric,
                 int32_t reduce_range, float clip_ratio, int32_t mse, void* q_out, float* scale_out, void* zp_out, int32_t layout, void* workspace,
                 size_t workspace_bytes, void* stream, bool emit_q);

constexpr int kAwqMaxGrid = 64;
constexpr int kColChunks = 64;   // row chunks of the column reductions

// ---- column sums of |x| (awq.py:47-50) / column max of |x| (smooth_quant.py:62-69): partial[chunk][k]
template <bool MAX>
__global__ __launch_bounds__(256) void col_abs_partial_kernel(const float* __restrict__ X, int64_t T, int64_t K, int64_t ldx, float* __restrict__ partial) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (k >= K) return;
    const int64_t per = (T + static_cast<int64_t>(gridDim.y) - 1) / static_cast<int64_t>(gridDim.y);
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * per, t1 = t0 + per < T ? t0 + per : T;
    float acc = 0.f;
    int64_t t = t0;
    for (; t + 3 < t1; t += 4) {   // four loads in flight per lane
        const float a = fabsf(X[t * ldx + k]), b = fabsf(X[(t + 1) * ldx + k]), c = fabsf(X[(t + 2) * ldx + k]), d = fabsf(X[(t + 3) * ldx + k]);
        if constexpr (MAX) acc = nmax(nmax(acc, a), nmax(b, nmax(c, d)));
        else acc += (a + b) + (c + d);
    }
    for (; t < t1; ++t) {
        const float a = fabsf(X[t * ldx + k]);
        if constexpr (MAX) acc = nmax(acc, a); else acc += a;
    }
    partial[static_cast<int64_t>(blockIdx.y) * K + k] = acc;
}

template <bool MAX>
__global__ __launch_bounds__(256) void col_abs_finish_kernel(const float* __restrict__ partial, int chunks, int64_t K, float inv_count, float* __restrict__ out) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (k >= K) return;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float v = partial[static_cast<int64_t>(c) * K + k];
        if constexpr (MAX) acc = nmax(acc, v); else acc += v;
    }
    out[k] = MAX ? acc : acc * inv_count;
}

__global__ __launch_bounds__(256) void col_abs_accumulate_kernel(const float* __restrict__ partial, int chunks, int64_t K, int accumulate, float* __restrict__ out) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (k >= K) return;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) acc += partial[static_cast<int64_t>(c) * K + k];
    out[k] = accumulate ? out[k] + acc : acc;
}

// ---- awq.py:52-72: |w| / absmax of its quantization group, mean over the output channels -> ws[k].
// gmax[kg][n] = absmax over rows [kg g, kg g + g) of column n (g = K: one row of maxima per column).
__global__ __launch_bounds__(256) void group_absmax_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, int64_t g, float* __restrict__ gmax) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t kg = blockIdx.y;
    const int64_t r0 = kg * g, r1 = r0 + g < K ? r0 + g : K;
    float m = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4)
        m = nmax(nmax(m, fabsf(W[r * ldw + n])), nmax(fabsf(W[(r + 1) * ldw + n]), nmax(fabsf(W[(r + 2) * ldw + n]), fabsf(W[(r + 3) * ldw + n]))));
    for (; r < r1; ++r) m = nmax(m, fabsf(W[r * ldw + n]));
    gmax[kg * N + n] = m;
}

// tensor strategy: one maximum for all of W (folds the per-column maxima in place into gmax[0..N) = the same value)
__global__ __launch_bounds__(1024) void fold_to_scalar_kernel(float* v, int64_t count) {
    __shared__ float sm[16];
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) m = nmax(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = threadIdx.x < (blockDim.x >> 6) ? sm[threadIdx.x] : 0.f;
        t = wave_max(t);
        if (threadIdx.x == 0) sm[0] = t;
    }
    __syncthreads();
    m = sm[0];
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) v[i] = m;
}

__global__ __launch_bounds__(256) void weight_scale_rows_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, int64_t g,
                                                                const float* __restrict__ gmax, float* __restrict__ ws) {
    __shared__ float sm[4];
    const int64_t k = blockIdx.x;
    const float* row = W + k * ldw;
    const float* mx = gmax + (k / g) * N;
    float acc = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += 256) acc += fabsf(row[n]) / mx[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[k] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / static_cast<float>(N);
}

// ---- awq.py:143-150: the n_grid candidate scales, one block per candidate:
//   s = clip(act^r / ws^(1 - r), 1e-4), r = i / n_grid;  s /= sqrt(max(s) * min(s))
__global__ __launch_bounds__(1024) void grid_scales_kernel(const float* __restrict__ act, const float* __restrict__ ws, int64_t K, int n_grid,
                                                           float* __restrict__ scales /* [n_grid][K] */) {
    __shared__ float s_mx[16], s_mn[16];
    const int i = blockIdx.x;
    const float ratio = static_cast<float>(static_cast<double>(i) / static_cast<double>(n_grid));   // Python float -> fp32 operand of np.power
    const float one_minus = static_cast<float>(1.0 - static_cast<double>(i) / static_cast<double>(n_grid));
    float* out = scales + static_cast<int64_t>(i) * K;
    float mx = 0.f, mn = INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += blockDim.x) {
        float s = powf(act[k], ratio) / powf(ws[k], one_minus);
        s = nmax(s, 1e-4f);
        out[k] = s;
        mx = nmax(mx, s);
        mn = nmin(mn, s);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_mn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < static_cast<int>(blockDim.x >> 6); ++w) { mx = nmax(mx, s_mx[w]); mn = nmin(mn, s_mn[w]); }
        s_mx[0] = sqrtf(mx * mn);
    }
    __syncthreads();
    const float norm = s_mx[0];
    for (int64_t k = threadIdx.x; k < K; k += blockDim.x) out[k] = out[k] / norm;
}

// ---- W * s[:, None] (awq.py:156) -> Ws, four columns per thread
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, const float* __restrict__ s,
                                                         float* __restrict__ out) {
    const int64_t n4 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    const int64_t k = blockIdx.y;
    if (n4 >= N) return;
    const float sk = s[k];
    if (n4 + 3 < N && (ldw & 3) == 0 && (N & 3) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(W + k * ldw + n4);
        *reinterpret_cast<float4*>(out + k * N + n4) = make_float4(v.x * sk, v.y * sk, v.z * sk, v.w * sk);
    } else {
        for (int64_t n = n4; n < N && n < n4 + 4; ++n) out[k * N + n] = W[k * ldw + n] * sk;
    }
}

// ---- D = W - dequant(q) / s[:, None]  (awq.py:166-175; s == null: the clip search's D = W - dequant(q), awq.py:236-245),
// with the block's max |D| for the fp16 pieces' power-of-two scale.  One thread = one column x 8 rows.
__global__ __launch_bounds__(256) void awq_diff_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, const uint8_t* __restrict__ q,
                                                       const float* __restrict__ qscale, const uint8_t* __restrict__ qzp, ParamIndex pi, int32_t is_signed,
                                                       const float* __restrict__ s, float* __restrict__ D, float* __restrict__ absmax_partial) {
    __shared__ float sm[4];
    const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 8;
    float m = 0.f;
    if (n < N) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = r0 + u;
            if (r >= K) break;
            const int64_t p = pi(r, n);
            const uint32_t byte = q[r * N + n];
            const int32_t qi = is_signed ? static_cast<int32_t>(static_cast<int8_t>(byte)) : static_cast<int32_t>(byte);
            const int32_t zp = is_signed ? static_cast<int32_t>(static_cast<int8_t>(qzp[p])) : static_cast<int32_t>(qzp[p]);
            float w_hat = dequantize_one(qi, qscale[p], zp);          // utils.py:130-132
            if (s != nullptr) w_hat = w_hat / s[r];                    // awq.py:175
            const float d = W[r * ldw + n] - w_hat;
            D[r * N + n] = d;
            m = nmax(m, fabsf(d));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) absmax_partial[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
}

// The same for vector-aligned operands: four neighbouring columns per thread (16-byte loads of W, 4-byte loads of q, 16-byte
// stores of D), parameters fetched once for the thread's 8 rows (8 divides every group size this path accepts).
__global__ __launch_bounds__(256) void awq_diff4_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, const uint8_t* __restrict__ q,
                                                        const float* __restrict__ qscale, const uint8_t* __restrict__ qzp, ParamIndex pi, int32_t is_signed,
                                                        const float* __restrict__ s, float* __restrict__ D, float* __restrict__ absmax_partial) {
    __shared__ float sm[4];
    const int64_t n = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 8;
    float m = 0.f;
    if (n < N) {
        float sc[4];
        int32_t zp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t p = pi(r0, n + e);
            sc[e] = qscale[p];
            zp[e] = is_signed ? static_cast<int32_t>(static_cast<int8_t>(qzp[p])) : static_cast<int32_t>(qzp[p]);
        }
        float4 w[8];
        uint32_t qb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = r0 + u < K ? r0 + u : K - 1;   // clamped, never predicated
            w[u] = *reinterpret_cast<const float4*>(W + r * ldw + n);
            qb[u] = *reinterpret_cast<const uint32_t*>(q + r * N + n);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = r0 + u;
            if (r < K) {
                const float inv = s != nullptr ? s[r] : 1.0f;
                const float wv[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t byte = (qb[u] >> (8 * e)) & 0xffu;
                    const int32_t qi = is_signed ? static_cast<int32_t>(static_cast<int8_t>(byte)) : static_cast<int32_t>(byte);
                    float w_hat = dequantize_one(qi, sc[e], zp[e]);
                    if (s != nullptr) w_hat = w_hat / inv;
                    d[e] = wv[e] - w_hat;
                    m = nmax(m, fabsf(d[e]));
                }
                *reinterpret_cast<float4*>(D + r * N + n) = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) absmax_partial[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
}

// ---- group strategy, 16 <= g <= 128 rows: scale rows -> RTN -> W - dequant / s in ONE pass over W.  The three steps of a
// candidate that only move data -- W s[:, None] written and read back, the integers written and read back, W read a third
// time for the difference -- fall away: the block of rtn_group_fused (8 waves x 16 rows x 256 columns, the group's range
// through LDS) keeps its rows in registers from the load to the difference.  Same arithmetic in the same order as the
// separate kernels (fp32 product w s, utils.py R1 / Q1 / K1 through the helpers of oq_common.hpp, (q - zp) * scale, / s,
// w - .): the D it writes is theirs bit for bit.
//
// PIECES (round 5): the difference leaves the kernel as what the loss product reads -- the FIRST fp16 pieces of D in the
// GEMM's operand layout (gemm_tn.hpp: rows 8c .. 8c + 7 of a column are one 16-byte vector; a lane's 16 rows x 4 columns are
// eight of them, a wave-instruction writes 4 KB in one piece) -- instead of 64 MB of fp32 that a second kernel re-reads to
// split: no D, no split launch (27 us per candidate on 4096 x 4096), no partial maxima.  The power-of-two scale of the
// pieces then has to exist BEFORE the differences do: it comes from an upper bound of |D| (awq_bound_kernel) instead of
// their maximum, which changes nothing but an exponent as long as no piece leaves fp16's normal range -- same mantissas,
// same products, same loss bits.
typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2a __attribute__((ext_vector_type(2)));
template <bool PIECES>
__global__ __launch_bounds__(512, 2) void awq_group_diff_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, int64_t g, QGrid grid, int wpg,
                                                                const float* __restrict__ row_scale, float* __restrict__ D,
                                                                float* __restrict__ absmax_partial, const float* __restrict__ piece_scale,
                                                                float* __restrict__ piece_header, u32x4a* __restrict__ P, int64_t Np) {
    __shared__ float4 s_mn[8][64];
    __shared__ float4 s_mx[8][64];
    __shared__ float s_abs[8];
    __shared__ u32x4a s_stage[PIECES ? 8 : 1][PIECES ? 256 : 1];      // PIECES: 4 KB per wave, the store transposition below
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int wig = wave % wpg, gib = wave / wpg, gpb = 8 / wpg;
    const int64_t kgroups = K / g, kg = static_cast<int64_t>(blockIdx.y) * gpb + gib;
    const bool group_ok = kg < kgroups;
    const int64_t row0 = (group_ok ? kg : 0) * g + static_cast<int64_t>(wig) * 16;   // a surplus group re-reads group 0: loads are never predicated
    // the column tile rotates with the row of blocks: with a multiple of eight tiles (N = 4096: 16) an XCD -- linear block id
    // % 8 -- would otherwise own the same tiles in every row of blocks, a fixed residue of the address inside a row
    // (docs/LAB_NOTES_r05.md; scale search 3.15-3.32 -> 3.08 ms).  A permutation of the tiles per row of blocks: speed only.
    const int64_t tile_col0 = static_cast<int64_t>((blockIdx.x + blockIdx.y) % gridDim.x) * 256;
    const bool col_ok = tile_col0 + lane * 4 < N;
    const int64_t lcol = col_ok ? tile_col0 + lane * 4 : N - 4;
    float4 w[16];
    float sr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        w[r] = *reinterpret_cast<const float4*>(W + (row0 + r) * ldw + lcol);
        sr[r] = row_scale != nullptr ? row_scale[row0 + r] : 1.0f;     // uniform: scalar loads
    }
    auto scaled = [&](int r, int i) -> float {      // what the candidate quantizes: w * s (awq.py:156), or w (the clip search)
        const float x = i == 0 ? w[r].x : i == 1 ? w[r].y : i == 2 ? w[r].z : w[r].w;
        return row_scale != nullptr ? x * sr[r] : x;
    };
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mn[i] = mx[i] = scaled(0, i);
#pragma unroll
    for (int r = 1; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = scaled(r, i);
            mn[i] = nmin(mn[i], x);
            mx[i] = nmax(mx[i], x);
        }
    if (wpg > 1) {   // uniform over the block: the group's waves meet in LDS
        s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
        __syncthreads();
        const int w0 = gib * wpg;
        for (int v = 0; v < wpg; ++v) {
            const float4 tn = s_mn[w0 + v][lane], tx = s_mx[w0 + v][lane];
            mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y); mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
            mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y); mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
        }
    }
    ColQ cq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cq[i] = make_colq(qparam_from_minmax(mn[i], mx[i], grid), mn[i], mx[i], 0);
    const float lo = static_cast<float>(grid.qmin), hi = static_cast<float>(grid.qmax);
    float m = 0.f;
    float* out = D + row0 * N + tile_col0 + lane * 4;
    // K1's fast path is proven two rows at a time (rtn.hip, round 6): a running NaN-propagating maximum of |t - k| against the
    // narrowest band of the lane's four columns, one ballot per pair of rows
    const float thr_min = nmin(nmin(cq[0].thr, cq[1].thr), nmin(cq[2].thr, cq[3].thr));
#pragma unroll
    for (int rg = 0; rg < 16; rg += 2) {
        float x[2][4], f[2][4];
        float far = 0.0f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[r][i] = scaled(rg + r, i);
                const float t = x[r][i] * cq[i].rinv;
                const float k = rintf(t);
                far = nmax(far, fabsf(t - k));
                f[r][i] = __builtin_amdgcn_fmed3f(k + cq[i].zpb, lo, hi);
            }
        if (__builtin_amdgcn_ballot_w64(!(far < thr_min)) != 0) {   // wave-uniform, rare: redo these rows with the IEEE divide
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) f[r][i] = quantize_exact_biased(x[r][i], cq[i], grid.qmin, grid.qmax, 0);
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int r = rg + r2;
            const float wv[4] = {w[r].x, w[r].y, w[r].z, w[r].w};
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float w_hat = (f[r2][i] - cq[i].zpb) * cq[i].scale;          // utils.py:130-132 on exact small integers
                if (row_scale != nullptr) w_hat = w_hat / sr[r];
                d[i] = wv[i] - w_hat;
                m = nmax(m, fabsf(d[i]));
            }
            if constexpr (PIECES) w[r] = make_float4(d[0], d[1], d[2], d[3]);      // the row's weights are dead: its differences take their place
            else if (col_ok && group_ok) *reinterpret_cast<float4*>(out + r * N) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    if constexpr (PIECES) {
        const float sc = piece_scale[0];                         // a power of two: the products are exact
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {      // the product behind this launch reads its operand's scale from the header
            piece_header[0] = piece_scale[0]; piece_header[1] = piece_scale[1]; piece_header[2] = piece_scale[2];
        }
        // A lane holds the vectors of columns 4 l .. 4 l + 3; stored from there a wave-instruction would write 64 pieces of
        // 16 bytes, 64 bytes apart (measured: the kernel at 60 us instead of 39).  Through 4 KB of LDS per wave, one chunk
        // at a time, every instruction writes 1 KB in one piece: lane l takes vector 64 j + l.  Written and read by the same
        // wave only: no barrier (the compiler's own lgkmcnt waits order a wave's LDS accesses).
        if (!group_ok) return;                                   // uniform per wave; nothing below synchronises
        u32x4a* const stage = &s_stage[wave][0];
        u32x4a* o = P + (row0 / 8) * 2 * Np + tile_col0;        // columns past N (inside the padded width) hold zeros
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4a v;
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) {
                    const float4 a = w[8 * c + 2 * rp], b = w[8 * c + 2 * rp + 1];
                    const float xa = i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w;
                    const float xb = i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w;
                    const f16x2a h2 = {static_cast<_Float16>(col_ok ? xa * sc : 0.f), static_cast<_Float16>(col_ok ? xb * sc : 0.f)};
                    v[rp] = __builtin_bit_cast(uint32_t, h2);
                }
                stage[lane * 4 + i] = v;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(stage[j * 64 + lane], o + c * 2 * Np + j * 64 + lane);
        }
        return;
    }
    if (!(col_ok && group_ok)) m = 0.f;
    m = wave_max(m);
    if (lane == 0) s_abs[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = s_abs[0];
#pragma unroll
        for (int v = 1; v < 8; ++v) t = nmax(t, s_abs[v]);
        absmax_partial[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}

// ---- losses[i] = sum of the GEMM's per-block sums / (T N), in block order
// (one block per candidate: all candidates of a search in ONE launch behind the last product)
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ partial_all, int64_t nblocks, int64_t stride, double inv_count,
                                                          float* __restrict__ loss_all) {
    __shared__ double sm[4];
    const float* partial = partial_all + static_cast<int64_t>(blockIdx.x) * stride;
    float* loss = loss_all + blockIdx.x;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) acc += static_cast<double>(partial[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = static_cast<float>(((sm[0] + sm[1]) + (sm[2] + sm[3])) * inv_count);
}

// ---- first minimum (`loss < best_error`, awq.py:178 / :250)
__global__ void argmin_first_kernel(const float* __restrict__ losses, int n, int32_t* __restrict__ best) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int b = 0;
    float v = INFINITY;
    for (int i = 0; i < n; ++i)
        if (losses[i] < v) { v = losses[i]; b = i; }
    *best = b;
}

// smooth_quant.py:111-113: act^alpha / (w + 1e-9)^(1 - alpha), act clamped to >= 1e-5 (:66-67)
__global__ __launch_bounds__(256) void smooth_scale_kernel(const float* __restrict__ act, const float* __restrict__ wmax, int64_t K, float alpha, float one_minus,
                                                           float* __restrict__ out) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (k >= K) return;
    out[k] = powf(nmax(act[k], 1e-5f), alpha) / powf(wmax[k] + 1e-9f, one_minus);
}

__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ W, int64_t K, int64_t N, int64_t ldw, float* __restrict__ out) {
    __shared__ float sm[4];
    const int64_t k = blockIdx.x;
    float m = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += 256) m = nmax(m, fabsf(W[k * ldw + n]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
}

// Upper bound of |D| = |w - dequant(quant(w s)) / s| for a candidate, from two K-vectors: inside a group the quantized
// value lies in the group's range extended to zero (+ half a step), so |w s - w^| <= 2.07 max_{k' in group} |w_k' s_k'| for
// every element, and dividing by the row's own s_k gives  |D_kn| <= 2.2 max_{k' in group}(rowmax_k' s_k') / min_{k in group} s_k.
// ALL candidates of a search in two launches ahead of the loop (a tiny kernel costs ~5 us of launch each, twice per
// candidate that was 6 % of the search): blockIdx.y = candidate (row scales at row_scales + y * K; nullptr: the clip
// search, whose ten candidates share one bound), one wave per group, then one block per candidate folds the groups into
// the pieces' scale triple [s, 1 / s^2, 1 / s] (the form of syrk_bf16x3.hip::absmax_scale_body).
__global__ __launch_bounds__(64) void awq_bound_kernel(const float* __restrict__ rowmax, const float* __restrict__ row_scales, int64_t K, int64_t g,
                                                       float* __restrict__ bound) {
    const float* row_scale = row_scales != nullptr ? row_scales + static_cast<int64_t>(blockIdx.y) * K : nullptr;
    const int64_t k0 = static_cast<int64_t>(blockIdx.x) * g;
    float top = 0.f, smin = INFINITY;
    for (int64_t k = k0 + threadIdx.x; k < k0 + g; k += 64) {
        const float sk = row_scale != nullptr ? fabsf(row_scale[k]) : 1.0f;
        top = nmax(top, rowmax[k] * sk);
        smin = nmin(smin, sk);
    }
    top = wave_max(top);
    smin = -wave_max(-smin);
    if (threadIdx.x == 0) bound[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = 2.2f * top / smin;
}

__global__ __launch_bounds__(256) void awq_piece_scales_kernel(const float* __restrict__ bound, const int ngroups, float* __restrict__ triples) {
    __shared__ float sm[4];
    const float* b = bound + static_cast<int64_t>(blockIdx.x) * ngroups;
    float m = 0.f;
    for (int i = threadIdx.x; i < ngroups; i += 256) m = nmax(m, b[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
        int ex = 0;
        float sc = 1.0f, inv2 = 1.0f;
        if (m > 0.0f && m < INFINITY) {
            (void)frexpf(m, &ex);                      // m = f * 2^ex, f in [0.5, 1): s m < 2^15
            int e = 15 - ex;
            e = e > 60 ? 60 : (e < -60 ? -60 : e);     // 1 / s^2 must stay a normal float
            sc = ldexpf(1.0f, e);
            inv2 = ldexpf(1.0f, -2 * e);
        }
        float* t = triples + 4 * blockIdx.x;
        t[0] = sc; t[1] = inv2; t[2] = 1.0f / sc; t[3] = 0.f;
    }
}

struct AwqWs {   // carving of the caller's workspace
    char* pieces_x;     // fp16 pieces of X^T (made once)
    char* pieces_d;     // fp16 pieces of D (per candidate)
    float* Ws;          // [K, N] scaled weights
    float* D;           // [K, N]
    uint8_t* q;         // [K, N]
    float* qscale;      // [N * K/g] (>= N, >= 1)
    uint8_t* qzp;
    float* act;         // [K]
    float* wsc;         // [K]
    float* gmax;        // [K/g, N]
    float* colpart;     // [kColChunks, K]
    float* gemm_part;   // [tiles]
    float* diff_part;   // absmax partials of awq_diff_kernel
    float* rowmax;      // [K] max |w| of every row, once per search (the bound of the fused pieces route)
    float* bounds;      // [candidates][K / g] bounds of |D|, and
    float* triples;     // [candidates][4] the pieces' scale triples made from them, once per search
    char* rtn_ws;
    size_t rtn_ws_bytes;
    // Gram route (T >= kAwqGramRatio K): sum_n ||X d_n||^2 = <D, (X^T X) D>, the K x K Gram matrix made ONCE per search
    bool gram;
    float* G;           // [K, K] = (2 / T) X^T X (the GPTQ Hessian kernel, gptq.py:246-260 with n = T)
    char* pieces_g;     // its fp16 pieces as the product's first operand
    char* hess_ws;
    size_t hess_ws_bytes;
};

// The reference's loss needs two [T, K] x [K, N] products per candidate (awq.py:166-177); here ONE product X D with
// D = W - W^.  With a long calibration set (T rows >> K channels: 128 sequences of 2048 tokens against K = 4096) even that is
// T / K times more matrix work than the quadratic form <D, G D> with G = X^T X, which is made once per search by the
// Hessian kernels: per candidate a [K, K] x [K, N] product whose epilogue takes the dot product with D.  G enters with both
// fp16 pieces (22 bits: rounding G to 11 bits would move every column's loss the same way, nothing averages out), D with
// its first (its rounding errors are independent per column): two products.
// Break-even: two products of 2 K^2 N against one of 2 T K N, plus the Gram matrix over ~20 candidates.
#ifndef OQ_AWQ_GRAM_RATIO
#define OQ_AWQ_GRAM_RATIO 3   /* measured on 4096^2 (scripts/lab_awq_routes.py): T = 2 K direct 6.8 / 3.3 ms vs Gram 7.1 / 3.8; T = 3 K 9.0 / 4.4 vs 7.3 / 3.9 */
#endif
constexpr int64_t kAwqGramRatio = OQ_AWQ_GRAM_RATIO;
static bool awq_use_gram(int64_t T, int64_t K) { return T >= kAwqGramRatio * K; }
static int64_t loss_tiles(int64_t T, int64_t K, int64_t N) { return gemm_f16x3_tiles(awq_use_gram(T, K) ? K : T, N); }
static int64_t loss_stride(int64_t T, int64_t K, int64_t N) { return (loss_tiles(T, K, N) + 63) / 64 * 64; }
// all candidates' partial sums -> their losses, one launch
static int32_t finish_losses(const float* gemm_part, int n_cand, int64_t T, int64_t K, int64_t N, float* losses_out, hipStream_t s);

#ifndef OQ_AWQ_HI_ONLY
#define OQ_AWQ_HI_ONLY 1   /* lab: 0 = the three-product (22-bit) loss of round 3 */
#endif
constexpr bool kAwqHiPiecesOnly = OQ_AWQ_HI_ONLY != 0;
#ifndef OQ_AWQ_FUSED_PIECES
#define OQ_AWQ_FUSED_PIECES 1   /* lab: 0 = D in fp32 + the split launch of round 4 */
#endif
constexpr bool kAwqFusedPieces = OQ_AWQ_FUSED_PIECES != 0;
static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// Smallest group the searches take (the reference's own AWQ tests use 8, test/pre_passes/test_awq.py:68); the parameter
// buffers of the workspace are sized for it.
constexpr int kAwqMinGroup = 4;

static int64_t diff_blocks(int64_t K, int64_t N) { return ceil_div(N, 256) * ceil_div(K, 8); }

static size_t awq_workspace(int64_t T, int64_t K, int64_t N, AwqWs* w, char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    const bool gram = awq_use_gram(T, K);
    char* px = take(gram ? 0 : gemm_f16x3_pieces_bytes(K, T));
    char* pd = take(gemm_f16x3_pieces_bytes(K, N));
    char* G = take(gram ? static_cast<size_t>(K) * K * 4 : 0);
    char* pg = take(gram ? gemm_f16x3_pieces_bytes(K, K) : 0);
    const size_t hess_bytes = gram ? oq_hessian_workspace_bytes(T, K) : 0;
    char* hws = take(hess_bytes);
    char* Ws = take(static_cast<size_t>(K) * N * 4);
    char* D = take(static_cast<size_t>(K) * N * 4);
    char* q = take(static_cast<size_t>(K) * N);
    // parameters of the general route: one per group of >= kAwqMinGroup rows (the fused route needs 16 and more)
    char* qs = take(static_cast<size_t>(K) * N * 4 / kAwqMinGroup + static_cast<size_t>(N) * 4 + 1024);
    char* qz = take(static_cast<size_t>(K) * N / kAwqMinGroup + static_cast<size_t>(N) + 1024);
    char* act = take(static_cast<size_t>(K) * 4);
    char* wsc = take(static_cast<size_t>(K) * 4);
    char* gmax = take(static_cast<size_t>(K) * N * 4 / kAwqMinGroup + static_cast<size_t>(N) * 4 + 1024);
    char* colpart = take(static_cast<size_t>(kColChunks) * K * 4);
    char* gpart = take(static_cast<size_t>(kAwqMaxGrid) * loss_stride(T, K, N) * 4 + 1024);   // every candidate its own partial sums
    char* dpart = take(static_cast<size_t>(diff_blocks(K, N)) * 4 + 1024);
    char* rmax = take(static_cast<size_t>(K) * 4 + 256);
    char* bnd = take(static_cast<size_t>(kAwqMaxGrid) * (K / 16 + 1) * 4 + 256);
    char* trp = take(static_cast<size_t>(kAwqMaxGrid) * 16 + 256);
    const size_t rtn_bytes = oq_rtn_workspace_bytes(K, N, OQ_GROUP, kAwqMinGroup, 0) + oq_rtn_workspace_bytes(K, N, OQ_GROUP, 16, 0) +
                             oq_rtn_workspace_bytes(K, N, OQ_TENSOR, -1, 0) + 1024;
    char* rtn = take(rtn_bytes);
    if (w) {
        w->pieces_x = px; w->pieces_d = pd; w->Ws = reinterpret_cast<float*>(Ws); w->D = reinterpret_cast<float*>(D);
        w->q = reinterpret_cast<uint8_t*>(q); w->qscale = reinterpret_cast<float*>(qs); w->qzp = reinterpret_cast<uint8_t*>(qz);
        w->act = reinterpret_cast<float*>(act); w->wsc = reinterpret_cast<float*>(wsc); w->gmax = reinterpret_cast<float*>(gmax);
        w->colpart = reinterpret_cast<float*>(colpart); w->gemm_part = reinterpret_cast<float*>(gpart); w->diff_part = reinterpret_cast<float*>(dpart);
        w->rowmax = reinterpret_cast<float*>(rmax); w->bounds = reinterpret_cast<float*>(bnd); w->triples = reinterpret_cast<float*>(trp);
        w->rtn_ws = rtn; w->rtn_ws_bytes = rtn_bytes;
        w->gram = gram; w->G = reinterpret_cast<float*>(G); w->pieces_g = pg; w->hess_ws = hws; w->hess_ws_bytes = hess_bytes;
    }
    return off + 256;
}

static int32_t param_index(int32_t strategy, int64_t K, int64_t g, ParamIndex* pi) {
    if (strategy == OQ_TENSOR) *pi = ParamIndex{1, 0, 0};
    else if (strategy == OQ_CHANNEL) *pi = ParamIndex{1, 0, 1};
    else *pi = ParamIndex{g, 1, K / g};   // rtn.py:98-109: entry n * (K / g) + kg
    return OQ_OK;
}

// the fused route of candidate_loss: direct product (the Gram route reads D itself in its epilogue), first pieces only, and a
// contraction length that fills its last stage (no zero chunks to write behind the rows)
static bool group_fused(int32_t strategy, int64_t g, int64_t K, int64_t N, int64_t ldw, const float* W) {      // the fused quantize-residual kernel takes the candidate
    return N % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0 && strategy == OQ_GROUP && (g == 16 || g == 32 || g == 64 || g == 128) && K % g == 0;
}
static bool pieces_route(const AwqWs& w, int64_t K) {
    return kAwqFusedPieces && kAwqHiPiecesOnly && !w.gram && gemm_f16x3_chunks(K) * 8 == K;
}

// once per search, fused route: row maxima of W, the bounds and the pieces' scales of all candidates (three launches)
static int32_t prepare_piece_scales(const AwqWs& w, const float* W, int64_t K, int64_t N, int64_t ldw, int64_t g, const float* row_scales, int n_cand,
                                    hipStream_t s) {
    hipLaunchKernelGGL(row_absmax_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N, ldw, w.rowmax);
    const int nc = row_scales != nullptr ? n_cand : 1;
    hipLaunchKernelGGL(awq_bound_kernel, dim3(static_cast<uint32_t>(K / g), static_cast<uint32_t>(nc)), dim3(64), 0, s, w.rowmax, row_scales, K, g, w.bounds);
    hipLaunchKernelGGL(awq_piece_scales_kernel, dim3(static_cast<uint32_t>(nc)), dim3(256), 0, s, w.bounds, static_cast<int>(K / g), w.triples);
    return check_launch("awq piece scales");
}

// one candidate: quantize `Wq` (the weights as the candidate sees them), D = W - dequant (/ s), loss -> loss_out
static int32_t candidate_loss(const AwqWs& w, const float* W, int64_t ldw, const float* row_scale, int64_t T, int64_t K,
                              int64_t N, int32_t qtype, int32_t strategy, int64_t group_size, int64_t g, int32_t symmetric, int32_t reduce_range,
                              float clip_ratio, int candidate, hipStream_t s) {
    int32_t st;
    int nparts;
    const bool vec = N % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0 && (strategy != OQ_GROUP || g % 8 == 0);
    if (vec && strategy == OQ_GROUP && (g == 16 || g == 32 || g == 64 || g == 128) && K % g == 0) {
        QGrid grid;
        st = make_grid(qtype, symmetric, reduce_range, clip_ratio, &grid);
        if (st != OQ_OK) return st;
        const int wpg = static_cast<int>(g / 16), gpb = 8 / wpg;
        const dim3 dgrid(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(ceil_div(K / g, gpb)));
        if (pieces_route(w, K)) {
            // the first pieces of D straight from the quantize-residual kernel (see it); their scale from a bound of |D|
            // (prepare_piece_scales: all candidates ahead of the loop)
            hipLaunchKernelGGL(awq_group_diff_kernel<true>, dgrid, dim3(512), 0, s, W, K, N, ldw, g, grid, wpg, row_scale, static_cast<float*>(nullptr),
                               static_cast<float*>(nullptr), w.triples + 4 * (row_scale != nullptr ? candidate : 0), reinterpret_cast<float*>(w.pieces_d),
                               reinterpret_cast<u32x4a*>(w.pieces_d + gemm_f16x3_header_bytes()), gemm_f16x3_padded_cols(N));
            st = check_launch("awq_group_diff_kernel (pieces)");
            if (st != OQ_OK) return st;
            float* part = w.gemm_part + static_cast<int64_t>(candidate) * loss_stride(T, K, N);
            return launch_gemm_f16x3(w.pieces_x, w.pieces_d, T, N, K, 1.0f, 0.0f, nullptr, 0, part, s, kAwqHiPiecesOnly);
        }
        hipLaunchKernelGGL(awq_group_diff_kernel<false>, dgrid, dim3(512), 0, s, W, K, N, ldw, g, grid, wpg, row_scale, w.D, w.diff_part,
                           static_cast<const float*>(nullptr), static_cast<float*>(nullptr), static_cast<u32x4a*>(nullptr), static_cast<int64_t>(0));
        nparts = static_cast<int>(dgrid.x * dgrid.y);
    } else {
        const float* Wq = W;
        int64_t ldq = ldw;
        if (row_scale != nullptr) {   // awq.py:156: the candidate's weights, materialised for the general RTN entry point
            hipLaunchKernelGGL(scale_rows_kernel, dim3(static_cast<uint32_t>(ceil_div(ceil_div(N, 4), 256)), static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N,
                               ldw, row_scale, w.Ws);
            Wq = w.Ws;
            ldq = N;
        }
        st = rtn_impl(Wq, K, N, ldq, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, 0, w.q, w.qscale, w.qzp, OQ_LAYOUT_KN, w.rtn_ws,
                      w.rtn_ws_bytes, s, true);
        if (st != OQ_OK) return st;
        ParamIndex pi;
        param_index(strategy, K, g, &pi);
        const int32_t is_signed = (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0;
        if (vec) {
            const dim3 dgrid(static_cast<uint32_t>(ceil_div(N, 1024)), static_cast<uint32_t>(ceil_div(K, 8)));
            hipLaunchKernelGGL(awq_diff4_kernel, dgrid, dim3(256), 0, s, W, K, N, ldw, w.q, w.qscale, w.qzp, pi, is_signed, row_scale, w.D, w.diff_part);
            nparts = static_cast<int>(dgrid.x * dgrid.y);
        } else {
            const dim3 dgrid(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(ceil_div(K, 8)));
            hipLaunchKernelGGL(awq_diff_kernel, dgrid, dim3(256), 0, s, W, K, N, ldw, w.q, w.qscale, w.qzp, pi, is_signed, row_scale, w.D, w.diff_part);
            nparts = static_cast<int>(dgrid.x * dgrid.y);
        }
    }
    st = check_launch("awq_diff_kernel");
    if (st != OQ_OK) return st;
    // the pieces of D with the scale from awq_diff_kernel's partial maxima (instead of a second pass over D)
    st = make_f16x2_pieces_from_partials(w.D, K, N, N, w.diff_part, nparts, w.pieces_d, s, kAwqHiPiecesOnly);   // both routes read D's first pieces only
    if (st != OQ_OK) return st;
    // first pieces only: every term of the product carries a relative rounding error <= 2^-10, the loss is a sum of T N
    // squared K-term dot products -- its error (~1e-7 relative, measured against the three-product form) is four orders of
    // magnitude below what separates neighbouring grid points; a third of the matrix work (264 -> ~100 us per candidate)
    float* part = w.gemm_part + static_cast<int64_t>(candidate) * loss_stride(T, K, N);
    if (w.gram)   // <D, G D>, see finish_losses
        return launch_gemm_f16x3(w.pieces_g, w.pieces_d, K, N, K, 1.0f, 0.0f, w.D, N, part, s, false, true, kAwqHiPiecesOnly);   // G 22 bits, D 11
    return launch_gemm_f16x3(w.pieces_x, w.pieces_d, T, N, K, 1.0f, 0.0f, nullptr, 0, part, s, kAwqHiPiecesOnly);
}

static int32_t finish_losses(const float* gemm_part, int n_cand, int64_t T, int64_t K, int64_t N, float* losses_out, hipStream_t s) {
    // direct: mean((X D)^2) = sum / (T N).  Gram: <D, X^T X D> / (T N) = <D, G D> / (2 N) with G = (2 / T) X^T X
    const double inv = awq_use_gram(T, K) ? 1.0 / (2.0 * static_cast<double>(N)) : 1.0 / (static_cast<double>(T) * static_cast<double>(N));
    hipLaunchKernelGGL(loss_finish_kernel, dim3(static_cast<uint32_t>(n_cand)), dim3(256), 0, s, gemm_part, loss_tiles(T, K, N), loss_stride(T, K, N), inv,
                       losses_out);
    return check_launch("loss_finish_kernel");
}

// once per search: the activations as the loss product's first operand -- X^T in pieces, or the Gram matrix in pieces
static int32_t prepare_activations(const AwqWs& w, const float* X, int64_t T, int64_t K, int64_t ldx, void* stream, hipStream_t s) {
    if (!w.gram) return make_f16x2_pieces(X, K, T, ldx, true, w.pieces_x, s);   // pieces over the contraction index k
    int32_t st = oq_hessian_accumulate_f32(X, T, K, ldx, 0, T, w.G, OQ_HESSIAN_AUTO, w.hess_ws, w.hess_ws_bytes, stream);
    if (st != OQ_OK) return st;
    return make_f16x2_pieces(w.G, K, K, K, false, w.pieces_g, s);
}

static int32_t check_common(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                            int64_t group_size, int64_t* g) {
    OQ_REQUIRE(X && W && matrix_ok(T, K, ldx) && matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT, "awq: bad argument");
    OQ_REQUIRE(qtype == OQ_INT4 || qtype == OQ_UINT4 || qtype == OQ_INT8 || qtype == OQ_UINT8, OQ_ERR_UNSUPPORTED, "awq: 4- and 8-bit types only");
    OQ_REQUIRE(strategy == OQ_TENSOR || strategy == OQ_CHANNEL || strategy == OQ_GROUP, OQ_ERR_INVALID_ARGUMENT, "awq: unknown strategy %d", strategy);
    *g = K;
    if (strategy == OQ_GROUP) {
        int64_t gs = group_size > K ? K : group_size;
        if (gs == -1) gs = K;
        OQ_REQUIRE(gs >= kAwqMinGroup && K % gs == 0, OQ_ERR_UNSUPPORTED, "awq: group_size must divide K and be >= %d (got %lld for K = %lld)", kAwqMinGroup,
                   (long long)group_size, (long long)K);
        *g = gs;
    }
    OQ_REQUIRE(ceil_div(K, 8) <= 65535 && K <= 65535 && ceil_div(K, *g) <= 65535, OQ_ERR_UNSUPPORTED, "awq: K too large");
    return OQ_OK;
}

}  // namespace oq

extern "C" {

using namespace oq;

size_t oq_awq_workspace_bytes(int64_t T, int64_t K, int64_t N) {
    if (!oq::matrix_ok(T, K, K) || !oq::matrix_ok(K, N, N) || K > 65535) return 0;
    return awq_workspace(T, K, N, nullptr, nullptr);
}

int32_t oq_awq_scale_search_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw, int32_t qtype,
                                int32_t strategy, int64_t group_size, int32_t symmetric, int32_t reduce_range, int32_t n_grid, float* scales_out,
                                float* losses_out, int32_t* best_out, void* workspace, size_t workspace_bytes, void* stream) {
    int64_t g;
    int32_t st = check_common(X, T, K, ldx, W, N, ldw, qtype, strategy, group_size, &g);
    if (st != OQ_OK) return st;
    OQ_REQUIRE(scales_out && losses_out && best_out && n_grid >= 1 && n_grid <= kAwqMaxGrid, OQ_ERR_INVALID_ARGUMENT, "oq_awq_scale_search_f32: bad argument");
    const size_t need = oq_awq_workspace_bytes(T, K, N);
    OQ_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, OQ_ERR_WORKSPACE,
               "oq_awq_scale_search_f32: 256-byte aligned workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    hipStream_t s = as_stream(stream);
    AwqWs w;
    awq_workspace(T, K, N, &w, static_cast<char*>(workspace));
    // awq.py:47-50: mean |x| per input channel
    const dim3 cgrid(static_cast<uint32_t>(ceil_div(K, 256)), static_cast<uint32_t>(T < kColChunks ? T : kColChunks));
    hipLaunchKernelGGL(col_abs_partial_kernel<false>, cgrid, dim3(256), 0, s, X, T, K, ldx, w.colpart);
    hipLaunchKernelGGL(col_abs_finish_kernel<false>, dim3(cgrid.x), dim3(256), 0, s, w.colpart, static_cast<int>(cgrid.y), K, 1.0f / static_cast<float>(T), w.act);
    // awq.py:52-72: weight scale
    const int64_t kgroups = K / g;
    hipLaunchKernelGGL(group_absmax_kernel, dim3(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(kgroups)), dim3(256), 0, s, W, K, N, ldw, g, w.gmax);
    if (strategy == OQ_TENSOR) hipLaunchKernelGGL(fold_to_scalar_kernel, dim3(1), dim3(1024), 0, s, w.gmax, N);
    hipLaunchKernelGGL(weight_scale_rows_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N, ldw, g, w.gmax, w.wsc);
    hipLaunchKernelGGL(grid_scales_kernel, dim3(static_cast<uint32_t>(n_grid)), dim3(1024), 0, s, w.act, w.wsc, K, n_grid, scales_out);
    st = check_launch("awq statistics");
    if (st != OQ_OK) return st;
    st = prepare_activations(w, X, T, K, ldx, stream, s);
    if (st != OQ_OK) return st;
    if (pieces_route(w, K) && group_fused(strategy, g, K, N, ldw, W)) {
        st = prepare_piece_scales(w, W, K, N, ldw, g, scales_out, n_grid, s);
        if (st != OQ_OK) return st;
    }
    for (int i = 0; i < n_grid; ++i) {
        const float* si = scales_out + static_cast<int64_t>(i) * K;
        st = candidate_loss(w, W, ldw, si, T, K, N, qtype, strategy, group_size, g, symmetric, reduce_range, 1.0f, i, s);
        if (st != OQ_OK) return st;
    }
    st = finish_losses(w.gemm_part, n_grid, T, K, N, losses_out, s);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(64), 0, s, losses_out, n_grid, best_out);
    return check_launch("argmin_first_kernel");
}

int32_t oq_awq_clip_search_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw, int32_t qtype,
                               int32_t strategy, int64_t group_size, int32_t symmetric, int32_t reduce_range, float* losses_out, int32_t* best_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    int64_t g;
    int32_t st = check_common(X, T, K, ldx, W, N, ldw, qtype, strategy, group_size, &g);
    if (st != OQ_OK) return st;
    OQ_REQUIRE(losses_out && best_out, OQ_ERR_INVALID_ARGUMENT, "oq_awq_clip_search_f32: bad argument");
    const size_t need = oq_awq_workspace_bytes(T, K, N);
    OQ_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, OQ_ERR_WORKSPACE,
               "oq_awq_clip_search_f32: 256-byte aligned workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    hipStream_t s = as_stream(stream);
    AwqWs w;
    awq_workspace(T, K, N, &w, static_cast<char*>(workspace));
    st = prepare_activations(w, X, T, K, ldx, stream, s);
    if (st != OQ_OK) return st;
    if (pieces_route(w, K) && group_fused(strategy, g, K, N, ldw, W)) {
        st = prepare_piece_scales(w, W, K, N, ldw, g, nullptr, 10, s);
        if (st != OQ_OK) return st;
    }
    for (int i = 0; i < 10; ++i) {
        const float ratio = static_cast<float>(1.0 - static_cast<double>(i) / 100.0);   // awq.py:227
        st = candidate_loss(w, W, ldw, nullptr, T, K, N, qtype, strategy, group_size, g, symmetric, reduce_range, ratio, i, s);
        if (st != OQ_OK) return st;
    }
    st = finish_losses(w.gemm_part, 10, T, K, N, losses_out, s);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(64), 0, s, losses_out, 10, best_out);
    return check_launch("argmin_first_kernel");
}

// ---- the searches from STREAMED statistics (next row N1 x N2): a calibration walk that consumes its batches as they come cannot
// hand X over, and does not have to -- the loss is <D, G D> with G = (2 / T) X^T X (the Gram route above, which long calibration sets
// take anyway) and the candidate scales need mean |x| per input channel only.  Both are running sums over batches.
static int64_t stats_rows(int64_t K) { return kAwqGramRatio * K; }   // a row count that selects the Gram layout of the workspace

size_t oq_awq_stats_workspace_bytes(int64_t K, int64_t N) {
    if (!oq::matrix_ok(K, N, N) || K > 65535 || !oq::extent_ok(stats_rows(K))) return 0;
    return awq_workspace(stats_rows(K), K, N, nullptr, nullptr);
}

// act_sum[k] = sum over all T calibration rows of |x[t, k]|; G [K, K] = (2 / T) X^T X (oq_hessian_accumulate_f32 with n counting rows)
static int32_t stats_begin(const float* act_sum, const float* G, int64_t T, int64_t K, const float* W, int64_t N, int64_t ldw, int32_t qtype,
                           int32_t strategy, int64_t group_size, int64_t* g, void* workspace, size_t workspace_bytes, AwqWs* w, hipStream_t s,
                           const char* who) {
    OQ_REQUIRE(G && W && T > 0 && count_ok(T, kMaxSamples) && matrix_ok(K, K, K) && matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
    OQ_REQUIRE(qtype == OQ_INT4 || qtype == OQ_UINT4 || qtype == OQ_INT8 || qtype == OQ_UINT8, OQ_ERR_UNSUPPORTED, "awq: 4- and 8-bit types only");
    OQ_REQUIRE(strategy == OQ_TENSOR || strategy == OQ_CHANNEL || strategy == OQ_GROUP, OQ_ERR_INVALID_ARGUMENT, "awq: unknown strategy %d", strategy);
    *g = K;
    if (strategy == OQ_GROUP) {
        int64_t gs = group_size > K ? K : group_size;
        if (gs == -1) gs = K;
        OQ_REQUIRE(gs >= kAwqMinGroup && K % gs == 0, OQ_ERR_UNSUPPORTED, "awq: group_size must divide K and be >= %d (got %lld for K = %lld)", kAwqMinGroup,
                   (long long)group_size, (long long)K);
        *g = gs;
    }
    OQ_REQUIRE(ceil_div(K, 8) <= 65535 && K <= 65535 && ceil_div(K, *g) <= 65535, OQ_ERR_UNSUPPORTED, "awq: K too large");
    const size_t need = oq_awq_stats_workspace_bytes(K, N);
    OQ_REQUIRE(need != 0 && workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, OQ_ERR_WORKSPACE,
               "%s: 256-byte aligned workspace of %zu bytes needed, %zu given", who, need, workspace_bytes);
    awq_workspace(stats_rows(K), K, N, w, static_cast<char*>(workspace));
    OQ_REQUIRE(w->gram, OQ_ERR_LAUNCH, "%s: the Gram layout was not selected", who);
    OQ_REQUIRE(hipMemcpyAsync(w->G, G, static_cast<size_t>(K) * K * 4, hipMemcpyDeviceToDevice, s) == hipSuccess, OQ_ERR_LAUNCH, "%s: copy of G failed", who);
    if (act_sum != nullptr)   // awq.py:47-50: mean |x| per input channel
        hipLaunchKernelGGL(col_abs_finish_kernel<false>, dim3(static_cast<uint32_t>(ceil_div(K, 256))), dim3(256), 0, s, act_sum, 1, K,
                           static_cast<float>(1.0 / static_cast<double>(T)), w->act);
    return make_f16x2_pieces(w->G, K, K, K, false, w->pieces_g, s);
}

int32_t oq_awq_scale_search_stats_f32(const float* act_sum, const float* G, int64_t T, int64_t K, const float* W, int64_t N, int64_t ldw, int32_t qtype,
                                      int32_t strategy, int64_t group_size, int32_t symmetric, int32_t reduce_range, int32_t n_grid, float* scales_out,
                                      float* losses_out, int32_t* best_out, void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(act_sum && scales_out && losses_out && best_out && n_grid >= 1 && n_grid <= kAwqMaxGrid, OQ_ERR_INVALID_ARGUMENT,
               "oq_awq_scale_search_stats_f32: bad argument");
    hipStream_t s = as_stream(stream);
    AwqWs w;
    int64_t g;
    int32_t st = stats_begin(act_sum, G, T, K, W, N, ldw, qtype, strategy, group_size, &g, workspace, workspace_bytes, &w, s, "oq_awq_scale_search_stats_f32");
    if (st != OQ_OK) return st;
    const int64_t Tg = stats_rows(K), kgroups = K / g;
    hipLaunchKernelGGL(group_absmax_kernel, dim3(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(kgroups)), dim3(256), 0, s, W, K, N, ldw, g, w.gmax);
    if (strategy == OQ_TENSOR) hipLaunchKernelGGL(fold_to_scalar_kernel, dim3(1), dim3(1024), 0, s, w.gmax, N);
    hipLaunchKernelGGL(weight_scale_rows_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N, ldw, g, w.gmax, w.wsc);
    hipLaunchKernelGGL(grid_scales_kernel, dim3(static_cast<uint32_t>(n_grid)), dim3(1024), 0, s, w.act, w.wsc, K, n_grid, scales_out);
    st = check_launch("awq statistics");
    if (st != OQ_OK) return st;
    for (int i = 0; i < n_grid; ++i) {
        st = candidate_loss(w, W, ldw, scales_out + static_cast<int64_t>(i) * K, Tg, K, N, qtype, strategy, group_size, g, symmetric, reduce_range, 1.0f, i, s);
        if (st != OQ_OK) return st;
    }
    st = finish_losses(w.gemm_part, n_grid, Tg, K, N, losses_out, s);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(64), 0, s, losses_out, n_grid, best_out);
    return check_launch("argmin_first_kernel");
}

int32_t oq_awq_clip_search_stats_f32(const float* G, int64_t T, int64_t K, const float* W, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                                     int64_t group_size, int32_t symmetric, int32_t reduce_range, float* losses_out, int32_t* best_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(losses_out && best_out, OQ_ERR_INVALID_ARGUMENT, "oq_awq_clip_search_stats_f32: bad argument");
    hipStream_t s = as_stream(stream);
    AwqWs w;
    int64_t g;
    int32_t st = stats_begin(nullptr, G, T, K, W, N, ldw, qtype, strategy, group_size, &g, workspace, workspace_bytes, &w, s, "oq_awq_clip_search_stats_f32");
    if (st != OQ_OK) return st;
    const int64_t Tg = stats_rows(K);
    for (int i = 0; i < 10; ++i) {
        const float ratio = static_cast<float>(1.0 - static_cast<double>(i) / 100.0);   // awq.py:227
        st = candidate_loss(w, W, ldw, nullptr, Tg, K, N, qtype, strategy, group_size, g, symmetric, reduce_range, ratio, i, s);
        if (st != OQ_OK) return st;
    }
    st = finish_losses(w.gemm_part, 10, Tg, K, N, losses_out, s);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(64), 0, s, losses_out, 10, best_out);
    return check_launch("argmin_first_kernel");
}

// sum over the T rows of |x[t, k]| added to `sum_inout` [K] (a running statistic over calibration batches; awq.py:47-50 divides by T)
size_t oq_abs_sum_cols_workspace_bytes(int64_t K) {
    if (!oq::extent_ok(K)) return 0;
    return oq::align256(static_cast<size_t>(oq::kColChunks) * K * 4) + 512;
}

int32_t oq_abs_sum_cols_f32(const float* X, int64_t T, int64_t K, int64_t ldx, float* sum_inout, int32_t accumulate, void* workspace,
                            size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(X && sum_inout && matrix_ok(T, K, ldx), OQ_ERR_INVALID_ARGUMENT, "oq_abs_sum_cols_f32: bad argument");
    const size_t need = oq_abs_sum_cols_workspace_bytes(K);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_abs_sum_cols_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    hipStream_t s = as_stream(stream);
    char* base = static_cast<char*>(workspace);
    base += (256 - reinterpret_cast<uintptr_t>(base) % 256) % 256;
    float* colpart = reinterpret_cast<float*>(base);
    const dim3 cgrid(static_cast<uint32_t>(ceil_div(K, 256)), static_cast<uint32_t>(T < kColChunks ? T : kColChunks));
    hipLaunchKernelGGL(col_abs_partial_kernel<false>, cgrid, dim3(256), 0, s, X, T, K, ldx, colpart);
    hipLaunchKernelGGL(col_abs_accumulate_kernel, dim3(cgrid.x), dim3(256), 0, s, colpart, static_cast<int>(cgrid.y), K, accumulate, sum_inout);
    return check_launch("col_abs_accumulate_kernel");
}

int32_t oq_smooth_quant_scale_f32(const float* X, int64_t T, int64_t K, int64_t ldx, const float* W, int64_t N, int64_t ldw, float alpha, float* scale_out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(X && W && scale_out && matrix_ok(T, K, ldx) && matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT,
               "oq_smooth_quant_scale_f32: bad argument");
    const size_t need = align256(static_cast<size_t>(kColChunks) * K * 4) + 2 * align256(static_cast<size_t>(K) * 4) + 256;
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_smooth_quant_scale_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    hipStream_t s = as_stream(stream);
    char* base = static_cast<char*>(workspace);
    base += (256 - reinterpret_cast<uintptr_t>(base) % 256) % 256;
    float* colpart = reinterpret_cast<float*>(base);
    float* act = reinterpret_cast<float*>(base + align256(static_cast<size_t>(kColChunks) * K * 4));
    float* wmax = act + align256(static_cast<size_t>(K) * 4) / 4;
    const dim3 cgrid(static_cast<uint32_t>(ceil_div(K, 256)), static_cast<uint32_t>(T < kColChunks ? T : kColChunks));
    hipLaunchKernelGGL(col_abs_partial_kernel<true>, cgrid, dim3(256), 0, s, X, T, K, ldx, colpart);                 // smooth_quant.py:62-69
    hipLaunchKernelGGL(col_abs_finish_kernel<true>, dim3(cgrid.x), dim3(256), 0, s, colpart, static_cast<int>(cgrid.y), K, 1.0f, act);
    hipLaunchKernelGGL(row_absmax_kernel, dim3(static_cast<uint32_t>(K)), dim3(256), 0, s, W, K, N, ldw, wmax);      // :71-74
    hipLaunchKernelGGL(smooth_scale_kernel, dim3(static_cast<uint32_t>(ceil_div(K, 256))), dim3(256), 0, s, act, wmax, K, alpha,
                       static_cast<float>(1.0 - static_cast<double>(alpha)), scale_out);
    return check_launch("smooth_scale_kernel");
}

size_t oq_smooth_quant_workspace_bytes(int64_t K) {
    if (!oq::extent_ok(K)) return 0;
    return oq::align256(static_cast<size_t>(oq::kColChunks) * K * 4) + 2 * oq::align256(static_cast<size_t>(K) * 4) + 512;
}

}  // extern "C"
