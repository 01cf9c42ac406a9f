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

#include <cstdlib>
#include <utility>

namespace oq {

constexpr int kLoopMaxRows = 128;

struct LoopArgs {
    float* W;          // [K, N] working copy: the rows the per-group parameters are derived from (gptq.py:168-184 reads `W`)
    const float* Wt;   // where the rows of the current sub-block come from (gptq.py:157 `W1`): row r at Wt + (r - wt_row0) * N.
    int64_t wt_row0;   //   = W, 0 unless block_size > 128: then the block's working copy, which alone receives the in-block updates
    int64_t err_row0;  // row of the Err buffer that belongs to row i1 (the buffer holds a whole super-block)
    const float* coef; // rows-over-lanes kernel: this launch's coefficient image (gptq_coef_image_kernel), kCoefFloats floats
    int32_t params_from_tile;  // the rows a group's parameters are read from are the launch's own rows when the group spans the launch
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
    int32_t packed4;          // OQ_LAYOUT_KN_PACKED4: q_int is [K, N/2], two columns per byte (core/_pack.py:8-22 order)
    int64_t pre_first_group;  // index (row / g) of the first group that starts in this block
};

// The integer of (row, column c): one byte of [K, N], or its nibble of byte (row * N + c) / 2 -- the even column's lane takes its
// right-hand neighbour's level (XOR = lane distance between neighbouring columns: 1, or 16 where a column is a DPP row) and
// stores both.  N is even (checked by the host), so a pair is in or out together; the low nibble of the two's-complement byte IS
// the two's-complement nibble.  Every lane of the pair must reach this call.
template <int XOR>
__device__ __forceinline__ void store_level(const LoopArgs& a, int64_t row, int64_t c, bool live, uint32_t level) {
    if (!a.packed4) {
        if (live) a.q_int[row * a.N + c] = static_cast<uint8_t>(level);
        return;
    }
    const uint32_t partner = static_cast<uint32_t>(__shfl_xor(static_cast<int>(level), XOR, 64));
    if (live && (c & 1) == 0) a.q_int[(row * a.N + c) >> 1] = static_cast<uint8_t>((level & 0xfu) | ((partner & 0xfu) << 4));
}

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
            x[u] = a.Wt[(a.i1 - a.wt_row0 + (i < count ? i : count - 1)) * a.N + cc];
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
                    store_level<1>(a, row, c, live, static_cast<uint32_t>(qi));
                    if (live) {
                        a.q_deq[row * a.N + c] = q;
                        a.err[(a.err_row0 + s0 + i) * a.N + c] = e;
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

// ---------------------------------------------------------------------------------------------------------------------
// CORRECTED mode, round 3: rows over lanes.
//
// The sequential kernel above gives a lane one column and ALL rows of a sub-block: N / 64 waves (64 at N = 4096) walk
// the K steps while 15/16 of the chip idles, and every step is ~100 dependent vector instructions.  Here 16 lanes (one
// DPP row) share a column and hold the 128 rows of the block interleaved (lane r holds rows r, 16 + r, ...: 8 registers),
// so N / 4 waves -- one per SIMD at N = 4096 -- walk the steps, and a step is
//     v_mov_dpp row_newbcast:R        the row's value to the 16 lanes of its column
//     mul, fma x 4                    y = fl(w / scale), the IEEE quotient from the refined reciprocal of scale (below)
//     add, med3                       + (1.5 * 2^23 + zp): round-half-even and zero point in one rounding; clamp
//     sub, mul, sub                   dequantize, residual
//     mul, fma x 4                    the IEEE quotient (w - q) / d from the refined reciprocal of d
//     (LIVE + 1) / 2 x pk_mul, pk_add every lane updates its own later rows (product, then subtraction: gptq.py:198-200)
// straight-line, no branch; the row's owner picks its (error, level) out of the 16 steps' values after the slab.
// ~28 instructions = ~110 cycles with one wave per SIMD, against ~10 k cycles per row in round 1 and 600 in round 2.
// The block is walked as 8 slabs of 16 rows; after a slab every lane stores the error / integer / dequantized value of
// its own row and the registers rotate (w[k] = w[k + 1]), so the current slab is always w[0] and the code of a slab is
// specialised only by the number of later slabs (8 instantiations).
//
// Division without the ~11-instruction v_div_* expansion on the chain: for d = U[i][i] (one per row, wave-uniform) the
// block prologue computes r0 = rcp(d), r1 = fma(fma(-d, r0, 1), r0, r0) -- the first four instructions of the compiler's
// own fp32 division sequence -- and the step runs its remaining five on the numerator (mul, fma, fma, fma, fma).  That IS
// the correctly rounded quotient whenever v_div_scale would not rescale (numerator and quotient in [2^-100, 2^100]);
// outside (degenerate data) the result may differ from IEEE in the last bit.
// Requires group_size <= 0 or group_size % 16 == 0 and i1 % 16 == 0 (groups start at slab starts); the host falls back
// to gptq_block_kernel otherwise.
// ---------------------------------------------------------------------------------------------------------------------
constexpr float kMagicF = 12582912.0f;       // 1.5 * 2^23: fp32 spacing 1 around it, level = low byte of the bits

template <int R>
__device__ __forceinline__ float bcast16(float v) {   // lane R of every row of 16 lanes to all lanes of that row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + R /* row_newbcast:R */, 0xf, 0xf, true));
}
template <int S>
__device__ __forceinline__ float ror16(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + S /* row_ror:S */, 0xf, 0xf, true));
}

struct ColState {
    float scale, neg_scale, rinv, zm;   // rinv = refined reciprocal of scale (div_refined), zm = float(zp) + 1.5 * 2^23
    int32_t zp;
};
__device__ __forceinline__ ColState make_colstate(float scale, int32_t zp) {
    ColState c;
    c.scale = scale;
    c.neg_scale = -scale;
    c.zp = zp;
    c.rinv = refined_rcp(scale);
    c.zm = static_cast<float>(zp) + kMagicF;
    return c;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image of the block's coefficients, three planes (kCoefFloats floats = 64 KB):
//   A [128 rows][64]  coefficient of row i's error for block row j = 16 * (i / 16 + m) + r at [i][r * 4 + m],  m = 0..3
//   B [ 64 rows][64]  the same for m = 4..7 at [i][r * 4 + (m - 4)]  (rows 64.. have no such block rows)
//   D [128 rows][32]  (-d, refined 1 / d) of row i, once per r (so that a lane reads it at a fixed offset from its own base)
// m = 0 is the slab of row i itself.  A lane reads one step's coefficients as two 16-byte vectors + one 8-byte vector,
// all conflict-free, at immediate offsets from two address registers.
constexpr int kCoefA = 0, kCoefB = kLoopMaxRows * 64, kCoefD = kCoefB + 64 * 64, kCoefFloats = kCoefD + kLoopMaxRows * 32;

struct SlabIn {
    const float* crow;    // LDS: plane A, row of the slab's first step, at this lane's r (plane B: + kCoefB)
    const float* drow;    // LDS: plane D, the same
    ColState cs;
    float lo_m, hi_m;
    int r16, rows;
};

struct StepCoef {   // what one step reads from LDS
    f32x4 ca, cb;
    f32x2 dd;
};
template <int LIVE, int R>
__device__ __forceinline__ void load_coef(StepCoef& c, const SlabIn& in) {
    c.ca = *reinterpret_cast<const f32x4*>(in.crow + R * 64);
    if constexpr (LIVE >= 4) c.cb = *reinterpret_cast<const f32x4*>(in.crow + kCoefB + R * 64);
    c.dd = *reinterpret_cast<const f32x2*>(in.drow + R * 32);
}

// w2[p] = block rows of slots (2p, 2p + 1) relative to the current slab: packed mul / add, two rows per instruction.
// Straight-line: no branch, no VALU -> SALU hand-off inside a step (a ballot + taken branch per step cost more than the
// arithmetic it guarded).
constexpr int kCoefAhead = 2;   // LDS reads are issued this many steps ahead
template <int LIVE, bool RAGGED, int R>
__device__ __forceinline__ void row_step(f32x2 (&w2)[4], const SlabIn& in, StepCoef (&ring)[kCoefAhead + 1], float& save_e, float& save_cl) {
    if constexpr (RAGGED) {
        if (R >= in.rows) return;   // uniform
    }
    // software pipeline: the LDS reads of step R + kCoefAhead are issued before this step's chain.  The sched_barrier at
    // the end keeps the compiler from hoisting all 16 steps' reads to the top of the slab.
    if constexpr (R + kCoefAhead < 16) load_coef<LIVE, R + kCoefAhead>(ring[(R + kCoefAhead) % (kCoefAhead + 1)], in);
    const StepCoef& cur = ring[R % (kCoefAhead + 1)];
    const f32x4 ca = cur.ca, cb = cur.cb;
    const f32x2 dd = cur.dd;
    const float wi = bcast16<R>(w2[0].x);
    // K1 (gptq.py:186-188 = utils.py:72-79): y = fl(wi / s) (IEEE); adding M + zp (M = 1.5 * 2^23: fp32 spacing 1) rounds y
    // half-to-even and adds the zero point in one correctly rounded operation; the clamp happens on M + level.
    const float y = div_refined(wi, in.cs.neg_scale, in.cs.rinv);
    const float cl = __builtin_amdgcn_fmed3f(y + in.cs.zm, in.lo_m, in.hi_m);
    const float q = (cl - in.cs.zm) * in.cs.scale;                      // gptq.py:189 = utils.py:130-132
    const float t = wi - q;
    const float e = div_refined(t, dd.x, dd.y);                         // gptq.py:197: err = (w - q) / d
    const bool mine = in.r16 == R;
    save_e = mine ? e : save_e;
    save_cl = mine ? cl : save_cl;
    // gptq.py:198-200: W1[i:, :] -= outer(U[i, i:], err): one rounding for the product, one for the subtraction
    const f32x2 ee = {e, e};
    w2[0] = w2[0] - f32x2{ca.x, ca.y} * ee;
    if constexpr (LIVE >= 2) w2[1] = w2[1] - f32x2{ca.z, ca.w} * ee;
    if constexpr (LIVE >= 4) w2[2] = w2[2] - f32x2{cb.x, cb.y} * ee;
    if constexpr (LIVE >= 6) w2[3] = w2[3] - f32x2{cb.z, cb.w} * ee;
    // Pin the updated rows here.  Nothing reads rows of later slabs before their own slab, so the optimiser sinks their 16
    // updates to the end of the slab and keeps 16 steps' coefficients alive meanwhile (140 spilled registers).
    asm volatile("" : "+v"(w2[0]));
    if constexpr (LIVE >= 2) asm volatile("" : "+v"(w2[1]));
    if constexpr (LIVE >= 4) asm volatile("" : "+v"(w2[2]));
    if constexpr (LIVE >= 6) asm volatile("" : "+v"(w2[3]));
    __builtin_amdgcn_sched_barrier(0);
}

template <int LIVE, bool RAGGED, int... Rs>
__device__ __forceinline__ void slab_steps(f32x2 (&w2)[4], const SlabIn& in, float& save_e, float& save_cl, std::integer_sequence<int, Rs...>) {
    StepCoef ring[kCoefAhead + 1];
#pragma unroll
    for (int i = 0; i <= kCoefAhead; ++i) ring[i].cb = f32x4{0.f, 0.f, 0.f, 0.f};
    load_coef<LIVE, 0>(ring[0], in);
    load_coef<LIVE, 1>(ring[1], in);
    static_assert(kCoefAhead == 2, "prologue loads steps 0 .. kCoefAhead - 1");
    (row_step<LIVE, RAGGED, Rs>(w2, in, ring, save_e, save_cl), ...);
}
template <int LIVE, bool RAGGED>
__device__ __forceinline__ void slab16(f32x2 (&w2)[4], const SlabIn& in, float& save_e, float& save_cl) {
    slab_steps<LIVE, RAGGED>(w2, in, save_e, save_cl, std::make_integer_sequence<int, 16>{});
}

// The LDS image of every launch of a call (layout above), built once: launch `ord` walks rows [i1, i1 + cnt) with
// i1 = (ord / per_block) * bs + (ord % per_block) * 128 (blocks of `bs` rows, taller ones as chains of 128-row launches).
// One thread per float: blockIdx.x = row i, 160 threads = A (64) | B (64) | D (32).
__global__ __launch_bounds__(160) void gptq_coef_image_kernel(const float* __restrict__ U, int64_t K, int64_t bs, int per_block,
                                                               float* __restrict__ image) {
    const int64_t ord = blockIdx.y;
    const int i = blockIdx.x, pos = threadIdx.x;
    const int64_t sub = ord % per_block;
    const int64_t i1 = (ord / per_block) * bs + sub * kLoopMaxRows;
    int64_t cnt = bs - sub * kLoopMaxRows;
    if (cnt > kLoopMaxRows) cnt = kLoopMaxRows;
    if (cnt > K - i1) cnt = K - i1;
    float* img = image + ord * kCoefFloats;
    if (pos < 128) {
        const int r = (pos & 63) >> 2, m = (pos & 3) + 4 * (pos >> 6);
        const int j = 16 * ((i >> 4) + m) + r;
        const float v = (i < cnt && j < cnt) ? U[(i1 + i) * K + i1 + j] : 0.0f;
        if (pos < 64) img[kCoefA + i * 64 + pos] = v;
        else if (i < 64) img[kCoefB + i * 64 + (pos - 64)] = v;
    } else {
        const int64_t ic = i < cnt ? i : cnt - 1;
        const float d = U[(i1 + ic) * K + i1 + ic];
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float f0 = __builtin_fmaf(-d, r0, 1.0f);
        img[kCoefD + i * 32 + (pos - 128)] = (pos & 1) == 0 ? -d : __builtin_fmaf(f0, r0, r0);
    }
}

#ifdef OQ_LOOP_STAMPS   // lab builds only (python -m onnx_quantize_amd._build --define OQ_LOOP_STAMPS): s_memtime per phase
__device__ unsigned long long g_loop_stamps[16];
#define OQ_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_loop_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OQ_STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(1024) void gptq_rows16_kernel(const LoopArgs a) {
    __shared__ float P[kCoefFloats];   // 64 KB
    OQ_STAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, c4 = lane >> 4;
    const int64_t c = (static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + wave) * 4 + c4;
    const bool live = c < a.N;
    const int64_t cc = live ? c : a.N - 1;
    const int count = static_cast<int>(a.count);
    const int nslabs = (count + 15) >> 4;

    // ---- prologue: ONE round trip to memory (a launch is ~15 us, a round trip ~1 us with every workgroup starting at once).
    // Issued back to back: the block's rows (lane (r16, c4) holds rows 16 k + r16 of column c), the carried parameters, then
    // the launch's coefficient image (built once per call by gptq_coef_image_kernel; every workgroup reads the same 64 KB)
    // through registers into LDS, all of a thread's 16-byte vectors in flight at once (16 with 256 threads).  LDS-DMA needs no registers but ~170
    // cycles of ISSUE per 1 KB piece here (4150 cycles for a wave's 24 pieces, in-kernel stamps); loads that are predicated
    // get an exec branch and a vmcnt(0) each and run one after the other (19000 cycles): everything below is unconditional,
    // indices past the end are clamped (the same bytes are written again).
    f32x2 w2[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int row = 16 * k + r16;
        const float x = a.Wt[(a.i1 - a.wt_row0 + (row < count ? row : count - 1)) * a.N + cc];
        if (k & 1) w2[k >> 1].y = x; else w2[k >> 1].x = x;
    }
    OQ_STAMP(14);
    float scale;
    int32_t zp;
    if (a.i1 == 0) {  // gptq.py:104-116
        const int64_t pi = a.init_count == 1 ? 0 : cc;
        scale = a.init_scale[pi];
        zp = a.init_zp[pi];
    } else {
        scale = a.carry_scale[cc];
        zp = a.carry_zp[cc];
    }
    {
        constexpr int kVec = kCoefFloats / 4;               // 16-byte vectors in the image
        constexpr int kBatch = 16;                          // 256 threads: the whole image in one batch
        const float4* src = reinterpret_cast<const float4*>(a.coef);
        float4* dst = reinterpret_cast<float4*>(P);
        const int nthr = static_cast<int>(blockDim.x);
        for (int base = threadIdx.x; base < kVec; base += kBatch * nthr) {
            float4 v[kBatch];
            int idx[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                idx[u] = base + u * nthr < kVec ? base + u * nthr : kVec - 1;
                v[u] = src[idx[u]];
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) dst[idx[u]] = v[u];
        }
    }
    OQ_STAMP(13);
    SlabIn in;
    in.cs = make_colstate(scale, zp);
    in.lo_m = static_cast<float>(a.grid.qmin) + kMagicF;
    in.hi_m = static_cast<float>(a.grid.qmax) + kMagicF;
    in.r16 = r16;
    int64_t grp = a.g > 0 ? a.i1 / a.g : 0;
    int64_t rem = a.g > 0 ? a.i1 - grp * a.g : 1;   // position of the slab's first row inside its group
    OQ_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this wave's LDS-DMAs have landed
    OQ_STAMP(2);
    __syncthreads();
    OQ_STAMP(3);

    for (int k0 = 0; k0 < nslabs; ++k0) {
        const int64_t row0 = a.i1 + 16 * k0;
        if (a.g > 0 && rem == 0) {
            // gptq.py:168-184: a group starts here: parameters from rows [row0, row0 + g) of W (channel strategy), or the
            // ones the MSE search left
            if (a.pre_scale != nullptr) {
                const int64_t o = (grp - a.pre_first_group) * a.N + cc;
                scale = a.pre_scale[o];
                zp = a.zp_signed ? static_cast<int32_t>(static_cast<int8_t>(a.pre_zp[o])) : static_cast<int32_t>(a.pre_zp[o]);
            } else {
                float mn = INFINITY, mx = -INFINITY;
                if (a.params_from_tile && k0 == 0 && a.g == count) {
                    // the group is exactly this launch's rows, still untouched in the registers (rows past `count` are
                    // duplicates of the last one)
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) {
                        mn = nmin(mn, nmin(w2[p2].x, w2[p2].y));
                        mx = nmax(mx, nmax(w2[p2].x, w2[p2].y));
                    }
                } else {
                    const int64_t rend = row0 + a.g < a.K ? row0 + a.g : a.K;
                    for (int64_t r = row0 + r16; r < rend; r += 128) {
                        float x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = a.W[(r + 16 * u < rend ? r + 16 * u : rend - 1) * a.N + cc];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { mn = nmin(mn, x[u]); mx = nmax(mx, x[u]); }
                    }
                }
                mn = nmin(mn, ror16<8>(mn)); mx = nmax(mx, ror16<8>(mx));
                mn = nmin(mn, ror16<4>(mn)); mx = nmax(mx, ror16<4>(mx));
                mn = nmin(mn, ror16<2>(mn)); mx = nmax(mx, ror16<2>(mx));
                mn = nmin(mn, ror16<1>(mn)); mx = nmax(mx, ror16<1>(mx));
                const QParam p = qparam_from_minmax(mn, mx, a.grid);
                scale = p.scale;
                zp = p.zp;
            }
            in.cs = make_colstate(scale, zp);
            if (live && r16 == 0 && a.used_scale != nullptr) {
                a.used_scale[grp * a.N + c] = scale;
                a.used_zp[grp * a.N + c] = zp;
            }
        }
        in.crow = P + kCoefA + (16 * k0) * 64 + r16 * 4;
        in.drow = P + kCoefD + (16 * k0) * 32 + r16 * 2;
        in.rows = count - 16 * k0 < 16 ? count - 16 * k0 : 16;
        float save_e = 0.0f, save_cl = kMagicF;
        if (in.rows < 16) {
            slab16<0, true>(w2, in, save_e, save_cl);
        } else {
            switch (nslabs - 1 - k0) {
                case 0: slab16<0, false>(w2, in, save_e, save_cl); break;
                case 1: slab16<1, false>(w2, in, save_e, save_cl); break;
                case 2: slab16<2, false>(w2, in, save_e, save_cl); break;
                case 3: slab16<3, false>(w2, in, save_e, save_cl); break;
                case 4: slab16<4, false>(w2, in, save_e, save_cl); break;
                case 5: slab16<5, false>(w2, in, save_e, save_cl); break;
                case 6: slab16<6, false>(w2, in, save_e, save_cl); break;
                default: slab16<7, false>(w2, in, save_e, save_cl); break;
            }
        }
        store_level<16>(a, row0 + r16, c, live && r16 < in.rows, __float_as_uint(save_cl) & 0xffu);   // lanes 16 apart: columns c, c + 1, same row
        if (live && r16 < in.rows) {
            const int64_t row = row0 + r16;
            a.err[(a.err_row0 + 16 * k0 + r16) * a.N + c] = save_e;
            a.q_deq[row * a.N + c] = (save_cl - in.cs.zm) * in.cs.scale;
        }
        OQ_STAMP(4 + k0);
        // the next slab becomes slot 0
        w2[0] = f32x2{w2[0].y, w2[1].x};
        w2[1] = f32x2{w2[1].y, w2[2].x};
        w2[2] = f32x2{w2[2].y, w2[3].x};
        w2[3] = f32x2{w2[3].y, w2[3].y};
        if (a.g > 0) {
            rem += 16;
            if (rem >= a.g) { rem = 0; ++grp; }
        }
    }
    if (live && r16 == 0) {
        a.carry_scale[c] = scale;
        a.carry_zp[c] = zp;
    }
    OQ_STAMP(12);
}

// ---------------------------------------------------------------------------------------------------------------------
// The launch's errors to the remaining rows of its super-block: C[m, n] -= sum_k A[k][m] * Err[k][n], k < Kd <= 128,
// m < M <= 384.  gemm_tn_kernel's 128 x 128 tiles put such a product on 32-96 workgroups, each MFMA-bound for 7.8 us
// (128 x 128 x 128 on one CU's fp32 matrix rate) plus a pipelined stage loop that never gets going in 4 stages: 21-35 us
// measured.  Here a workgroup owns 64 x 64 (4 waves x 32 x 32: ONE v_mfma_f32_32x32x2_f32 per k-pair and wave), all Kd rows
// of both operands are staged at once (64 KB of LDS, two workgroups per CU), and the grid is (M / 64) x (N / 64) = 128-384
// workgroups for N = 4096.  Same arithmetic as gemm_tn_kernel (an fmaf chain over k in ascending order, then C - sum).
// Requires 16-byte aligned operands and rows (the host falls back to gemm_tn_kernel otherwise).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPanelTile = 64;
typedef float f32x16p __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void panel_update_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc, int64_t M, int64_t N, int Kd, int ntn) {
    __shared__ float sA[kLoopMaxRows][kPanelTile];
    __shared__ float sB[kLoopMaxRows][kPanelTile];
    const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x - tile_m * ntn;
    const int64_t m0 = static_cast<int64_t>(tile_m) * kPanelTile, n0 = static_cast<int64_t>(tile_n) * kPanelTile;
    const int t = threadIdx.x;
    {   // 128 rows x 64 floats per operand = 2048 float4: 8 per thread and operand, all in flight; clamped, never predicated
        const int c4 = (t & 15) * 4, r0 = t >> 4;
        const int64_t ca = m0 + c4 < M ? m0 + c4 : M - 4, cb = n0 + c4 < N ? n0 + c4 : N - 4;
        float4 va[8], vb[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = r0 + 16 * p;
            const int kc = k < Kd ? k : Kd - 1;
            va[p] = *reinterpret_cast<const float4*>(A + kc * lda + ca);
            vb[p] = *reinterpret_cast<const float4*>(B + kc * ldb + cb);
        }
        const bool oka = m0 + c4 < M, okb = n0 + c4 < N;
        auto keep = [](float4 x, bool ok) {   // zeros outside the operands, on the bits (a select between two float4 goes through scratch)
            const uint32_t m = ok ? 0xffffffffu : 0u;
            return make_float4(__uint_as_float(__float_as_uint(x.x) & m), __uint_as_float(__float_as_uint(x.y) & m),
                               __uint_as_float(__float_as_uint(x.z) & m), __uint_as_float(__float_as_uint(x.w) & m));
        };
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = r0 + 16 * p;
            *reinterpret_cast<float4*>(&sA[k][c4]) = keep(va[p], oka && k < Kd);
            *reinterpret_cast<float4*>(&sB[k][c4]) = keep(vb[p], okb && k < Kd);
        }
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, kl = lane >> 5, cl = lane & 31;
    f32x16p acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int kd8 = (Kd + 7) & ~7;   // rows past Kd are zeros (all 128 rows of the LDS tiles were written)
    for (int k = 0; k < kd8; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; u += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[k + u + kl][wm * 32 + cl], sB[k + u + kl][wn * 32 + cl], acc, 0, 0, 0);
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int64_t col = n0 + wn * 32 + cl;
    if (col < N) {
        float cv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
            cv[e] = C[(row < M ? row : M - 1) * ldc + col];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
            const float v = -1.0f * acc[e];                      // gemm_tn_kernel's epilogue with alpha = -1, beta = 1
            if (row < M) C[row * ldc + col] = 1.0f * cv[e] + v;
        }
    }
}

static bool panel_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t M, int64_t N, int64_t Kd) {
    return Kd >= 1 && Kd <= kLoopMaxRows && M >= 4 && N >= 4 && M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(B) & 15u) == 0 && (reinterpret_cast<uintptr_t>(C) & 3u) == 0 &&
           ceil_div(M, kPanelTile) * ceil_div(N, kPanelTile) < (1 << 30);
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
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    for (int64_t r = r0; r < r1; ++r) {   // lanes past N walk along on a clamped column (the packed layout pairs neighbouring lanes)
        const int32_t qi = quantize_one(a.W[r * a.N + cc], scale, zp, qmin, qmax);  // gptq.py:186-188
        store_level<1>(a, r, c, live, static_cast<uint32_t>(qi));
        if (live) a.q_deq[r * a.N + c] = dequantize_one(qi, scale, zp);             // :189
    }
}

int32_t rtn_impl(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy, int64_t group_size,
                 int32_t symmetric, int32_t reduce_range, float clip_ratio, int32_t mse, void* q_out, float* scale_out, void* zp_out,
                 int32_t layout, void* workspace, size_t workspace_bytes, void* stream, bool emit_q);

}  // namespace oq

extern "C" {

using namespace oq;

// Rows whose lazy batch update (gptq.py:208) is deferred to ONE GEMM.  block_size <= 128: up to kSuperRows rows = several
// reference blocks (two-level blocking, see oq_gptq_loop_f32); taller blocks: the block itself.
constexpr int64_t kSuperRows = 512;

static int64_t super_rows_bound(int64_t K, int64_t block_size) {
    const int64_t s = block_size > kSuperRows ? block_size : kSuperRows;
    return s < K ? s : K;
}

static int64_t launches_per_block(int64_t block_size) { return block_size > kLoopMaxRows ? ceil_div(block_size, kLoopMaxRows) : 1; }
static size_t coef_image_bytes(int64_t K, int64_t block_size) {
    return static_cast<size_t>(ceil_div(K, block_size) * launches_per_block(block_size)) * kCoefFloats * sizeof(float);
}

// operand pieces of the lazy batch update behind a super-block on the fp16-piece GEMM (gemm_tn.hpp): U rows x all columns, Err
static size_t piece_gemm_bytes(int64_t K, int64_t N, int64_t block_size) {
    const int64_t s = super_rows_bound(K, block_size);
    return gemm_f16x3_pieces_bytes(s, K) + gemm_f16x3_pieces_bytes(s, N) + 1024;
}

size_t oq_gptq_loop_workspace_bytes(int64_t K, int64_t N, int64_t block_size) {
    if (!matrix_ok(K, N, N) || K > kMaxHessianWidth || block_size <= 0 || block_size > kMaxExtent) return 0;
    // Err [super-block rows, N] + carried (scale, zp) [N] + (mse) per-group parameters of one launch [128, N] x (4 + 1) B
    // + (block_size > 128) the block's working copy [block_size, N] + the MSE search's own workspace for one [group, N] slice
    // + the coefficient images of all launches (rows-over-lanes kernel)
    const size_t tall = block_size > kLoopMaxRows ? static_cast<size_t>(block_size < K ? block_size : K) * N * 4 : 0;
    return static_cast<size_t>(super_rows_bound(K, block_size)) * N * 4 + static_cast<size_t>(N) * 8 + static_cast<size_t>(kLoopMaxRows) * N * 5 + tall +
           coef_image_bytes(K, block_size) + piece_gemm_bytes(K, N, block_size) + oq_rtn_workspace_bytes(K, N, OQ_CHANNEL, -1, 1) + 4096;
}

int32_t oq_gptq_loop_f32(float* W, int64_t K, int64_t N, const float* U, int32_t qtype, int64_t group_size, int32_t symmetric,
                         int32_t reduce_range, float clip_ratio, int32_t mse, int64_t block_size, int32_t mode, int32_t method,
                         const float* init_scale, const int32_t* init_zp, int64_t init_count, void* q_int_out, int32_t q_layout,
                         float* q_deq_out, float* used_scale, int32_t* used_zp, void* workspace, size_t workspace_bytes,
                         void* stream) {
    OQ_REQUIRE(W && U && init_scale && init_zp && q_int_out && q_deq_out && K > 0 && N > 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_gptq_loop_f32: bad argument");
    OQ_REQUIRE(matrix_ok(K, N, N) && K <= kMaxHessianWidth, OQ_ERR_UNSUPPORTED, "oq_gptq_loop_f32: matrix too large (K=%lld N=%lld)", (long long)K, (long long)N);
    OQ_REQUIRE(init_count == 1 || init_count == N, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: init_count must be 1 or N");
    OQ_REQUIRE(block_size > 0 && block_size <= kMaxExtent && group_size <= kMaxExtent, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: block_size must be positive");
    OQ_REQUIRE(mode == OQ_GPTQ_PARITY || mode == OQ_GPTQ_CORRECTED || mode == OQ_GPTQ_CORRECTED_COLUMNS, OQ_ERR_INVALID_ARGUMENT,
               "oq_gptq_loop_f32: bad mode %d", mode);
    OQ_REQUIRE(method >= OQ_HESSIAN_AUTO && method <= OQ_HESSIAN_F16X3, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: unknown method %d", method);
    OQ_REQUIRE(q_layout == OQ_LAYOUT_KN || q_layout == OQ_LAYOUT_KN_PACKED4, OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: bad q_layout %d", q_layout);
    OQ_REQUIRE(q_layout == OQ_LAYOUT_KN || ((qtype == OQ_INT4 || qtype == OQ_UINT4) && N % 2 == 0), OQ_ERR_UNSUPPORTED,
               "oq_gptq_loop_f32: the packed layout takes a 4-bit type and an even N");
    const bool columns_kernel = mode == OQ_GPTQ_CORRECTED_COLUMNS;      // same bytes as OQ_GPTQ_CORRECTED, the one-column-per-lane kernel
    if (columns_kernel) mode = OQ_GPTQ_CORRECTED;
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
    a.q_int = static_cast<uint8_t*>(q_int_out); a.q_deq = q_deq_out; a.packed4 = q_layout == OQ_LAYOUT_KN_PACKED4 ? 1 : 0;
    a.used_scale = (group_size > 0) ? used_scale : nullptr;
    a.used_zp = used_zp;
    if (a.used_scale != nullptr && used_zp == nullptr) return fail(OQ_ERR_INVALID_ARGUMENT, "oq_gptq_loop_f32: used_zp missing");
    const int64_t err_rows = super_rows_bound(K, block_size);
    const bool tall = block_size > kLoopMaxRows;
    char* wsp = static_cast<char*>(workspace);
    a.err = reinterpret_cast<float*>(wsp);                              wsp += static_cast<size_t>(err_rows) * N * 4;
    a.carry_scale = reinterpret_cast<float*>(wsp);                      wsp += static_cast<size_t>(N) * 4;
    a.carry_zp = reinterpret_cast<int32_t*>(wsp);                       wsp += static_cast<size_t>(N) * 4;
    float* pre_scale = reinterpret_cast<float*>(wsp);                   wsp += static_cast<size_t>(kLoopMaxRows) * N * 4;
    uint8_t* pre_zp = reinterpret_cast<uint8_t*>(wsp);                  wsp += static_cast<size_t>(kLoopMaxRows) * N;
    wsp += (256 - reinterpret_cast<uintptr_t>(wsp) % 256) % 256;
    float* wb = reinterpret_cast<float*>(wsp);                          // block_size > 128: the block's working copy W1
    if (tall) wsp += static_cast<size_t>(block_size < K ? block_size : K) * N * 4;
    wsp += (256 - reinterpret_cast<uintptr_t>(wsp) % 256) % 256;
    float* image = reinterpret_cast<float*>(wsp);                       // coefficient images of all launches
    wsp += coef_image_bytes(K, block_size);
    wsp += (256 - reinterpret_cast<uintptr_t>(wsp) % 256) % 256;
    char* pieces_a = wsp;                                               // fp16 pieces of U[s0:s_end, s_end:] ...
    wsp += (gemm_f16x3_pieces_bytes(err_rows, K) + 255) / 256 * 256;
    char* pieces_b = wsp;                                               // ... and of the super-block's Err
    wsp += (gemm_f16x3_pieces_bytes(err_rows, N) + 255) / 256 * 256;
    char* mse_ws = wsp + (256 - reinterpret_cast<uintptr_t>(wsp) % 256) % 256;
    const size_t mse_ws_bytes = static_cast<size_t>(static_cast<char*>(workspace) + workspace_bytes - mse_ws);
    a.pre_scale = nullptr; a.pre_zp = nullptr; a.pre_first_group = 0;
    a.zp_signed = (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0;
    a.Wt = W; a.wt_row0 = 0; a.err_row0 = 0; a.coef = nullptr; a.params_from_tile = tall ? 0 : 1;
    if (mode == OQ_GPTQ_PARITY && !(mse && a.g > 0) && ceil_div(K, a.g > 0 ? a.g : 128) <= 65535) {
        // no row depends on another one: one elementwise launch instead of K sequential steps
        const int64_t band_rows = a.g > 0 ? a.g : 128;
        a.i1 = 0; a.count = 0;
        hipLaunchKernelGGL(gptq_parity_kernel, dim3(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(ceil_div(K, band_rows))),
                           dim3(256), 0, s, a, band_rows);
        return check_launch("gptq_parity_kernel");
    }

    // Sequential path.  The reference walks blocks of `block_size` rows: inside a block every row's error reaches the later
    // rows of the block at once (on the copy W1, gptq.py:157, 198-200), the rows behind the block in one product at its end
    // (gptq.py:208), and group parameters are read from W as it stands when the group starts (gptq.py:168-184).
    //   * A kernel launch walks `sub` <= 128 rows.  block_size > 128: the block is a chain of such launches on a working
    //     copy `wb` of its rows; after each launch one GEMM carries its errors to the block's remaining rows IN THE COPY (W
    //     itself stays as the reference's `W` does until the block ends), then one GEMM with Kd = block_size updates W behind
    //     the block.
    //   * block_size <= 128, CORRECTED: the same two levels for speed.  A product with Kd = 128 over all rows behind a block
    //     reads and writes C once per 128 k-steps and is bound by that traffic (4096 x 4096: 33 TFLOP/s measured); so
    //     `S` = up to 512 rows (a multiple of block_size and of the group size, so that the rows a group's parameters are read
    //     from are complete when it starts) form a super-block: after each block a small GEMM updates the super-block's
    //     remaining rows, and ONE GEMM with Kd = S updates everything behind it.  Same sums in a different order.
    const int64_t sub = tall ? kLoopMaxRows : block_size;
    int64_t S = block_size;
    if (!tall && mode == OQ_GPTQ_CORRECTED) {
        int64_t unit = block_size;                                     // lcm(block_size, g)
        if (a.g > 0) {
            int64_t x = unit, y = a.g;
            while (y) { const int64_t t = x % y; x = y; y = t; }
            unit = unit / x * a.g;
        }
        if (unit <= kSuperRows) S = kSuperRows / unit * unit;
    }
    const bool rows16 = !columns_kernel && mode == OQ_GPTQ_CORRECTED && (a.g <= 0 || a.g % 16 == 0) && sub % 16 == 0;
    const uint32_t nblk = static_cast<uint32_t>(ceil_div(N, kLoopColsV2));
    // rows-over-lanes kernel: 4 columns per wave; as many waves per block as keep the grid within one block per CU
    int wpb16 = 4 * static_cast<int>(ceil_div(ceil_div(N, 16), 256));
    if (wpb16 > 16) wpb16 = 16;
    const uint32_t nblk16 = static_cast<uint32_t>(ceil_div(N, 4 * wpb16));
    const int per_block = static_cast<int>(launches_per_block(block_size));
    if (rows16) {
        hipLaunchKernelGGL(gptq_coef_image_kernel, dim3(kLoopMaxRows, static_cast<uint32_t>(ceil_div(K, block_size) * per_block)), dim3(160), 0, s,
                           U, K, block_size, per_block, image);
        st = check_launch("gptq_coef_image_kernel");
        if (st != OQ_OK) return st;
    }
    for (int64_t s0 = 0; s0 < K; s0 += S) {
        const int64_t s_end = s0 + S < K ? s0 + S : K;
        if (tall) {
            if (hipMemcpyAsync(wb, W + s0 * N, static_cast<size_t>(s_end - s0) * N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return fail(OQ_ERR_LAUNCH, "oq_gptq_loop_f32: copy of the block failed");
            a.Wt = wb; a.wt_row0 = s0;
        }
        for (int64_t i1 = s0; i1 < s_end; i1 += sub) {
            const int64_t count = (s_end - i1) < sub ? (s_end - i1) : sub;
            a.i1 = i1; a.count = count; a.err_row0 = i1 - s0;
            if (mse && a.g > 0) {
                // gptq.py:168-184 with mse=True: the MSE search (channel strategy) on rows [r, r+g) of the working
                // matrix for every group that starts inside this launch, before the sequential kernel runs.
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
            if (rows16 && i1 % 16 == 0) {
                const int64_t ord = (i1 / block_size) * per_block + (i1 % block_size) / kLoopMaxRows;
                a.coef = image + ord * kCoefFloats;
                hipLaunchKernelGGL(gptq_rows16_kernel, dim3(nblk16), dim3(static_cast<uint32_t>(64 * wpb16)), 0, s, a);
                st = check_launch("gptq_rows16_kernel");
            } else {
                hipLaunchKernelGGL(gptq_block_kernel, dim3(nblk), dim3(256), 0, s, a);
                st = check_launch("gptq_block_kernel");
            }
            if (st != OQ_OK) return st;
            const int64_t i2 = i1 + count;
            if (mode == OQ_GPTQ_CORRECTED && i2 < s_end) {
                // the launch's errors to the remaining rows of the (super-)block: rows [i2, s_end) -= U[i1:i2, i2:s_end]^T Err
                GemmTN g;
                g.At = U + i1 * K + i2; g.lda = K; g.M = s_end - i2;
                g.B = a.err + (i1 - s0) * N; g.ldb = N; g.N = N;
                g.C = tall ? wb + (i2 - s0) * N : W + i2 * N; g.ldc = N;
                g.Kd = count; g.alpha = -1.0f; g.beta = 1.0f; g.sa = 1.0f; g.sb = 1.0f; g.upper_only = 0; g.mirror = 0;
                if (panel_eligible(g.At, g.lda, g.B, g.ldb, g.C, g.M, g.N, g.Kd)) {
                    const int ntn = static_cast<int>(ceil_div(g.N, kPanelTile));
                    hipLaunchKernelGGL(panel_update_kernel, dim3(static_cast<uint32_t>(ceil_div(g.M, kPanelTile) * ntn)), dim3(256), 0, s, g.At, g.lda,
                                       g.B, g.ldb, g.C, g.ldc, g.M, g.N, static_cast<int>(g.Kd), ntn);
                    st = check_launch("panel_update_kernel");
                } else {
                    st = launch_gemm_tn(g, s);
                }
                if (st != OQ_OK) return st;
            }
        }
        if (mode == OQ_GPTQ_CORRECTED && s_end < K) {
            // gptq.py:208 with the intended operand, for the whole (super-)block: W[s_end:, :] -= U[s0:s_end, s_end:]^T Err
            GemmTN g;
            g.At = U + s0 * K + s_end; g.lda = K; g.M = K - s_end;
            g.B = a.err; g.ldb = N; g.N = N;
            g.C = W + s_end * N; g.ldc = N;
            g.Kd = s_end - s0; g.alpha = -1.0f; g.beta = 1.0f; g.sa = 1.0f; g.sb = 1.0f; g.upper_only = 0; g.mirror = 0;
            // Large products run on the fp16-piece GEMM (22-bit operands, fp32 accumulate: the Hessian's arithmetic, 3-5 x
            // the fp32 MFMA rate); with OQ_HESSIAN_F32 selected -- the reference's arithmetic class throughout -- and for
            // small problems the fp32 MFMA kernel.
            if (method != OQ_HESSIAN_F32 && g.Kd >= 64 && g.M * g.N >= (int64_t{1} << 20)) {
                st = make_f16x2_pieces(g.At, g.Kd, g.M, g.lda, false, pieces_a, s);
                if (st == OQ_OK) st = make_f16x2_pieces(g.B, g.Kd, g.N, g.ldb, false, pieces_b, s);
                if (st == OQ_OK) st = launch_gemm_f16x3(pieces_a, pieces_b, g.M, g.N, g.Kd, g.alpha, g.beta, g.C, g.ldc, nullptr, s);
            } else {
                st = launch_gemm_tn(g, s);
            }
            if (st != OQ_OK) return st;
        }
    }
    return OQ_OK;
}

#ifdef OQ_LOOP_STAMPS
void oq_lab_loop_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oq::g_loop_stamps), sizeof(unsigned long long) * 16); }
#endif

}  // extern "C"
