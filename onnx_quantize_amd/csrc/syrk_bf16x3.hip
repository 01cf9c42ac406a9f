// G1 on the bf16 matrix cores with fp32-exact operands:  C = beta C + alpha X^T X  for tall X [T, K]  (gptq.py:246-260).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (157 vs 2 500 TFLOP/s dense), and the Hessian is 90 % of a GPTQ
// run.  An fp32 number is the exact sum of three bf16 numbers: hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (8 + 8
// + 8 significand bits; the subtractions are exact).  So x y = sum over the nine piece products, each of which is EXACT
// in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Relative to x y the products weigh
//   hi.hi 1 | hi.mid, mid.hi 2^-8 | mid.mid, hi.lo, lo.hi 2^-16 | mid.lo, lo.mid 2^-24 | lo.lo 2^-32.
// terms = 6 keeps everything down to 2^-16 (what is dropped is <= 2^-23 |x y|, the size of ONE fp32 rounding of the
// product, and of either sign because the pieces are rounded to nearest); terms = 9 keeps all and is then MORE exact
// than an fp32 fma chain (no product rounding at all).  6 MFMAs at 16 x the fp32 rate = 2.7 x the fp32 peak.
// (bf16 has fp32's exponent range, so the split needs no scaling; inf / NaN inputs give NaN where fp32 gives inf / NaN.)
//
// Two kernels.
//  1. split_bf16x3_kernel: X -> pieces P, once (HBM-bound: 4 B read + 6 B written per element).  The MFMA wants, per
//     lane, 8 consecutive k (= rows t of X) of ONE column, so a thread takes one column x 8 rows (coalesced dword loads
//     across the wave), splits them in registers and writes three 16-byte vectors {t0..t7}:
//         P[chunk = t / 8][piece][column (padded to 256)] x 16 B,   zero rows / columns behind T / K.
//     Splitting inside the GEMM was measured first: every block re-splits its two panels (K / 256 times each element),
//     and ~100 dependent VALU instructions per wave and stage do not hide under 48 MFMAs -- 212 against 336 TFLOP/s
//     (fp32-equivalent) with the split compiled out.
//  2. syrk_pieces_kernel: 256 x 256 tile of C per block, 8 waves (4 x 2), each 64 x 128 = 2 x 4 MFMA tiles (128
//     accumulator registers).  A stage = 16 rows = [operand A|B][piece][chunk 0|1][256 columns] x 16 B = 48 KB of LDS, in
//     exactly the order P holds them, so staging is 48 global_load_lds_dwordx4 (1 KB each, 6 per wave): no staging
//     registers, no ds_write, no VALU.  LDS is a ring of three stages (144 KB, one block per CU).
//
// Round 2, `terms` = 3 ("f16x3"): the same machinery on the fp16 matrix instructions (syrk_f16_m16_kernel, 16x16x32, by
// default; syrk_pieces_kernel<3>, 32x32x16, with OQ_SYRK_F16_M16=0) with TWO fp16 pieces per element,
// hi = f16(x s), lo = f16(x s - hi): 11 + 11 significand bits, i.e. operands rounded to 22 bits (relative 2^-23), and the
// three products hi.hi, hi.lo, lo.hi (what is dropped -- lo.lo and the operand roundings -- is <= 3 * 2^-23 |x y|, the
// size of the product roundings of an fp32 fma chain and of either sign).  HALF the matrix-core work of terms = 6.
// fp16 has five exponent bits, so the batch is scaled by a power of two s that puts max |x| into [2^14, 2^15) (one extra
// read of X for the absmax: 3 % of the call) and 1 / s^2 -- exact -- goes onto alpha on the device.  Elements below
// 2^-29 max |x| lose low-order bits of their lo piece to fp16's subnormal spacing: an ABSOLUTE error below 2^-39 max |x|.
#include "gemm_tn.hpp"

#include <type_traits>

#ifndef OQ_SYRK_F16_STRIDE
#define OQ_SYRK_F16_STRIDE 3   /* measured on K = 11008: stride 5 21.7, 4 21.4, 3 21.1, 2 21.1, 1 21.3 ms per 65 536 rows */
#endif
#define OQ_SYRK_F16_STRIDE_EXPR (F16 ? OQ_SYRK_F16_STRIDE : 4)
#ifndef OQ_SYRK_M16_STRIDE
#define OQ_SYRK_M16_STRIDE 4   /* K = 11008, ms per 65 536 rows: stride 2 19.0-19.2, 4 18.83, 6 18.9, 8 18.9-19.0, 10 19.2-19.3 */
#endif
#ifndef OQ_SYRK_F16_DEEP
#define OQ_SYRK_F16_DEEP 1
#endif

namespace oq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kST = 256;                          // block tile edge
constexpr int kSThreads = 512;
// Geometry of a stage by piece kind.  bf16: three pieces, 16 rows (two 8-row chunks = the k of one MFMA) per stage, ring of
// three stages = 144 KB.  fp16: two pieces and half the MFMAs per row, so a stage holds 32 rows (four chunks, two MFMA
// k-steps) to keep 48 MFMAs per wave between two barriers, and the ring has two stages = 128 KB.
template <int TERMS> struct StageGeom {
    static constexpr bool F16 = TERMS == 3;
    static constexpr int PIECES = F16 ? 2 : 3;
    static constexpr int CH = (F16 && OQ_SYRK_F16_DEEP) ? 4 : 2;                     // 8-row chunks per stage
    static constexpr int RING = (F16 && OQ_SYRK_F16_DEEP) ? 2 : 3;
    static constexpr int ROWS = 8 * CH;
    static constexpr int PLANE = CH * kST * 16;                // one piece of one operand: [chunk][256 columns] x 16 B
    static constexpr int OPERAND = PIECES * PLANE;
    static constexpr int STAGE = 2 * OPERAND;                  // A | B
    static constexpr int LDS = RING * STAGE;
    static constexpr int NDMA = STAGE / 1024 / 8;              // 1 KB pieces of a stage per wave: 6 (bf16) / 8 (fp16)
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, v);
}

// two consecutive rows of one column -> the three packed piece pairs
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);      // exact
    mid = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);    // exact, <= 8 bits
    lo = pk_bf16(s0, s1);
}

// ---- 1. X -> pieces
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                           const int64_t Kp, const int64_t nchunks, u32x4* __restrict__ P) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const bool col_ok = k < K;
    const float* src = X + (col_ok ? k : K - 1);
#pragma unroll 1
    for (int64_t c = static_cast<int64_t>(blockIdx.y) * 4; c < nchunks && c < static_cast<int64_t>(blockIdx.y) * 4 + 4; ++c) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t t = c * 8 + r;
            const float x = src[(t < T ? t : T - 1) * ldx];
            v[r] = (t < T && col_ok) ? x : 0.f;
        }
        u32x4 hi, mid, lo;
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            uint32_t h, m, l;
            split_pair(v[2 * rp], v[2 * rp + 1], h, m, l);
            hi[rp] = h; mid[rp] = m; lo[rp] = l;
        }
        u32x4* o = P + (c * 3) * Kp + k;
        __builtin_nontemporal_store(hi, o);
        __builtin_nontemporal_store(mid, o + Kp);
        __builtin_nontemporal_store(lo, o + 2 * Kp);
    }
}

// ---- 1b. fp16 pieces: scale, then hi = f16(x s), lo = f16(x s - hi)
#ifdef OQ_PREP_NT
#define OQ_PREP_LOAD(p) __builtin_nontemporal_load(p)
#else
#define OQ_PREP_LOAD(p) (*(p))
#endif
__device__ __forceinline__ void absmax_partial_body(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                    float* __restrict__ partial, const int64_t nblocks, const int64_t block) {
    float m = 0.f;
    const int64_t rows_per = (T + nblocks - 1) / nblocks;
    const int64_t t0 = block * rows_per, t1 = t0 + rows_per < T ? t0 + rows_per : T;
    for (int64_t t = t0; t < t1; ++t)
        for (int64_t k = threadIdx.x; k < K; k += 256) m = nmax(m, fabsf(OQ_PREP_LOAD(X + t * ldx + k)));
    m = wave_max(m);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[block] = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
}

__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                             float* __restrict__ partial) {
    absmax_partial_body(X, T, K, ldx, partial, static_cast<int64_t>(gridDim.x), static_cast<int64_t>(blockIdx.x));
}

// partial maxima -> scale[0] = s = 2^(15 - exponent of max) (1 for an all-zero or non-finite batch), scale[1] = 1 / s^2, scale[2] = 1 / s
template <int CLAMP = 60>
__device__ __forceinline__ void absmax_scale_body(const float* __restrict__ partial, const int n, float* __restrict__ scale) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = nmax(m, partial[i]);
    m = wave_max(m);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
        int ex = 0;
        float s = 1.0f, inv2 = 1.0f;
        if (m > 0.0f && m < INFINITY) {
            (void)frexpf(m, &ex);                      // m = f * 2^ex, f in [0.5, 1)
            int e = 15 - ex;
            e = e > CLAMP ? CLAMP : (e < -CLAMP ? -CLAMP : e);     // 60: 1 / s^2 must stay a normal float (the consumers that read it)
            s = ldexpf(1.0f, e);
            inv2 = CLAMP <= 60 ? ldexpf(1.0f, -2 * e) : 0.0f;       // wide range (plain GEMM operands): 1 / s^2 is not read
        }
        scale[0] = s;
        scale[1] = inv2;
        scale[2] = 1.0f / s;   // exact: a power of two
    }
}

__global__ __launch_bounds__(256) void absmax_scale_kernel(const float* __restrict__ partial, const int n, float* __restrict__ scale) {
    absmax_scale_body(partial, n, scale);
}

// the same over (nearly) the whole exponent range of fp32: for operands of a plain product C = A B, whose epilogue applies 1 / s_a
// and 1 / s_b one after the other (an operand whose largest magnitude is 1e-30 or 1e30 must not vanish or overflow in the pieces)
__global__ __launch_bounds__(256) void absmax_scale_wide_kernel(const float* __restrict__ partial, const int n, float* __restrict__ scale) {
    absmax_scale_body<110>(partial, n, scale);
}

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
    const f16x2 v = {static_cast<_Float16>(a), static_cast<_Float16>(b)};   // v_cvt_f16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, v);
}

// alpha > 0: the dead-channel guard's factor; alpha <= 0: no guard (callers pass 0 or their product's -1).  The one value
// kSplitHiOnly: no guard AND the lo plane is not written (a consumer that reads first pieces only).
constexpr float kSplitHiOnly = -3.0e38f;
__device__ __forceinline__ void split_f16x2_body(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                 const int64_t Kp, const int64_t nchunks, const float* __restrict__ scale,
                                                 const float alpha, u32x4* __restrict__ P, const int64_t bx, const int64_t by) {
    const int64_t k = bx * 256 + threadIdx.x;
    const bool col_ok = k < K;
    const float* src = X + (col_ok ? k : K - 1);
    const float sc = scale[0];
#pragma unroll 1
    for (int64_t c = by * 4; c < nchunks && c < by * 4 + 4; ++c) {
        float v[8];
        uint32_t alive = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t t = c * 8 + r;
            const float x = OQ_PREP_LOAD(src + (t < T ? t : T - 1) * ldx);
            v[r] = (t < T && col_ok) ? x * sc : 0.f;                       // a power of two: exact
            // a sample whose square the fp32 path still sees (H[k][k] > 0: the channel is NOT dead, gptq.py:284-286) must not
            // vanish below fp16's last subnormal (|x| < 2^-40 max|x|): it keeps one unit there
            alive |= ((x * x) * alpha > 0.0f && t < T && col_ok) ? (1u << r) : 0u;
        }
        u32x4 hi, lo;
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const f16x2 h2 = {static_cast<_Float16>(v[2 * rp]), static_cast<_Float16>(v[2 * rp + 1])};
            uint32_t hb = __builtin_bit_cast(uint32_t, h2);
            if ((hb & 0x7fffu) == 0 && (alive >> (2 * rp) & 1u)) hb |= 1u;
            if ((hb & 0x7fff0000u) == 0 && (alive >> (2 * rp + 1) & 1u)) hb |= 0x10000u;
            hi[rp] = hb;
            lo[rp] = pk_f16(v[2 * rp] - static_cast<float>(h2[0]), v[2 * rp + 1] - static_cast<float>(h2[1]));   // differences exact
        }
        u32x4* o = P + (c * 2) * Kp + k;
        __builtin_nontemporal_store(hi, o);
        if (alpha != kSplitHiOnly) __builtin_nontemporal_store(lo, o + Kp);      // kSplitHiOnly: the consumer reads first pieces only
    }
}

__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                          const int64_t Kp, const int64_t nchunks, const float* __restrict__ scale,
                                                          const float alpha, u32x4* __restrict__ P) {
    split_f16x2_body(X, T, K, ldx, Kp, nchunks, scale, alpha, P, static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(blockIdx.y));
}

// one matrix-core instruction on a piece pair, chosen by the piece type
__device__ __forceinline__ void mfma_pieces(const bf16x8& x, const bf16x8& y, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }
__device__ __forceinline__ void mfma_pieces(const f16x8& x, const f16x8& y, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); }

// The piece products of one output tile, small terms first, and the slot in front of which the B operand of the next
// tile column is fetched.  bf16: 0 = hi, 1 = mid, 2 = lo; fp16: 0 = hi, 1 = lo.
template <int TERMS> struct TermSeq;
template <> struct TermSeq<3> { static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0}, PF = 1; };
template <> struct TermSeq<6> { static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0}, PF = 3; };
template <> struct TermSeq<9> { static constexpr int A[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, B[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, PF = 6; };

// ---- 2. C += pieces^T pieces
// Stream of one stage (sched_barrier(0) pins it; what sits BETWEEN two MFMAs issues while the first one runs):
//   slots 0-23  this wave's six global_load_lds of stage s + 2 into ring slot (s + 2) % 3 (free since the last barrier),
//               one every fourth slot
//   per j       the B operand of tile column j + 1 while the MFMAs of column j run
//   tail        stage s + 1 landed before the last barrier, so its first operands are fetched behind the last MFMAs of
//               this stage: no wave starts a stage by waiting for LDS while all eight queue 9 KB each on it
//   end         vmcnt(0) (the DMAs were issued ~40 MFMAs ago), lgkmcnt(0), s_barrier.
template <int TERMS>
__global__ __launch_bounds__(kSThreads) void syrk_pieces_kernel(const u32x4* __restrict__ P, const int64_t K, const int64_t Kp, const int64_t nstages_all,
                                                                const float alpha_in, const float beta, float* __restrict__ C,
                                                                float* __restrict__ slab, const int64_t stages_per_slice, const int ntiles,
                                                                const float* __restrict__ post_scale) {
    using G = StageGeom<TERMS>;
    constexpr bool F16 = G::F16;
    constexpr int PIECES = G::PIECES, CH = G::CH, RING = G::RING, NDMA = G::NDMA;
    constexpr int kPlaneBytes = G::PLANE, kOperandBytes = G::OPERAND, kStageBytes = G::STAGE;
    using frag = std::conditional_t<F16, f16x8, bf16x8>;
    const float alpha = post_scale ? alpha_in * post_scale[1] : alpha_in;      // fp16 pieces: 1 / s^2, a power of two
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int tile_m, tile_n;
    upper_tile_of(static_cast<int>(xcd_remap(blockIdx.x, gridDim.x)), ntiles, tile_m, tile_n);
    const int64_t m0 = static_cast<int64_t>(tile_m) * kST, n0 = static_cast<int64_t>(tile_n) * kST;
    const int64_t s_begin = static_cast<int64_t>(blockIdx.y) * stages_per_slice;
    const int64_t s_end = s_begin + stages_per_slice < nstages_all ? s_begin + stages_per_slice : nstages_all;
    const int64_t nstages = s_end - s_begin;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kc = lane >> 5, cl = lane & 31;

    // ---- loader role: pieces q = NDMA wave .. NDMA wave + NDMA - 1 of the stage; piece q = 1 KB =
    // (operand, piece, chunk, quarter of 64 columns) in LDS order [operand][piece][chunk][256 columns]
    // bounds are the maxima over both piece kinds, not NDMA / PIECES: a dependent-size array captured by a lambda silently
    // invalidates the host side of this kernel (clang 22, HIP: no stub is emitted and the library does not load)
    const char* gsrc[8];
    uint32_t ldst[8];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int q = wave * NDMA + i;
        const int quarter = q & 3, cc = (q >> 2) % CH, pc = (q / (4 * CH)) % PIECES, op = q / (4 * CH * PIECES);
        ldst[i] = static_cast<uint32_t>(q) * 1024u;
        const int64_t colq = (op ? n0 : m0) + quarter * 64 + lane;
        gsrc[i] = reinterpret_cast<const char*>(P + ((s_begin * CH + cc) * PIECES + pc) * Kp + colq);
    }
    const int64_t stage_bytes = static_cast<int64_t>(CH) * PIECES * Kp * 16;
    auto stage_dma = [&](int64_t s_rel, int slot3, int i) {
        __builtin_amdgcn_global_load_lds(gsrc[i] + s_rel * stage_bytes,
                                         (__attribute__((address_space(3))) void*)(lds + slot3 * kStageBytes + ldst[i]), 16, 0, 0);
    };

    // prologue: the first RING - 1 stages (clamped to the last stage of the slice when the slice is shorter: harmless re-reads)
#pragma unroll
    for (int st = 0; st < RING - 1; ++st)
#pragma unroll
        for (int i = 0; i < NDMA; ++i) stage_dma(st < nstages ? st : nstages - 1, st, i);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int kSlots = 8 * TERMS * (CH / 2);
    static_assert(kSlots >= 2 + (NDMA - 1) * (OQ_SYRK_F16_STRIDE_EXPR) + 1, "the DMAs must fit into the stream");
    const uint32_t rd_a = static_cast<uint32_t>((kc * kST + wm * 64 + cl) * 16);
    const uint32_t rd_b = static_cast<uint32_t>(kOperandBytes + (kc * kST + wn * 128 + cl) * 16);
    constexpr int kStepBytes = 2 * kST * 16;            // the two chunks of one MFMA k-step
    frag a[2][3], b[2][3];
    auto read_a = [&](int slot3, int h, int i) {
#pragma unroll
        for (int p = PIECES - 1; p >= 0; --p)
            a[i][p] = *reinterpret_cast<const frag*>(lds + slot3 * kStageBytes + rd_a + p * kPlaneBytes + h * kStepBytes + i * 32 * 16);
    };
    auto read_b = [&](int slot3, int h, int j, int which) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            b[which][p] = *reinterpret_cast<const frag*>(lds + slot3 * kStageBytes + rd_b + p * kPlaneBytes + h * kStepBytes + j * 32 * 16);
    };
    auto stage_body = [&](auto phase_tag, int64_t s, int cur3, int nxt3, int wr3) {
        constexpr int STRIDE = OQ_SYRK_F16_STRIDE_EXPR;                  // one DMA every fourth MFMA
        constexpr int DMA0 = decltype(phase_tag)::value ? 2 : 0;         // first slot of this wave's DMAs
        const int64_t s_dma = s + RING - 1 < nstages ? s + RING - 1 : nstages - 1;   // behind the slice: re-read its last stage into a slot nobody reads
        int slot = 0;
        auto mm = [&](const frag& x, const frag& y, f32x16& c) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_pieces(x, y, c);
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= DMA0 && slot < DMA0 + NDMA * STRIDE && (slot - DMA0) % STRIDE == 0) stage_dma(s_dma, wr3, (slot - DMA0) / STRIDE);
            ++slot;
        };
        using seq = TermSeq<TERMS>;
#pragma unroll
        for (int h = 0; h < CH / 2; ++h) {
            constexpr bool kLookAhead = RING >= 3;        // stage s + 1 landed before the last barrier only with a ring of three
            const bool last_step = h == CH / 2 - 1;
            read_a(cur3, h, 1);                           // a[0], b[0] came with the previous k-step
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cur = j & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x16 c = acc[i][j];
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) {
                        if (t == seq::PF && i == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (j < 3) read_b(cur3, h, j + 1, cur ^ 1);
                            else if (!last_step) read_b(cur3, h + 1, 0, 0);
                            else if (kLookAhead) read_b(nxt3, 0, 0, 0);
                        }
                        mm(a[i][seq::A[t]], b[cur][seq::B[t]], c);
                    }
                    acc[i][j] = c;
                    if (i == 0 && j == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (!last_step) read_a(cur3, h + 1, 0);
                        else if (kLookAhead) read_a(nxt3, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    read_b(0, 0, 0, 0);
    read_a(0, 0, 0);
    int cur3 = 0, nxt3 = RING >= 3 ? 1 : 1, wr3 = RING - 1;
    // Neighbouring waves (the two of a SIMD, and neighbouring SIMDs) run the same stream with their DMAs two slots apart,
    // so that they do not all pay the issue cost of a global_load_lds -- M0, address, ~100 cycles on a busy CU -- in the
    // same MFMA gap (measured: all in slots 0-5 199, half a stage apart 213-219, one every fourth slot and two slots
    // apart 225 TFLOP/s fp32-equivalent, K = 11008).  Two copies of the loop: a branch inside it would cost the
    // accumulators their registers.
    auto rotate = [&]() {
        if (RING >= 3) { const int t3 = cur3; cur3 = nxt3; nxt3 = wr3; wr3 = t3; }
        else {           // ring of two: the stage just filled becomes current, and its first operands are fetched now
            const int t3 = cur3; cur3 = wr3; nxt3 = t3; wr3 = t3;
            read_b(cur3, 0, 0, 0);
            read_a(cur3, 0, 0);
        }
    };
    const int phase = __builtin_amdgcn_readfirstlane((wave >> 2) ^ (wave & 1));
    if (phase) {
        for (int64_t s = 0; s < nstages; ++s) {
            stage_body(std::true_type{}, s, cur3, nxt3, wr3);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_barrier();
            rotate();
        }
    } else {
        for (int64_t s = 0; s < nstages; ++s) {
            stage_body(std::false_type{}, s, cur3, nxt3, wr3);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_barrier();
            rotate();
        }
    }

    // ---- epilogue.  C/D map of a 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
    float* out = slab ? slab + static_cast<int64_t>(blockIdx.y) * K * K : C;
    const bool direct = slab == nullptr;
    const bool diag = tile_m == tile_n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t colj = n0 + wn * 128 + j * 32 + cl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kc;
                if (row < K && colj < K) {
                    float val = acc[i][j][e];
                    if (direct) {
                        // a diagonal tile holds both (r, c) and (c, r), summed in different orders: keep the upper one
                        // and mirror it, so that C is symmetric bit for bit
                        if (diag && colj < row) continue;
                        val = alpha * val;
                        if (beta != 0.0f) val = beta * out[row * K + colj] + val;
                        if (colj != row) out[colj * K + row] = val;
                    }
                    out[row * K + colj] = val;
                }
            }
        }
}

// ---- 2b. The fp16-piece GEMM on v_mfma_f32_16x16x32_f16 (experiment: OQ_SYRK_F16_M16=1).  Same block tile, stage geometry
// (32 rows = four 8-row chunks = ONE k-step of this instruction), ring of two and LDS-DMA staging as syrk_pieces_kernel<3>;
// a wave's 64 x 128 tile is 4 x 8 tiles of 16 x 16.  Lane l supplies rows / columns i = l & 15 and the eight k-values of
// chunk l >> 4 -- the same 16-byte vector P holds.  C/D map: col = l & 15, row = 4 (l >> 4) + e, e = 0..3.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The stage loop of the 16x16x32 kernels, shared by the SYRK (A = B = one piece array) and the two-operand GEMM
// (gemm_f16x3_kernel).  gsrc[i]: where this wave's i-th 1 KB piece of stage 0 comes from; a stage further is
// gsrc[i] + stage_bytes[i >= split ? 1 : 0] (the two operands may have different padded widths).  acc: the wave's 64 x 128
// tile as 4 x 8 tiles of 16 x 16, zeroed here.
// PRODUCTS = 3: hi.hi + hi.lo + lo.hi (22-bit operands).  PRODUCTS = 2: hi.hi + lo.hi -- A with both pieces, B with its first
// piece only (the Gram route of the AWQ searches: a 22-bit G against an 11-bit D whose rounding errors are independent per
// column); B's lo plane is neither fetched nor read.  PRODUCTS = 1 (hi.hi only) has a stage loop of its own, f16_hi_mainloop.
template <int NDMA, int PRODUCTS = 3>
__device__ __forceinline__ void f16_m16_mainloop(const char* const (&gsrc)[8], const int64_t (&stage_bytes)[8], const int64_t nstages,
                                                 unsigned char* lds, f32x4v (&acc)[4][8]) {
    using G = StageGeom<3>;
    static_assert(G::CH == 4 && G::RING == 2, "one 32-deep k-step per stage");
    constexpr int PIECES = 2;
    constexpr int kPlaneBytes = G::PLANE, kOperandBytes = G::OPERAND, kStageBytes = G::STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kc = lane >> 4, cl = lane & 15;
    uint32_t ldst[8];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) ldst[i] = static_cast<uint32_t>(wave * NDMA + i) * 1024u;
    static_assert(PRODUCTS == 3 || NDMA == 8, "operand of a wave's DMA pieces: wave / 4, piece (wave / 2) % 2, with eight per wave");
    // uniform: PRODUCTS = 1 fetches hi planes only, PRODUCTS = 2 everything but B's lo plane (waves 6 and 7)
    const bool dma_on = PRODUCTS == 3 || (PRODUCTS == 1 ? ((wave >> 1) & 1) == 0 : wave < 6);
    auto stage_dma = [&](int64_t s_rel, int slot, int i) {
        if (dma_on)
            __builtin_amdgcn_global_load_lds(gsrc[i] + s_rel * stage_bytes[i],
                                             (__attribute__((address_space(3))) void*)(lds + slot * kStageBytes + ldst[i]), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NDMA; ++i) stage_dma(0, 0, i);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const uint32_t rd_a = static_cast<uint32_t>((kc * kST + wm * 64 + cl) * 16);
    const uint32_t rd_b = static_cast<uint32_t>(kOperandBytes + (kc * kST + wn * 128 + cl) * 16);
    f16x8 a[4][2], b[2][2];
    auto read_a = [&](int slot, int i) {
#pragma unroll
        for (int p = (PRODUCTS >= 2 ? PIECES : 1) - 1; p >= 0; --p) a[i][p] = *reinterpret_cast<const f16x8*>(lds + slot * kStageBytes + rd_a + p * kPlaneBytes + i * 16 * 16);
    };
    auto read_b = [&](int slot, int j, int which) {
#pragma unroll
        for (int p = 0; p < (PRODUCTS == 3 ? PIECES : 1); ++p) b[which][p] = *reinterpret_cast<const f16x8*>(lds + slot * kStageBytes + rd_b + p * kPlaneBytes + j * 16 * 16);   // B's first piece only unless all three products run
    };
    auto stage_body = [&](auto phase_tag, int64_t s, int cur, int wr) {
        constexpr int STRIDE = PRODUCTS == 3 ? OQ_SYRK_M16_STRIDE : PRODUCTS == 2 ? 3 : 2;   // in 16-cycle MFMA slots (32 per product and stage)
        constexpr int DMA0 = decltype(phase_tag)::value ? (PRODUCTS == 3 ? 4 : 2) : 0;
        const int64_t s_dma = s + 1 < nstages ? s + 1 : nstages - 1;
        int slot = 0;
        auto mm = [&](const f16x8& x, const f16x8& y, f32x4v& c) {
            __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= DMA0 && slot < DMA0 + NDMA * STRIDE && (slot - DMA0) % STRIDE == 0) stage_dma(s_dma, wr, (slot - DMA0) / STRIDE);
            ++slot;
        };
        read_a(cur, 1); read_a(cur, 2); read_a(cur, 3);      // a[0], b[0] were fetched right behind the barrier
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cb = j & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4v c = acc[i][j];
                if (i == 1 && j < 7) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_b(cur, j + 1, cb ^ 1);
                }
                if constexpr (PRODUCTS >= 2) mm(a[i][1], b[cb][0], c);     // lo . hi
                if constexpr (PRODUCTS == 3) mm(a[i][0], b[cb][1], c);     // hi . lo
                mm(a[i][0], b[cb][0], c);     // hi . hi
                acc[i][j] = c;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    read_b(0, 0, 0);
    read_a(0, 0);
    int cur = 0, wr = 1;
    const int phase = __builtin_amdgcn_readfirstlane((wave >> 2) ^ (wave & 1));
    if (phase) {
        for (int64_t s = 0; s < nstages; ++s) {
            stage_body(std::true_type{}, s, cur, wr);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_barrier();
            const int t = cur; cur = wr; wr = t;
            read_b(cur, 0, 0);
            read_a(cur, 0);
        }
    } else {
        for (int64_t s = 0; s < nstages; ++s) {
            stage_body(std::false_type{}, s, cur, wr);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_barrier();
            const int t = cur; cur = wr; wr = t;
            read_b(cur, 0, 0);
            read_a(cur, 0);
        }
    }
}

// The stage loop for the FIRST pieces alone (the AWQ / clip searches' loss product: one MFMA product instead of three).
// With a third of the matrix work per stage (64 MFMAs per SIMD = 0.5 us) a ring of two stages no longer hides the operand
// DMA (~1 us from L2 / HBM): the generic loop above ran this product at 143 us per 4096^3, half of it waiting.  Here only
// the hi planes are staged -- 32 KB per stage, compact in LDS -- in a ring of FOUR, i.e. the DMAs of stage s + 3 are issued
// while stage s is multiplied and a wave waits for all but its eight youngest (vmcnt(8)) before the barrier.  Every wave
// fetches four 1 KB pieces per stage.  Global layout of the pieces unchanged ([chunk][2 planes][width] x 16 B).
constexpr int kHiPlane = 4 * kST * 16;        // [4 chunks][256 columns] x 16 B
constexpr int kHiStage = 2 * kHiPlane;        // A | B
constexpr int kHiRing = 4;
static_assert(kHiRing * kHiStage <= StageGeom<3>::LDS, "the hi-only ring fits the LDS reserved for the generic loop");

__device__ __forceinline__ void f16_hi_mainloop(const char* const (&gsrc)[4], const int64_t (&stage_bytes)[4], const int64_t nstages,
                                                unsigned char* lds, f32x4v (&acc)[4][8]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kc = lane >> 4, cl = lane & 15;
    uint32_t ldst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;                     // 32 pieces: operand q / 16, chunk (q / 4) % 4, quarter q % 4
        ldst[i] = static_cast<uint32_t>((q >> 4) * kHiPlane + (q & 15) * 1024);
    }
    auto stage_dma = [&](int64_t s_abs, int slot) {
        const int64_t sc = s_abs < nstages ? s_abs : nstages - 1;     // past the end: a harmless re-read of the last stage
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(gsrc[i] + sc * stage_bytes[i],
                                             (__attribute__((address_space(3))) void*)(lds + slot * kHiStage + ldst[i]), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < kHiRing - 1; ++st) stage_dma(st, st);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // stage 0 has landed (stages 1 and 2 may still fly)
    __builtin_amdgcn_s_barrier();
    const uint32_t rd_a = static_cast<uint32_t>((kc * kST + wm * 64 + cl) * 16);
    const uint32_t rd_b = static_cast<uint32_t>(kHiPlane + (kc * kST + wn * 128 + cl) * 16);
    int cur = 0;
    f16x8 a[4], b[2];
    {   // the first operands of stage 0
        a[0] = *reinterpret_cast<const f16x8*>(lds + rd_a);
        b[0] = *reinterpret_cast<const f16x8*>(lds + rd_b);
    }
    for (int64_t s = 0; s < nstages; ++s) {
        int wr = cur + kHiRing - 1;
        wr = wr >= kHiRing ? wr - kHiRing : wr;
        const unsigned char* base = lds + cur * kHiStage;
#pragma unroll
        for (int i = 1; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8*>(base + rd_a + i * 16 * 16);
        stage_dma(s + kHiRing - 1, wr);                  // slot `wr` was multiplied in stage s - 1: everybody has passed its barrier
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == 1 && j < 7) {
                    __builtin_amdgcn_sched_barrier(0);
                    b[(j + 1) & 1] = *reinterpret_cast<const f16x8*>(base + rd_b + (j + 1) * 16 * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j & 1], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage s + 1 has landed; the DMAs of s + 2 and s + 3 stay in flight
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == kHiRing ? 0 : cur + 1;
        // the next stage's first operands right behind the barrier (the last stage re-reads a slot that holds a copy of it)
        a[0] = *reinterpret_cast<const f16x8*>(lds + cur * kHiStage + rd_a);
        b[0] = *reinterpret_cast<const f16x8*>(lds + cur * kHiStage + rd_b);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-reads past the end: the epilogue reuses the ring
    __builtin_amdgcn_s_barrier();
}

// The block's 256 x 256 accumulator tile -> memory through LDS, in four passes of 64 rows (the rows of the waves wm == pass).
// In the MFMA's C/D layout a store instruction covers 4 rows x 64 bytes and a read-modify-write of C costs 128 loads + 128
// stores of 4 bytes per lane; short products (the factor's and the GPTQ loop's 512-deep updates: 16 stages, 13 us of matrix
// work per tile) spent most of their time there.  Staged, every wave handles whole rows: one ds_read_b128 and one 16-byte
// global access per lane and row (1 KB per wave-instruction).  `rows(r, c4, v)`: row r of the tile (0..255), the lane's four
// columns c4..c4+3 with their values; returns what `cols` should see for them (written back to LDS).  `cols(c, r, x)`, only
// when TRANSPOSED: column c of the tile, the lane's row r, for the mirrored half of a symmetric result -- 64 consecutive
// rows per wave-instruction.  The ring buffers of the stage loop are dead when this runs (its last barrier has passed).
constexpr int kEpiLd = 260;   // floats per staged row: 4 kc * 260 = 16 kc (mod 64 banks), the 16 cl lanes consecutive: no conflicts
template <bool TRANSPOSED, typename Rows, typename Cols>
__device__ __forceinline__ void staged_tile_epilogue(const f32x4v (&acc)[4][8], unsigned char* lds, Rows&& rows, Cols&& cols) {
    static_assert(64 * kEpiLd * 4 <= StageGeom<3>::LDS, "a 64-row pass fits the stage ring");
    float* t = reinterpret_cast<float*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kc = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        __syncthreads();
        if (wm == p) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[(i * 16 + 4 * kc + e) * kEpiLd + wn * 128 + j * 16 + cl] = acc[i][j][e];
        }
        __syncthreads();
#pragma unroll 2
        for (int rr = 0; rr < 8; ++rr) {
            const int r = wave * 8 + rr;
            f32x4v* cell = reinterpret_cast<f32x4v*>(t + r * kEpiLd + lane * 4);
            const f32x4v v = rows(p * 64 + r, lane * 4, *cell);
            if constexpr (TRANSPOSED) *cell = v;
        }
        if constexpr (TRANSPOSED) {
            __syncthreads();
#pragma unroll 4
            for (int cc = 0; cc < 32; ++cc) {
                const int c = wave * 32 + cc;
                cols(c, p * 64 + lane, t[lane * kEpiLd + c]);
            }
        }
    }
}

// `tile`: the block's (XCD-remapped) index among the upper-triangle tiles of this matrix; `slice`: its T-slice
__device__ __forceinline__ void syrk_f16_m16_body(const u32x4* __restrict__ P, const int64_t K, const int64_t Kp, const int64_t nstages_all,
                                                  const float alpha_in, const float beta, float* __restrict__ C,
                                                  float* __restrict__ slab, const int64_t stages_per_slice, const int ntiles,
                                                  const float* __restrict__ post_scale, const int tile, const int64_t slice, unsigned char* lds,
                                                  const int64_t ldc,    // leading dimension of C (slabs are always K x K)
                                                  const bool mirror_all = true) {   // false: only diagonal tiles write their lower half
    using G = StageGeom<3>;
    constexpr int PIECES = 2, CH = 4, NDMA = G::NDMA;
    const float alpha = post_scale ? alpha_in * post_scale[1] : alpha_in;
    int tile_m, tile_n;
    upper_tile_of(tile, ntiles, tile_m, tile_n);
    const int64_t m0 = static_cast<int64_t>(tile_m) * kST, n0 = static_cast<int64_t>(tile_n) * kST;
    const int64_t s_begin = slice * stages_per_slice;
    const int64_t s_end = s_begin + stages_per_slice < nstages_all ? s_begin + stages_per_slice : nstages_all;
    const int64_t nstages = s_end - s_begin;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const char* gsrc[8];
    int64_t stage_bytes[8];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int q = wave * NDMA + i;
        const int quarter = q & 3, cc = (q >> 2) % CH, pc = (q / (4 * CH)) % PIECES, op = q / (4 * CH * PIECES);
        const int64_t colq = (op ? n0 : m0) + quarter * 64 + lane;
        gsrc[i] = reinterpret_cast<const char*>(P + ((s_begin * CH + cc) * PIECES + pc) * Kp + colq);
        stage_bytes[i] = static_cast<int64_t>(CH) * PIECES * Kp * 16;
    }
    f32x4v acc[4][8];
    f16_m16_mainloop<NDMA>(gsrc, stage_bytes, nstages, lds, acc);

    float* out = slab ? slab + slice * K * K : C;
    const bool direct = slab == nullptr;
    const int64_t ld = direct ? ldc : K;
    const bool diag = tile_m == tile_n;
    const bool vec_ok = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    // rows: slab mode stores the tile as it is (the reduction applies alpha / beta and the symmetry); direct mode applies alpha
    // and beta, writes the upper part (diagonal tiles: from the diagonal on) and hands the final values to the mirror pass
    auto rows = [&](int r, int c4, f32x4v v) -> f32x4v {
        const int64_t row = m0 + r, col = n0 + c4;
        if (row >= K || col >= K) return v;
        float* o = out + row * ld + col;
        const bool full = col + 3 < K;
        if (!direct) {
            if (full && vec_ok) *reinterpret_cast<f32x4v*>(o) = v;
            else for (int q = 0; q < 4 && col + q < K; ++q) o[q] = v[q];
            return v;
        }
        f32x4v old = {0.f, 0.f, 0.f, 0.f};
        if (beta != 0.0f) {
            if (full && vec_ok) old = *reinterpret_cast<const f32x4v*>(o);
            else for (int q = 0; q < 4 && col + q < K; ++q) old[q] = o[q];
        }
        f32x4v val;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = alpha * v[q];
            if (beta != 0.0f) x = beta * old[q] + x;
            val[q] = x;
        }
        if (full && vec_ok && !(diag && col < row)) {
            *reinterpret_cast<f32x4v*>(o) = val;
        } else {
            for (int q = 0; q < 4 && col + q < K; ++q)
                if (!(diag && col + q < row)) o[q] = val[q];
        }
        return val;
    };
    // the mirrored half: element (r, c) of the tile goes to C[n0 + c][m0 + r]; diagonal tiles mirror their strictly upper part
    auto cols = [&](int c, int r, float x) {
        const int64_t row = m0 + r, col = n0 + c;
        if (row < K && col < K && (diag ? col > row : true)) out[col * ld + row] = x;
    };
    if (direct && (diag || mirror_all)) staged_tile_epilogue<true>(acc, lds, rows, cols);
    else staged_tile_epilogue<false>(acc, lds, rows, cols);
}

__global__ __launch_bounds__(kSThreads) void syrk_f16_m16_kernel(const u32x4* __restrict__ P, const int64_t K, const int64_t Kp, const int64_t nstages_all,
                                                                 const float alpha_in, const float beta, float* __restrict__ C,
                                                                 float* __restrict__ slab, const int64_t stages_per_slice, const int ntiles,
                                                                 const float* __restrict__ post_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    syrk_f16_m16_body(P, K, Kp, nstages_all, alpha_in, beta, C, slab, stages_per_slice, ntiles, post_scale,
                      static_cast<int>(xcd_remap(blockIdx.x, gridDim.x)), static_cast<int64_t>(blockIdx.y), lds, K);
}

// ---- 2c. Many Hessian updates in one launch chain (oq_hessian_accumulate_many_f32): the activations a calibration batch
// taps (72 tensors of 5120 rows on gemma-3-270m shapes) are each far too small to fill 256 CUs -- 6 to 36 tiles -- and five
// launches per tensor made the per-tensor route launch-bound.  One table of items, one launch per step: blockIdx -> (item,
// block of that item) by a binary search over the items' first block ids (uniform: scalar loads).  Every item takes the
// fp16-piece route with ONE T-slice, so the product writes H directly (alpha / beta in the epilogue, no slabs).
struct SyrkItem {
    const float* X;
    float* C;
    u32x4* P;
    float* scale;            // the item's header in front of P: [s, 1 / s^2, 1 / s, -, absmax partials ...]
    int64_t T, K, ldx, Kp, nchunks, nstages;
    int64_t ldc;             // leading dimension of C (K for a Hessian; the factor's trailing blocks live inside a larger matrix)
    int64_t split0;          // first block of this item in the split launch; its split grid is (Kp / 256) x ceil(nchunks / 4)
    int64_t tile0;           // first block of this item in the product launch
    float alpha, beta;
    int32_t tn, absmax_blocks;
    int32_t mirror_all;      // 1: H comes out exactly symmetric, both triangles written; 0: lower halves only inside diagonal tiles
    int32_t pad_;
};

constexpr int kManyAbsmaxBlocks = 256;   // row slices per item in the grouped absmax (64 left an 8192 x 8192 operand of the factor to 64 blocks: 0.5 ms)

// public items (oq_hip.h: int64 {X, H, T, K, ldx, n_seen, n_add, 0}) -> SyrkItem table; one thread, count is small
__global__ void syrk_many_plan_kernel(const int64_t* __restrict__ pub, const int count, unsigned char* __restrict__ pieces_base, SyrkItem* __restrict__ items) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t split0 = 0, tile0 = 0;
    size_t off = 0;
    for (int m = 0; m < count; ++m) {
        const int64_t* it = pub + static_cast<int64_t>(m) * 8;
        SyrkItem o;
        o.X = reinterpret_cast<const float*>(it[0]);
        o.C = reinterpret_cast<float*>(it[1]);
        o.T = it[2]; o.K = it[3]; o.ldx = it[4]; o.ldc = it[3];
        const int64_t n_seen = it[5], n_total = it[5] + it[6];
        o.Kp = (o.K + kST - 1) / kST * kST;
        o.nstages = (o.T + StageGeom<3>::ROWS - 1) / StageGeom<3>::ROWS;
        o.nchunks = o.nstages * StageGeom<3>::CH;
        o.scale = reinterpret_cast<float*>(pieces_base + off);
        o.P = reinterpret_cast<u32x4*>(pieces_base + off + 16384);
        off += 16384 + static_cast<size_t>(o.nchunks) * 2 * static_cast<size_t>(o.Kp) * 16;
        o.split0 = split0;
        split0 += (o.Kp / 256) * ((o.nchunks + 3) / 4);
        o.tn = static_cast<int32_t>(o.Kp / kST);
        o.tile0 = tile0;
        tile0 += static_cast<int64_t>(o.tn) * (o.tn + 1) / 2;
        o.alpha = static_cast<float>(2.0 / static_cast<double>(n_total));                                               // as oq_hessian_accumulate_f32
        o.beta = n_seen == 0 ? 0.0f : static_cast<float>(static_cast<double>(n_seen) / static_cast<double>(n_total));   // gptq.py:254
        o.absmax_blocks = static_cast<int32_t>(o.T < kManyAbsmaxBlocks ? o.T : kManyAbsmaxBlocks);
        o.mirror_all = 1; o.pad_ = 0;
        items[m] = o;
    }
}

// The deferred trailing update of the blocked Cholesky (factor.hip) as `count` items of the same shape:
// C_m[pend:, pend:] -= A_m^T A_m with A_m = Lt_m[O:pend, pend:K] (Kd = pend - O rows, k-major like a batch of activations),
// both inside K x K matrices `ms` floats apart.  alpha = -1, beta = 1, both triangles written (the factor keeps them current).
__global__ void syrk_factor_plan_kernel(const float* __restrict__ Lt, float* __restrict__ P, const int64_t ms, const int count, const int64_t K,
                                        const int64_t O, const int64_t pend, unsigned char* __restrict__ pieces_base, SyrkItem* __restrict__ items) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= count) return;
    SyrkItem o;
    o.X = Lt + m * ms + O * K + pend;
    o.C = P + m * ms + pend * K + pend;
    o.T = pend - O; o.K = K - pend; o.ldx = K; o.ldc = K;
    o.Kp = (o.K + kST - 1) / kST * kST;
    o.nstages = (o.T + StageGeom<3>::ROWS - 1) / StageGeom<3>::ROWS;
    o.nchunks = o.nstages * StageGeom<3>::CH;
    const size_t item_bytes = 16384 + static_cast<size_t>(o.nchunks) * 2 * static_cast<size_t>(o.Kp) * 16;
    o.scale = reinterpret_cast<float*>(pieces_base + m * item_bytes);
    o.P = reinterpret_cast<u32x4*>(pieces_base + m * item_bytes + 16384);
    o.split0 = m * ((o.Kp / 256) * ((o.nchunks + 3) / 4));
    o.tn = static_cast<int32_t>(o.Kp / kST);
    o.tile0 = m * (static_cast<int64_t>(o.tn) * (o.tn + 1) / 2);
    o.alpha = -1.0f;
    o.beta = 1.0f;
    o.absmax_blocks = static_cast<int32_t>(o.T < kManyAbsmaxBlocks ? o.T : kManyAbsmaxBlocks);
    // what reads the trailing square later: rows of the upper triangle (panel GEMM) and whole 128 x 128 diagonal blocks
    // (chol_diag_kernel) -- the lower halves of off-diagonal tiles (a 4-byte scatter down columns) are never read
    o.mirror_all = 0; o.pad_ = 0;
    items[m] = o;
}

// the item whose block range holds `id`: FIRST(items[m]) <= id < FIRST(items[m + 1])
template <typename First>
__device__ __forceinline__ int item_of_block(const SyrkItem* __restrict__ items, const int count, const int64_t id, First first) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first(items[mid]) <= id) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void absmax_partial_many_kernel(const SyrkItem* __restrict__ items) {
    const SyrkItem& it = items[blockIdx.y];
    if (static_cast<int>(blockIdx.x) >= it.absmax_blocks) return;
    absmax_partial_body(it.X, it.T, it.K, it.ldx, it.scale + 4, it.absmax_blocks, static_cast<int64_t>(blockIdx.x));
}

__global__ __launch_bounds__(256) void absmax_scale_many_kernel(const SyrkItem* __restrict__ items) {
    const SyrkItem& it = items[blockIdx.x];
    absmax_scale_body(it.scale + 4, it.absmax_blocks, it.scale);
}

__global__ __launch_bounds__(256) void split_f16x2_many_kernel(const SyrkItem* __restrict__ items, const int count) {
    const int64_t id = blockIdx.x;
    const int m = item_of_block(items, count, id, [](const SyrkItem& i) { return i.split0; });
    const SyrkItem& it = items[m];
    const int64_t local = id - it.split0, nbx = it.Kp / 256;
    split_f16x2_body(it.X, it.T, it.K, it.ldx, it.Kp, it.nchunks, it.scale, it.alpha, it.P, local % nbx, local / nbx);
}

__global__ __launch_bounds__(kSThreads) void syrk_f16_m16_many_kernel(const SyrkItem* __restrict__ items, const int count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int m = item_of_block(items, count, id, [](const SyrkItem& i) { return i.tile0; });
    const SyrkItem& it = items[m];
    syrk_f16_m16_body(it.P, it.K, it.Kp, it.nstages, it.alpha, it.beta, it.C, nullptr, it.nstages, it.tn, it.scale,
                      static_cast<int>(id - it.tile0), 0, lds, it.ldc, it.mirror_all != 0);
}

// ---- 3. Two-operand GEMM on fp16 pieces: C = beta C + alpha A^T B, A [Kd, M] and B [Kd, N] both k-major (gemm_tn.hpp's
// convention), each split into two fp16 pieces of its own power-of-two scaled elements (same pieces, same three products and
// the same stage loop as the SYRK above: 22-bit operands, fp32 accumulate).  Users: the lazy batch updates of the corrected
// GPTQ loop (gptq.py:208: W[i2:] -= U[i1:i2, i2:]^T Err, Kd = 512) and the AWQ / clip searches' X (W - W^) products
// (pre_passes/awq.py:177, :247), whose squared-error loss is reduced in the epilogue without ever writing the [T, N] product.
struct PieceGemm {
    const u32x4* PA;        // pieces of A: [stage chunks][2][Mp] x 16 B   (Mp, Np: M, N padded to 256)
    const u32x4* PB;
    int64_t M, N, Mp, Np, nstages;
    const float* scale_a;   // device: [s, 1 / s^2, 1 / s] of A (absmax_scale_kernel)
    const float* scale_b;
    const float* row_unscale_a = nullptr;   // optional: 1 / s_m per row m of the result (A split with one scale per column of A = row of its source)
    float alpha, beta;
    float* C;               // EPI 0: [M, N], leading dimension ldc
    int64_t ldc;
    float* partial;         // EPI 1: per-block sum of squares of the block's part of A^T B; EPI 2: per-block sum of (A^T B) o C
    float* Ct = nullptr;    // EPI 0, optional: the same values transposed, Ct[n][m], leading dimension ldct
    int64_t ldct = 0;
    int32_t k_from_n = 0;   // B[k][n] = 0 for k < n (lower triangular B): a column tile starts its contraction at its first column
    int32_t k_to_m = 0;     // A[k][m] = 0 for k > m (upper triangular A^T): a row tile ends its contraction behind its last row
};

template <int EPI, int PRODUCTS = 3>
__device__ __forceinline__ void gemm_f16x3_body(const PieceGemm& g, const int tile, unsigned char* lds) {
    using G = StageGeom<3>;
    constexpr int PIECES = 2, CH = 4, NDMA = G::NDMA;
    const int tiles_n = static_cast<int>(g.Np / kST);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int64_t m0 = static_cast<int64_t>(tile_m) * kST, n0 = static_cast<int64_t>(tile_n) * kST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // triangular operands (the recursive-doubling inverse of factor.hip): stages that only meet structural zeros are skipped
    int64_t s_begin = 0, s_end = g.nstages;
    if (g.k_from_n) s_begin = n0 / G::ROWS;
    if (g.k_to_m) { const int64_t e = (m0 + kST + G::ROWS - 1) / G::ROWS; s_end = e < s_end ? e : s_end; }
    if (s_begin > s_end) s_begin = s_end;
    const char* gsrc[8];
    int64_t stage_bytes[8];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int q = wave * NDMA + i;
        const int quarter = q & 3, cc = (q >> 2) % CH, pc = (q / (4 * CH)) % PIECES, op = q / (4 * CH * PIECES);
        const int64_t width = op ? g.Np : g.Mp;
        const int64_t colq = (op ? n0 : m0) + quarter * 64 + lane;
        stage_bytes[i] = static_cast<int64_t>(CH) * PIECES * width * 16;
        gsrc[i] = reinterpret_cast<const char*>((op ? g.PB : g.PA) + (static_cast<int64_t>(cc) * PIECES + pc) * width + colq) + s_begin * stage_bytes[i];
    }
    f32x4v acc[4][8];
    if (s_end > s_begin) {
        if constexpr (PRODUCTS == 1) {
            const char* hsrc[4];
            int64_t hbytes[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = wave * 4 + i;
                const int quarter = q & 3, cc = (q >> 2) & 3, op = q >> 4;
                const int64_t width = op ? g.Np : g.Mp;
                hbytes[i] = static_cast<int64_t>(CH) * PIECES * width * 16;
                hsrc[i] = reinterpret_cast<const char*>((op ? g.PB : g.PA) + static_cast<int64_t>(cc) * PIECES * width + (op ? n0 : m0) + quarter * 64 + lane) +
                          s_begin * hbytes[i];
            }
            f16_hi_mainloop(hsrc, hbytes, s_end - s_begin, lds, acc);
        } else {
            f16_m16_mainloop<NDMA, PRODUCTS>(gsrc, stage_bytes, s_end - s_begin, lds, acc);
        }
    } else {   // block-uniform: nothing to contract
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    const float unscale = g.scale_a[2] * g.scale_b[2];   // 1 / (s_a s_b): exact, powers of two
    if constexpr (EPI == 0) {
        // the two powers of two one after the other (alpha 2^-b, then 2^-a on the sum): the same bits as alpha 2^-(a + b) whenever
        // that product is a normal float, and no vanishing / overflowing factor when the operands' scales are far apart or extreme
        const float alpha = g.alpha * g.scale_b[2], ua = g.scale_a[2];
        const bool vec_ok = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15u) == 0;
        auto rows = [&](int r, int c4, f32x4v v) -> f32x4v {
            const int64_t row = m0 + r, col = n0 + c4;
            if (row >= g.M || col >= g.N) return v;
            float* o = g.C + row * g.ldc + col;
            const bool vec = vec_ok && col + 3 < g.N;
            f32x4v old = {0.f, 0.f, 0.f, 0.f};
            if (g.beta != 0.0f) {
                if (vec) old = *reinterpret_cast<const f32x4v*>(o);
                else for (int q = 0; q < 4 && col + q < g.N; ++q) old[q] = o[q];
            }
            f32x4v val;
            const float ua_row = g.row_unscale_a != nullptr ? g.row_unscale_a[row] : ua;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = alpha * (v[q] * ua_row);
                if (g.beta != 0.0f) x = g.beta * old[q] + x;
                val[q] = x;
            }
            if (vec) *reinterpret_cast<f32x4v*>(o) = val;
            else for (int q = 0; q < 4 && col + q < g.N; ++q) o[q] = val[q];
            return val;
        };
        auto cols = [&](int c, int r, float x) {
            if (m0 + r < g.M && n0 + c < g.N) g.Ct[(n0 + c) * g.ldct + m0 + r] = x;
        };
        if (g.Ct != nullptr) staged_tile_epilogue<true>(acc, lds, rows, cols);
        else staged_tile_epilogue<false>(acc, lds, rows, cols);
    } else if constexpr (EPI == 2) {
        // sum over the block's part of (A^T B) o C, C [M, N] read only (the quadratic form <D, G D> of the AWQ searches' Gram
        // route: A = G, B = D, C = D); the tile goes through LDS so that every lane reads 16 contiguous bytes of C
        float sum = 0.f;
        const bool vec_ok = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15u) == 0;
        auto rows = [&](int r, int c4, f32x4v v) -> f32x4v {
            const int64_t row = m0 + r, col = n0 + c4;
            if (row >= g.M || col >= g.N) return v;
            const float* o = g.C + row * g.ldc + col;
            if (vec_ok && col + 3 < g.N) {
                const f32x4v d = *reinterpret_cast<const f32x4v*>(o);
#pragma unroll
                for (int q = 0; q < 4; ++q) sum += (v[q] * unscale) * d[q];
            } else {
                for (int q = 0; q < 4 && col + q < g.N; ++q) sum += (v[q] * unscale) * o[q];
            }
            return v;
        };
        auto cols = [&](int, int, float) {};
        staged_tile_epilogue<false>(acc, lds, rows, cols);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        __syncthreads();                                  // the staged tile has been read by everybody
        float* red = reinterpret_cast<float*>(lds);
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < kSThreads / 64; ++w) t += red[w];   // fixed order: deterministic
            g.partial[tile] = t;
        }
    } else {
        // rows / columns past M / N are zero pieces: their products are exact zeros, no masks
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][e] * unscale;
                    sum += v * v;
                }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        float* red = reinterpret_cast<float*>(lds);   // the ring is free behind the stage loop's last barrier
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < kSThreads / 64; ++w) t += red[w];   // fixed order: deterministic
            g.partial[tile] = t;
        }
    }
}

template <int EPI, int PRODUCTS = 3>
__global__ __launch_bounds__(kSThreads) void gemm_f16x3_kernel(const PieceGemm g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    gemm_f16x3_body<EPI, PRODUCTS>(g, static_cast<int>(blockIdx.x), lds);
}

// many products of one shape in one launch: blockIdx.y = problem (its PieceGemm in device memory: uniform scalar loads)
__global__ __launch_bounds__(kSThreads) void gemm_f16x3_many_kernel(const PieceGemm* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const PieceGemm g = items[blockIdx.y];
    gemm_f16x3_body<0>(g, static_cast<int>(blockIdx.x), lds);
}

// ---- 3b. One level of the recursive-doubling inverse of factor.hip on the fp16-piece kernels.  With L = [[L11, 0], [L21, L22]]
// and X = inv(L):  X21 = -X22 (L21 X11).  Per problem q (pair p of the level, matrix m of the batch), b = size of the first
// block, b2 = rows of the second:
//   S   [b2, b]  = sum_k Lt[o1 + k][o2 + r] X[o1 + k][o1 + j]      (X11 lower triangular: k >= j)
//   X21 [b2, b]  = -sum_c Y[o2 + c][o2 + r] S[c][j]                 (Y22 = X22^T upper triangular: c <= r),  Y12 = X21^T
// Operand pieces come from the grouped preparation kernels of section 2c (SyrkItem tables: only X, T, K, ldx, P, scale, split0
// and absmax_blocks are read by them); this kernel lays out the tables.  One thread per problem.
struct InverseLevel {
    const float* Lt; float* X; float* Y; float* S;
    int64_t ms, K, b, b2, first, pairs;
    int32_t count;
};

__device__ __forceinline__ size_t piece_bytes_dev(int64_t T, int64_t cols) {
    const int64_t st = (T + StageGeom<3>::ROWS - 1) / StageGeom<3>::ROWS, cp = (cols + kST - 1) / kST * kST;
    return 16384 + static_cast<size_t>(st) * StageGeom<3>::CH * 2 * static_cast<size_t>(cp) * 16;
}

__device__ __forceinline__ SyrkItem prep_item(const float* X, int64_t T, int64_t cols, int64_t ldx, unsigned char* where, int64_t split0) {
    SyrkItem o;
    o.X = X; o.C = nullptr; o.T = T; o.K = cols; o.ldx = ldx; o.ldc = 0;
    o.Kp = (cols + kST - 1) / kST * kST;
    o.nstages = (T + StageGeom<3>::ROWS - 1) / StageGeom<3>::ROWS;
    o.nchunks = o.nstages * StageGeom<3>::CH;
    o.scale = reinterpret_cast<float*>(where);
    o.P = reinterpret_cast<u32x4*>(where + 16384);
    o.split0 = split0; o.tile0 = 0; o.tn = 0;
    o.alpha = -1.0f; o.beta = 0.0f;          // alpha <= 0: no dead-channel bookkeeping in the split (a Hessian matter)
    o.absmax_blocks = static_cast<int32_t>(T < kManyAbsmaxBlocks ? T : kManyAbsmaxBlocks);
    o.mirror_all = 0; o.pad_ = 0;
    return o;
}

__device__ __forceinline__ int64_t split_blocks_of(int64_t T, int64_t cols) {
    const int64_t st = (T + StageGeom<3>::ROWS - 1) / StageGeom<3>::ROWS, cp = (cols + kST - 1) / kST * kST;
    return (cp / 256) * ((st * StageGeom<3>::CH + 3) / 4);
}

__global__ void inverse_level_plan_kernel(const InverseLevel lv, unsigned char* __restrict__ pieces_base, SyrkItem* __restrict__ prep,
                                          PieceGemm* __restrict__ gemms) {
    const int64_t n = lv.pairs * lv.count;
    const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int64_t m = q / lv.pairs, p = q - m * lv.pairs;
    const int64_t b = lv.b, b2 = lv.b2, K = lv.K;
    const int64_t o1 = (lv.first + p) * 2 * b, o2 = o1 + b;
    const float* Lt = lv.Lt + m * lv.ms;
    float* X = lv.X + m * lv.ms;
    float* Y = lv.Y + m * lv.ms;
    float* S = lv.S + m * lv.ms + (lv.first + p) * b * b;
    const size_t bA1 = piece_bytes_dev(b, b2), bB1 = piece_bytes_dev(b, b), bA2 = piece_bytes_dev(b2, b2), bS = piece_bytes_dev(b2, b);
    unsigned char* w = pieces_base + static_cast<size_t>(q) * (bA1 + bB1 + bA2 + bS);
    const int64_t sA1 = split_blocks_of(b, b2), sB1 = split_blocks_of(b, b), sA2 = split_blocks_of(b2, b2), sS = split_blocks_of(b2, b);
    // first preparation launch: A1 | B1 | A2 (3 n items), second: S (n items, block ids from 0 again)
    prep[q] = prep_item(Lt + o1 * K + o2, b, b2, K, w, q * sA1);
    prep[n + q] = prep_item(X + o1 * K + o1, b, b, K, w + bA1, n * sA1 + q * sB1);
    prep[2 * n + q] = prep_item(Y + o2 * K + o2, b2, b2, K, w + bA1 + bB1, n * (sA1 + sB1) + q * sA2);
    prep[3 * n + q] = prep_item(S, b2, b, b, w + bA1 + bB1 + bA2, q * sS);
    PieceGemm g1;
    g1.PA = prep[q].P; g1.scale_a = prep[q].scale;
    g1.PB = prep[n + q].P; g1.scale_b = prep[n + q].scale;
    g1.M = b2; g1.N = b; g1.Mp = prep[q].Kp; g1.Np = prep[n + q].Kp; g1.nstages = prep[q].nstages;
    g1.alpha = 1.0f; g1.beta = 0.0f; g1.C = S; g1.ldc = b; g1.partial = nullptr; g1.Ct = nullptr; g1.ldct = 0;
    g1.k_from_n = 1; g1.k_to_m = 0;
    gemms[q] = g1;
    PieceGemm g2;
    g2.PA = prep[2 * n + q].P; g2.scale_a = prep[2 * n + q].scale;
    g2.PB = prep[3 * n + q].P; g2.scale_b = prep[3 * n + q].scale;
    g2.M = b2; g2.N = b; g2.Mp = prep[2 * n + q].Kp; g2.Np = prep[3 * n + q].Kp; g2.nstages = prep[2 * n + q].nstages;
    g2.alpha = -1.0f; g2.beta = 0.0f; g2.C = X + o2 * K + o1; g2.ldc = K; g2.partial = nullptr;
    g2.Ct = Y + o1 * K + o2; g2.ldct = K;
    g2.k_from_n = 0; g2.k_to_m = 1;
    gemms[n + q] = g2;
}

// One power-of-two scale per ROW of a [T, K] row-major source (one wave per row): s_t = 2^(15 - exponent of the row's largest
// magnitude), clamped to 2^+-110; 1 for an all-zero or non-finite row.  scales[t] = s_t, scales[T + t] = 1 / s_t (exact).
__global__ __launch_bounds__(256) void row_pow2_scales_kernel(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                              float* __restrict__ scales) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int lane = threadIdx.x & 63;
    const float* src = X + t * ldx;
    float m = 0.f;
    for (int64_t k = lane; k < K; k += 64) m = nmax(m, fabsf(src[k]));
    m = wave_max(m);
    if (lane == 0) {
        float sc = 1.0f;
        if (m > 0.0f && m < INFINITY) {
            int ex = 0;
            (void)frexpf(m, &ex);
            int e = 15 - ex;
            e = e > 110 ? 110 : (e < -110 ? -110 : e);
            sc = ldexpf(1.0f, e);
        }
        scales[t] = sc;
        scales[T + t] = 1.0f / sc;
    }
}

// Source with the contraction index as its FAST axis (X [T, K] row-major used as A = X^T: contraction over k, columns t):
// pieces P[k / 8][piece][t padded to 256].  A thread takes one row t and 8 consecutive k (32 contiguous bytes).
__global__ __launch_bounds__(256) void split_f16x2_fast_axis_kernel(const float* __restrict__ X, const int64_t T, const int64_t K, const int64_t ldx,
                                                                    const int64_t Tp, const int64_t nchunks, const float* __restrict__ scale,
                                                                    u32x4* __restrict__ P, const float* __restrict__ row_scale = nullptr) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const bool ok = t < T;
    const float* src = X + (ok ? t : T - 1) * ldx;
    const float sc = row_scale != nullptr ? row_scale[ok ? t : T - 1] : scale[0];   // one power of two per row (plain GEMM operands) or per operand
    for (int64_t c = static_cast<int64_t>(blockIdx.y) * 4; c < nchunks && c < static_cast<int64_t>(blockIdx.y) * 4 + 4; ++c) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t k = c * 8 + r;
            const float x = src[k < K ? k : K - 1];
            v[r] = (k < K && ok) ? x * sc : 0.f;
        }
        u32x4 hi, lo;
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const f16x2 h2 = {static_cast<_Float16>(v[2 * rp]), static_cast<_Float16>(v[2 * rp + 1])};
            hi[rp] = __builtin_bit_cast(uint32_t, h2);
            lo[rp] = pk_f16(v[2 * rp] - static_cast<float>(h2[0]), v[2 * rp + 1] - static_cast<float>(h2[1]));
        }
        u32x4* o = P + (c * 2) * Tp + t;
        o[0] = hi;
        o[Tp] = lo;
    }
}

constexpr int kAbsmaxBlocks = 2048;
constexpr size_t kScaleHeaderBytes = 16384;   // fp16 pieces: [scale, 1 / scale^2, -, -] + the absmax partials, in front of P
static int64_t padded_k(int64_t K) { return ceil_div(K, kST) * kST; }
static int64_t stages_of(int64_t T, int rows) { return ceil_div(T, rows); }

size_t syrk_bf16x3_pieces_bytes(int64_t T, int64_t K) {
    if (T <= 0 || K <= 0) return 0;
    // room for either kind: three bf16 pieces of 16-row stages, or the scale header + two fp16 pieces of 32-row stages
    const size_t b16 = static_cast<size_t>(stages_of(T, StageGeom<6>::ROWS)) * StageGeom<6>::CH * StageGeom<6>::PIECES;
    const size_t f16 = static_cast<size_t>(stages_of(T, StageGeom<3>::ROWS)) * StageGeom<3>::CH * StageGeom<3>::PIECES;
    return (b16 > f16 ? b16 : f16) * padded_k(K) * 16 + kScaleHeaderBytes;
}

// phases: 1 = X -> pieces (absmax / scale / split), 2 = pieces -> C (GEMM + slab reduce), 3 = both.  `base`: the 256-byte
// aligned pieces region (syrk_bf16x3_pieces_bytes); `slab`: room for the T-slice partial sums (may be null).  The two phases
// of one batch may run on different streams (oq_hessian_prepare_f32 / oq_hessian_accumulate_prepared_f32): phase 1 is
// HBM-bound, phase 2 matrix-core bound, so the preparation of the next batch hides behind the product of this one.
int32_t syrk_pieces_phases(const float* X, int64_t T, int64_t K, int64_t ldx, float alpha, float beta, float* C, unsigned char* base, float* slab,
                           size_t slab_bytes, int terms, int phases, hipStream_t s);

int32_t launch_syrk_bf16x3(const float* X, int64_t T, int64_t K, int64_t ldx, float alpha, float beta, float* C, void* workspace,
                           size_t workspace_bytes, int terms, hipStream_t s) {
    OQ_REQUIRE(X != nullptr && C != nullptr && T > 0 && K >= 1 && ldx >= K, OQ_ERR_INVALID_ARGUMENT, "syrk_bf16x3: bad argument");
    OQ_REQUIRE(terms == 3 || terms == 6 || terms == 9, OQ_ERR_INVALID_ARGUMENT, "syrk_bf16x3: terms must be 3 (fp16 pieces), 6 or 9");
    const size_t pieces = syrk_bf16x3_pieces_bytes(T, K);
    OQ_REQUIRE(workspace != nullptr && workspace_bytes >= pieces + 256, OQ_ERR_WORKSPACE,
               "syrk_bf16x3: workspace of %zu bytes needed for the operand pieces, %zu given", pieces + 256, workspace_bytes);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255u)) & 255u;
    return syrk_pieces_phases(X, T, K, ldx, alpha, beta, C, base, reinterpret_cast<float*>(base + pieces), workspace_bytes - pieces - 256, terms, 3, s);
}

int32_t syrk_pieces_phases(const float* X, int64_t T, int64_t K, int64_t ldx, float alpha, float beta, float* C, unsigned char* base, float* slab,
                           size_t slab_bytes, int terms, int phases, hipStream_t s) {
    const bool f16 = terms == 3;
    // fp16 pieces take 2/3 of the bf16 pieces' room: the scale header lives in the spare third
    float* scale = f16 ? reinterpret_cast<float*>(base) : nullptr;
    u32x4* P = reinterpret_cast<u32x4*>(base + (f16 ? kScaleHeaderBytes : 0));
    if (slab == nullptr) slab_bytes = 0;
    // per launch, not once: the attribute belongs to the current device's copy of the kernel
#ifdef OQ_SYRK_LAB
    static const bool f16_m16_env = [] { const char* v = getenv("OQ_SYRK_F16_M16"); return !v || atoi(v) != 0; }();   // lab builds only; 0: the 32x32x16 form
    const bool f16_m16 = f16 && f16_m16_env;
#else
    const bool f16_m16 = f16;      // the shipped library reads no environment here
#endif
    const void* kfn = f16_m16 ? reinterpret_cast<const void*>(&syrk_f16_m16_kernel) : terms == 9 ? reinterpret_cast<const void*>(&syrk_pieces_kernel<9>)
                                 : (terms == 6 ? reinterpret_cast<const void*>(&syrk_pieces_kernel<6>) : reinterpret_cast<const void*>(&syrk_pieces_kernel<3>));
    const int lds_bytes = f16 ? StageGeom<3>::LDS : StageGeom<6>::LDS;
    const int stage_rows = f16 ? StageGeom<3>::ROWS : StageGeom<6>::ROWS, stage_chunks = stage_rows / 8;
    hipError_t e1 = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    OQ_REQUIRE(e1 == hipSuccess, OQ_ERR_LAUNCH, "syrk_bf16x3: cannot reserve %d bytes of LDS", lds_bytes);
    const int64_t Kp = padded_k(K), nstages = stages_of(T, stage_rows), nchunks = nstages * stage_chunks;
    OQ_REQUIRE(ceil_div(nchunks, 4) <= 65535, OQ_ERR_UNSUPPORTED, "syrk_bf16x3: at most 2 097 120 rows per call, %lld given", (long long)T);
    int32_t st = OQ_OK;
    if (!(phases & 1)) {
        // pieces prepared by an earlier call
    } else if (f16) {
        const int nb = static_cast<int>(T < kAbsmaxBlocks ? T : kAbsmaxBlocks);
        hipLaunchKernelGGL(absmax_partial_kernel, dim3(static_cast<uint32_t>(nb)), dim3(256), 0, s, X, T, K, ldx, scale + 4);
        hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(256), 0, s, scale + 4, nb, scale);
        hipLaunchKernelGGL(split_f16x2_kernel, dim3(static_cast<uint32_t>(Kp / 256), static_cast<uint32_t>(ceil_div(nchunks, 4))), dim3(256), 0, s,
                           X, T, K, ldx, Kp, nchunks, scale, alpha, P);
        st = check_launch("split_f16x2_kernel");
    } else {
        hipLaunchKernelGGL(split_bf16x3_kernel, dim3(static_cast<uint32_t>(Kp / 256), static_cast<uint32_t>(ceil_div(nchunks, 4))), dim3(256), 0, s,
                           X, T, K, ldx, Kp, nchunks, P);
        st = check_launch("split_bf16x3_kernel");
    }
    if (st != OQ_OK || !(phases & 2)) return st;

    const int tn = static_cast<int>(Kp / kST);
    const int64_t tiles = static_cast<int64_t>(tn) * (tn + 1) / 2;
    // T-slices: one block per CU (128-144 KB of LDS), 256 CUs.  Choose the slice count (<= 16, slices of >= 512 rows, slab
    // permitting) whose block count fills whole rounds of 256 best; ties go to fewer slices.  Measured on K = 4096, 65 536
    // rows (scripts/lab_syrk_splits.py, ms per product): 1 slice 3.26, 3: 2.72, 5: 2.57, 7: 2.54, 9: 2.53, 15: 2.61; K = 11008
    // is flat (1 slice 17.04, 4: 17.02).  A cost model that prefers fewer slices (5 and 1 for these shapes) was tried and
    // measured equal in time, but every slice is also a shorter fp32 accumulation chain: with one slice instead of four the
    // error against float64 of an 8192-row K = 11008 call rose from 8e-7 to 4e-6 of max |H| (bound: 1e-5).  More slices stay.
    int splits = 1;
    {
        const int64_t min_stages = 512 / stage_rows;                 // slices of >= 512 rows
        const int64_t by_rows = nstages / min_stages > 0 ? nstages / min_stages : 1;
        int64_t cap = by_rows < 16 ? by_rows : 16;
        while (cap > 1 && static_cast<size_t>(cap) * K * K * sizeof(float) > slab_bytes) --cap;
        double best = -1.0;
        for (int c = 1; c <= cap; ++c) {
            const int64_t blocks = tiles * c;
            const double fill = static_cast<double>(blocks) / static_cast<double>(ceil_div(blocks, 256) * 256);
            if (fill > best + 0.02) { best = fill; splits = c; }      // more slices only for a real gain (each costs a K x K pass)
        }
    }
#ifdef OQ_SYRK_LAB
    if (const char* v = getenv("OQ_SYRK_SPLITS")) {      // lab builds only (scripts/lab_syrk_splits.sh): force the slice count
        const int c = atoi(v);
        if (c >= 1 && (c == 1 || static_cast<size_t>(c) * K * K * sizeof(float) <= slab_bytes)) splits = c;
    }
#endif
    int64_t per = ceil_div(nstages, splits);
    splits = static_cast<int>(ceil_div(nstages, per));
    float* slab_f = splits > 1 ? slab : nullptr;
    const dim3 grid(static_cast<uint32_t>(tiles), static_cast<uint32_t>(splits));
    const float* no_scale = nullptr;
    if (terms == 9)
        hipLaunchKernelGGL(syrk_pieces_kernel<9>, grid, dim3(kSThreads), lds_bytes, s, P, K, Kp, nstages, alpha, beta, C, slab_f, per, tn, no_scale);
    else if (terms == 6)
        hipLaunchKernelGGL(syrk_pieces_kernel<6>, grid, dim3(kSThreads), lds_bytes, s, P, K, Kp, nstages, alpha, beta, C, slab_f, per, tn, no_scale);
    else if (f16_m16)
        hipLaunchKernelGGL(syrk_f16_m16_kernel, grid, dim3(kSThreads), lds_bytes, s, P, K, Kp, nstages, alpha, beta, C, slab_f, per, tn,
                           static_cast<const float*>(scale));
    else
        hipLaunchKernelGGL(syrk_pieces_kernel<3>, grid, dim3(kSThreads), lds_bytes, s, P, K, Kp, nstages, alpha, beta, C, slab_f, per, tn,
                           static_cast<const float*>(scale));
    st = check_launch("syrk_pieces_kernel");
    if (st != OQ_OK || splits == 1) return st;
    return launch_syrk_reduce(slab_f, splits, K, alpha, beta, C, kST, s, f16 ? scale : nullptr);
}


// ---- many Hessian updates, one launch chain (section 2c)
static size_t syrk_many_item_bytes(int64_t T, int64_t K) {
    return kScaleHeaderBytes + static_cast<size_t>(stages_of(T, StageGeom<3>::ROWS)) * StageGeom<3>::CH * 2 * static_cast<size_t>(padded_k(K)) * 16;
}
static size_t syrk_many_table_bytes(int64_t count) { return (static_cast<size_t>(count) * sizeof(SyrkItem) + 255) / 256 * 256; }

size_t syrk_f16x3_many_workspace_bytes(const int64_t* items_host, int64_t count) {
    if (items_host == nullptr || count <= 0 || count > 65535) return 256;
    size_t total = syrk_many_table_bytes(count) + 512;
    for (int64_t m = 0; m < count; ++m) {
        const int64_t T = items_host[m * 8 + 2], K = items_host[m * 8 + 3];
        if (!matrix_ok(T, K, K) || K > kMaxHessianWidth) return 256;      // refused by the call itself
        total += syrk_many_item_bytes(T, K);
    }
    return total;
}

int32_t launch_syrk_f16x3_many(const int64_t* items_host, const int64_t* items_device, int64_t count, void* workspace, size_t workspace_bytes,
                               hipStream_t s) {
    OQ_REQUIRE(items_host && items_device && count > 0 && count <= 65535, OQ_ERR_INVALID_ARGUMENT, "hessian_many: 1 to 65535 items, in host and device memory");
    int64_t split_blocks = 0, tiles = 0;
    for (int64_t m = 0; m < count; ++m) {
        const int64_t* it = items_host + m * 8;
        OQ_REQUIRE(it[0] != 0 && it[1] != 0 && it[2] > 0 && it[3] > 0 && it[4] >= it[3] && it[5] >= 0 && it[6] > 0, OQ_ERR_INVALID_ARGUMENT,
                   "hessian_many: item %lld: X, H, T > 0, K > 0, ldx >= K, n_seen >= 0, n_add > 0 expected", (long long)m);
        OQ_REQUIRE(matrix_ok(it[2], it[3], it[4]) && it[3] <= kMaxHessianWidth && it[5] <= kMaxSamples && it[6] <= kMaxSamples, OQ_ERR_UNSUPPORTED,
                   "hessian_many: item %lld is too large", (long long)m);
        const int64_t Kp = padded_k(it[3]), nchunks = stages_of(it[2], StageGeom<3>::ROWS) * StageGeom<3>::CH, tn = Kp / kST;
        split_blocks += (Kp / 256) * ceil_div(nchunks, 4);
        tiles += tn * (tn + 1) / 2;
    }
    OQ_REQUIRE(split_blocks < (1ll << 31) && tiles < (1ll << 31), OQ_ERR_UNSUPPORTED, "hessian_many: too many blocks for one launch");
    const size_t need = syrk_f16x3_many_workspace_bytes(items_host, count);
    OQ_REQUIRE(workspace != nullptr && workspace_bytes >= need, OQ_ERR_WORKSPACE, "hessian_many: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255u)) & 255u;
    SyrkItem* table = reinterpret_cast<SyrkItem*>(base);
    unsigned char* pieces = base + syrk_many_table_bytes(count);
    const int n = static_cast<int>(count);
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk_f16_m16_many_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, StageGeom<3>::LDS);
    OQ_REQUIRE(e1 == hipSuccess, OQ_ERR_LAUNCH, "hessian_many: cannot reserve %d bytes of LDS", StageGeom<3>::LDS);
    hipLaunchKernelGGL(syrk_many_plan_kernel, dim3(1), dim3(64), 0, s, items_device, n, pieces, table);
    hipLaunchKernelGGL(absmax_partial_many_kernel, dim3(kManyAbsmaxBlocks, static_cast<uint32_t>(n)), dim3(256), 0, s, table);
    hipLaunchKernelGGL(absmax_scale_many_kernel, dim3(static_cast<uint32_t>(n)), dim3(256), 0, s, table);
    hipLaunchKernelGGL(split_f16x2_many_kernel, dim3(static_cast<uint32_t>(split_blocks)), dim3(256), 0, s, table, n);
    int32_t st = check_launch("split_f16x2_many_kernel");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(syrk_f16_m16_many_kernel, dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), StageGeom<3>::LDS, s, table, n);
    return check_launch("syrk_f16_m16_many_kernel");
}

size_t syrk_f16x3_factor_update_bytes(int64_t K, int64_t kd_max, int64_t count) {
    if (K <= 0 || kd_max <= 0 || count <= 0) return 0;
    return syrk_many_table_bytes(count) + static_cast<size_t>(count) * syrk_many_item_bytes(kd_max, K) + 512;
}

int32_t launch_syrk_f16x3_factor_update(const float* Lt, float* P, int64_t ms, int64_t count, int64_t K, int64_t O, int64_t pend, void* workspace,
                                        size_t workspace_bytes, hipStream_t s) {
    const int64_t rest = K - pend, kd = pend - O;
    OQ_REQUIRE(Lt && P && count > 0 && count <= 65535 && rest > 0 && kd > 0, OQ_ERR_INVALID_ARGUMENT, "factor update: bad argument");
    const size_t need = syrk_f16x3_factor_update_bytes(rest, kd, count);
    OQ_REQUIRE(workspace != nullptr && workspace_bytes >= need, OQ_ERR_WORKSPACE, "factor update: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255u)) & 255u;
    SyrkItem* table = reinterpret_cast<SyrkItem*>(base);
    unsigned char* pieces = base + syrk_many_table_bytes(count);
    const int n = static_cast<int>(count);
    const int64_t Kp = padded_k(rest), nchunks = stages_of(kd, StageGeom<3>::ROWS) * StageGeom<3>::CH, tn = Kp / kST;
    const int64_t split_blocks = count * (Kp / 256) * ceil_div(nchunks, 4), tiles = count * (tn * (tn + 1) / 2);
    OQ_REQUIRE(split_blocks < (1ll << 31) && tiles < (1ll << 31), OQ_ERR_UNSUPPORTED, "factor update: too many blocks for one launch");
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk_f16_m16_many_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, StageGeom<3>::LDS);
    OQ_REQUIRE(e1 == hipSuccess, OQ_ERR_LAUNCH, "factor update: cannot reserve %d bytes of LDS", StageGeom<3>::LDS);
    hipLaunchKernelGGL(syrk_factor_plan_kernel, dim3(static_cast<uint32_t>(ceil_div(count, 64))), dim3(64), 0, s, Lt, P, ms, n, K, O, pend, pieces, table);
    hipLaunchKernelGGL(absmax_partial_many_kernel, dim3(kManyAbsmaxBlocks, static_cast<uint32_t>(n)), dim3(256), 0, s, table);
    hipLaunchKernelGGL(absmax_scale_many_kernel, dim3(static_cast<uint32_t>(n)), dim3(256), 0, s, table);
    hipLaunchKernelGGL(split_f16x2_many_kernel, dim3(static_cast<uint32_t>(split_blocks)), dim3(256), 0, s, table, n);
    hipLaunchKernelGGL(syrk_f16_m16_many_kernel, dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), StageGeom<3>::LDS, s, table, n);
    return check_launch("syrk_f16_m16_many_kernel (factor update)");
}

static size_t piece_bytes_host(int64_t T, int64_t cols) {
    return kScaleHeaderBytes + static_cast<size_t>(stages_of(T, StageGeom<3>::ROWS)) * StageGeom<3>::CH * 2 * static_cast<size_t>(padded_k(cols)) * 16;
}
static int64_t split_blocks_host(int64_t T, int64_t cols) { return (padded_k(cols) / 256) * ceil_div(stages_of(T, StageGeom<3>::ROWS) * StageGeom<3>::CH, 4); }

size_t inverse_level_f16x3_bytes(int64_t b, int64_t b2, int64_t problems) {
    if (b <= 0 || b2 <= 0 || problems <= 0) return 0;
    const size_t tables = (static_cast<size_t>(problems) * (4 * sizeof(SyrkItem) + 2 * sizeof(PieceGemm)) + 255) / 256 * 256;
    return tables + static_cast<size_t>(problems) * (piece_bytes_host(b, b2) + piece_bytes_host(b, b) + piece_bytes_host(b2, b2) + piece_bytes_host(b2, b)) + 512;
}

int32_t launch_inverse_level_f16x3(const float* Lt, float* X, float* Y, float* S, int64_t ms, int64_t count, int64_t K, int64_t b, int64_t first,
                                   int64_t pairs, int64_t b2, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const int64_t n = pairs * count;
    OQ_REQUIRE(Lt && X && Y && S && n > 0 && n <= 65535 && b > 0 && b2 > 0 && b2 <= b, OQ_ERR_INVALID_ARGUMENT, "inverse level: bad argument");
    const size_t need = inverse_level_f16x3_bytes(b, b2, n);
    OQ_REQUIRE(workspace != nullptr && workspace_bytes >= need, OQ_ERR_WORKSPACE, "inverse level: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255u)) & 255u;
    SyrkItem* prep = reinterpret_cast<SyrkItem*>(base);
    PieceGemm* gemms = reinterpret_cast<PieceGemm*>(base + static_cast<size_t>(n) * 4 * sizeof(SyrkItem));
    unsigned char* pieces = base + (static_cast<size_t>(n) * (4 * sizeof(SyrkItem) + 2 * sizeof(PieceGemm)) + 255) / 256 * 256;
    const int64_t split1 = n * (split_blocks_host(b, b2) + split_blocks_host(b, b) + split_blocks_host(b2, b2)), split2 = n * split_blocks_host(b2, b);
    const int64_t tiles = (padded_k(b2) / kST) * (padded_k(b) / kST);
    OQ_REQUIRE(split1 < (1ll << 31) && split2 < (1ll << 31) && tiles < (1ll << 31), OQ_ERR_UNSUPPORTED, "inverse level: too many blocks for one launch");
    OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_many_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, StageGeom<3>::LDS) == hipSuccess,
               OQ_ERR_LAUNCH, "inverse level: cannot reserve %d bytes of LDS", StageGeom<3>::LDS);
    InverseLevel lv;
    lv.Lt = Lt; lv.X = X; lv.Y = Y; lv.S = S; lv.ms = ms; lv.K = K; lv.b = b; lv.b2 = b2; lv.first = first; lv.pairs = pairs;
    lv.count = static_cast<int32_t>(count);
    const uint32_t n3 = static_cast<uint32_t>(3 * n), n1 = static_cast<uint32_t>(n);
    hipLaunchKernelGGL(inverse_level_plan_kernel, dim3(static_cast<uint32_t>(ceil_div(n, 64))), dim3(64), 0, s, lv, pieces, prep, gemms);
    hipLaunchKernelGGL(absmax_partial_many_kernel, dim3(kManyAbsmaxBlocks, n3), dim3(256), 0, s, prep);
    hipLaunchKernelGGL(absmax_scale_many_kernel, dim3(n3), dim3(256), 0, s, prep);
    hipLaunchKernelGGL(split_f16x2_many_kernel, dim3(static_cast<uint32_t>(split1)), dim3(256), 0, s, prep, static_cast<int>(n3));
    hipLaunchKernelGGL(gemm_f16x3_many_kernel, dim3(static_cast<uint32_t>(tiles), n1), dim3(kSThreads), StageGeom<3>::LDS, s, gemms);
    hipLaunchKernelGGL(absmax_partial_many_kernel, dim3(kManyAbsmaxBlocks, n1), dim3(256), 0, s, prep + 3 * n);
    hipLaunchKernelGGL(absmax_scale_many_kernel, dim3(n1), dim3(256), 0, s, prep + 3 * n);
    hipLaunchKernelGGL(split_f16x2_many_kernel, dim3(static_cast<uint32_t>(split2)), dim3(256), 0, s, prep + 3 * n, static_cast<int>(n1));
    hipLaunchKernelGGL(gemm_f16x3_many_kernel, dim3(static_cast<uint32_t>(tiles), n1), dim3(kSThreads), StageGeom<3>::LDS, s, gemms + n);
    return check_launch("inverse level (fp16 pieces)");
}

// ---- host side of the two-operand GEMM (gemm_tn.hpp)
size_t gemm_f16x3_pieces_bytes(int64_t Kd, int64_t cols) {
    if (Kd <= 0 || cols <= 0) return 0;
    return static_cast<size_t>(stages_of(Kd, StageGeom<3>::ROWS)) * StageGeom<3>::CH * 2 * padded_k(cols) * 16 + kScaleHeaderBytes;
}

int32_t make_f16x2_pieces(const float* X, int64_t Kd, int64_t cols, int64_t ldx, bool contraction_is_fast_axis, void* pieces, hipStream_t s,
                          bool wide_range, float* row_scales) {
    OQ_REQUIRE(X && pieces && Kd > 0 && cols > 0 && (reinterpret_cast<uintptr_t>(pieces) & 255u) == 0, OQ_ERR_INVALID_ARGUMENT, "make_f16x2_pieces: bad argument");
    float* scale = static_cast<float*>(pieces);
    u32x4* P = reinterpret_cast<u32x4*>(static_cast<unsigned char*>(pieces) + kScaleHeaderBytes);
    const int64_t Cp = padded_k(cols), nstages = stages_of(Kd, StageGeom<3>::ROWS), nchunks = nstages * StageGeom<3>::CH;
    OQ_REQUIRE(ceil_div(nchunks, 4) <= 65535, OQ_ERR_UNSUPPORTED, "make_f16x2_pieces: contraction too long (%lld)", (long long)Kd);
    // absmax over the whole operand: rows x row length as stored
    const int64_t rows = contraction_is_fast_axis ? cols : Kd, rowlen = contraction_is_fast_axis ? Kd : cols;
    const int nb = static_cast<int>(rows < kAbsmaxBlocks ? rows : kAbsmaxBlocks);
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(static_cast<uint32_t>(nb)), dim3(256), 0, s, X, rows, rowlen, ldx, scale + 4);
    if (wide_range) hipLaunchKernelGGL(absmax_scale_wide_kernel, dim3(1), dim3(256), 0, s, scale + 4, nb, scale);
    else hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(256), 0, s, scale + 4, nb, scale);
    const dim3 grid(static_cast<uint32_t>(Cp / 256), static_cast<uint32_t>(ceil_div(nchunks, 4)));
    if (row_scales != nullptr) {   // one scale per row of the source (= per row of the product): row_scales[t] = s_t, [cols + t] = 1 / s_t
        OQ_REQUIRE(contraction_is_fast_axis, OQ_ERR_INVALID_ARGUMENT, "make_f16x2_pieces: per-row scales need the contraction on the fast axis");
        hipLaunchKernelGGL(row_pow2_scales_kernel, dim3(static_cast<uint32_t>(ceil_div(cols, 4))), dim3(256), 0, s, X, cols, Kd, ldx, row_scales);
        hipLaunchKernelGGL(split_f16x2_fast_axis_kernel, grid, dim3(256), 0, s, X, cols, Kd, ldx, Cp, nchunks, scale, P, static_cast<const float*>(row_scales));
        return check_launch("split_f16x2 (gemm pieces, per-row scales)");
    }
    if (contraction_is_fast_axis)
        hipLaunchKernelGGL(split_f16x2_fast_axis_kernel, grid, dim3(256), 0, s, X, cols, Kd, ldx, Cp, nchunks, scale, P, static_cast<const float*>(nullptr));
    else   // alpha = 0: no dead-channel guard (only the Hessian needs a vanishing sample to stay visible)
        hipLaunchKernelGGL(split_f16x2_kernel, grid, dim3(256), 0, s, X, Kd, cols, ldx, Cp, nchunks, scale, 0.0f, P);
    return check_launch("split_f16x2 (gemm pieces)");
}

// The same for a row-major [Kd, cols] source whose maximum |x| the caller already has as `npart` partial maxima on the
// device (a producer kernel that folded them while writing X): no second pass over X for the scale.
int32_t make_f16x2_pieces_from_partials(const float* X, int64_t Kd, int64_t cols, int64_t ldx, const float* absmax_partials, int npart, void* pieces,
                                        hipStream_t s, bool first_pieces_only) {
    OQ_REQUIRE(X && pieces && absmax_partials && npart > 0 && Kd > 0 && cols > 0 && (reinterpret_cast<uintptr_t>(pieces) & 255u) == 0, OQ_ERR_INVALID_ARGUMENT,
               "make_f16x2_pieces_from_partials: bad argument");
    float* scale = static_cast<float*>(pieces);
    u32x4* P = reinterpret_cast<u32x4*>(static_cast<unsigned char*>(pieces) + kScaleHeaderBytes);
    const int64_t Cp = padded_k(cols), nstages = stages_of(Kd, StageGeom<3>::ROWS), nchunks = nstages * StageGeom<3>::CH;
    OQ_REQUIRE(ceil_div(nchunks, 4) <= 65535, OQ_ERR_UNSUPPORTED, "make_f16x2_pieces_from_partials: contraction too long (%lld)", (long long)Kd);
    hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(256), 0, s, absmax_partials, npart, scale);
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(static_cast<uint32_t>(Cp / 256), static_cast<uint32_t>(ceil_div(nchunks, 4))), dim3(256), 0, s, X, Kd, cols, ldx,
                       Cp, nchunks, scale, first_pieces_only ? kSplitHiOnly : 0.0f, P);
    return check_launch("split_f16x2 (gemm pieces, given maxima)");
}

size_t gemm_f16x3_header_bytes() { return kScaleHeaderBytes; }
int64_t gemm_f16x3_padded_cols(int64_t cols) { return padded_k(cols); }
int64_t gemm_f16x3_chunks(int64_t Kd) { return stages_of(Kd, StageGeom<3>::ROWS) * StageGeom<3>::CH; }
int32_t launch_gemm_f16x3(const void* pieces_a, const void* pieces_b, int64_t M, int64_t N, int64_t Kd, float alpha, float beta, float* C,
                          int64_t ldc, float* loss_partial, hipStream_t s, bool hi_pieces_only, bool dot_with_c, bool b_first_piece_only,
                          const float* row_unscale_a) {
    OQ_REQUIRE(pieces_a && pieces_b && M > 0 && N > 0 && Kd > 0 && ((C != nullptr) != (loss_partial != nullptr) || dot_with_c), OQ_ERR_INVALID_ARGUMENT,
               "gemm_f16x3: bad argument");
    OQ_REQUIRE(!hi_pieces_only || (loss_partial && !dot_with_c), OQ_ERR_INVALID_ARGUMENT, "gemm_f16x3: the one-product form exists for the sum-of-squares epilogue only");
    OQ_REQUIRE(!dot_with_c || (C && loss_partial && ldc >= N), OQ_ERR_INVALID_ARGUMENT, "gemm_f16x3: the dot epilogue reads C [M, N] and writes one partial per block");
    PieceGemm g;
    g.scale_a = static_cast<const float*>(pieces_a);
    g.scale_b = static_cast<const float*>(pieces_b);
    g.PA = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(pieces_a) + kScaleHeaderBytes);
    g.PB = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(pieces_b) + kScaleHeaderBytes);
    g.M = M; g.N = N; g.Mp = padded_k(M); g.Np = padded_k(N); g.nstages = stages_of(Kd, StageGeom<3>::ROWS);
    g.alpha = alpha; g.beta = beta; g.C = C; g.ldc = ldc; g.partial = loss_partial;
    OQ_REQUIRE(row_unscale_a == nullptr || (C != nullptr && loss_partial == nullptr), OQ_ERR_INVALID_ARGUMENT, "gemm_f16x3: per-row scales exist for the store form only");
    g.row_unscale_a = row_unscale_a;
    const int64_t tiles = (g.Mp / kST) * (g.Np / kST);
    OQ_REQUIRE(tiles < (1 << 30), OQ_ERR_UNSUPPORTED, "gemm_f16x3: too many tiles");
    const int lds_bytes = StageGeom<3>::LDS;
    OQ_REQUIRE(!b_first_piece_only || dot_with_c, OQ_ERR_INVALID_ARGUMENT, "gemm_f16x3: the two-product form exists for the dot epilogue only");
    if (dot_with_c && b_first_piece_only) {
        OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess,
                   OQ_ERR_LAUNCH, "gemm_f16x3: cannot reserve %d bytes of LDS", lds_bytes);
        hipLaunchKernelGGL((gemm_f16x3_kernel<2, 2>), dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), lds_bytes, s, g);
    } else if (dot_with_c) {
        OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess,
                   OQ_ERR_LAUNCH, "gemm_f16x3: cannot reserve %d bytes of LDS", lds_bytes);
        hipLaunchKernelGGL(gemm_f16x3_kernel<2>, dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), lds_bytes, s, g);
    } else if (loss_partial && hi_pieces_only) {
        OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess,
                   OQ_ERR_LAUNCH, "gemm_f16x3: cannot reserve %d bytes of LDS", lds_bytes);
        hipLaunchKernelGGL((gemm_f16x3_kernel<1, 1>), dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), lds_bytes, s, g);
    } else if (loss_partial) {
        OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess,
                   OQ_ERR_LAUNCH, "gemm_f16x3: cannot reserve %d bytes of LDS", lds_bytes);
        hipLaunchKernelGGL(gemm_f16x3_kernel<1>, dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), lds_bytes, s, g);
    } else {
        OQ_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess,
                   OQ_ERR_LAUNCH, "gemm_f16x3: cannot reserve %d bytes of LDS", lds_bytes);
        hipLaunchKernelGGL(gemm_f16x3_kernel<0>, dim3(static_cast<uint32_t>(tiles)), dim3(kSThreads), lds_bytes, s, g);
    }
    return check_launch("gemm_f16x3_kernel");
}

int64_t gemm_f16x3_tiles(int64_t M, int64_t N) { return (padded_k(M) / kST) * (padded_k(N) / kST); }

}  // namespace oq
