// Internal helpers shared by the gfx950 kernels of liboq_hip.so.  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/oq_hip.h"

namespace oq {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded: no wave-size macro on gfx950)

// ------------------------------------------------------------------ errors (host side)
void set_error(const char* fmt, ...);
int32_t fail(int32_t status, const char* fmt, ...);
int32_t check_launch(const char* what);

#define OQ_REQUIRE(cond, status, ...)                 \
    do {                                              \
        if (!(cond)) return ::oq::fail((status), __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ------------------------------------------------------------------ argument ranges (host side)
// Every extent a caller passes is checked against these BEFORE any arithmetic on it, so that no product of two of them can
// wrap around and no grid dimension can be truncated: hostile arguments end in a status, never in a fault, a division by
// zero or a silently shortened launch (tests/test_hostile_arguments.py sweeps every entry point).
constexpr int64_t kMaxExtent = (1LL << 31) - 1;   // rows, columns, leading dimensions, tokens of one operand
constexpr int64_t kMaxElements = 1LL << 40;       // elements of one operand (4 TiB of fp32: far beyond the 288 GB of one GPU)
constexpr int64_t kMaxThreads = 1LL << 38;        // one thread per element, 256 per block: the grid's x extent stays below 2^31
constexpr int64_t kMaxHessianWidth = 1LL << 17;  // K of a [K, K] Hessian / factor: 64 GiB of fp32 at this bound
constexpr int64_t kMaxSamples = 1LL << 52;        // running sample counts: sums stay exact in a double
inline bool extent_ok(int64_t v) { return v > 0 && v <= kMaxExtent; }
inline bool count_ok(int64_t n, int64_t limit = kMaxElements) { return n > 0 && n <= limit; }
// rows x cols with leading dimension ld (both factors below 2^31: the product cannot overflow)
inline bool matrix_ok(int64_t rows, int64_t cols, int64_t ld) {
    return extent_ok(rows) && extent_ok(cols) && ld >= cols && ld <= kMaxExtent && rows * ld <= kMaxElements;
}

// ------------------------------------------------------------------ quantization grid (T1)
// core/_dtypes.py:8-30, :61-70.  Host-evaluated once per call and passed to kernels by value.
struct QGrid {
    int32_t qmin;      // qrange(symmetric, reduce_range): clamp range AND the range the scale is derived from
    int32_t qmax;
    int32_t symmetric;
    int32_t zero;      // symmetric zero point: round_half_even((qmax+qmin)/2)
    double levels;     // symmetric: min(qmax - zero, zero - qmin)   (float64 in the reference)
    float clip_ratio;  // utils.py:63-64
    int32_t bits;
};

bool qrange_host(int32_t qtype, int32_t symmetric, int32_t reduce_range, int64_t* qmin, int64_t* qmax);
int32_t make_grid(int32_t qtype, int32_t symmetric, int32_t reduce_range, float clip_ratio, QGrid* out);

// ------------------------------------------------------------------ device arithmetic
// All of this file is compiled with -ffp-contract=off: the reference rounds every product and sum
// separately, a fused multiply-add would change results.

// np.min / np.max / np.minimum / np.maximum / np.clip propagate NaN; fminf / fmaxf (IEEE minNum / maxNum) drop it and
// would turn a corrupted tensor into plausible finite ranges.  gfx950 has the IEEE-754-2019 forms as single instructions
// (v_minimum3_f32 / v_maximum3_f32), so NaN-propagating reductions cost the same as the dropping ones.
__host__ __device__ __forceinline__ float nmin(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__host__ __device__ __forceinline__ float nmax(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__host__ __device__ __forceinline__ double nmin(double a, double b) { return __builtin_elementwise_minimum(a, b); }
__host__ __device__ __forceinline__ double nmax(double a, double b) { return __builtin_elementwise_maximum(a, b); }

// Exchange with lane (l ^ OFF) for OFF = 8 / 16 / 32 without the LDS crossbar (`__shfl_xor` becomes ds_bpermute_b32):
// a DPP rotate inside the rows of 16 lanes, and gfx950's v_permlane16_swap / v_permlane32_swap, which exchange the odd
// rows (halves) of one register with the even rows (halves) of another -- fed the same value twice, the two results hold
// "mine" and "my partner's" in one order or the other, which is all a min / max fold needs.
template <int OFF, typename F>
__device__ __forceinline__ float xor_fold(float v, F f) {
    static_assert(OFF == 8 || OFF == 16 || OFF == 32, "lane distance");
    if constexpr (OFF == 8) {
        const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        return f(v, __int_as_float(t));
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return f(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return f(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}
template <int OFF> __device__ __forceinline__ float xor_min(float v) { return xor_fold<OFF>(v, [](float a, float b) { return nmin(a, b); }); }
template <int OFF> __device__ __forceinline__ float xor_max(float v) { return xor_fold<OFF>(v, [](float a, float b) { return nmax(a, b); }); }

struct QParam {
    float scale;
    int32_t zp;
};

// R1 tail + Q1: raw (min, max) of a row -> (scale, zero point).
//   utils.py:63-69   lo = min(min*clip, 0), hi = max(max*clip, 0)
//   utils.py:258-271 asymmetric   utils.py:273-298 symmetric (float64 division, see oracle)
__device__ __forceinline__ QParam qparam_from_range(float lo, float hi, const QGrid& g) {
    QParam p;
    if (g.symmetric) {
        float amax = nmax(fabsf(lo), fabsf(hi));
        double s = static_cast<double>(amax) / g.levels;
        if (s < static_cast<double>(FLT_MIN)) s = 1.0;
        p.scale = static_cast<float>(s);
        p.zp = g.zero;
    } else {
        float s = (hi - lo) / static_cast<float>(g.qmax - g.qmin);
        if (s < FLT_MIN) s = 1.0f;
        float z = static_cast<float>(g.qmin) - lo / s;
        z = nmin(nmax(z, static_cast<float>(g.qmin)), static_cast<float>(g.qmax));
        p.scale = s;
        p.zp = static_cast<int32_t>(rintf(z));
    }
    return p;
}

__device__ __forceinline__ QParam qparam_from_minmax(float mn, float mx, const QGrid& g) {
    float lo = nmin(mn * g.clip_ratio, 0.0f);
    float hi = nmax(mx * g.clip_ratio, 0.0f);
    return qparam_from_range(lo, hi, g);
}

// K1 utils.py:72-79: fp32 divide (IEEE, correctly rounded), round-half-even, int32, + zp, clamp.
__device__ __forceinline__ int32_t quantize_one(float x, float scale, int32_t zp, int32_t qmin, int32_t qmax) {
    int32_t r = static_cast<int32_t>(rintf(x / scale));
    r = static_cast<int32_t>(static_cast<uint32_t>(r) + static_cast<uint32_t>(zp));
    return min(max(r, qmin), qmax);
}

// ---- K1 fast path ---------------------------------------------------------------------------
// The IEEE divide of utils.py:73 costs ~11 VALU instructions per element, which makes the fused
// kernel VALU-bound instead of HBM-bound.  Per column we therefore keep rinv = fl(1/s) and compute
// t = fl(x * rinv), k = rint(t).  |t - fl(x/s)| <= |x/s| * (2^-23 + 2^-24) (two correctly rounded
// operations vs one), so whenever t is farther than `thr` = 0.5 - B*2^-21 from a half-integer
// (B >= max |x/s| of the column's group) no tie can separate rint(t) from rint(fl(x/s)) and k is
// exactly the reference's integer.  The rare elements inside that band (and any NaN/inf) are redone
// with the true division, so the result is bit-identical by construction, not by tolerance.
struct ColQ {
    float scale;
    float rinv;  // fl(1 / scale)
    float zpb;   // float(zp + bias), bias maps signed ranges onto 0..255 / 0..15 for v_cvt_pk_u8_f32
    float thr;   // <= 0 forces the exact path
    int32_t zp;
};

__device__ __forceinline__ ColQ make_colq(const QParam& p, float raw_min, float raw_max, int32_t bias) {
    ColQ c;
    c.scale = p.scale;
    c.zp = p.zp;
    c.rinv = 1.0f / p.scale;
    c.zpb = static_cast<float>(p.zp + bias);
    const float bound = nmax(fabsf(raw_min), fabsf(raw_max)) * c.rinv * 1.0001f;
    float thr = 0.5f - bound * 4.76837158203125e-07f /* 2^-21 */ - 1e-30f;
    if (!(p.scale < 1e30f) || !(bound < 4194304.0f /* 2^22: k + zp must stay exact in fp32 */)) thr = -1.0f;
    c.thr = thr;
    return c;
}

// Returns the clamped, biased level as an exact small float; sets `unsafe` when the fast product
// could not be proven equal to the reference (caller redoes the row with quantize_exact_biased).
__device__ __forceinline__ float quantize_fast_biased(float x, const ColQ& c, float lo_b, float hi_b, bool& unsafe) {
    const float t = x * c.rinv;
    const float k = rintf(t);
    unsafe = unsafe || !(fabsf(t - k) < c.thr);
    return __builtin_amdgcn_fmed3f(k + c.zpb, lo_b, hi_b);
}

__device__ __forceinline__ float quantize_exact_biased(float x, const ColQ& c, int32_t qmin, int32_t qmax, int32_t bias) {
    return static_cast<float>(quantize_one(x, c.scale, c.zp, qmin, qmax) + bias);
}

// Division by a divisor known ahead of the numerators (a row's scale, a pivot): r1 = refined_rcp(s) is the first four
// instructions of the compiler's own fp32 division sequence (rcp, fma(-s, r0, 1), fma(., r0, r0)), div_refined runs its
// remaining five on the numerator (mul, fma x 4) -- no v_div_scale / v_div_fmas / v_div_fixup, no branch.  That IS the
// correctly rounded quotient whenever v_div_scale would not rescale (numerator and quotient in [2^-100, 2^100], or a zero
// numerator); outside (degenerate data) the result may differ from IEEE in the last bit.
__device__ __forceinline__ float div_refined(float x, float neg_s, float r1) {
    const float q0 = x * r1;
    const float e1 = __builtin_fmaf(neg_s, q0, x);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(neg_s, q1, x);
    return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float refined_rcp(float s) {
    const float r0 = __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
}

// K2 utils.py:130-132: (f32(q) - f32(zp)) * scale, two roundings.
__device__ __forceinline__ float dequantize_one(int32_t q, float scale, int32_t zp) {
    return (static_cast<float>(q) - static_cast<float>(zp)) * scale;
}

// 64-lane butterfly reductions (no LDS).
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Blocks that share blockIdx % 8 share an XCD (and its L2) under the observed round-robin
// placement.  Remap so that each XCD works on one CONTIGUOUS range of logical block ids:
// speed only (partial lines of the scattered scale / zero-point outputs meet in one L2),
// never correctness.  Bijective for any nblk (cdna_hip_programming.md section 5, XCD swizzle).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7u, x = b & 7u;
    const uint32_t base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace oq
