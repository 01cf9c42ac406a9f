// HQQ zero-point optimisation (core/_algorithms/hqq.py:106-213 of the reference) for gfx950 -- SURVEY.md 8f, row N2.
//
// The reference alternates, for up to `iters` rounds over the WHOLE preprocessed array [N*K/g, g]:
//     w_q = clip(round(w * (1/s) + z));  w_r = (w_q - z) / (1/s);  w_e = shrink(w - w_r, beta, p)
//     err = mean |w - w_r|  (ONE number for the whole array: it steers `best` and the early stop)
//     z   = mean_row(w_q - (w - w_e) * (1/s))
// Rows (one k-group of one output column) are independent inside a round, the rounds are coupled only through
// `err`.  So a round is one elementwise kernel over W (read once per round: 4 B / element, HBM / Infinity-Cache
// bound with a division and a power per element) that leaves a per-block partial of sum|w - w_r|, and a one-block kernel that
// folds the partials in a fixed order and takes the reference's decision ON THE DEVICE (no host round trip):
// the next round's kernel applies it.  One thread = one row; neighbouring lanes = neighbouring columns, so every
// load of W [K, N] is coalesced and no transposed copy (utils.py:24) exists.
//
// Parity: np.power (fp32 powf) and NumPy's pairwise fp32 means are not bit-reproducible on a GPU, so zero points
// agree to a few ulp and an integer may move where w / s + z lands within that distance of a tie; the row means
// use NumPy's summation tree (8 strided partial sums per <= 128 elements, halves above) to stay as close as
// possible.  tests/test_hqq_gpu.py states the tolerances.
#include "oq_common.hpp"

namespace oq {

struct HqqCtrl {
    double best_err;
    int32_t improved;  // the round that was just judged lowered the error: its zero points become `best`
    int32_t stopped;   // early stop taken (hqq.py:136-137): later rounds are no-ops
    int32_t rounds;    // rounds actually evaluated (diagnostic)
    int32_t pad;
};

struct HqqArgs {
    const float* W;
    int64_t K, N, ldw, g, kgroups;
    const float* scale;   // [N * kgroups], entry n * kgroups + kg (rtn.py:98-109 layout)
    float* zp_cur;        // zero points of the round being evaluated
    float* zp_next;       // hqq.py:140 result of that round
    float* zp_best;
    double* partial;      // [gridDim.x * gridDim.y] sums of |w - w_r|
    HqqCtrl* ctrl;
    float qmin, qmax;
    float inv_beta;       // float32(1 / beta) of this round (beta *= kappa in float64 on the host, hqq.py:128)
    float expo;           // float32(lp_norm - 1)
    int32_t round;
};

// NumPy's pairwise sum of a row (numpy/core/src/umath/loops_utils.h.src::pairwise_sum): `value(i)` is evaluated
// once per element, in index order; the result is what np.add.reduce returns for the row.
template <typename F>
__device__ __forceinline__ float pairwise_leaf(int64_t n, int64_t base, F&& value) {
    // n <= 128 (loops_utils: n < 8 -> plain loop; else 8 strided partial sums + tail)
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; ++i) res += value(base + i);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = value(base + j);
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += value(base + i + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += value(base + i);
    return res;
}

// Whole row of length n: recursion of loops_utils (n > 128: halves, the left one a multiple of 8) unrolled with an
// explicit stack; -1 on the work stack means "add the two newest sub-tree sums".
template <typename F>
__device__ __forceinline__ float pairwise_row(int64_t n, F&& value) {
    if (n <= 128) return pairwise_leaf(n, 0, value);
    int64_t work[40];
    float vals[24];
    int wt = 0, vt = 0;
    int64_t pos = 0;
    work[wt++] = n;
    while (wt > 0) {
        const int64_t len = work[--wt];
        if (len < 0) {
            const float b = vals[--vt], a = vals[--vt];
            vals[vt++] = a + b;
        } else if (len <= 128) {
            vals[vt++] = pairwise_leaf(len, pos, value);
            pos += len;
        } else {
            int64_t n2 = len / 2;
            n2 -= n2 % 8;
            work[wt++] = -1;
            work[wt++] = len - n2;
            work[wt++] = n2;
        }
    }
    return vals[0];
}

__device__ __forceinline__ float hqq_shrink(float d, float inv_beta, float expo) {
    // hqq.py:102-103: sign(x) * relu(|x| - (1/beta) * (|x| + 1e-8)^(p - 1)), every step rounded to fp32
    const float a = fabsf(d);
    // (|x| + 1e-8)^(p - 1) = 2^((p - 1) log2(.)) on the hardware log / exp units (~2e-6 relative; np.power is not
    // reproduced bit for bit either way, see the header)
    const float t = a - inv_beta * __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf(a + 1e-8f));
    const float m = nmax(0.0f, t);
    // np.sign(d) * m with m >= 0 (or NaN): m under d's sign bit -- the same bits, +-0 and NaN included (d = +-0 gives m = 0)
    return __builtin_copysignf(m, d);
}

__global__ __launch_bounds__(256) void hqq_round_kernel(const HqqArgs a) {
    __shared__ double s_part[4];
    const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t kg = blockIdx.y;
    const bool live = col < a.N;
    const int64_t r = (live ? col : a.N - 1) * a.kgroups + kg;
    double abs_sum = 0.0;
    const HqqCtrl c = *a.ctrl;
    bool active = live;
    if (a.round > 0) {   // apply the decision taken on the previous round (hqq.py:132-140)
        if (live && c.improved) a.zp_best[r] = a.zp_cur[r];
        if (c.stopped) active = false;
        else if (live) a.zp_cur[r] = a.zp_next[r];
    }
    if (active) {
        const float z = a.zp_cur[r];
        const float inv = 1.0f / a.scale[r];                               // hqq.py:120
        const float ninv = -inv, rr = refined_rcp(inv);                    // the row's divisor of :125, see div_refined
        const float* w = a.W + kg * a.g * a.ldw + col;
        const float zmean = pairwise_row(a.g, [&](int64_t t) {
            const float x = w[t * a.ldw];
            const float wq = nmin(nmax(rintf(x * inv + z), a.qmin), a.qmax);   // :124
            const float wr = div_refined(wq - z, ninv, rr);                    // :125 (wq - z) / inv
            const float d = x - wr;
            abs_sum += static_cast<double>(fabsf(d));                          // :131
            const float we = hqq_shrink(d, a.inv_beta, a.expo);                // :126
            return wq - (x - we) * inv;                                        // :140
        }) / static_cast<float>(a.g);
        a.zp_next[r] = zmean;
    }
    // block partial of sum |w - w_r|, folded in a fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) abs_sum += __shfl_xor(abs_sum, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = abs_sum;
    __syncthreads();
    if (threadIdx.x == 0)
        a.partial[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// ---------------------------------------------------------------------------------------------
// All rounds of a row in ONE pass over W.  Inside a round the rows are independent and a row's next
// zero point depends only on its own current one (hqq.py:140); the global error steers only WHICH round's zero points
// are kept and when the walk stops (hqq.py:131-137) -- never the trajectory itself.  So a thread keeps its group's G values
// in registers, walks all `iters` rounds, writes its zero point of every round (`traj[round][row]`, coalesced) and its share
// of every round's sum |w - w_r|; `hqq_decide_all_kernel` then replays the reference's decisions over the per-round sums and
// names the round whose zero points win.  W is read once instead of `iters` times, 3 launches instead of 2 * iters + 1.
// Per element, per round and per sum the arithmetic and its order are those of hqq_round_kernel / hqq_decide_kernel: the two
// routes give the same bits (tests/test_hqq.py compares them).
// ---------------------------------------------------------------------------------------------
constexpr int kHqqFusedMaxIters = 32;

struct HqqBetas { float inv_beta[kHqqFusedMaxIters]; };

template <int G>
__global__ __launch_bounds__(256, 2) void hqq_rounds_reg_kernel(const HqqArgs a, const float* zero_point_in, float* traj /* [iters + 1][rows] */,
                                                               double* partial /* [iters][parts] */, const HqqBetas betas, int32_t iters) {
    __shared__ double s_part[4];
    const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t kg = blockIdx.y;
    const bool live = col < a.N;
    const int64_t cc = live ? col : a.N - 1;
    const int64_t r = cc * a.kgroups + kg;
    const int64_t rows = a.N * a.kgroups;
    const int64_t parts = static_cast<int64_t>(gridDim.x) * gridDim.y;
    const int64_t part = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    // wave-uniform row base + one 32-bit lane offset for all G loads (64-bit per-lane addresses for 128 rows would need 256
    // registers before the first value has landed)
    const char* base = reinterpret_cast<const char*>(a.W + kg * G * a.ldw);
    const uint32_t lane_off = static_cast<uint32_t>(cc) * 4u;
    const int64_t row_bytes = a.ldw * 4;
    float x[G];
#pragma unroll
    for (int t = 0; t < G; ++t) x[t] = *reinterpret_cast<const float*>(base + t * row_bytes + lane_off);
    float z = zero_point_in[r];
    const float inv = 1.0f / a.scale[r];                                       // hqq.py:120
    const float ninv = -inv, rr = refined_rcp(inv);                            // the row's divisor of :125, see div_refined
    if (live) traj[r] = z;
#pragma unroll 1
    for (int32_t it = 0; it < iters; ++it) {
        const float inv_beta = betas.inv_beta[it];
        double abs_sum = 0.0;
        // x[t] * inv does not change from round to round and the optimiser would keep all G products in registers next to the
        // G values themselves (169 registers at G = 32, spills at G = 128): hide the loop-invariance of `inv`
        float inv_r = inv;
        asm volatile("" : "+v"(inv_r));
        // NumPy's pairwise sum of a row of G <= 128 elements (8 strided partial sums, see pairwise_leaf), `value` evaluated
        // once per element in index order
        // One 8-element group at a time: every group reads the zero point and 1 / scale through its own opaque copy, so its
        // work cannot be hoisted in front of the previous group's (left alone, all G / 8 independent groups are interleaved and
        // G = 128 spills 600+ bytes per lane).
        float acc[8];
#pragma unroll
        for (int t0 = 0; t0 < G; t0 += 8) {
            float zg = z, ig = inv_r, ng = ninv, rg = rr;
            if (t0 == 0) asm volatile("" : "+v"(zg), "+v"(ig), "+v"(ng), "+v"(rg));
            else asm volatile("" : "+v"(zg), "+v"(ig), "+v"(ng), "+v"(rg) : "v"(acc[0]), "v"(acc[3]), "v"(acc[7]));   // ... and AFTER the previous group is folded
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = x[t0 + j];
                const float wq = nmin(nmax(rintf(xv * ig + zg), a.qmin), a.qmax);   // :124
                const float wr = div_refined(wq - zg, ng, rg);                        // :125 (wq - zg) / inv
                const float d = xv - wr;
                abs_sum += static_cast<double>(fabsf(d));                             // :131
                const float we = hqq_shrink(d, inv_beta, a.expo);                     // :126
                const float v = wq - (xv - we) * ig;                                  // :140
                acc[j] = t0 == 0 ? v : acc[j] + v;     // NumPy's pairwise sum of a row of <= 128 elements: 8 strided partial sums
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float zmean = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) / static_cast<float>(G);
        // the row mean is complete HERE: left alone the optimiser sinks the shrink / mean half of every element below the
        // error reduction and its barrier and keeps d and w_q of all G elements alive across it (2 G registers: spills at 128)
        asm volatile("" : "+v"(zmean) : : "memory");
        if (!live) abs_sum = 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) abs_sum += __shfl_xor(abs_sum, off, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = abs_sum;
        __syncthreads();
        if (threadIdx.x == 0) partial[static_cast<int64_t>(it) * parts + part] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        __syncthreads();
        z = zmean;
        if (live) traj[static_cast<int64_t>(it + 1) * rows + r] = z;
    }
}

// hqq.py:131-137 replayed over the per-round sums: ctrl->rounds = rounds evaluated, ctrl->pad = the round whose zero points win
// (0 = the given ones: the first round evaluates them and an infinite best_err always yields to it unless the error is NaN).
__global__ __launch_bounds__(1024) void hqq_decide_all_kernel(const double* partial, int64_t nparts, double count, int32_t early_stop, int32_t iters,
                                                              HqqCtrl* ctrl) {
    __shared__ double s[16];
    __shared__ int s_stop;
    double best_err = INFINITY;
    int32_t best_round = 0, rounds = 0;
    for (int32_t it = 0; it < iters; ++it) {
        double acc = 0.0;
        for (int64_t i = threadIdx.x; i < nparts; i += blockDim.x) acc += partial[static_cast<int64_t>(it) * nparts + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t += s[w];
            const double err = static_cast<double>(static_cast<float>(t / count));   // np.mean of an fp32 array is fp32
            rounds += 1;
            s_stop = 0;
            if (err < best_err) {
                best_err = err;
                best_round = it;
            } else if (early_stop) {
                s_stop = 1;
            }
        }
        __syncthreads();
        if (s_stop) break;
    }
    if (threadIdx.x == 0) {
        ctrl->best_err = best_err;
        ctrl->improved = 0;
        ctrl->stopped = 0;
        ctrl->rounds = rounds;
        ctrl->pad = best_round;
    }
}

// hqq.py:131-137 on the device: err = float32 mean; better -> remember, else early stop.
__global__ __launch_bounds__(1024) void hqq_decide_kernel(const double* partial, int64_t nparts, double count, int32_t early_stop,
                                                          HqqCtrl* ctrl) {
    __shared__ double s[16];
    if (ctrl->stopped) {
        if (threadIdx.x == 0) ctrl->improved = 0;
        return;
    }
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < nparts; i += blockDim.x) acc += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t += s[w];
        const double err = static_cast<double>(static_cast<float>(t / count));   // np.mean of an fp32 array is fp32
        ctrl->rounds += 1;
        if (err < ctrl->best_err) {
            ctrl->best_err = err;
            ctrl->improved = 1;
        } else {
            ctrl->improved = 0;
            if (early_stop) ctrl->stopped = 1;
        }
    }
}

__global__ void hqq_init_kernel(HqqCtrl* ctrl) {
    ctrl->best_err = INFINITY;
    ctrl->improved = 0;
    ctrl->stopped = 0;
    ctrl->rounds = 0;
    ctrl->pad = 0;
}

// After the last round: settle `best` and quantize with it (hqq.py:163-171: round(w / s + z), float zero point
// inside the rounding, no int32 cast).  q is the [K, N] one-value-per-byte array `_post_process_array` returns.
__global__ __launch_bounds__(256) void hqq_finish_kernel(const HqqArgs a, uint8_t* q, float* zp_out, int32_t layout) {
    const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t kg = blockIdx.y;
    if (col >= a.N) return;
    const int64_t r = col * a.kgroups + kg;
    // fused route (a.zp_next == nullptr): a.zp_cur is the trajectory [iters + 1][rows], ctrl->pad the winning round
    const float z = a.zp_next == nullptr ? a.zp_cur[static_cast<int64_t>(a.ctrl->pad) * (a.N * a.kgroups) + r]
                                         : ((a.round > 0 && a.ctrl->improved) ? a.zp_cur[r] : a.zp_best[r]);
    zp_out[r] = z;
    if (q == nullptr) return;
    const float s = a.scale[r];
    const float* w = a.W + kg * a.g * a.ldw + col;
    if (layout == OQ_LAYOUT_NBITS) {
        // qrules/_common.py:72-87: the (column, k-group) chunk is g / 2 consecutive bytes, even k in the low nibble
        uint8_t* o = q + r * (a.g / 2);
        for (int64_t t = 0; t < a.g; t += 8) {   // g is a power of two >= 16 for HQQ (hqq.py:66-70): whole 4-byte words
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = nmin(nmax(rintf(w[(t + j) * a.ldw] / s + z), a.qmin), a.qmax);
                word |= (static_cast<uint32_t>(v) & 0xfu) << (4 * j);
            }
            *reinterpret_cast<uint32_t*>(o + t / 2) = word;
        }
        return;
    }
    for (int64_t t = 0; t < a.g; ++t) {
        const float v = nmin(nmax(rintf(w[t * a.ldw] / s + z), a.qmin), a.qmax);
        q[(kg * a.g + t) * a.N + col] = static_cast<uint8_t>(v);
    }
}

}  // namespace oq

extern "C" {

using namespace oq;

size_t oq_hqq_workspace_bytes(int64_t K, int64_t N, int64_t group_size) {
    if (!oq::matrix_ok(K, N, N)) return 0;
    int64_t g = group_size > K ? K : group_size;
    if (g == -1) g = K;
    if (g <= 0 || K % g != 0) return 0;
    const int64_t rows = N * (K / g);
    const int64_t parts = ceil_div(N, 256) * (K / g);
    // per-round route: three zero-point arrays + one partial per workgroup; one-pass route (g = 16 .. 128, <= 32 rounds): the
    // zero points of every round + the partials of every round
    const size_t per_round = static_cast<size_t>(rows) * 4 * 3 + static_cast<size_t>(parts) * 8;
    const size_t one_pass = static_cast<size_t>(rows) * 4 * (kHqqFusedMaxIters + 1) + static_cast<size_t>(parts) * 8 * kHqqFusedMaxIters;
    return (per_round > one_pass ? per_round : one_pass) + 1024;
}

int32_t oq_hqq_optimize_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t group_size, int32_t reduce_range,
                            const float* scale, const float* zero_point_in, double lp_norm, double beta, double kappa, int32_t iters,
                            int32_t early_stop, int32_t per_round_launches, void* q_out, int32_t layout, float* zero_point_out, int32_t* rounds_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(W && scale && zero_point_in && zero_point_out && matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT,
               "oq_hqq_optimize_f32: bad argument");
    OQ_REQUIRE(iters >= 0 && beta > 0.0, OQ_ERR_INVALID_ARGUMENT, "oq_hqq_optimize_f32: iters >= 0 and beta > 0 needed");
    OQ_REQUIRE(layout == OQ_LAYOUT_KN || layout == OQ_LAYOUT_NBITS, OQ_ERR_INVALID_ARGUMENT, "oq_hqq_optimize_f32: bad layout %d", layout);
    int64_t g = group_size > K ? K : group_size;   // utils.py:19-22
    if (g == -1) g = K;
    OQ_REQUIRE(g > 0, OQ_ERR_INVALID_ARGUMENT, "oq_hqq_optimize_f32: bad group_size %lld", (long long)group_size);
    OQ_REQUIRE(K % g == 0, OQ_ERR_UNSUPPORTED, "oq_hqq_optimize_f32: groups that straddle columns (K %% group_size != 0) are not supported");
    const int64_t kgroups = K / g;
    OQ_REQUIRE(kgroups <= 65535, OQ_ERR_UNSUPPORTED, "oq_hqq_optimize_f32: more than 65535 groups per column");
    const size_t need = oq_hqq_workspace_bytes(K, N, group_size);
    OQ_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, OQ_ERR_WORKSPACE,
               "oq_hqq_optimize_f32: 8-byte aligned workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    int64_t qmin, qmax;
    OQ_REQUIRE(qrange_host(OQ_UINT4, 0, reduce_range, &qmin, &qmax), OQ_ERR_INVALID_ARGUMENT, "oq_hqq_optimize_f32: qrange");
    hipStream_t s = as_stream(stream);
    const int64_t rows = N * kgroups;
    const dim3 grid(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(kgroups));
    const int64_t parts = static_cast<int64_t>(grid.x) * grid.y;

    char* base = static_cast<char*>(workspace);
    HqqArgs a;
    a.ctrl = reinterpret_cast<HqqCtrl*>(base);
    a.partial = reinterpret_cast<double*>(base + 64);
    a.zp_cur = reinterpret_cast<float*>(base + 64 + parts * 8);
    a.zp_next = a.zp_cur + rows;
    a.zp_best = a.zp_next + rows;
    a.W = W; a.K = K; a.N = N; a.ldw = ldw; a.g = g; a.kgroups = kgroups; a.scale = scale;
    a.qmin = static_cast<float>(qmin); a.qmax = static_cast<float>(qmax);
    a.expo = static_cast<float>(lp_norm - 1.0);
    a.round = 0; a.inv_beta = 0.f;

    // One pass over W for all rounds where the group fits the register tile (the reference's only HQQ configurations: group
    // sizes that are powers of two >= 16, hqq.py:66-70) and the trajectory fits the workspace; same bits as the loop below.
    // iters < 0 never happens (checked above); iters == 0 takes the loop (no round: the given zero points).
    if ((g == 16 || g == 32 || g == 64 || g == 128) && iters >= 1 && iters <= kHqqFusedMaxIters && !per_round_launches) {
        float* traj = reinterpret_cast<float*>(base + 64 + parts * 8 * kHqqFusedMaxIters);
        HqqBetas betas;
        double b2 = beta;
        for (int32_t it = 0; it < kHqqFusedMaxIters; ++it) {
            betas.inv_beta[it] = static_cast<float>(1.0 / b2);   // (1.0 / beta) meets an fp32 array: weak scalar -> fp32
            b2 *= kappa;                                         // :128
        }
        a.zp_cur = traj; a.zp_next = nullptr; a.zp_best = nullptr;
        switch (g) {
            case 16: hipLaunchKernelGGL((hqq_rounds_reg_kernel<16>), grid, dim3(256), 0, s, a, zero_point_in, traj, a.partial, betas, iters); break;
            case 32: hipLaunchKernelGGL((hqq_rounds_reg_kernel<32>), grid, dim3(256), 0, s, a, zero_point_in, traj, a.partial, betas, iters); break;
            case 64: hipLaunchKernelGGL((hqq_rounds_reg_kernel<64>), grid, dim3(256), 0, s, a, zero_point_in, traj, a.partial, betas, iters); break;
            default: hipLaunchKernelGGL((hqq_rounds_reg_kernel<128>), grid, dim3(256), 0, s, a, zero_point_in, traj, a.partial, betas, iters); break;
        }
        hipLaunchKernelGGL(hqq_decide_all_kernel, dim3(1), dim3(1024), 0, s, a.partial, parts, static_cast<double>(K) * static_cast<double>(N),
                           early_stop, iters, a.ctrl);
        int32_t st1 = check_launch("hqq_rounds_reg_kernel");
        if (st1 != OQ_OK) return st1;
        OQ_REQUIRE(layout == OQ_LAYOUT_KN || q_out == nullptr || (reinterpret_cast<uintptr_t>(q_out) & 3u) == 0, OQ_ERR_UNSUPPORTED,
                   "oq_hqq_optimize_f32: NBITS layout needs a 4-byte aligned output");
        a.round = iters;
        hipLaunchKernelGGL(hqq_finish_kernel, grid, dim3(256), 0, s, a, static_cast<uint8_t*>(q_out), zero_point_out, layout);
        st1 = check_launch("hqq_finish_kernel");
        if (st1 != OQ_OK) return st1;
        if (rounds_out != nullptr &&
            hipMemcpyAsync(rounds_out, &a.ctrl->rounds, sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(OQ_ERR_LAUNCH, "oq_hqq_optimize_f32: device copy failed");
        return OQ_OK;
    }

    // hqq.py:115-116: best = zero_point.copy(); the first round evaluates the given zero points
    if (hipMemcpyAsync(a.zp_cur, zero_point_in, static_cast<size_t>(rows) * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(a.zp_best, zero_point_in, static_cast<size_t>(rows) * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(OQ_ERR_LAUNCH, "oq_hqq_optimize_f32: device copy failed");
    hipLaunchKernelGGL(hqq_init_kernel, dim3(1), dim3(1), 0, s, a.ctrl);
    double b = beta;
    for (int32_t it = 0; it < iters; ++it) {
        a.round = it;
        a.inv_beta = static_cast<float>(1.0 / b);   // (1.0 / beta) meets an fp32 array: weak scalar -> fp32
        b *= kappa;                                 // :128
        hipLaunchKernelGGL(hqq_round_kernel, grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(hqq_decide_kernel, dim3(1), dim3(1024), 0, s, a.partial, parts, static_cast<double>(K) * static_cast<double>(N),
                           early_stop, a.ctrl);
    }
    int32_t st = check_launch("hqq_round_kernel");
    if (st != OQ_OK) return st;
    a.round = iters;
    OQ_REQUIRE(layout == OQ_LAYOUT_KN || q_out == nullptr || (g % 8 == 0 && (reinterpret_cast<uintptr_t>(q_out) & 3u) == 0), OQ_ERR_UNSUPPORTED,
               "oq_hqq_optimize_f32: NBITS layout needs group_size %% 8 == 0 and a 4-byte aligned output");
    hipLaunchKernelGGL(hqq_finish_kernel, grid, dim3(256), 0, s, a, static_cast<uint8_t*>(q_out), zero_point_out, layout);
    st = check_launch("hqq_finish_kernel");
    if (st != OQ_OK) return st;
    if (rounds_out != nullptr &&
        hipMemcpyAsync(rounds_out, &a.ctrl->rounds, sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(OQ_ERR_LAUNCH, "oq_hqq_optimize_f32: device copy failed");
    return OQ_OK;
}

}  // extern "C"
