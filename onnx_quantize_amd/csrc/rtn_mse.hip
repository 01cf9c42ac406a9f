// M1: the MSE range search of utils.py:140-239 for gfx950.
//
// For every row of the reference's preprocessed layout (a k-group of one output channel, a whole output
// channel, or the whole tensor) the reference shrinks the (min, max) range by p = 1 - i/100, i = 0..19,
// fake-quantizes the row with the parameters of the shrunk range and keeps the range with the smallest
// sum |q - x|^2.4.  Its stop rule is GLOBAL: it counts the iterations in which no row at all improved and
// stops at the fifth (the counter is never reset, utils.py:232-237).
//
// Here every row records a 20-bit mask of the iterations in which it would improve (rows are independent
// until the stop), the masks are OR-ed into one device word, and a resolve kernel derives the stop
// iteration from that word and picks, per row, the last improving iteration not after it.  The search is
// ALU-bound (an IEEE divide and |d|^2.4 per element and candidate, 20 candidates); W is re-read from
// L2 / Infinity Cache per candidate instead of being tiled (g > 128), or held in registers (g <= 128, mse_rows_reg_kernel: the division
// there is the proven-equal reciprocal product of the RTN kernels with an exact redo inside the tie band).  The power runs on the hardware log / exp units (the library
// powf was 3/4 of the kernel: 4.6 -> 1.2 ms on the 4096 x 11008 matrix).
//
// Numerics: per-element arithmetic follows the reference (divide, rint, clamp, (q - zp) * s, subtract, abs,
// |.|^2.4), but NumPy's pow kernel and its pairwise summation order cannot be reproduced bit for bit,
// so two candidates whose errors differ in the last bits can swap; tests/test_mse_gpu.py is tolerance-aware.
#include "oq_common.hpp"

namespace oq {

constexpr int kMseSteps = 20;      // int(maxshrink * grid) = int(0.20 * 100.0), utils.py:197
constexpr int kMsePatience = 5;    // utils.py:150
constexpr float kMseNorm = 2.4f;   // utils.py:152
constexpr int kMseChunk = 16;      // elements per tie-band check of the register kernel

// candidate i: p = 1 - i / 100.0 (Python float), applied to fp32 ranges as a weak scalar -> fp32 product
__device__ __forceinline__ float shrink_factor(int i) { return static_cast<float>(1.0 - static_cast<double>(i) / 100.0); }

__device__ __forceinline__ float fake_quant_error(float x, const QParam& p, const QGrid& g) {
    const int32_t q = quantize_one(x, p.scale, p.zp, g.qmin, g.qmax);
    const float d = dequantize_one(q, p.scale, p.zp) - x;
    // |d|^2.4 = 2^(2.4 log2 |d|) on the hardware log / exp units (~2e-6 relative; the library powf costs several times
    // the rest of the candidate, and NumPy's own float32 pow is not reproduced bit for bit either way, see above)
    return __builtin_amdgcn_exp2f(kMseNorm * __builtin_amdgcn_logf(fabsf(d)));
}

struct MseRow {
    float lo0, hi0;   // utils.py:188 range with clip_ratio = 1.0, zero included
    uint32_t mask;    // bit i: iteration i improves this row (no early stop assumed)
};

// channel / group: one thread per (column, k-group); lanes walk neighbouring columns -> coalesced rows
__global__ __launch_bounds__(256) void mse_rows_kernel(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t g,
                                                       int64_t kgroups, QGrid grid, MseRow* rows, uint32_t* any_mask) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t kg = blockIdx.y;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    const float* col = W + kg * g * ldw + nc;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t r = 0; r < g; ++r) {
        const float x = col[r * ldw];
        mn = nmin(mn, x);
        mx = nmax(mx, x);
    }
    const float lo0 = nmin(mn, 0.0f), hi0 = nmax(mx, 0.0f);
    float best = FLT_MAX;   // np.finfo(float32).max, utils.py:190
    uint32_t mask = 0;
    for (int i = 0; i < kMseSteps; ++i) {
        const float p = shrink_factor(i);
        const QParam qp = qparam_from_range(p * lo0, p * hi0, grid);
        float err = 0.f;
        for (int64_t r = 0; r < g; ++r) err += fake_quant_error(col[r * ldw], qp, grid);
        if (err < best) {  // utils.py:225
            best = err;
            mask |= 1u << i;
        }
    }
    if (live) {
        MseRow o;
        o.lo0 = lo0; o.hi0 = hi0; o.mask = mask;
        rows[n * kgroups + kg] = o;
    }
    // block-wide OR, one atomic per wave
    uint32_t m = live ? mask : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(any_mask, m);
}

// The same search with the group's G values held in REGISTERS across the 20 candidates: W is read once
// (G strided 4-byte loads per thread, coalesced across the lanes of a row) instead of 21 times from L2 / Infinity Cache,
// and the candidate loop is pure ALU.  Same per-element arithmetic and the same summation order (r = 0 .. G-1 into one fp32
// accumulator) as mse_rows_kernel: bit-identical masks.  G <= 128 keeps the tile within 256 VGPRs (two waves per SIMD: an
// ALU-bound loop with G independent chains per candidate needs no more).
template <int G>
__global__ __launch_bounds__(256, 2) void mse_rows_reg_kernel(const float* W, int64_t N, int64_t ldw, int64_t kgroups, QGrid grid,
                                                             MseRow* rows, uint32_t* any_mask) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t kg = blockIdx.y;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    // wave-uniform row base + one 32-bit lane offset for all G loads (no 64-bit per-lane address per row)
    const char* base = reinterpret_cast<const char*>(W + kg * G * ldw);
    const uint32_t lane_off = static_cast<uint32_t>(nc) * 4u;
    const int64_t row_bytes = ldw * 4;
    float x[G];
#pragma unroll
    for (int r = 0; r < G; ++r) x[r] = *reinterpret_cast<const float*>(base + r * row_bytes + lane_off);
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int r = 1; r < G; ++r) {
        mn = nmin(mn, x[r]);
        mx = nmax(mx, x[r]);
    }
    const float lo0 = nmin(mn, 0.0f), hi0 = nmax(mx, 0.0f);
    float best = FLT_MAX;   // np.finfo(float32).max, utils.py:190
    uint32_t mask = 0;
#pragma unroll 1
    for (int i = 0; i < kMseSteps; ++i) {
        const float p = shrink_factor(i);
        const QParam qp = qparam_from_range(p * lo0, p * hi0, grid);
        // the division-free level of oq_common.hpp (proved equal to the reference's integer away from rounding ties), in
        // chunks of 16 elements: a chunk in which any lane of the wave sits inside the tie band (~1 % of the chunks) is
        // redone with the true division by the whole wave.  Same values, same summation order as the plain loop.
        const ColQ c = make_colq(qp, mn, mx, 0);
        const float lo_f = static_cast<float>(grid.qmin), hi_f = static_cast<float>(grid.qmax);
        float err = 0.f;
#pragma unroll
        for (int r0 = 0; r0 < G; r0 += kMseChunk) {
            float e[kMseChunk];
            float off = 0.f;   // max |t - rint(t)| of the chunk (t is finite whenever c.thr > 0, make_colq)
            // one chunk at a time: its reciprocal is opaque until the previous chunk's sum exists (left alone, the compiler
            // runs several chunks' temporaries at once and spills the tile at G = 128)
            float rinv = c.rinv;
            asm volatile("" : "+v"(rinv), "+v"(err));
#pragma unroll
            for (int r = 0; r < kMseChunk; ++r) {
                const float t = x[r0 + r] * rinv;
                const float k = rintf(t);
                off = fmaxf(off, fabsf(t - k));
                const float lvl = __builtin_amdgcn_fmed3f(k + c.zpb, lo_f, hi_f);
                const float d = (lvl - c.zpb) * c.scale - x[r0 + r];
                e[r] = __builtin_amdgcn_exp2f(kMseNorm * __builtin_amdgcn_logf(fabsf(d)));
            }
            if (__builtin_amdgcn_ballot_w64(!(off < c.thr)) != 0) {
#pragma unroll
                for (int r = 0; r < kMseChunk; ++r) e[r] = fake_quant_error(x[r0 + r], qp, grid);
            }
#pragma unroll
            for (int r = 0; r < kMseChunk; ++r) err += e[r];
        }
        if (err < best) {  // utils.py:225
            best = err;
            mask |= 1u << i;
        }
    }
    if (live) {
        MseRow o;
        o.lo0 = lo0; o.hi0 = hi0; o.mask = mask;
        rows[n * kgroups + kg] = o;
    }
    uint32_t m = live ? mask : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(any_mask, m);
}

// tensor: per-block partial sums of all 20 candidate errors
__global__ __launch_bounds__(256) void mse_tensor_partial(const float* W, int64_t K, int64_t N, int64_t ldw, const float* range,
                                                          QGrid grid, float* partial /* [blocks][20] */) {
    __shared__ float s_sum[4][kMseSteps];
    const float lo0 = nmin(range[0], 0.0f), hi0 = nmax(range[1], 0.0f);
    QParam qp[kMseSteps];
#pragma unroll
    for (int i = 0; i < kMseSteps; ++i) qp[i] = qparam_from_range(shrink_factor(i) * lo0, shrink_factor(i) * hi0, grid);
    float acc[kMseSteps];
#pragma unroll
    for (int i = 0; i < kMseSteps; ++i) acc[i] = 0.f;
    const int64_t total = K * N;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t k = t / N, n = t - k * N;
        const float x = W[k * ldw + n];
#pragma unroll
        for (int i = 0; i < kMseSteps; ++i) acc[i] += fake_quant_error(x, qp[i], grid);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kMseSteps; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) s_sum[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kMseSteps)
        partial[static_cast<int64_t>(blockIdx.x) * kMseSteps + threadIdx.x] =
            (s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + (s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x]);
}

__global__ void mse_tensor_mask(const float* partial, int nblocks, const float* range, MseRow* row, uint32_t* any_mask) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float best = FLT_MAX;
    uint32_t mask = 0;
    for (int i = 0; i < kMseSteps; ++i) {
        float err = 0.f;
        for (int b = 0; b < nblocks; ++b) err += partial[b * kMseSteps + i];
        if (err < best) { best = err; mask |= 1u << i; }
    }
    row->lo0 = nmin(range[0], 0.0f);
    row->hi0 = nmax(range[1], 0.0f);
    row->mask = mask;
    *any_mask = mask;
}

// raw (min, max) of the whole tensor -> range[2] (single block; the tensor strategy is the plumbing case)
__global__ __launch_bounds__(1024) void tensor_minmax_kernel(const float* W, int64_t K, int64_t N, int64_t ldw, float* range) {
    __shared__ float s_mn[16], s_mx[16];
    float mn = INFINITY, mx = -INFINITY;
    const int64_t total = K * N;
    for (int64_t t = threadIdx.x; t < total; t += blockDim.x) {
        const int64_t k = t / N, n = t - k * N;
        const float x = W[k * ldw + n];
        mn = nmin(mn, x);
        mx = nmax(mx, x);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { mn = nmin(mn, s_mn[w]); mx = nmax(mx, s_mx[w]); }
        range[0] = mn;
        range[1] = mx;
    }
}

// Stop iteration from the global mask (utils.py:232-237), then per row: last improving iteration not after it.
__device__ __forceinline__ int stop_iteration(uint32_t any_mask) {
    int stale = 0;
    for (int i = 0; i < kMseSteps; ++i) {
        if (!((any_mask >> i) & 1u)) ++stale;
        if (stale >= kMsePatience) return i;   // iteration i is the last one executed
    }
    return kMseSteps - 1;
}

__global__ void mse_resolve_kernel(const MseRow* rows, int64_t count, const uint32_t* any_mask, QGrid grid, float* scale,
                                   uint8_t* zp) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const int last = stop_iteration(*any_mask);
    const uint32_t m = rows[r].mask & (last >= 31 ? 0xffffffffu : ((2u << last) - 1u));
    // best_min/max start as the unshrunk range (utils.py:191-192): that is also candidate 0
    const int idx = m ? 31 - __builtin_clz(m) : 0;
    const float p = shrink_factor(idx);
    const QParam qp = qparam_from_range(p * rows[r].lo0, p * rows[r].hi0, grid);  // utils.py:345-347 (clip ratio not applied)
    scale[r] = qp.scale;
    zp[r] = static_cast<uint8_t>(qp.zp);
}

size_t rtn_mse_workspace(int64_t K, int64_t N, int32_t strategy, int64_t g) {
    const int64_t rows = strategy == OQ_TENSOR ? 1 : N * (K / g);
    return static_cast<size_t>(rows) * sizeof(MseRow) + 1024 * kMseSteps * sizeof(float) + 512;
}

// quantize pass shared with the two-pass RTN path (rtn.hip)
int32_t launch_quantize_kn(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t g, int64_t kgroups, const float* scale,
                           const uint8_t* zp, uint8_t* q, const QGrid& grid, int32_t zp_signed, bool tensor, hipStream_t s,
                           int32_t layout);

int32_t rtn_mse_impl(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid_in, int32_t strategy, int64_t g,
                     void* q_out, float* scale_out, void* zp_out, int32_t zp_signed, void* workspace, size_t workspace_bytes,
                     hipStream_t s, bool emit_q) {
    const size_t need = rtn_mse_workspace(K, N, strategy, g);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "rtn(mse): workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    QGrid grid = grid_in;
    grid.clip_ratio = 1.0f;  // utils.py:188 and :334-344: the MSE range replaces the clipped one
    const int64_t kgroups = strategy == OQ_TENSOR ? 1 : K / g;
    const int64_t rows = strategy == OQ_TENSOR ? 1 : N * kgroups;
    char* base = static_cast<char*>(workspace);
    uint32_t* any_mask = reinterpret_cast<uint32_t*>(base);
    float* range = reinterpret_cast<float*>(base + 16);
    MseRow* rowbuf = reinterpret_cast<MseRow*>(base + 256);
    float* partial = reinterpret_cast<float*>(base + 256 + static_cast<size_t>(rows) * sizeof(MseRow));
    if (hipMemsetAsync(any_mask, 0, 4, s) != hipSuccess) return fail(OQ_ERR_LAUNCH, "rtn(mse): memset failed");
    int32_t st;
    if (strategy == OQ_TENSOR) {
        const int nblocks = 1024;
        hipLaunchKernelGGL(tensor_minmax_kernel, dim3(1), dim3(1024), 0, s, W, K, N, ldw, range);
        hipLaunchKernelGGL(mse_tensor_partial, dim3(nblocks), dim3(256), 0, s, W, K, N, ldw, range, grid, partial);
        hipLaunchKernelGGL(mse_tensor_mask, dim3(1), dim3(64), 0, s, partial, nblocks, range, rowbuf, any_mask);
        st = check_launch("mse_tensor");
    } else {
        const dim3 gd(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(kgroups));
        if (g == 128) hipLaunchKernelGGL((mse_rows_reg_kernel<128>), gd, dim3(256), 0, s, W, N, ldw, kgroups, grid, rowbuf, any_mask);
        else if (g == 64) hipLaunchKernelGGL((mse_rows_reg_kernel<64>), gd, dim3(256), 0, s, W, N, ldw, kgroups, grid, rowbuf, any_mask);
        else if (g == 32) hipLaunchKernelGGL((mse_rows_reg_kernel<32>), gd, dim3(256), 0, s, W, N, ldw, kgroups, grid, rowbuf, any_mask);
        else hipLaunchKernelGGL(mse_rows_kernel, gd, dim3(256), 0, s, W, K, N, ldw, g, kgroups, grid, rowbuf, any_mask);
        st = check_launch("mse_rows_kernel");
    }
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(mse_resolve_kernel, dim3(static_cast<uint32_t>(ceil_div(rows, 256))), dim3(256), 0, s, rowbuf, rows, any_mask, grid,
                       scale_out, static_cast<uint8_t*>(zp_out));
    st = check_launch("mse_resolve_kernel");
    if (st != OQ_OK || !emit_q) return st;
    return launch_quantize_kn(W, K, N, ldw, g, kgroups, scale_out, static_cast<const uint8_t*>(zp_out), static_cast<uint8_t*>(q_out),
                              grid, zp_signed, strategy == OQ_TENSOR, s, OQ_LAYOUT_KN);
}

}  // namespace oq
