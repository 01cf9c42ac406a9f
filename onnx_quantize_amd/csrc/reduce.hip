// C1 (calibration min/max, minmax.py:40-64) and S1 (absmax, smooth_quant.py:62-74) for gfx950.
// HBM-bound reductions: every element is read once with 16-byte loads, reduced in registers, across
// the wave with butterfly shuffles, across the block through LDS; the running (min, max) state of a
// tensor name lives in device memory and is updated by the last stage, so a calibration loop never
// synchronises with the host.
#include "oq_common.hpp"

namespace oq {

constexpr int kRedBlock = 512;      // 8 waves
constexpr int kRedMaxBlocks = 2048;  // <= 256 CUs x 8 blocks (cdna_hip_programming.md Guideline 11)

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4))); };

template <typename T>
__device__ __forceinline__ T t_min(T a, T b) { return nmin(a, b); }

template <typename T>
__device__ __forceinline__ void block_minmax(T& mn, T& mx, T* s_mn, T* s_mx) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < static_cast<int>(blockDim.x >> 6); ++w) {
            mn = nmin(mn, s_mn[w]);
            mx = nmax(mx, s_mx[w]);
        }
    }
}

// Grid-stride sweep over 16-byte (fp32) / 32-byte (fp64) vectors with EIGHT independent non-temporal loads in flight per
// lane: activations are read exactly once, and a linear read needs that depth to reach the HBM rate (scripts/membench:
// 4 loads in flight 5.4-5.5 TB/s, 8 loads 6.2 TB/s on a 180 MB array).
template <typename T>
__device__ __forceinline__ void stream_minmax(const typename Vec4<T>::type* xv, int64_t nvec, int64_t tid, int64_t stride, T& mn, T& mx) {
    using V = typename Vec4<T>::type;
    constexpr int kDepth = 8;
    int64_t i = tid;
    for (; i + (kDepth - 1) * stride < nvec; i += kDepth * stride) {
        V a[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) a[u] = __builtin_nontemporal_load(xv + i + u * stride);
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            mn = nmin(nmin(nmin(mn, a[u].x), nmin(a[u].y, a[u].z)), a[u].w);
            mx = nmax(nmax(nmax(mx, a[u].x), nmax(a[u].y, a[u].z)), a[u].w);
        }
    }
    for (; i < nvec; i += stride) {
        const V a = __builtin_nontemporal_load(xv + i);
        mn = nmin(nmin(mn, a.x), nmin(nmin(a.y, a.z), a.w));
        mx = nmax(nmax(mx, a.x), nmax(nmax(a.y, a.z), a.w));
    }
}

// Stage 1: per-block partial (min, max).  x must be element-aligned only; the 16/32-byte body is
// peeled by the host into [head | vector body | tail] through `vec_off`.
template <typename T>
__global__ __launch_bounds__(kRedBlock) void minmax_partial(const T* x, int64_t count, int64_t vec_off, T* partial) {
    using V = typename Vec4<T>::type;
    __shared__ T s_mn[kRedBlock / 64], s_mx[kRedBlock / 64];
    T mn = INFINITY, mx = -INFINITY;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t nvec = (count - vec_off) / 4;
    const V* xv = reinterpret_cast<const V*>(x + vec_off);
    stream_minmax<T>(xv, nvec, tid, stride, mn, mx);
    // head and tail scalars
    for (int64_t j = tid; j < vec_off; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    for (int64_t j = vec_off + nvec * 4 + tid; j < count; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    block_minmax(mn, mx, s_mn, s_mx);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = mn;
        partial[2 * blockIdx.x + 1] = mx;
    }
}

// Stage 2: fold the partials and apply minmax.py:50-64 to the device-resident state.
template <typename T>
__global__ __launch_bounds__(kRedBlock) void minmax_update(const T* partial, int nblocks, T* state, double momentum) {
    __shared__ T s_mn[kRedBlock / 64], s_mx[kRedBlock / 64];
    T mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        mn = nmin(mn, partial[2 * i]);
        mx = nmax(mx, partial[2 * i + 1]);
    }
    block_minmax(mn, mx, s_mn, s_mx);
    if (threadIdx.x != 0) return;
    if (state[2] == T(0)) {            // minmax.py:50-51 first sight
        state[0] = mn;
        state[1] = mx;
        state[2] = T(1);
    } else if (momentum > 0.0) {       // minmax.py:53-60 EMA; products rounded separately (no FMA)
        const T m = static_cast<T>(momentum), om = static_cast<T>(1.0 - momentum);
        state[0] = m * state[0] + om * mn;
        state[1] = m * state[1] + om * mx;
    } else {                           // minmax.py:63-64
        state[0] = nmin(state[0], mn);
        state[1] = nmax(state[1], mx);
    }
}

template <typename T>
static int32_t minmax_collect(const T* x, int64_t count, T* state, double momentum, void* ws, size_t ws_bytes,
                              void* stream) {
    OQ_REQUIRE(x && state && count_ok(count), OQ_ERR_INVALID_ARGUMENT, "oq_minmax_collect: bad argument");
    OQ_REQUIRE(momentum >= 0.0 && momentum < 1.0, OQ_ERR_INVALID_ARGUMENT, "Momentum must be in the range [0, 1).");
    OQ_REQUIRE((reinterpret_cast<uintptr_t>(x) % sizeof(T)) == 0, OQ_ERR_INVALID_ARGUMENT, "oq_minmax_collect: misaligned input");
    int64_t nblocks = ceil_div(count, static_cast<int64_t>(kRedBlock) * 8);
    if (nblocks > kRedMaxBlocks) nblocks = kRedMaxBlocks;
    if (nblocks < 1) nblocks = 1;
    const size_t need = static_cast<size_t>(kRedMaxBlocks) * 2 * sizeof(double);
    OQ_REQUIRE(ws && ws_bytes >= need, OQ_ERR_WORKSPACE, "oq_minmax_collect: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    // peel to the vector alignment (4 elements)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    const uintptr_t valign = 4 * sizeof(T);
    int64_t vec_off = static_cast<int64_t>(((valign - addr % valign) % valign) / sizeof(T));
    if (vec_off > count) vec_off = count;
    hipStream_t s = as_stream(stream);
    T* partial = static_cast<T*>(ws);
    hipLaunchKernelGGL(minmax_partial<T>, dim3(static_cast<uint32_t>(nblocks)), dim3(kRedBlock), 0, s, x, count, vec_off, partial);
    int32_t st = check_launch("minmax_partial");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(minmax_update<T>, dim3(1), dim3(kRedBlock), 0, s, partial, static_cast<int>(nblocks), state, momentum);
    return check_launch("minmax_update");
}

// ---------------------------------------------------------------------------------- C1 for a list of tensors
// One calibration batch hands over dozens of activation tensors (gemma-3-270m: 72 per batch of 13-42 MB each): at
// 2 launches per tensor the loop is launch-bound (2.0 TB/s end to end against 5.5 TB/s for the reduction itself).
// Here ONE launch pair serves the whole list: blockIdx.y = tensor, blockIdx.x = slice of it.
struct ManyDesc {   // device-resident, 24 bytes per tensor (oq_hip.h: oq_minmax_desc)
    const float* x;
    int64_t count;
    float* state;
};

__global__ __launch_bounds__(kRedBlock) void minmax_many_partial(const ManyDesc* desc, float* partial /* [n][gridDim.x][2] */) {
    __shared__ float s_mn[kRedBlock / 64], s_mx[kRedBlock / 64];
    const ManyDesc d = desc[blockIdx.y];
    const float* x = d.x;
    const int64_t count = d.count;
    // peel to 16-byte alignment: [head | float4 body | tail]
    int64_t vec_off = static_cast<int64_t>(((16 - reinterpret_cast<uintptr_t>(x) % 16) % 16) / 4);
    if (vec_off > count) vec_off = count;
    const int64_t nvec = (count - vec_off) / 4;
    const Vec4<float>::type* xv = reinterpret_cast<const Vec4<float>::type*>(x + vec_off);
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    float mn = INFINITY, mx = -INFINITY;
    stream_minmax<float>(xv, nvec, tid, stride, mn, mx);
    for (int64_t j = tid; j < vec_off; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    for (int64_t j = vec_off + nvec * 4 + tid; j < count; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    block_minmax(mn, mx, s_mn, s_mx);
    if (threadIdx.x == 0) {
        float* o = partial + (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 2;
        o[0] = mn;
        o[1] = mx;
    }
}

// one wave per tensor: fold its slices and apply minmax.py:50-64 to its state
__global__ __launch_bounds__(64) void minmax_many_update(const ManyDesc* desc, const float* partial, int slices, double momentum) {
    const float* p = partial + static_cast<int64_t>(blockIdx.x) * slices * 2;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < slices; i += 64) {
        mn = nmin(mn, p[2 * i]);
        mx = nmax(mx, p[2 * i + 1]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (threadIdx.x != 0) return;
    float* state = desc[blockIdx.x].state;
    if (state[2] == 0.0f) {
        state[0] = mn;
        state[1] = mx;
        state[2] = 1.0f;
    } else if (momentum > 0.0) {
        const float m = static_cast<float>(momentum), om = static_cast<float>(1.0 - momentum);
        state[0] = m * state[0] + om * mn;
        state[1] = m * state[1] + om * mx;
    } else {
        state[0] = nmin(state[0], mn);
        state[1] = nmax(state[1], mx);
    }
}

static int many_slices(int64_t n) {   // ~4096 blocks in total, 4..64 slices per tensor
    int64_t s = 4096 / (n > 0 ? n : 1);
    if (s < 4) s = 4;
    if (s > 64) s = 64;
    return static_cast<int>(s);
}

// ------------------------------------------------------------------------------------- absmax
constexpr int kAbsChunkRows = 128;

__global__ __launch_bounds__(512) void absmax_cols_partial(const float* x, int64_t R, int64_t C, int64_t ldx, bool vec4,
                                                           float* partial, uint32_t ncol_tiles) {
    __shared__ float4 s_mx[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col_tile = blockIdx.x % ncol_tiles, chunk = blockIdx.x / ncol_tiles;
    const int64_t row0 = static_cast<int64_t>(chunk) * kAbsChunkRows + wave * 16;
    const int64_t col0 = static_cast<int64_t>(col_tile) * 256;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec4) {
        const int64_t c = col0 + lane * 4;
        if (c < C) {
            float4 t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                t[r] = (row0 + r < R) ? *reinterpret_cast<const float4*>(x + (row0 + r) * ldx + c) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mx[0] = nmax(mx[0], fabsf(t[r].x)); mx[1] = nmax(mx[1], fabsf(t[r].y));
                mx[2] = nmax(mx[2], fabsf(t[r].z)); mx[3] = nmax(mx[3], fabsf(t[r].w));
            }
        }
    } else {
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t c = col0 + i * 64 + lane;
                if (c < C && row0 + r < R) mx[i] = nmax(mx[i], fabsf(x[(row0 + r) * ldx + c]));
            }
    }
    s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < 8; ++w) {
        const float4 t = s_mx[w][lane];
        mx[0] = nmax(mx[0], t.x); mx[1] = nmax(mx[1], t.y); mx[2] = nmax(mx[2], t.z); mx[3] = nmax(mx[3], t.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = vec4 ? col0 + lane * 4 + i : col0 + i * 64 + lane;
        if (c < C) partial[static_cast<int64_t>(chunk) * C + c] = mx[i];
    }
}

__global__ void absmax_cols_finalize(const float* partial, int64_t chunks, int64_t C, float* out) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m = 0.f;
    for (int64_t k = 0; k < chunks; ++k) m = nmax(m, partial[k * C + c]);
    out[c] = m;
}

// one wave per row
__global__ __launch_bounds__(256) void absmax_rows(const float* x, int64_t R, int64_t C, int64_t ldx, bool vec4, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = x + r * ldx;
    float m = 0.f;
    if (vec4) {
        for (int64_t c = lane * 4; c < C; c += 256) {
            const float4 t = *reinterpret_cast<const float4*>(row + c);
            m = nmax(nmax(m, fabsf(t.x)), nmax(nmax(fabsf(t.y), fabsf(t.z)), fabsf(t.w)));
        }
    } else {
        for (int64_t c = lane; c < C; c += 64) m = nmax(m, fabsf(row[c]));
    }
    m = wave_max(m);
    if (lane == 0) out[r] = m;
}

// per-row (min, max): one wave per row
__global__ __launch_bounds__(256) void minmax_rows(const float* x, int64_t R, int64_t C, int64_t ldx, bool vec4, float* mn_out,
                                                   float* mx_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = x + r * ldx;
    float mn = INFINITY, mx = -INFINITY;
    if (vec4) {
        for (int64_t c = lane * 4; c < C; c += 256) {
            const float4 t = *reinterpret_cast<const float4*>(row + c);
            mn = nmin(nmin(mn, t.x), nmin(nmin(t.y, t.z), t.w));
            mx = nmax(nmax(mx, t.x), nmax(nmax(t.y, t.z), t.w));
        }
    } else {
        for (int64_t c = lane; c < C; c += 64) { mn = nmin(mn, row[c]); mx = nmax(mx, row[c]); }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) { mn_out[r] = mn; mx_out[r] = mx; }
}

// 64-bit content fingerprint of a byte range in HBM (seam.py: is this calibration input the one whose Hessian is cached?).
// Every 16-byte word w at index i contributes mix(i, w) to a sum modulo 2^64, so the order in which lanes and blocks add is
// irrelevant and one agent-scope atomic per wave ends the kernel.  mix = the splitmix64 finaliser over the four lanes of the
// word chained with its index: an edit of any bit of any word changes its term (the finaliser is a bijection of the chained
// state), so two contents collide with probability 2^-64, position swaps included.  The tail (< 16 bytes) is read byte by byte.
__device__ __forceinline__ uint64_t fp_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kRedBlock) void fingerprint_kernel(const uint4* x, int64_t nvec, const uint8_t* tail, int ntail, int64_t nbytes,
                                                                 unsigned long long* out) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kRedBlock;
    uint64_t acc = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kRedBlock + threadIdx.x; i < nvec; i += stride) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x) + i);
        uint64_t h = fp_mix(static_cast<uint64_t>(i) + 0x9E3779B97F4A7C15ull);
        h = fp_mix(h ^ (static_cast<uint64_t>(w[0]) | (static_cast<uint64_t>(w[1]) << 32)));
        h = fp_mix(h ^ (static_cast<uint64_t>(w[2]) | (static_cast<uint64_t>(w[3]) << 32)));
        acc += h;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t h = fp_mix(static_cast<uint64_t>(nbytes) ^ 0xD1B54A32D192ED03ull);      // the length is part of the content
        for (int j = 0; j < ntail; ++j) h = fp_mix(h ^ (static_cast<uint64_t>(tail[j]) + 0x100ull * static_cast<uint64_t>(j + 1)));
        acc += h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(acc)), off, 64));
        const uint32_t hi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(acc >> 32)), off, 64));
        acc += static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, static_cast<unsigned long long>(acc));
}

}  // namespace oq

extern "C" {

using namespace oq;

int32_t oq_fingerprint64(const void* data, int64_t nbytes, uint64_t* out, void* stream) {
    OQ_REQUIRE(data && out && nbytes > 0 && nbytes <= kMaxElements, OQ_ERR_INVALID_ARGUMENT, "oq_fingerprint64: bad argument (1 <= nbytes <= 2^40)");
    OQ_REQUIRE((reinterpret_cast<uintptr_t>(data) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_fingerprint64: data must be 16-byte aligned, out 8-byte aligned");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(out, 0, sizeof(uint64_t), s) != hipSuccess) return fail(OQ_ERR_LAUNCH, "oq_fingerprint64: cannot clear the result");
    const int64_t nvec = nbytes / 16;
    int64_t nblocks = ceil_div(nvec > 0 ? nvec : 1, static_cast<int64_t>(kRedBlock) * 8);
    if (nblocks > kRedMaxBlocks) nblocks = kRedMaxBlocks;
    hipLaunchKernelGGL(fingerprint_kernel, dim3(static_cast<uint32_t>(nblocks)), dim3(kRedBlock), 0, s, static_cast<const uint4*>(data), nvec,
                       static_cast<const uint8_t*>(data) + nvec * 16, static_cast<int>(nbytes - nvec * 16), nbytes,
                       reinterpret_cast<unsigned long long*>(out));
    return check_launch("fingerprint_kernel");
}

size_t oq_minmax_workspace_bytes(int64_t count) {
    (void)count;
    return static_cast<size_t>(kRedMaxBlocks) * 2 * sizeof(double);
}

int32_t oq_minmax_collect_f32(const float* x, int64_t count, float* state, double momentum, void* workspace,
                              size_t workspace_bytes, void* stream) {
    return minmax_collect<float>(x, count, state, momentum, workspace, workspace_bytes, stream);
}

int32_t oq_minmax_collect_f64(const double* x, int64_t count, double* state, double momentum, void* workspace,
                              size_t workspace_bytes, void* stream) {
    return minmax_collect<double>(x, count, state, momentum, workspace, workspace_bytes, stream);
}

size_t oq_minmax_many_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return static_cast<size_t>(n) * many_slices(n) * 2 * sizeof(float) + 256;
}

int32_t oq_minmax_collect_many_f32(const void* desc, int64_t n, double momentum, void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(desc && n > 0 && n <= 65535, OQ_ERR_INVALID_ARGUMENT, "oq_minmax_collect_many_f32: bad argument (1 <= n <= 65535)");
    OQ_REQUIRE(momentum >= 0.0 && momentum < 1.0, OQ_ERR_INVALID_ARGUMENT, "Momentum must be in the range [0, 1).");
    const size_t need = oq_minmax_many_workspace_bytes(n);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_minmax_collect_many_f32: workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
    const int slices = many_slices(n);
    hipStream_t s = as_stream(stream);
    const ManyDesc* d = static_cast<const ManyDesc*>(desc);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(minmax_many_partial, dim3(static_cast<uint32_t>(slices), static_cast<uint32_t>(n)), dim3(kRedBlock), 0, s, d, partial);
    int32_t st = check_launch("minmax_many_partial");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(minmax_many_update, dim3(static_cast<uint32_t>(n)), dim3(64), 0, s, d, partial, slices, momentum);
    return check_launch("minmax_many_update");
}

size_t oq_absmax_workspace_bytes(int64_t R, int64_t C, int32_t transposed) {
    if (transposed || !matrix_ok(R, C, C)) return 256;
    return static_cast<size_t>(ceil_div(R, kAbsChunkRows) * C) * sizeof(float) + 256;
}

int32_t oq_absmax_f32(const float* x, int64_t R, int64_t C, int64_t ldx, int32_t transposed, float* out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(x && out && matrix_ok(R, C, ldx), OQ_ERR_INVALID_ARGUMENT, "oq_absmax_f32: bad argument");
    const bool vec4 = (C % 4 == 0) && (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    hipStream_t s = as_stream(stream);
    if (transposed) {
        hipLaunchKernelGGL(absmax_rows, dim3(static_cast<uint32_t>(ceil_div(R, 4))), dim3(256), 0, s, x, R, C, ldx, vec4, out);
        return check_launch("absmax_rows");
    }
    const int64_t chunks = ceil_div(R, kAbsChunkRows);
    const size_t need = static_cast<size_t>(chunks * C) * sizeof(float);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_absmax_f32: workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
    const uint32_t ncol_tiles = static_cast<uint32_t>(ceil_div(C, 256));
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(absmax_cols_partial, dim3(ncol_tiles * static_cast<uint32_t>(chunks)), dim3(512), 0, s, x, R, C, ldx,
                       vec4, partial, ncol_tiles);
    int32_t st = check_launch("absmax_cols_partial");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(absmax_cols_finalize, dim3(static_cast<uint32_t>(ceil_div(C, 256))), dim3(256), 0, s, partial, chunks, C, out);
    return check_launch("absmax_cols_finalize");
}

int32_t oq_minmax_rows_f32(const float* x, int64_t R, int64_t C, int64_t ldx, float* min_out, float* max_out, void* stream) {
    OQ_REQUIRE(x && min_out && max_out && matrix_ok(R, C, ldx), OQ_ERR_INVALID_ARGUMENT, "oq_minmax_rows_f32: bad argument");
    const bool vec4 = (C % 4 == 0) && (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    hipLaunchKernelGGL(minmax_rows, dim3(static_cast<uint32_t>(ceil_div(R, 4))), dim3(256), 0, as_stream(stream), x, R, C, ldx,
                       vec4, min_out, max_out);
    return check_launch("minmax_rows");
}

}  // extern "C"
