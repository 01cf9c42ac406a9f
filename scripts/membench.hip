// Stand-alone HBM access-pattern micro-benchmarks used to choose the tiling of the fused RTN kernel.
//   hipcc --offload-arch=gfx950 -O3 -o membench scripts/membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int K = 4096, N = 11008;

__device__ __forceinline__ float f4max(float m, float4 t) { return fmaxf(fmaxf(m, t.x), fmaxf(fmaxf(t.y, t.z), t.w)); }

// A: linear grid-stride read, UNROLL independent 16-B loads in flight per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void read_linear(const float4* x, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) m = f4max(m, t[u]);
    }
    for (; i < n4; i += stride) m = f4max(m, x[i]);
    if (m == 12345.f) out[0] = m;
}

// B: tile pattern of the fused kernel: block = WAVES waves, each wave RPW rows x 1 KiB; tiles in id order
template <int RPW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void read_tiles(const float* W, int ncol_tiles, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_tile = blockIdx.x / ncol_tiles, col_tile = blockIdx.x % ncol_tiles;
    const size_t row0 = (size_t)row_tile * (RPW * WAVES) + wave * RPW;
    const float* p = W + row0 * N + col_tile * 256 + lane * 4;
    float4 t[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) m = f4max(m, t[r]);
    if (m == 12345.f) out[0] = m;
}

// C: persistent: grid = nblk blocks, each walks tiles id = blockIdx + j*gridDim with the next tile prefetched
template <int RPW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void read_tiles_persistent(const float* W, int ncol_tiles, int ntiles, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = 0.f;
    float4 cur[RPW], nxt[RPW];
    int tile = blockIdx.x;
    auto addr = [&](int t) {
        const int row_tile = t / ncol_tiles, col_tile = t % ncol_tiles;
        return W + ((size_t)row_tile * (RPW * WAVES) + wave * RPW) * N + col_tile * 256 + lane * 4;
    };
    if (tile < ntiles) {
        const float* p = addr(tile);
#pragma unroll
        for (int r = 0; r < RPW; ++r) cur[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
    }
    while (tile < ntiles) {
        const int nt = tile + gridDim.x;
        if (nt < ntiles) {
            const float* p = addr(nt);
#pragma unroll
            for (int r = 0; r < RPW; ++r) nxt[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) m = f4max(m, cur[r]);
#pragma unroll
        for (int r = 0; r < RPW; ++r) cur[r] = nxt[r];
        tile = nt;
    }
    if (m == 12345.f) out[0] = m;
}

// D: linear copy fp32 -> fp32 ; E: read fp32, write 1 byte per element (the KN output volume) ; F: write 0.5 B
__global__ __launch_bounds__(256) void copy_linear(const float4* x, float4* y, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + stride < n4; i += 2 * stride) {
        float4 a = x[i], b = x[i + stride];
        y[i] = a; y[i + stride] = b;
    }
    for (; i < n4; i += stride) y[i] = x[i];
}
// NOTE: rounds 1's first version of this kernel had no remainder loop and silently skipped up to 26 % of the
// elements (its "33 us" was for 74 % of the work); the remainder loop below makes it complete.
template <int UNROLL>
__global__ __launch_bounds__(256) void read4_write1(const float4* x, uint32_t* y, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i + (n4 / (UNROLL * stride)) * (UNROLL * stride); j < n4; j += stride) {
        const float4 t = x[j];
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(t.x, 0, 0);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t.y, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t.z, 2, w);
        y[j] = __builtin_amdgcn_cvt_pk_u8_f32(t.w, 3, w);
    }
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].x, 0, 0);
            w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].y, 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].z, 2, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].w, 3, w);
            y[i + u * stride] = w;
        }
    }
}

__global__ void fill_random(float* x, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        x[i] = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * 3.0f;
    }
}


// G: build-up from B towards the real fused kernel, one feature per LEVEL
//   1: min and max   2: + LDS exchange + barrier   3: + scattered scale/zp stores   4: + runtime stride (ldw) and 64-bit args
template <int LEVEL>
__global__ __launch_bounds__(512) void buildup(const float* W, long ldw_rt, int ncol_tiles, float* scale, unsigned char* zp, float* out) {
    __shared__ float4 s_mn[8][64];
    __shared__ float4 s_mx[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_tile = blockIdx.x / ncol_tiles, col_tile = blockIdx.x % ncol_tiles;
    const size_t row0 = (size_t)row_tile * 128 + wave * 16;
    const long ldw = LEVEL >= 4 ? ldw_rt : N;
    const float* p = W + row0 * ldw + col_tile * 256 + lane * 4;
    float4 t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * ldw);
    float4 mn = t[0], mx = t[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        mn.x = fminf(mn.x, t[r].x); mn.y = fminf(mn.y, t[r].y); mn.z = fminf(mn.z, t[r].z); mn.w = fminf(mn.w, t[r].w);
        mx.x = fmaxf(mx.x, t[r].x); mx.y = fmaxf(mx.y, t[r].y); mx.z = fmaxf(mx.z, t[r].z); mx.w = fmaxf(mx.w, t[r].w);
    }
    if (LEVEL >= 2) {
        s_mn[wave][lane] = mn; s_mx[wave][lane] = mx;
        __syncthreads();
        for (int w = 0; w < 8; ++w) {
            const float4 a = s_mn[w][lane], b = s_mx[w][lane];
            mn.x = fminf(mn.x, a.x); mn.y = fminf(mn.y, a.y); mn.z = fminf(mn.z, a.z); mn.w = fminf(mn.w, a.w);
            mx.x = fmaxf(mx.x, b.x); mx.y = fmaxf(mx.y, b.y); mx.z = fmaxf(mx.z, b.z); mx.w = fmaxf(mx.w, b.w);
        }
    }
    if (LEVEL >= 3) {
        if (wave == 0) {
            const float mns[4] = {mn.x, mn.y, mn.z, mn.w}, mxs[4] = {mx.x, mx.y, mx.z, mx.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t o = (size_t)(col_tile * 256 + lane * 4 + i) * 32 + row_tile;
                const float sc = (mxs[i] - mns[i]) / 15.0f;
                scale[o] = sc;
                zp[o] = (unsigned char)rintf(-mns[i] / sc);
            }
        }
    } else {
        if (mn.x + mx.y + mn.z + mx.w == 12345.f) out[0] = mn.x;
    }
}

// H: tile read (as B) + byte output, no reduction: isolates the cost of the strided store patterns
//   MODE 0: KN  4 B/lane per row (256 B runs)          MODE 1: KN 16 B/lane after a quad transpose (4 rows x 256 B per instr)
//   MODE 2: NBITS-like 8 B/lane per column at 2 KiB stride   MODE 3: no store
template <int MODE>
__global__ __launch_bounds__(512) void tile_rw(const float* W, int ncol_tiles, unsigned char* q, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_tile = blockIdx.x / ncol_tiles, col_tile = blockIdx.x % ncol_tiles;
    const size_t row0 = (size_t)row_tile * 128 + wave * 16;
    const size_t col0 = (size_t)col_tile * 256 + lane * 4;
    const float* p = W + row0 * N + col0;
    float4 t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
    uint32_t w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
        w[r] = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
    }
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<uint32_t*>(q + (row0 + r) * N + col0) = w[r];
    } else if (MODE == 1) {
        const int ql = lane & 3;
#pragma unroll
        for (int r4 = 0; r4 < 16; r4 += 4) {
            // 4x4 transpose inside each quad: lane ql collects row r4+ql from the 4 lanes of its quad
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // value held by quad-lane j for row (r4 + ql)
                uint32_t mine = ql == 0 ? w[r4] : (ql == 1 ? w[r4 + 1] : (ql == 2 ? w[r4 + 2] : w[r4 + 3]));
                (void)mine;
                uint32_t src = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const uint32_t v = __shfl(w[r4 + rr], (lane & ~3) | j, 64);
                    if (rr == ql) src = v;
                }
                o[j] = src;
            }
            const size_t c16 = (size_t)col_tile * 256 + (lane >> 2) * 16;
            *reinterpret_cast<uint4*>(q + (row0 + r4 + ql) * N + c16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = (w[0] >> (8 * i) & 15) | (w[1] >> (8 * i) & 15) << 4 | (w[2] >> (8 * i) & 15) << 8 | (w[3] >> (8 * i) & 15) << 12 |
                               (w[4] >> (8 * i) & 15) << 16 | (w[5] >> (8 * i) & 15) << 20 | (w[6] >> (8 * i) & 15) << 24 | (w[7] >> (8 * i) & 15) << 28;
            const uint32_t b = (w[8] >> (8 * i) & 15) | (w[9] >> (8 * i) & 15) << 4 | (w[10] >> (8 * i) & 15) << 8 | (w[11] >> (8 * i) & 15) << 12 |
                               (w[12] >> (8 * i) & 15) << 16 | (w[13] >> (8 * i) & 15) << 20 | (w[14] >> (8 * i) & 15) << 24 | (w[15] >> (8 * i) & 15) << 28;
            *reinterpret_cast<uint2*>(q + ((col0 + i) * 32 + row_tile) * 64 + wave * 8) = make_uint2(a, b);
        }
    } else {
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc ^= w[r];
        if (acc == 0x12345678u) out[0] = 1.f;
    }
}

__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7u, x = b & 7u;
    const uint32_t base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}
// I: KN store pattern under different block orders / store flavours
//   ORDER 0: column tiles fastest (plain)   1: column tiles fastest, XCD-chunked   2: K fastest, XCD-chunked
template <int ORDER, bool NTS>
__global__ __launch_bounds__(512) void tile_rw_order(const float* W, int ncol_tiles, int nrow_tiles, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nblk = ncol_tiles * nrow_tiles;
    uint32_t id = ORDER == 0 ? blockIdx.x : xcd_remap(blockIdx.x, nblk);
    int row_tile, col_tile;
    if (ORDER == 2) { col_tile = id / nrow_tiles; row_tile = id % nrow_tiles; }
    else { row_tile = id / ncol_tiles; col_tile = id % ncol_tiles; }
    const size_t row0 = (size_t)row_tile * 128 + wave * 16;
    const size_t col0 = (size_t)col_tile * 256 + lane * 4;
    const float* p = W + row0 * N + col0;
    float4 t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
        uint32_t* dst = reinterpret_cast<uint32_t*>(q + (row0 + r) * N + col0);
        if (NTS) __builtin_nontemporal_store(x, dst); else *dst = x;
    }
}

// J: same tile kernel, but the 32 KiB output tile of a block is written to a LINEAR location
// (wrong layout on purpose): separates "strided address pattern" from "kernel structure".
//   LIN 1: block-linear (tile contiguous)   LIN 2: wave rows contiguous but tiles in band order
template <int LIN>
__global__ __launch_bounds__(512) void tile_rw_linear(const float* W, int ncol_tiles, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_tile = blockIdx.x / ncol_tiles, col_tile = blockIdx.x % ncol_tiles;
    const size_t row0 = (size_t)row_tile * 128 + wave * 16;
    const size_t col0 = (size_t)col_tile * 256 + lane * 4;
    const float* p = W + row0 * N + col0;
    float4 t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
        size_t off;
        if (LIN == 1) off = (size_t)blockIdx.x * 32768 + (size_t)wave * 4096 + r * 256 + lane * 4;
        else off = ((size_t)row_tile * 128 + wave * 16 + r) * N + (size_t)col_tile * 256 + lane * 4;   // == real layout (control)
        *reinterpret_cast<uint32_t*>(q + off) = x;
    }
}

// K: persistent tile kernel with the rolling register reload (row r of the next unit is loaded right
// after row r of the current unit has been converted and stored).  STRIDE 1: units b, b+G, b+2G (column
// fastest, grid-stride)   STRIDE 0: contiguous chunk of units per block.
template <int WAVES, int STRIDE, bool NTS, int LINOUT = 0>
__global__ __launch_bounds__(WAVES * 64) void tile_stream(const float* W, int ncol_tiles, int ntiles, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWS = WAVES * 16;
    int u, u_end, step;
    if (STRIDE) { u = blockIdx.x; u_end = ntiles; step = gridDim.x; }
    else { u = (int)((long)blockIdx.x * ntiles / gridDim.x); u_end = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x); step = 1; }
    if (u >= u_end) return;
    auto base = [&](int t, size_t& row0, size_t& col0) {
        const int row_tile = t / ncol_tiles, col_tile = t % ncol_tiles;
        row0 = (size_t)row_tile * ROWS + wave * 16; col0 = (size_t)col_tile * 256 + lane * 4;
    };
    size_t row0, col0;
    base(u, row0, col0);
    float4 t[16];
    {
        const float* p = W + row0 * N + col0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
    }
    auto body = [&](auto reload_tag) {
        constexpr bool RELOAD = decltype(reload_tag)::value;
        size_t nrow0 = row0, ncol0 = col0;
        if (RELOAD) base(u + step, nrow0, ncol0);
        const float* np = W + nrow0 * N + ncol0;
        unsigned char* o = LINOUT ? q + ((size_t)u * WAVES + wave) * 4096 + lane * 4 : q + row0 * N + col0;
        const size_t ostride = LINOUT ? 256 : N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
            uint32_t* dst = reinterpret_cast<uint32_t*>(o + (size_t)r * ostride);
            if (NTS) __builtin_nontemporal_store(x, dst); else *dst = x;
            if constexpr (RELOAD) t[r] = *reinterpret_cast<const float4*>(np + (size_t)r * N);
        }
        row0 = nrow0; col0 = ncol0;
    };
    for (; u + step < u_end; u += step) body(std::true_type{});
    body(std::false_type{});
}

// M: linear (non grid-stride) read4_write1: block handles one contiguous chunk, each wave UNROLL KiB-rows deep,
// no loop: the "one tile per short-lived block" structure, but with linear addresses
template <int UNROLL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void linear_oneshot(const float4* x, uint32_t* y, size_t n4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * WAVES + wave) * UNROLL * 64 + lane;
    float4 t[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) t[u] = x[base + u * 64];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].x, 0, 0);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].y, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].z, 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].w, 3, w);
        y[base + u * 64] = w;
    }
}

// N: one-shot tile kernel, 128 rows x 256 cols per block, WAVES waves x RPW rows (WAVES*RPW = 128), byte stores
template <int WAVES, int RPW, bool NTS>
__global__ __launch_bounds__(WAVES * 64) void tile_rw_shape(const float* W, int ncol_tiles, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_tile = blockIdx.x / ncol_tiles, col_tile = blockIdx.x % ncol_tiles;
    const size_t row0 = (size_t)row_tile * 128 + wave * RPW;
    const size_t col0 = (size_t)col_tile * 256 + lane * 4;
    const float* p = W + row0 * N + col0;
    float4 t[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
        uint32_t* dst = reinterpret_cast<uint32_t*>(q + (row0 + r) * N + col0);
        if (NTS) __builtin_nontemporal_store(x, dst); else *dst = x;
    }
}

// P: looping tile kernel WITHOUT the rolling reload: per unit 16 loads, then 16 x (cvt + store), like E<16> but
// with tile addresses (TILE=1) or linear addresses (TILE=0); 4 waves per block, grid-stride over units.
template <int TILE, bool NTS>
__global__ __launch_bounds__(256) void loop_units(const float* W, int ncol_tiles, int nunits, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        size_t roff, woff, rstride, wstride;
        if (TILE) {
            const int row_tile = u / ncol_tiles, col_tile = u % ncol_tiles;     // unit = 64 rows x 256 cols
            const size_t row0 = (size_t)row_tile * 64 + wave * 16, col0 = (size_t)col_tile * 256 + lane * 4;
            roff = row0 * N + col0; woff = roff; rstride = N; wstride = N;
        } else {
            roff = ((size_t)u * 4 + wave) * 4096 + lane * 4; woff = roff; rstride = 256; wstride = 256;
        }
        float4 t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(W + roff + r * rstride);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
            x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
            uint32_t* dst = reinterpret_cast<uint32_t*>(q + woff + r * wstride);
            if (NTS) __builtin_nontemporal_store(x, dst); else *dst = x;
        }
    }
}

// Q: one-shot tile kernel with a different wave -> sub-tile map: the 8 waves of a block take CW adjacent
// column tiles x (8/CW) row slices of 16 rows (CW=1: the fused-kernel geometry, CW=8: 16 rows x 2048 columns)
template <int CW, bool NTS>
__global__ __launch_bounds__(512) void tile_rw_map(const float* W, unsigned char* q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RS = 8 / CW;                       // row slices per block
    const int tiles_x = N / (256 * CW);              // 43 / CW must be handled: N=11008 -> 43 tiles; use ceil + guard
    const int bx = blockIdx.x % ((43 + CW - 1) / CW), by = blockIdx.x / ((43 + CW - 1) / CW);
    (void)tiles_x;
    const int col_tile = bx * CW + (wave % CW);
    if (col_tile >= 43) return;
    const size_t row0 = (size_t)by * (16 * RS) + (wave / CW) * 16;
    const size_t col0 = (size_t)col_tile * 256 + lane * 4;
    const float* p = W + row0 * N + col0;
    float4 t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = *reinterpret_cast<const float4*>(p + (size_t)r * N);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].x, 0, 0);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].y, 1, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].z, 2, x);
        x = __builtin_amdgcn_cvt_pk_u8_f32(t[r].w, 3, x);
        uint32_t* dst = reinterpret_cast<uint32_t*>(q + (row0 + r) * N + col0);
        if (NTS) __builtin_nontemporal_store(x, dst); else *dst = x;
    }
}

// R: one-shot, but with E's address map: adjacent waves take adjacent KiB, the U loads of one wave are
// (total waves) KiB apart.  Separates "looping" from "who touches neighbouring addresses when".
template <int U, int WAVES, bool NTS>
__global__ __launch_bounds__(WAVES * 64) void oneshot_interleaved(const float4* x, uint32_t* y, size_t n4) {
    const size_t lane = threadIdx.x & 63, g = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const size_t wv = n4 / (64 * U);
    float4 t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = x[((size_t)u * wv + g) * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].x, 0, 0);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].y, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].z, 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[u].w, 3, w);
        uint32_t* dst = &y[((size_t)u * wv + g) * 64 + lane];
        if (NTS) __builtin_nontemporal_store(w, dst); else *dst = w;
    }
}

int main(int argc, char** argv) {
    const bool rnd = argc > 1 && atoi(argv[1]) == 1;
    printf("data = %s\n", rnd ? "random" : "constant");
    const size_t elems = (size_t)K * N, bytes = elems * 4;
    const int NB = 4;
    std::vector<float*> in(NB); std::vector<float*> outb(NB);
    for (int b = 0; b < NB; ++b) { CK(hipMalloc(&in[b], bytes)); CK(hipMalloc(&outb[b], bytes)); CK(hipMemset(in[b], 0x3c, bytes)); if (rnd) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, in[b], elems, 1234u + b); }
    float* sink; CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto launch) {
        for (int i = 0; i < 10; ++i) launch(i % NB);
        CK(hipDeviceSynchronize());
        const int iters = 100;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch(i % NB);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("%-58s %7.2f us  %6.2f TB/s\n", name, us, moved / us / 1e6);
        fflush(stdout);
    };
    const size_t n4 = elems / 4;
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[128];
        snprintf(nm, 128, "A read_linear<4>  grid=%d x256", blocks);
        timeit(nm, bytes, [&](int b) { hipLaunchKernelGGL(read_linear<4>, dim3(blocks), dim3(256), 0, 0, (const float4*)in[b], n4, sink); });
        snprintf(nm, 128, "A read_linear<8>  grid=%d x256", blocks);
        timeit(nm, bytes, [&](int b) { hipLaunchKernelGGL(read_linear<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)in[b], n4, sink); });
    }
    timeit("A read_linear<16> grid=2048 x256", bytes, [&](int b) { hipLaunchKernelGGL(read_linear<16>, dim3(2048), dim3(256), 0, 0, (const float4*)in[b], n4, sink); });
    const int nct = N / 256;
    timeit("B read_tiles<16,8> (128x256 tiles, 1376 blocks x512)", bytes, [&](int b) { hipLaunchKernelGGL((read_tiles<16, 8>), dim3(nct * (K / 128)), dim3(512), 0, 0, in[b], nct, sink); });
    timeit("B read_tiles<8,8>  (64x256 tiles, 2752 blocks x512)", bytes, [&](int b) { hipLaunchKernelGGL((read_tiles<8, 8>), dim3(nct * (K / 64)), dim3(512), 0, 0, in[b], nct, sink); });
    timeit("B read_tiles<16,4> (64x256 tiles, 2752 blocks x256)", bytes, [&](int b) { hipLaunchKernelGGL((read_tiles<16, 4>), dim3(nct * (K / 64)), dim3(256), 0, 0, in[b], nct, sink); });
    timeit("B read_tiles<32,4> (128x256 tiles, 1376 blocks x256)", bytes, [&](int b) { hipLaunchKernelGGL((read_tiles<32, 4>), dim3(nct * (K / 128)), dim3(256), 0, 0, in[b], nct, sink); });
    {
        float* sc; unsigned char* zpb; CK(hipMalloc(&sc, (size_t)N * 32 * 4)); CK(hipMalloc(&zpb, (size_t)N * 32));
        timeit("G buildup<1> min+max", bytes, [&](int b) { hipLaunchKernelGGL(buildup<1>, dim3(nct * 32), dim3(512), 0, 0, in[b], (long)N, nct, sc, zpb, sink); });
        timeit("G buildup<2> + LDS exchange/barrier", bytes, [&](int b) { hipLaunchKernelGGL(buildup<2>, dim3(nct * 32), dim3(512), 0, 0, in[b], (long)N, nct, sc, zpb, sink); });
        timeit("G buildup<3> + scattered scale/zp stores", bytes, [&](int b) { hipLaunchKernelGGL(buildup<3>, dim3(nct * 32), dim3(512), 0, 0, in[b], (long)N, nct, sc, zpb, sink); });
        timeit("G buildup<4> + runtime ldw", bytes, [&](int b) { hipLaunchKernelGGL(buildup<4>, dim3(nct * 32), dim3(512), 0, 0, in[b], (long)N, nct, sc, zpb, sink); });
    }
    timeit("H tile_rw<3> read tiles + cvt, no store", bytes, [&](int b) { hipLaunchKernelGGL(tile_rw<3>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b], sink); });
    timeit("H tile_rw<0> + KN store 4 B/lane (45 MB)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(tile_rw<0>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b], sink); });
    timeit("H tile_rw<1> + KN store 16 B/lane quad-transposed", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(tile_rw<1>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b], sink); });
    timeit("H tile_rw<2> + NBITS store 8 B/lane @2KiB stride (22.5 MB)", bytes * 1.125, [&](int b) { hipLaunchKernelGGL(tile_rw<2>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b], sink); });
    timeit("I order0 plain stores", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<0, false>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("I order0 nt stores", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<0, true>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("I order1 (col fastest, XCD chunks) plain", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<1, false>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("I order1 (col fastest, XCD chunks) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<1, true>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("I order2 (K fastest, XCD chunks) plain", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<2, false>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("I order2 (K fastest, XCD chunks) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_order<2, true>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, 32, (unsigned char*)outb[b]); });
    timeit("J tile kernel, block-linear output (wrong layout)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(tile_rw_linear<1>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    timeit("J tile kernel, real layout (control)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(tile_rw_linear<2>, dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    timeit("N tile 8 waves x 16 rows, nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_shape<8, 16, true>), dim3(nct * 32), dim3(512), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    timeit("N tile 16 waves x 8 rows, nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_shape<16, 8, true>), dim3(nct * 32), dim3(1024), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    timeit("N tile 16 waves x 8 rows", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_shape<16, 8, false>), dim3(nct * 32), dim3(1024), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    timeit("N tile 4 waves x 32 rows, nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_shape<4, 32, true>), dim3(nct * 32), dim3(256), 0, 0, in[b], nct, (unsigned char*)outb[b]); });
    for (int nblk : {1024, 2048}) {
        char nm[128];
        snprintf(nm, 128, "P loop_units linear grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((loop_units<0, false>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "P loop_units linear nt grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((loop_units<0, true>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "P loop_units TILE grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((loop_units<1, false>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "P loop_units TILE nt grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((loop_units<1, true>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
    }
    timeit("Q map CW=1 (128 rows x 256 cols per block) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_map<1, true>), dim3(43 * (K / 128)), dim3(512), 0, 0, in[b], (unsigned char*)outb[b]); });
    timeit("Q map CW=2 (64 rows x 512 cols) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_map<2, true>), dim3(22 * (K / 64)), dim3(512), 0, 0, in[b], (unsigned char*)outb[b]); });
    timeit("Q map CW=4 (32 rows x 1024 cols) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_map<4, true>), dim3(11 * (K / 32)), dim3(512), 0, 0, in[b], (unsigned char*)outb[b]); });
    timeit("Q map CW=8 (16 rows x 2048 cols) nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_map<8, true>), dim3(6 * (K / 16)), dim3(512), 0, 0, in[b], (unsigned char*)outb[b]); });
    timeit("Q map CW=8 (16 rows x 2048 cols)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_rw_map<8, false>), dim3(6 * (K / 16)), dim3(512), 0, 0, in[b], (unsigned char*)outb[b]); });
    timeit("R oneshot_interleaved<16, 8 waves>", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((oneshot_interleaved<16, 8, false>), dim3(n4 / (64 * 16 * 8)), dim3(512), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("R oneshot_interleaved<16, 8 waves> nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((oneshot_interleaved<16, 8, true>), dim3(n4 / (64 * 16 * 8)), dim3(512), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("R oneshot_interleaved<16, 4 waves>", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((oneshot_interleaved<16, 4, false>), dim3(n4 / (64 * 16 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("R oneshot_interleaved<4, 4 waves>", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((oneshot_interleaved<4, 4, false>), dim3(n4 / (64 * 4 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("R oneshot_interleaved<4, 4 waves> nt", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((oneshot_interleaved<4, 4, true>), dim3(n4 / (64 * 4 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("E read4_write1<1> grid=8192", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(read4_write1<1>, dim3(8192), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("E read4_write1<2> grid=8192", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(read4_write1<2>, dim3(8192), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("E read4_write1<16> grid=2048", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(read4_write1<16>, dim3(2048), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("M linear_oneshot<16, 8 waves> (1376 blocks)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((linear_oneshot<16, 8>), dim3(n4 / (16 * 64 * 8)), dim3(512), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("M linear_oneshot<16, 4 waves> (2752 blocks)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((linear_oneshot<16, 4>), dim3(n4 / (16 * 64 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("M linear_oneshot<4, 4 waves> (11008 blocks)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((linear_oneshot<4, 4>), dim3(n4 / (4 * 64 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("M linear_oneshot<8, 4 waves> (5504 blocks)", bytes * 1.25, [&](int b) { hipLaunchKernelGGL((linear_oneshot<8, 4>), dim3(n4 / (8 * 64 * 4)), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    for (int nblk : {256, 512, 768, 1024}) {
        char nm[128];
        snprintf(nm, 128, "K stream<8 waves, grid-stride> grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<8, 1, false>), dim3(nblk), dim3(512), 0, 0, in[b], nct, nct * 32, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "K stream<8 waves, chunked> grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<8, 0, false>), dim3(nblk), dim3(512), 0, 0, in[b], nct, nct * 32, (unsigned char*)outb[b]); });
    }
    for (int nblk : {256, 512}) {
        char nm[128];
        snprintf(nm, 128, "K stream<8 waves, grid-stride, LINEAR OUT> grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<8, 1, false, 1>), dim3(nblk), dim3(512), 0, 0, in[b], nct, nct * 32, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "K stream<4 waves, grid-stride, LINEAR OUT> grid=%d", 2 * nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<4, 1, false, 1>), dim3(2 * nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
    }
    for (int nblk : {512, 1024, 2048}) {
        char nm[128];
        snprintf(nm, 128, "K stream<4 waves, grid-stride> grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<4, 1, false>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
        snprintf(nm, 128, "K stream<4 waves, grid-stride, nt stores> grid=%d", nblk);
        timeit(nm, bytes * 1.25, [&](int b) { hipLaunchKernelGGL((tile_stream<4, 1, true>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * 64, (unsigned char*)outb[b]); });
    }
    for (int nblk : {256, 512, 768, 1024}) {
        char nm[128];
        snprintf(nm, 128, "C persistent<16,8> prefetch, grid=%d", nblk);
        timeit(nm, bytes, [&](int b) { hipLaunchKernelGGL((read_tiles_persistent<16, 8>), dim3(nblk), dim3(512), 0, 0, in[b], nct, nct * (K / 128), sink); });
    }
    for (int nblk : {512, 1024, 2048}) {
        char nm[128];
        snprintf(nm, 128, "C persistent<16,4> prefetch, grid=%d", nblk);
        timeit(nm, bytes, [&](int b) { hipLaunchKernelGGL((read_tiles_persistent<16, 4>), dim3(nblk), dim3(256), 0, 0, in[b], nct, nct * (K / 64), sink); });
    }
    timeit("D copy_linear (R+W 360 MB) grid=4096", 2.0 * bytes, [&](int b) { hipLaunchKernelGGL(copy_linear, dim3(4096), dim3(256), 0, 0, (const float4*)in[b], (float4*)outb[b], n4); });
    timeit("E read4_write1<4> (R 180 + W 45 MB) grid=4096", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(read4_write1<4>, dim3(4096), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    timeit("E read4_write1<8> (R 180 + W 45 MB) grid=2048", bytes * 1.25, [&](int b) { hipLaunchKernelGGL(read4_write1<8>, dim3(2048), dim3(256), 0, 0, (const float4*)in[b], (uint32_t*)outb[b], n4); });
    return 0;
}
