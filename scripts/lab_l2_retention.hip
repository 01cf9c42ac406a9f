// Lab: does a tile read with default-policy loads stay in the XCD's L2 while the same CU streams more tiles with non-temporal loads?
// One 8-wave workgroup per CU (the per-tensor kernel's shape).  Each workgroup: read tile A (128 rows x 256 columns = 128 KB,
// default policy), then stream S more tiles (nt or default), then re-read A and time the re-read with the 100 MHz clock;
// compared with the time of reading a tile nobody has touched.   hipcc --offload-arch=gfx950 -O3 -o build/lab/l2_retention scripts/lab_l2_retention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t clk() { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
template <bool NT>
__device__ __forceinline__ float read_tile(const float* W, int64_t ld, int64_t row0, int64_t col0, int lane, int wave) {
    const float* p = W + (row0 + wave * 16) * ld + col0 + lane * 4;
    f32x4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = NT ? __builtin_nontemporal_load((const f32x4*)(p + r * ld)) : *(const f32x4*)(p + r * ld);
    float m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m += v[r][0] + v[r][1] + v[r][2] + v[r][3];
    return m;
}
template <bool STREAM_NT>
__global__ __launch_bounds__(512) void probe(const float* W, int64_t ld, int ncol, int stream_tiles, int ntiles, float* sink, uint64_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x, G = gridDim.x;
    auto origin = [&](int t, int64_t& r0, int64_t& c0) { r0 = (int64_t)(t / ncol) * 128; c0 = (int64_t)(t % ncol) * 256; };
    int64_t r0, c0;
    float acc = 0;
    origin(b, r0, c0);
    acc += read_tile<false>(W, ld, r0, c0, lane, wave);                      // tile A: default policy
    for (int s = 1; s <= stream_tiles; ++s) {                               // the stream: tiles b + s * G
        int64_t r1, c1; origin((b + s * G) % ntiles, r1, c1);
        acc += read_tile<STREAM_NT>(W, ld, r1, c1, lane, wave);
    }
    __syncthreads();
    uint64_t t0 = clk();
    acc += read_tile<true>(W, ld, r0, c0, lane, wave);                       // re-read A
    __syncthreads();
    uint64_t t1 = clk();
    int64_t r2, c2; origin((b + (stream_tiles + 1) * G) % ntiles, r2, c2);
    acc += read_tile<true>(W, ld, r2, c2, lane, wave);                       // a tile nobody has read
    __syncthreads();
    uint64_t t2 = clk();
    if (threadIdx.x == 0) { out[2 * b] = t1 - t0; out[2 * b + 1] = t2 - t1; }
    if (acc == 12345.678f) sink[0] = acc;
}
int main() {
    const int64_t K = 8192, N = 11008; const int ncol = N / 256, ntiles = (K / 128) * ncol, G = 256;
    float* W; hipMalloc(&W, K * N * 4); hipMemset(W, 0, K * N * 4);
    float* sink; hipMalloc(&sink, 4); uint64_t* out; hipMalloc(&out, G * 16);
    std::vector<uint64_t> h(2 * G);
    for (int nt = 0; nt < 2; ++nt) for (int s : {0, 1, 2, 3, 4, 6}) {
        for (int rep = 0; rep < 3; ++rep) {
            if (nt) hipLaunchKernelGGL(probe<true>, dim3(G), dim3(512), 0, 0, W, N, ncol, s, ntiles, sink, out);
            else hipLaunchKernelGGL(probe<false>, dim3(G), dim3(512), 0, 0, W, N, ncol, s, ntiles, sink, out);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, G * 16, hipMemcpyDeviceToHost);
        std::vector<double> a, c; for (int i = 0; i < G; ++i) { a.push_back(h[2 * i] / 100.0); c.push_back(h[2 * i + 1] / 100.0); }
        std::sort(a.begin(), a.end()); std::sort(c.begin(), c.end());
        printf("stream of %d tiles (%s): re-read of tile A median %.2f us (p90 %.2f)   untouched tile median %.2f us (p90 %.2f)\n", s, nt ? "nt" : "default",
               a[G / 2], a[G * 9 / 10], c[G / 2], c[G * 9 / 10]);
    }
    return 0;
}
