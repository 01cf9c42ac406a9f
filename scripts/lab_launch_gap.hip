// Lab (round 6): what lies between two dependent launches of one stream.  A kernel stamps the 100 MHz wall clock at the start and
// at the end of every workgroup; launched back to back, gap = (first start of launch i + 1) - (last end of launch i).
//   hipcc --offload-arch=gfx950 -O3 scripts/lab_launch_gap.hip -o build/lab/launch_gap && build/lab/launch_gap
// Variants: grid, block, dynamic LDS, bytes stored per workgroup (dirty lines for the end-of-kernel write-back), store policy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void gap_kernel(uint64_t* stamps, int launch, uint32_t* payload, int dwords_per_thread, int nt, int spin_us) {
    extern __shared__ unsigned char lds[];
    uint64_t t0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); t0 = wall_clock64(); }
    uint32_t* p = payload + (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x);
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (int i = 0; i < dwords_per_thread; ++i) {
        const uint32_t w = static_cast<uint32_t>(i + launch);
        uint32_t* q = p + i * stride;
        if (nt == 1) __builtin_nontemporal_store(w, q);
        else if (nt == 2) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(q), "v"(w) : "memory");
        else if (nt == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(q), "v"(w) : "memory");
        else if (nt == 4) asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(q), "v"(w) : "memory");
        else *q = w;
    }
    if (spin_us) {
        const uint64_t until = wall_clock64() + static_cast<uint64_t>(spin_us) * 100;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
    }
    if (dwords_per_thread < 0) lds[threadIdx.x] = 1;
    if (threadIdx.x == 0) {
        stamps[(static_cast<size_t>(launch) * gridDim.x + blockIdx.x) * 2] = t0;
        stamps[(static_cast<size_t>(launch) * gridDim.x + blockIdx.x) * 2 + 1] = wall_clock64();
    }
}

int main() {
    const int launches = 40;
    struct V { int grid, block, lds, dwords, nt, spin; const char* what; };
    const V vs[] = {
        {256, 512, 0, 0, 0, 0, "256 x 512, nothing"},
        {256, 512, 0, 0, 0, 20, "256 x 512, 20 us of sleep"},
        {256, 512, 144 * 1024, 0, 0, 20, "256 x 512, 144 KB of LDS, 20 us of sleep"},
        {2752, 512, 0, 0, 0, 0, "2752 x 512, nothing"},
        {2752, 512, 0, 0, 0, 2, "2752 x 512, 2 us of sleep each"},
        {256, 512, 0, 88, 1, 20, "256 x 512, 45 MB of nt stores, then 20 us of sleep"},
        {256, 512, 0, 88, 0, 20, "256 x 512, 45 MB of plain stores, then 20 us of sleep"},
        {256, 512, 144 * 1024, 88, 1, 0, "256 x 512, 144 KB LDS, 45 MB of nt stores at the very end"},
        {2752, 512, 0, 8, 1, 0, "2752 x 512, 45 MB of nt stores"},
        {256, 512, 0, 88, 2, 20, "256 x 512, 45 MB of sc1 stores, then 20 us of sleep"},
        {256, 512, 0, 88, 3, 20, "256 x 512, 45 MB of sc0 sc1 stores, then 20 us of sleep"},
        {256, 512, 0, 88, 4, 20, "256 x 512, 45 MB of sc1 nt stores, then 20 us of sleep"},
        {256, 512, 0, 88, 2, 0, "256 x 512, 45 MB of sc1 stores"},
        {256, 512, 0, 88, 1, 0, "256 x 512, 45 MB of nt stores"},
        {2752, 512, 0, 8, 2, 0, "2752 x 512, 45 MB of sc1 stores"},
        {2752, 512, 0, 4, 1, 0, "2752 x 512, 22 MB of nt stores"},
        {2752, 512, 0, 4, 2, 0, "2752 x 512, 22 MB of sc1 stores"},
        {2752, 512, 0, 1, 1, 0, "2752 x 512, 5.6 MB of nt stores"},
        {2752, 512, 0, 16, 1, 0, "2752 x 512, 90 MB of nt stores"},
    };
    uint64_t* d_stamps; uint32_t* d_payload;
    hipMalloc(&d_stamps, sizeof(uint64_t) * 2 * 2752 * launches);
    hipMalloc(&d_payload, sizeof(uint32_t) * 2752ull * 512 * 96);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    std::vector<uint64_t> h(2 * 2752 * launches);
    for (const V& v : vs) {
        hipMemset(d_stamps, 0, sizeof(uint64_t) * 2 * 2752 * launches);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(gap_kernel, dim3(v.grid), dim3(v.block), v.lds, 0, d_stamps, i, d_payload, v.dwords, v.nt, v.spin);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d_stamps, sizeof(uint64_t) * 2 * v.grid * launches, hipMemcpyDeviceToHost);
        std::vector<double> gaps, spans;
        for (int i = 5; i + 1 < launches; ++i) {
            uint64_t last_end = 0, first_start = ~0ull, this_first = ~0ull;
            for (int b = 0; b < v.grid; ++b) {
                last_end = std::max(last_end, h[(static_cast<size_t>(i) * v.grid + b) * 2 + 1]);
                this_first = std::min(this_first, h[(static_cast<size_t>(i) * v.grid + b) * 2]);
                first_start = std::min(first_start, h[(static_cast<size_t>(i + 1) * v.grid + b) * 2]);
            }
            gaps.push_back((static_cast<double>(first_start) - static_cast<double>(last_end)) / 100.0);
            spans.push_back((static_cast<double>(last_end) - static_cast<double>(this_first)) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
        printf("%-64s per launch %7.2f us   first start .. last end %7.2f us   gap to the next launch: median %5.2f  min %5.2f  max %5.2f us\n", v.what,
               ms * 1000.0 / launches, spans[spans.size() / 2], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
    return 0;
}
