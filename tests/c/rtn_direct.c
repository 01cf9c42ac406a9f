/* A torch-free consumer of the C ABI: plain C99 + the HIP runtime's C API.  Quantizes a K x N matrix of a fixed
 * pseudo-random sequence with oq_rtn_quantize_f32 (uint4, group 128, both layouts) and prints FNV-1a digests of the
 * outputs; tests/test_library_abi.py::test_c_program_quantizes_on_the_gpu compares them with the oracle on the same data.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include rtn_direct.c -L <lib dir> -loq_hip -L /opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "oq_hip.h"

static uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define OQ_OK_(call) do { int32_t s_ = (call); if (s_ != 0) { fprintf(stderr, "%s: %d %s\n", #call, (int)s_, oq_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    const int64_t K = argc > 1 ? atoll(argv[1]) : 512, N = argc > 2 ? atoll(argv[2]) : 384, g = 128;
    const int64_t groups = K / g * N;
    float* w = (float*)malloc((size_t)(K * N) * sizeof(float));
    uint32_t x = 12345u;                                   /* LCG: the Python side generates the same values */
    for (int64_t i = 0; i < K * N; ++i) {
        x = x * 1664525u + 1013904223u;
        w[i] = ((float)(x >> 8) / 16777216.0f - 0.5f) * 4.0f;
    }
    float *dw, *dscale;
    uint8_t *dq, *dzp, *dblob;
    void* ws;
    HIP_OK(hipMalloc((void**)&dw, (size_t)(K * N) * 4));
    HIP_OK(hipMalloc((void**)&dq, (size_t)(K * N)));
    HIP_OK(hipMalloc((void**)&dblob, (size_t)(K * N) / 2));
    HIP_OK(hipMalloc((void**)&dscale, (size_t)groups * 4));
    HIP_OK(hipMalloc((void**)&dzp, (size_t)groups));
    const size_t wsb = oq_rtn_workspace_bytes(K, N, OQ_GROUP, g, 0) + 256;
    HIP_OK(hipMalloc(&ws, wsb));
    HIP_OK(hipMemcpy(dw, w, (size_t)(K * N) * 4, hipMemcpyHostToDevice));
    uint8_t* q = (uint8_t*)malloc((size_t)(K * N));
    float* scale = (float*)malloc((size_t)groups * 4);
    uint8_t* zp = (uint8_t*)malloc((size_t)groups);
    OQ_OK_(oq_rtn_quantize_f32(dw, K, N, N, OQ_UINT4, OQ_GROUP, g, 0, 0, 1.0f, 0, dq, dscale, dzp, OQ_LAYOUT_KN, ws, wsb, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(q, dq, (size_t)(K * N), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scale, dscale, (size_t)groups * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(zp, dzp, (size_t)groups, hipMemcpyDeviceToHost));
    printf("kn q=%016llx scale=%016llx zp=%016llx\n", (unsigned long long)fnv1a(q, (size_t)(K * N)),
           (unsigned long long)fnv1a(scale, (size_t)groups * 4), (unsigned long long)fnv1a(zp, (size_t)groups));
    OQ_OK_(oq_rtn_quantize_f32(dw, K, N, N, OQ_UINT4, OQ_GROUP, g, 0, 0, 1.0f, 0, dblob, dscale, dzp, OQ_LAYOUT_NBITS, ws, wsb, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(q, dblob, (size_t)(K * N) / 2, hipMemcpyDeviceToHost));
    printf("nbits blob=%016llx\n", (unsigned long long)fnv1a(q, (size_t)(K * N) / 2));
    /* a loud error: the NBITS layout with the tensor strategy is not a thing */
    const int32_t st = oq_rtn_quantize_f32(dw, K, N, N, OQ_UINT4, OQ_TENSOR, -1, 0, 0, 1.0f, 0, dblob, dscale, dzp, OQ_LAYOUT_NBITS, ws, wsb, NULL);
    printf("error status=%d text=%s\n", (int)st, oq_last_error());
    return 0;
}
