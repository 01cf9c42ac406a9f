// A1 (rtn.py:54-109) with strategy = tensor for MANY weight tensors in three launches.
//
// A model like gemma-3-270m has 126 small MatMul weights (0.4-2.6 M parameters); per-tensor int8 RTN of one of them is
// three launches of a few microseconds of work each, so a loop over `oq_rtn_quantize_f32` is launch-bound (31 us per
// matrix, 3.9 ms for the model).  Here blockIdx.y walks the tensors of a device descriptor table:
//   1. rtn_many_partial   per (tensor, slice) min / max                         (utils.py:60-61)
//   2. rtn_many_finalize  one wave per tensor: fold, clip ratio, zero, qparams   (utils.py:63-69, 242-299)
//   3. rtn_many_quantize  q = clip(int32(rint(x / s)) + zp), one byte per value  (utils.py:72-79)
// The first pass uses the default cache policy on purpose: tensors this small are still in the Infinity Cache when the
// third pass reads them again.  Results are the same bits as oq_rtn_quantize_f32(strategy = tensor) for every tensor.
#include "oq_common.hpp"

namespace oq {

struct RtnTensorDesc {   // mirrors oq_rtn_tensor_desc (include/oq_hip.h)
    const float* w;
    int64_t count;
    uint8_t* q;
    float* scale;
    uint8_t* zp;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kManyBlock = 256;

__global__ __launch_bounds__(kManyBlock) void rtn_many_partial(const RtnTensorDesc* desc, float* partial /* [n][gridDim.x][2] */) {
    __shared__ float s_mn[kManyBlock / 64], s_mx[kManyBlock / 64];
    const RtnTensorDesc d = desc[blockIdx.y];
    const float* x = d.w;
    const int64_t count = d.count;
    int64_t head = static_cast<int64_t>(((16 - reinterpret_cast<uintptr_t>(x) % 16) % 16) / 4);   // peel to 16-byte alignment
    if (head > count) head = count;
    const int64_t nvec = (count - head) / 4;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + head);
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    float mn = INFINITY, mx = -INFINITY;
    int64_t i = tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {   // four independent 16-byte loads in flight per lane
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = xv[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mn = nmin(nmin(nmin(mn, a[u].x), nmin(a[u].y, a[u].z)), a[u].w);
            mx = nmax(nmax(nmax(mx, a[u].x), nmax(a[u].y, a[u].z)), a[u].w);
        }
    }
    for (; i < nvec; i += stride) {
        const f32x4 a = xv[i];
        mn = nmin(nmin(mn, a.x), nmin(nmin(a.y, a.z), a.w));
        mx = nmax(nmax(mx, a.x), nmax(nmax(a.y, a.z), a.w));
    }
    for (int64_t j = tid; j < head; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    for (int64_t j = head + nvec * 4 + tid; j < count; j += stride) { mn = nmin(mn, x[j]); mx = nmax(mx, x[j]); }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kManyBlock / 64; ++w) { mn = nmin(mn, s_mn[w]); mx = nmax(mx, s_mx[w]); }
        float* o = partial + (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 2;
        o[0] = mn;
        o[1] = mx;
    }
}

__global__ __launch_bounds__(64) void rtn_many_finalize(const RtnTensorDesc* desc, const float* partial, int slices, QGrid grid) {
    const float* p = partial + static_cast<int64_t>(blockIdx.x) * slices * 2;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < slices; i += 64) {
        mn = nmin(mn, p[2 * i]);
        mx = nmax(mx, p[2 * i + 1]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (threadIdx.x != 0) return;
    const QParam q = qparam_from_minmax(mn, mx, grid);
    const RtnTensorDesc d = desc[blockIdx.x];
    d.scale[0] = q.scale;
    d.zp[0] = static_cast<uint8_t>(q.zp);
}

__global__ __launch_bounds__(kManyBlock) void rtn_many_quantize(const RtnTensorDesc* desc, QGrid grid, int32_t zp_signed) {
    const RtnTensorDesc d = desc[blockIdx.y];
    const float scale = d.scale[0];
    const int32_t zp = zp_signed ? static_cast<int32_t>(static_cast<int8_t>(d.zp[0])) : static_cast<int32_t>(d.zp[0]);
    const int32_t qmin = grid.qmin, qmax = grid.qmax;
    const float* x = d.w;
    uint8_t* q = d.q;
    const int64_t count = d.count;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(q) & 3u) == 0;
    const int64_t nvec = vec ? count / 4 : 0;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x);
    uint32_t* qv = reinterpret_cast<uint32_t*>(q);
    for (int64_t i = tid; i < nvec; i += stride) {
        const f32x4 a = xv[i];
        const uint32_t b0 = static_cast<uint8_t>(quantize_one(a.x, scale, zp, qmin, qmax));   // signed levels: two's complement byte
        const uint32_t b1 = static_cast<uint8_t>(quantize_one(a.y, scale, zp, qmin, qmax));
        const uint32_t b2 = static_cast<uint8_t>(quantize_one(a.z, scale, zp, qmin, qmax));
        const uint32_t b3 = static_cast<uint8_t>(quantize_one(a.w, scale, zp, qmin, qmax));
        qv[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    for (int64_t j = nvec * 4 + tid; j < count; j += stride) q[j] = static_cast<uint8_t>(quantize_one(x[j], scale, zp, qmin, qmax));
}

static int many_slices_for(int64_t n) {   // ~4096 blocks in total, 2..64 slices per tensor
    int64_t s = 4096 / (n > 0 ? n : 1);
    if (s < 2) s = 2;
    if (s > 64) s = 64;
    return static_cast<int>(s);
}

}  // namespace oq

extern "C" {

using namespace oq;

size_t oq_rtn_tensor_many_workspace_bytes(int64_t n) {
    return n > 0 ? static_cast<size_t>(n) * many_slices_for(n) * 2 * sizeof(float) + 256 : 256;
}

int32_t oq_rtn_tensor_many_f32(const void* desc, int64_t n, int32_t qtype, int32_t symmetric, int32_t reduce_range, float clip_ratio,
                               void* workspace, size_t workspace_bytes, void* stream) {
    OQ_REQUIRE(desc != nullptr && n > 0 && n <= 65535, OQ_ERR_INVALID_ARGUMENT, "oq_rtn_tensor_many_f32: bad argument (1 <= n <= 65535)");
    OQ_REQUIRE(sizeof(RtnTensorDesc) == sizeof(oq_rtn_tensor_desc), OQ_ERR_INVALID_ARGUMENT, "oq_rtn_tensor_many_f32: descriptor layout");
    QGrid grid;
    int32_t st = make_grid(qtype, symmetric, reduce_range, clip_ratio, &grid);
    if (st != OQ_OK) return st;
    OQ_REQUIRE(grid.bits <= 8, OQ_ERR_UNSUPPORTED, "oq_rtn_tensor_many_f32: 4- and 8-bit types only (one byte per value)");
    const size_t need = oq_rtn_tensor_many_workspace_bytes(n);
    OQ_REQUIRE(workspace != nullptr && workspace_bytes >= need, OQ_ERR_WORKSPACE, "oq_rtn_tensor_many_f32: workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
    const int slices = many_slices_for(n);
    hipStream_t s = as_stream(stream);
    const RtnTensorDesc* d = static_cast<const RtnTensorDesc*>(desc);
    float* partial = static_cast<float*>(workspace);
    const dim3 grid2(static_cast<uint32_t>(slices), static_cast<uint32_t>(n));
    hipLaunchKernelGGL(rtn_many_partial, grid2, dim3(kManyBlock), 0, s, d, partial);
    st = check_launch("rtn_many_partial");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(rtn_many_finalize, dim3(static_cast<uint32_t>(n)), dim3(64), 0, s, d, partial, slices, grid);
    st = check_launch("rtn_many_finalize");
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(rtn_many_quantize, grid2, dim3(kManyBlock), 0, s, d, grid, (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0);
    return check_launch("rtn_many_quantize");
}

}  // extern "C"
