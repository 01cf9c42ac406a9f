// Q1 / K1 / K2 / A3 and the wire-format packers as standalone gfx950 kernels.  These are the small
// building blocks the reference calls by name (utils.py:72-137, :242-299; rtn.py:112-138;
// qrules/_common.py:96-121; _pack.py:8-22); the bulk RTN path fuses them in rtn.hip instead.
#include "oq_common.hpp"
#include "row_params.hpp"

namespace oq {

__global__ void qparams_kernel(const float* rmin, const float* rmax, int64_t count, QGrid grid, float* scale,
                               int32_t* zp) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // utils.py:242 takes ranges that already include zero and the clip ratio: no R1 tail here.
    const QParam p = qparam_from_range(rmin[i], rmax[i], grid);
    scale[i] = p.scale;
    zp[i] = p.zp;
}

// utils.py:242-299 evaluated in float64 (what NumPy does for float64 ranges); scale leaves as fp32.
__global__ void qparams_kernel_f64(const double* rmin, const double* rmax, int64_t count, QGrid grid, float* scale,
                                   int32_t* zp) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double lo = rmin[i], hi = rmax[i];
    if (grid.symmetric) {
        const double amax = nmax(fabs(lo), fabs(hi));
        double s = amax / grid.levels;
        if (s < DBL_MIN) s = 1.0;
        scale[i] = static_cast<float>(s);
        zp[i] = grid.zero;
    } else {
        double s = (hi - lo) / static_cast<double>(grid.qmax - grid.qmin);
        if (s < DBL_MIN) s = 1.0;
        double z = static_cast<double>(grid.qmin) - lo / s;
        z = nmin(nmax(z, static_cast<double>(grid.qmin)), static_cast<double>(grid.qmax));
        scale[i] = static_cast<float>(s);
        zp[i] = static_cast<int32_t>(rint(z));
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void quantize_kernel(const float* x, int64_t R, int64_t C, int64_t ldx,
                                                       const float* scale, const int32_t* zp, ParamIndex pi,
                                                       int64_t qmin, int64_t qmax, OutT* q) {
    const int64_t total = R * C;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t r = t / C, c = t - r * C;
        const int64_t p = pi(r, c);
        if constexpr (sizeof(OutT) == 1) {
            q[t] = static_cast<OutT>(quantize_one(x[r * ldx + c], scale[p], zp[p], static_cast<int32_t>(qmin),
                                                  static_cast<int32_t>(qmax)));
        } else {
            // 32-bit containers: int32(rint(x / s)) + zp is evaluated in int64 before the clamp, which is
            // what NumPy does for uint32 zero points and is equivalent for int32 ones inside the range.
            const int64_t v = static_cast<int64_t>(static_cast<int32_t>(rintf(x[r * ldx + c] / scale[p]))) + zp[p];
            q[t] = static_cast<OutT>(v < qmin ? qmin : (v > qmax ? qmax : v));
        }
    }
}

template <typename InT, typename ZpT = int32_t>
__global__ __launch_bounds__(256) void dequantize_kernel(const InT* q, int64_t R, int64_t C, const float* scale,
                                                         const ZpT* zp, ParamIndex pi, float* out, int64_t ldo) {
    const int64_t total = R * C;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t r = t / C, c = t - r * C;
        const int64_t p = pi(r, c);
        // utils.py:130-132: both operands go to fp32 first (exact for |q| < 2^24; int32 rounds like astype)
        out[r * ldo + c] = (static_cast<float>(q[t]) - static_cast<float>(zp[p])) * scale[p];
    }
}

// ---- one-byte types, vector-aligned rows: the fast path of K1 / K2 -------------------------------------------------------
// The generic kernels above pay two runtime 64-bit divisions per element (t / C, r / row_div), load 4 bytes per lane
// and fetch (scale, zp) per element -- 0.6 TB/s with group parameters, 2.1-2.4 TB/s with one (scale, zp), on the
// 4096 x 11008 matrix.  Here a thread owns 4 neighbouring columns of a chunk of 32 rows: 16-byte loads / 4-byte stores
// (or the reverse), 8 of them in flight, the parameter row `(r / row_div) * row_stride` advanced by counting, and the
// thread's four (scale, zp) pairs reloaded only when that row changes (once per group of rows).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTileRows = 32;

template <bool QUANT>
__global__ __launch_bounds__(256) void tile_kernel(const float* __restrict__ xin, const uint8_t* __restrict__ qin, int64_t R, int64_t C, int64_t ldx,
                                                   const float* __restrict__ scale, const int32_t* __restrict__ zp, ParamIndex pi,
                                                   int32_t qmin, int32_t qmax, int32_t is_signed, uint8_t* __restrict__ qout,
                                                   float* __restrict__ xout) {
    const int64_t c = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (c >= C) return;                                         // C % 4 == 0: a thread's four columns are all inside or all outside
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * kTileRows;
    const int64_t r1 = r0 + kTileRows < R ? r0 + kTileRows : R;
    float s[4];
    int32_t z[4];
    auto load_params = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = base + (c + u) * pi.col_stride;
            s[u] = scale[p];
            z[u] = zp[p];
        }
    };
    TileParamCursor<decltype(load_params)> params(r0, r1, pi, load_params);   // row_params.hpp: never loads behind the tile's last row
    params.start();
    for (int64_t r = r0; r < r1; r += 8) {
        f32x4 v[8];
        uint32_t b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t rr = r + u < r1 ? r + u : r1 - 1;
            if constexpr (QUANT) v[u] = *reinterpret_cast<const f32x4*>(xin + rr * ldx + c);
            else b[u] = *reinterpret_cast<const uint32_t*>(qin + rr * C + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r + u < r1) {
                if constexpr (QUANT) {
                    uint32_t o = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o |= static_cast<uint32_t>(static_cast<uint8_t>(quantize_one(v[u][e], s[e], z[e], qmin, qmax))) << (8 * e);
                    *reinterpret_cast<uint32_t*>(qout + (r + u) * C + c) = o;
                } else {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t byte = (b[u] >> (8 * e)) & 0xffu;
                        const int32_t qi = is_signed ? static_cast<int32_t>(static_cast<int8_t>(byte)) : static_cast<int32_t>(byte);
                        o[e] = dequantize_one(qi, s[e], z[e]);
                    }
                    *reinterpret_cast<f32x4*>(xout + (r + u) * ldx + c) = o;
                }
                params.row_done(r + u);   // uniform over the block
            }
        }
    }
}

static bool tile_eligible(const void* a, const void* b, int64_t R, int64_t C, int64_t ld) {
    return C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15u) == 0 && (reinterpret_cast<uintptr_t>(b) & 15u) == 0 &&
           ceil_div(R, kTileRows) <= 65535;
}

__global__ void bias_kernel(const float* bias, int64_t n, const float* w_scale, int64_t n_w, float x_scale,
                            int32_t* q, float* bscale) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = w_scale[n_w == 1 ? 0 : i] * x_scale;  // rtn.py:129 (fp32 product)
    bscale[i] = s;
    // rtn.py:130-137 -> utils.py:72-79 with QInt32 full range, zero point 0; the int32 cast of an
    // out-of-range quotient saturates here (NumPy's is platform-defined there).
    q[i] = static_cast<int32_t>(rintf(bias[i] / s));
}

__global__ void pack_zp_u4_kernel(const uint8_t* zp, int64_t N, int64_t blocks, uint8_t* out) {
    const int64_t half = (blocks + 1) / 2;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= N * half) return;
    const int64_t n = t / half, j = t - n * half;
    const uint32_t lo = zp[n * blocks + 2 * j] & 0x0fu;
    const uint32_t hi = (2 * j + 1 < blocks) ? (zp[n * blocks + 2 * j + 1] & 0x0fu) : 0x8u;  // _common.py:106-109
    out[t] = static_cast<uint8_t>(lo | (hi << 4));
}

// _pack.py:8-22: out[t] = v[2t] & 15 | (v[2t+1] & 15) << 4, zero pad.  A thread takes 16 values (one 16-byte load) and
// writes 8 bytes when the pointers allow it; the ragged end and unaligned calls go one output byte per thread.
__global__ void pack_nibbles_kernel(const uint8_t* v, int64_t count, uint8_t* out, int64_t vec16) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < vec16) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x4 a = *reinterpret_cast<const u32x4*>(v + 16 * t);
        u32x2 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t x = a[2 * h + e] & 0x0f0f0f0fu;              // four values, one per byte
                const uint32_t y = (x | (x >> 4)) & 0x00ff00ffu;            // bytes 0 and 2: neighbours joined
                w |= ((y & 0xffu) | ((y >> 8) & 0xff00u)) << (16 * e);
            }
            o[h] = w;
        }
        *reinterpret_cast<u32x2*>(out + 8 * t) = o;
        return;
    }
    const int64_t j = vec16 * 8 + (t - vec16);                              // output byte index behind the vector part
    if (j >= (count + 1) / 2) return;
    const uint32_t lo = v[2 * j] & 0x0fu;
    const uint32_t hi = (2 * j + 1 < count) ? (v[2 * j + 1] & 0x0fu) : 0u;  // _pack.py:15-17 zero pad
    out[j] = static_cast<uint8_t>(lo | (hi << 4));
}

// qrules/_common.py:72-87: B[n][kg][j] = q[kg*g + 2j][n] | q[kg*g + 2j + 1][n] << 4 (4 bits) or q[kg*g + j][n] (8 bits) from the
// [K, N] byte array the algorithms return.  A block transposes a tile of 64 rows x 64 columns through LDS: coalesced row
// reads, and per output column 32 (4 bits) / 64 (8 bits) consecutive blob bytes.  g % 64 == 0 or 64 % g == 0 (g >= 16, even).
__global__ __launch_bounds__(256) void pack_blob_kernel(const uint8_t* __restrict__ q, int64_t K, int64_t N, int64_t g, int32_t bits,
                                                        uint8_t* __restrict__ out) {
    __shared__ uint8_t t[64][65];
    const int64_t k0 = static_cast<int64_t>(blockIdx.y) * 64, n0 = static_cast<int64_t>(blockIdx.x) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int64_t k = k0 + r, n = n0 + tx;
        t[r][tx] = (k < K && n < N) ? q[k * N + n] : 0;
    }
    __syncthreads();
    const int64_t kgroups = K / g;
    if (bits == 4) {   // thread = (column c, byte pair index j2 in 0..31 of this 64-row slab)
        for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
            const int c = idx >> 5, j = idx & 31;
            const int64_t n = n0 + c, k = k0 + 2 * j;
            if (n < N && k < K) {
                const uint8_t b = static_cast<uint8_t>((t[2 * j][c] & 0x0f) | ((t[2 * j + 1][c] & 0x0f) << 4));
                out[(n * kgroups + k / g) * (g / 2) + (k % g) / 2] = b;
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
            const int c = idx >> 6, j = idx & 63;
            const int64_t n = n0 + c, k = k0 + j;
            if (n < N && k < K) out[(n * kgroups + k / g) * g + k % g] = t[j][c];
        }
    }
}

static uint32_t grid_for(int64_t work, int block = 256) {
    const int64_t b = ceil_div(work, block);
    return static_cast<uint32_t>(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace oq

extern "C" {

using namespace oq;

int32_t oq_qparams_f32(const float* rmin, const float* rmax, int64_t count, int32_t qtype, int32_t symmetric,
                       int32_t reduce_range, float* scale_out, int32_t* zp_out, void* stream) {
    OQ_REQUIRE(rmin && rmax && scale_out && zp_out && count_ok(count, kMaxThreads), OQ_ERR_INVALID_ARGUMENT, "oq_qparams_f32: bad argument");
    QGrid grid;
    const int32_t st = make_grid(qtype, symmetric, reduce_range, 1.0f, &grid);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(qparams_kernel, dim3(static_cast<uint32_t>(ceil_div(count, 256))), dim3(256), 0, as_stream(stream),
                       rmin, rmax, count, grid, scale_out, zp_out);
    return check_launch("qparams_kernel");
}

int32_t oq_qparams_f64(const double* rmin, const double* rmax, int64_t count, int32_t qtype, int32_t symmetric,
                       int32_t reduce_range, float* scale_out, int32_t* zp_out, void* stream) {
    OQ_REQUIRE(rmin && rmax && scale_out && zp_out && count_ok(count, kMaxThreads), OQ_ERR_INVALID_ARGUMENT, "oq_qparams_f64: bad argument");
    QGrid grid;
    const int32_t st = make_grid(qtype, symmetric, reduce_range, 1.0f, &grid);
    if (st != OQ_OK) return st;
    hipLaunchKernelGGL(qparams_kernel_f64, dim3(static_cast<uint32_t>(ceil_div(count, 256))), dim3(256), 0, as_stream(stream),
                       rmin, rmax, count, grid, scale_out, zp_out);
    return check_launch("qparams_kernel_f64");
}

int32_t oq_quantize_f32(const float* x, int64_t R, int64_t C, int64_t ldx, const float* scale, const int32_t* zp,
                        int64_t row_div, int64_t row_stride, int64_t col_stride, int32_t qtype, int32_t symmetric,
                        int32_t reduce_range, void* q_out, void* stream) {
    OQ_REQUIRE(x && scale && zp && q_out && matrix_ok(R, C, ldx) && row_div > 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_quantize_f32: bad argument");
    int64_t qmin, qmax;
    OQ_REQUIRE(qrange_host(qtype, symmetric, reduce_range, &qmin, &qmax), OQ_ERR_INVALID_ARGUMENT,
               "oq_quantize_f32: unknown quantization type %d", qtype);
    const ParamIndex pi{row_div, row_stride, col_stride};
    const dim3 grid(grid_for(R * C)), block(256);
    hipStream_t s = as_stream(stream);
    if (qtype <= OQ_UINT8 && tile_eligible(x, q_out, R, C, ldx)) {
        const dim3 tgrid(static_cast<uint32_t>(ceil_div(C, 1024)), static_cast<uint32_t>(ceil_div(R, kTileRows)));
        hipLaunchKernelGGL(tile_kernel<true>, tgrid, block, 0, s, x, static_cast<const uint8_t*>(nullptr), R, C, ldx, scale, zp, pi,
                           static_cast<int32_t>(qmin), static_cast<int32_t>(qmax), 0, static_cast<uint8_t*>(q_out), static_cast<float*>(nullptr));
        return check_launch("quantize tile_kernel");
    }
    switch (qtype) {
        case OQ_INT4: case OQ_INT8:
            hipLaunchKernelGGL(quantize_kernel<int8_t>, grid, block, 0, s, x, R, C, ldx, scale, zp, pi, qmin, qmax,
                               static_cast<int8_t*>(q_out));
            break;
        case OQ_UINT4: case OQ_UINT8:
            hipLaunchKernelGGL(quantize_kernel<uint8_t>, grid, block, 0, s, x, R, C, ldx, scale, zp, pi, qmin, qmax,
                               static_cast<uint8_t*>(q_out));
            break;
        case OQ_INT32:
            hipLaunchKernelGGL(quantize_kernel<int32_t>, grid, block, 0, s, x, R, C, ldx, scale, zp, pi, qmin, qmax,
                               static_cast<int32_t*>(q_out));
            break;
        default:
            hipLaunchKernelGGL(quantize_kernel<uint32_t>, grid, block, 0, s, x, R, C, ldx, scale, zp, pi, qmin, qmax,
                               static_cast<uint32_t*>(q_out));
            break;
    }
    return check_launch("quantize_kernel");
}

int32_t oq_dequantize_f32(const void* q, int64_t R, int64_t C, int32_t qtype, const float* scale, const int32_t* zp,
                          int64_t row_div, int64_t row_stride, int64_t col_stride, float* x_out, int64_t ldo,
                          void* stream) {
    OQ_REQUIRE(q && scale && zp && x_out && matrix_ok(R, C, ldo) && row_div > 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_dequantize_f32: bad argument");
    OQ_REQUIRE(qtype >= OQ_INT4 && qtype <= OQ_UINT32, OQ_ERR_INVALID_ARGUMENT, "oq_dequantize_f32: unknown type %d", qtype);
    const ParamIndex pi{row_div, row_stride, col_stride};
    const dim3 grid(grid_for(R * C)), block(256);
    hipStream_t s = as_stream(stream);
    if (qtype <= OQ_UINT8 && tile_eligible(x_out, q, R, C, ldo)) {
        const dim3 tgrid(static_cast<uint32_t>(ceil_div(C, 1024)), static_cast<uint32_t>(ceil_div(R, kTileRows)));
        hipLaunchKernelGGL(tile_kernel<false>, tgrid, block, 0, s, static_cast<const float*>(nullptr), static_cast<const uint8_t*>(q), R, C, ldo,
                           scale, zp, pi, 0, 0, (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0, static_cast<uint8_t*>(nullptr), x_out);
        return check_launch("dequantize tile_kernel");
    }
    switch (qtype) {
        case OQ_INT4: case OQ_INT8:
            hipLaunchKernelGGL(dequantize_kernel<int8_t>, grid, block, 0, s, static_cast<const int8_t*>(q), R, C, scale, zp, pi, x_out, ldo);
            break;
        case OQ_UINT4: case OQ_UINT8:
            hipLaunchKernelGGL(dequantize_kernel<uint8_t>, grid, block, 0, s, static_cast<const uint8_t*>(q), R, C, scale, zp, pi, x_out, ldo);
            break;
        case OQ_INT32:
            hipLaunchKernelGGL(dequantize_kernel<int32_t>, grid, block, 0, s, static_cast<const int32_t*>(q), R, C, scale, zp, pi, x_out, ldo);
            break;
        default:
            hipLaunchKernelGGL(dequantize_kernel<uint32_t>, grid, block, 0, s, static_cast<const uint32_t*>(q), R, C, scale, zp, pi, x_out, ldo);
            break;
    }
    return check_launch("dequantize_kernel");
}

int32_t oq_dequantize_fzp_f32(const void* q, int64_t R, int64_t C, int32_t qtype, const float* scale, const float* zp,
                              int64_t row_div, int64_t row_stride, int64_t col_stride, float* x_out, int64_t ldo,
                              void* stream) {
    OQ_REQUIRE(q && scale && zp && x_out && matrix_ok(R, C, ldo) && row_div > 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_dequantize_fzp_f32: bad argument");
    OQ_REQUIRE(qtype >= OQ_INT4 && qtype <= OQ_UINT8, OQ_ERR_UNSUPPORTED, "oq_dequantize_fzp_f32: 4- and 8-bit containers only, got type %d", qtype);
    const ParamIndex pi{row_div, row_stride, col_stride};
    const dim3 grid(grid_for(R * C)), block(256);
    hipStream_t s = as_stream(stream);
    if (qtype == OQ_INT4 || qtype == OQ_INT8)
        hipLaunchKernelGGL((dequantize_kernel<int8_t, float>), grid, block, 0, s, static_cast<const int8_t*>(q), R, C, scale, zp, pi, x_out, ldo);
    else
        hipLaunchKernelGGL((dequantize_kernel<uint8_t, float>), grid, block, 0, s, static_cast<const uint8_t*>(q), R, C, scale, zp, pi, x_out, ldo);
    return check_launch("dequantize_kernel (float zero points)");
}

int32_t oq_quantize_bias_f32(const float* bias, int64_t n, const float* w_scale, int64_t n_w_scale, float x_scale,
                             int32_t* q_out, float* bias_scale_out, void* stream) {
    OQ_REQUIRE(bias && w_scale && q_out && bias_scale_out && count_ok(n, kMaxThreads), OQ_ERR_INVALID_ARGUMENT, "oq_quantize_bias_f32: bad argument");
    OQ_REQUIRE(n_w_scale == 1 || n_w_scale == n, OQ_ERR_INVALID_ARGUMENT,
               "oq_quantize_bias_f32: weight scale must have 1 or %lld entries, got %lld", (long long)n, (long long)n_w_scale);
    hipLaunchKernelGGL(bias_kernel, dim3(static_cast<uint32_t>(ceil_div(n, 256))), dim3(256), 0, as_stream(stream), bias, n,
                       w_scale, n_w_scale, x_scale, q_out, bias_scale_out);
    return check_launch("bias_kernel");
}

int32_t oq_pack_zero_points_u4(const uint8_t* zp, int64_t N, int64_t blocks, uint8_t* out, void* stream) {
    OQ_REQUIRE(zp && out && extent_ok(N) && extent_ok(blocks) && N * ((blocks + 1) / 2) <= kMaxThreads, OQ_ERR_INVALID_ARGUMENT, "oq_pack_zero_points_u4: bad argument");
    const int64_t work = N * ((blocks + 1) / 2);
    hipLaunchKernelGGL(pack_zp_u4_kernel, dim3(static_cast<uint32_t>(ceil_div(work, 256))), dim3(256), 0, as_stream(stream), zp,
                       N, blocks, out);
    return check_launch("pack_zp_u4_kernel");
}

int32_t oq_pack_matmul_nbits(const void* q, int64_t K, int64_t N, int64_t group_size, int32_t bits, uint8_t* out, void* stream) {
    OQ_REQUIRE(q && out && matrix_ok(K, N, N), OQ_ERR_INVALID_ARGUMENT, "oq_pack_matmul_nbits: bad argument");
    OQ_REQUIRE(bits == 4 || bits == 8, OQ_ERR_UNSUPPORTED, "oq_pack_matmul_nbits: 4- or 8-bit values only");
    OQ_REQUIRE(group_size >= 2 && group_size % 2 == 0 && K % group_size == 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_pack_matmul_nbits: group_size must be even and divide K (%lld, %lld)", (long long)group_size, (long long)K);
    const dim3 grid(static_cast<uint32_t>(ceil_div(N, 64)), static_cast<uint32_t>(ceil_div(K, 64)));
    OQ_REQUIRE(grid.y <= 65535, OQ_ERR_UNSUPPORTED, "oq_pack_matmul_nbits: K up to 4 194 240");
    hipLaunchKernelGGL(pack_blob_kernel, grid, dim3(256), 0, as_stream(stream), static_cast<const uint8_t*>(q), K, N, group_size, bits, out);
    return check_launch("pack_blob_kernel");
}

int32_t oq_pack_nibbles(const void* values, int64_t count, uint8_t* out, void* stream) {
    OQ_REQUIRE(values && out && count_ok(count, kMaxThreads), OQ_ERR_INVALID_ARGUMENT, "oq_pack_nibbles: bad argument");
    const bool aligned = (reinterpret_cast<uintptr_t>(values) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
    const int64_t vec16 = aligned ? count / 16 : 0;
    const int64_t threads = vec16 + ((count + 1) / 2 - vec16 * 8);
    hipLaunchKernelGGL(pack_nibbles_kernel, dim3(static_cast<uint32_t>(ceil_div(threads, 256))), dim3(256), 0,
                       as_stream(stream), static_cast<const uint8_t*>(values), count, out, vec16);
    return check_launch("pack_nibbles_kernel");
}

}  // extern "C"
