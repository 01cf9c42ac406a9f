// Experiment harness for a persistent, software-pipelined form of the MatMulNBits RTN kernel (round 2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o gpurun_out/rtn_stream_bench scripts/rtn_stream_bench.hip \
//         -Lonnx_quantize_amd/lib -loq_hip -Wl,-rpath,'$ORIGIN/../onnx_quantize_amd/lib'
// Every variant is checked byte for byte against oq_rtn_quantize_f32 (the shipped kernel) before it is timed.
#include "../onnx_quantize_amd/csrc/oq_common.hpp"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace oq;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct StreamArgs {
    const float* W;
    int64_t K, N, ldw;
    int64_t kgroups;
    uint8_t* q;
    float* scale;
    uint8_t* zp;
    QGrid grid;
    uint32_t nstrips;      // ceil(N / 32)
    uint32_t kunits;       // ceil(kgroups / U)
    uint32_t nparts;       // partitions of the strips (8: one per XCD label)
    uint32_t waves_per_part;  // static schedule: persistent waves per partition
    uint32_t* queue;       // dynamic schedule: nparts counters + 1 done counter (all zero at launch, reset by the last wave)
    uint32_t total_waves;
    int32_t map;           // 0: strips fastest inside a partition, 1: k units fastest
    int32_t nt;            // bit 0 nt loads, bit 1 nt blob stores
};

// partition p of `n` items split into `parts` near-equal contiguous ranges
__device__ __forceinline__ uint32_t part_begin(uint32_t n, uint32_t parts, uint32_t p) { return static_cast<uint32_t>((static_cast<uint64_t>(n) * p) / parts); }

template <int U, bool COALESCE, int WPB>
__global__ __launch_bounds__(WPB * 64, 2) void rtn_stream(const StreamArgs a) {
    constexpr int LPR = 8, G = 128;
    static_assert(U == 2 || U == 4, "units of 2 or 4 k-groups");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane / LPR, cl = lane % LPR;
    const uint32_t part = blockIdx.x % a.nparts;
    const uint32_t s_begin = part_begin(a.nstrips, a.nparts, part), s_end = part_begin(a.nstrips, a.nparts, part + 1);
    const uint32_t pstrips = s_end - s_begin;
    const uint32_t punits = pstrips * a.kunits;
    const uint32_t wave_in_part = (blockIdx.x / a.nparts) * WPB + wave;

    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? (a.grid.bits == 4 ? 8 : 128) : 0;
    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
    const int64_t row_bytes = a.ldw * 4;

    // unit index inside the partition -> (strip, first k-group)
    auto unit_pos = [&](uint32_t r, uint32_t& strip, uint32_t& kg0) {
        uint32_t sl, ku;
        if (a.map == 0) { ku = r / pstrips; sl = r - ku * pstrips; }
        else { sl = r / a.kunits; ku = r - sl * a.kunits; }
        strip = s_begin + sl;
        kg0 = ku * U;
    };

    float bufA[16][4], bufB[16][4];
    auto load_tile = [&](float (&v)[16][4], uint32_t strip, uint32_t kg) {
        const int64_t strip0 = static_cast<int64_t>(strip) * 32;
        const bool col_ok = strip0 + cl * 4 < a.N;
        const char* base = reinterpret_cast<const char*>(a.W + static_cast<int64_t>(kg) * G * a.ldw + strip0);
        const uint32_t loff = static_cast<uint32_t>(h * 16 * a.ldw + (col_ok ? cl * 4 : a.N - 4 - strip0)) * 4u;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + r * row_bytes + loff));
            v[r][0] = u[0]; v[r][1] = u[1]; v[r][2] = u[2]; v[r][3] = u[3];
        }
        // pin the issue point: without it the scheduler moves the consumers of the OTHER buffer (and their vmcnt(0)) in
        // front of these loads, so the prefetch would start only after the previous tile has fully arrived
        __builtin_amdgcn_sched_barrier(0);
    };

    float acc_s[U], acc_z[U];
    // one tile: range -> parameters -> K1 -> blob stores; parameters kept for the unit's coalesced store
    auto process = [&](float (&v)[16][4], uint32_t strip, uint32_t kg, int t) {
        const int64_t c0 = static_cast<int64_t>(strip) * 32 + cl * 4;
        const bool col_ok = c0 < a.N;
        float mn[4], mx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mn[i] = mx[i] = v[0][i];
#pragma unroll
        for (int r = 1; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mn[i] = fminf(mn[i], v[r][i]);
                mx[i] = fmaxf(mx[i], v[r][i]);
            }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mn[i] = fminf(mn[i], __shfl_xor(mn[i], off, 64));
                mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], off, 64));
            }
        float sc[4], rinv[4], zpb[4], thr;
        {
            ColQ c = make_colq(qparam_from_minmax(mn[0], mx[0], a.grid), mn[0], mx[0], bias);
            sc[0] = c.scale; rinv[0] = c.rinv; zpb[0] = c.zpb; thr = c.thr;
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                c = make_colq(qparam_from_minmax(mn[i], mx[i], a.grid), mn[i], mx[i], bias);
                sc[i] = c.scale; rinv[i] = c.rinv; zpb[i] = c.zpb; thr = fminf(thr, c.thr);
            }
        }
        {
            const int j = h & 3;
            const float s = j == 0 ? sc[0] : j == 1 ? sc[1] : j == 2 ? sc[2] : sc[3];
            const float z = j == 0 ? zpb[0] : j == 1 ? zpb[1] : j == 2 ? zpb[2] : zpb[3];
            if constexpr (COALESCE) {
                acc_s[t] = s;
                acc_z[t] = z;
            } else {
                if (h < 4 && col_ok) {
                    const int64_t o = (c0 + j) * a.kgroups + kg;
                    a.scale[o] = s;
                    a.zp[o] = static_cast<uint8_t>(static_cast<int32_t>(z) - bias);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float f[4];
            bool unsafe = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tt = v[r][i] * rinv[i];
                const float k = rintf(tt);
                unsafe = unsafe || !(fabsf(tt - k) < thr);
                f[i] = __builtin_amdgcn_fmed3f(k + zpb[i], lo_b, hi_b);
            }
            if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    f[i] = static_cast<float>(quantize_one(v[r][i], sc[i], static_cast<int32_t>(zpb[i]) - bias, qmin, qmax) + bias);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[r][i] = f[i];
        }
        if (!col_ok) return;
        const uint32_t flip = bias ? 0x88888888u : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t words[2];
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                uint32_t ev = 0, od = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ev = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 8 + 2 * j][i], j, ev);
                    od = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 8 + 2 * j + 1][i], j, od);
                }
                words[wd] = (ev | (od << 4)) ^ flip;
            }
            u32x2* o = reinterpret_cast<u32x2*>(a.q + ((c0 + i) * a.kgroups + kg) * (G / 2) + h * 8);
            const u32x2 tv = {words[0], words[1]};
            if (a.nt & 2) __builtin_nontemporal_store(tv, o);
            else *o = tv;
        }
    };
    // the unit's parameters: lane sets 0-3 store U consecutive scales of column slot h, lane sets 4-7 the zero points
    auto flush = [&](uint32_t strip, uint32_t kg0) {
        if constexpr (COALESCE) {
            const int64_t c0 = static_cast<int64_t>(strip) * 32 + cl * 4;
            if (c0 >= a.N) return;
            const int64_t o = (c0 + (h & 3)) * a.kgroups + kg0;
            if (h < 4) {
                if constexpr (U == 4) *reinterpret_cast<float4*>(a.scale + o) = make_float4(acc_s[0], acc_s[1], acc_s[2], acc_s[3]);
                else *reinterpret_cast<float2*>(a.scale + o) = make_float2(acc_s[0], acc_s[1]);
            } else {
                uint32_t wz = 0;
#pragma unroll
                for (int t = 0; t < U; ++t) wz |= static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int32_t>(acc_z[t]) - bias)) << (8 * t);
                if constexpr (U == 4) *reinterpret_cast<uint32_t*>(a.zp + o) = wz;
                else *reinterpret_cast<uint16_t*>(a.zp + o) = static_cast<uint16_t>(wz);
            }
        }
    };

    // host guarantees kgroups % U == 0: every unit is full, the control flow below is static
    uint32_t cur = wave_in_part;
    if (cur >= punits) return;
    uint32_t strip, kg0;
    unit_pos(cur, strip, kg0);
    load_tile(bufA, strip, kg0);
    uint32_t nxt = cur + a.waves_per_part;
    while (nxt < punits) {
        uint32_t nstrip, nkg0;
        unit_pos(nxt, nstrip, nkg0);
        load_tile(bufB, strip, kg0 + 1);
        process(bufA, strip, kg0, 0);
        if constexpr (U == 4) {
            load_tile(bufA, strip, kg0 + 2);
            process(bufB, strip, kg0 + 1, 1);
            load_tile(bufB, strip, kg0 + 3);
            process(bufA, strip, kg0 + 2, 2);
            load_tile(bufA, nstrip, nkg0);
            process(bufB, strip, kg0 + 3, 3);
        } else {
            load_tile(bufA, nstrip, nkg0);
            process(bufB, strip, kg0 + 1, 1);
        }
        flush(strip, kg0);
        strip = nstrip;
        kg0 = nkg0;
        nxt += a.waves_per_part;
    }
    // last unit of this wave: nothing left to prefetch after its last tile
    load_tile(bufB, strip, kg0 + 1);
    process(bufA, strip, kg0, 0);
    if constexpr (U == 4) {
        load_tile(bufA, strip, kg0 + 2);
        process(bufB, strip, kg0 + 1, 1);
        load_tile(bufB, strip, kg0 + 3);
        process(bufA, strip, kg0 + 2, 2);
        process(bufB, strip, kg0 + 3, 3);
    } else {
        process(bufB, strip, kg0 + 1, 1);
    }
    flush(strip, kg0);
}

// Buffer roles: tile t of a unit lives in A (t even) or B (t odd); U is even, so the first tile of the next unit is A again.

extern "C" int32_t oq_rtn_quantize_f32(const float*, int64_t, int64_t, int64_t, int32_t, int32_t, int64_t, int32_t, int32_t, float, int32_t,
                                       void*, float*, void*, int32_t, void*, size_t, void*);
extern "C" size_t oq_rtn_workspace_bytes(int64_t, int64_t, int32_t, int64_t, int32_t);

__global__ void fill_normalish(float* x, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t s = static_cast<uint32_t>(i) * 2654435761u + seed;
        float acc = 0.f;
        for (int j = 0; j < 4; ++j) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            acc += static_cast<float>(s >> 8) * (1.0f / 16777216.0f) - 0.5f;
        }
        x[i] = acc * 1.7320508f;
    }
}

struct Variant {
    std::string name;
    int U; bool coalesce;
    int wpb;        // waves per block
    int blocks;     // grid
    int map, nt;
};

template <int U, bool C, int WPB>
static void launch(const StreamArgs& a, int blocks, hipStream_t s) {
    hipLaunchKernelGGL((rtn_stream<U, C, WPB>), dim3(blocks), dim3(WPB * 64), 0, s, a);
}

template <int U, bool C>
static void launch_w(int wpb, const StreamArgs& a, int blocks, hipStream_t s) {
    if (wpb == 1) launch<U, C, 1>(a, blocks, s);
    else if (wpb == 2) launch<U, C, 2>(a, blocks, s);
    else launch<U, C, 4>(a, blocks, s);
}

static void launch_variant(const Variant& v, StreamArgs a, hipStream_t s) {
    a.map = v.map; a.nt = v.nt;
    a.kunits = static_cast<uint32_t>(a.kgroups / v.U);
    a.nparts = 8;
    a.total_waves = v.blocks * v.wpb;
    a.waves_per_part = (v.blocks / 8) * v.wpb;
    if (v.U == 2) { if (v.coalesce) launch_w<2, true>(v.wpb, a, v.blocks, s); else launch_w<2, false>(v.wpb, a, v.blocks, s); }
    else { if (v.coalesce) launch_w<4, true>(v.wpb, a, v.blocks, s); else launch_w<4, false>(v.wpb, a, v.blocks, s); }
}

int main(int argc, char** argv) {
    const int64_t K = 4096, N = argc > 1 ? atoll(argv[1]) : 11008, g = 128;
    const int NBUF = 4, REPS = argc > 2 ? atoi(argv[2]) : 200;
    const size_t n = (size_t)K * N, groups = n / g;
    std::vector<float*> W(NBUF);
    for (int b = 0; b < NBUF; ++b) {
        CK(hipMalloc(&W[b], n * 4));
        hipLaunchKernelGGL(fill_normalish, dim3(4096), dim3(256), 0, 0, W[b], n, 1234u + b);
    }
    uint8_t *q_ref, *q, *zp_ref, *zp, *ws;
    float *s_ref, *sc;
    uint32_t* queue;
    CK(hipMalloc(&q_ref, n / 2)); CK(hipMalloc(&q, n / 2));
    CK(hipMalloc(&zp_ref, groups)); CK(hipMalloc(&zp, groups));
    CK(hipMalloc(&s_ref, groups * 4)); CK(hipMalloc(&sc, groups * 4));
    CK(hipMalloc(&queue, 64)); CK(hipMemset(queue, 0, 64));
    const size_t wsb = oq_rtn_workspace_bytes(K, N, OQ_GROUP, g, 0) + 256;
    CK(hipMalloc(&ws, wsb));
    CK(hipDeviceSynchronize());

    StreamArgs a{};
    a.K = K; a.N = N; a.ldw = N; a.kgroups = K / g; a.q = q; a.scale = sc; a.zp = zp;
    a.grid.qmin = 0; a.grid.qmax = 15; a.grid.symmetric = 0; a.grid.zero = 8; a.grid.levels = 7.0; a.grid.clip_ratio = 1.0f; a.grid.bits = 4;
    a.nstrips = static_cast<uint32_t>((N + 31) / 32);
    a.queue = queue;

    std::vector<Variant> vs;
    auto add = [&](int U, bool c, int wpb, int waves, int map, int nt) {
        char nm[128];
        snprintf(nm, sizeof nm, "U=%d %s wpb=%d waves=%d map%d nt%d", U, c ? "coalesce" : "scatter ", wpb, waves, map, nt);
        vs.push_back({nm, U, c, wpb, waves / wpb, map, nt});
    };
    // 11008 tiles: 1376 waves x 8 tiles, 2752 x 4, 688 x 16; 2048 = the unbalanced "fill every slot" grid
    for (int wpb : {1, 2, 4}) {
        add(2, true, wpb, 1376, 0, 1);
        add(4, true, wpb, 1376, 0, 1);
    }
    add(2, false, 1, 1376, 0, 1);
    add(2, true, 1, 1376, 1, 1);
    add(4, true, 1, 1376, 1, 1);
    add(2, true, 1, 1376, 0, 3);
    add(4, true, 1, 1376, 0, 3);
    add(2, true, 1, 1376, 0, 0);
    add(2, true, 1, 688, 0, 1);
    add(4, true, 1, 688, 0, 1);
    add(2, true, 1, 2752, 0, 1);
    add(4, true, 1, 2752, 0, 1);
    add(2, true, 4, 2048, 0, 1);
    add(4, true, 4, 2048, 0, 1);
    add(2, true, 1, 2048, 0, 1);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint8_t> hq(n / 2), hq_ref(n / 2), hz(groups), hz_ref(groups);
    std::vector<float> hs(groups), hs_ref(groups);

    auto time_ref = [&]() {
        for (int i = 0; i < 20; ++i) oq_rtn_quantize_f32(W[i % NBUF], K, N, N, OQ_UINT4, OQ_GROUP, g, 0, 0, 1.0f, 0, q_ref, s_ref, zp_ref, OQ_LAYOUT_NBITS, ws, wsb, nullptr);
        CK(hipEventRecord(e0));
        for (int i = 0; i < REPS; ++i) oq_rtn_quantize_f32(W[i % NBUF], K, N, N, OQ_UINT4, OQ_GROUP, g, 0, 0, 1.0f, 0, q_ref, s_ref, zp_ref, OQ_LAYOUT_NBITS, ws, wsb, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / REPS;
    };
    const double alg = (double)n * 4 + n / 2 + groups * 5;
    double t = time_ref();
    printf("%-58s %8.2f us  %6.3f TB/s  frac %.3f\n", "shipped rtn_group_wave<8> (library)", t, alg / t / 1e6, alg / t / 1e6 / 8.0);

    for (const Variant& v : vs) {
        // correctness on buffer 0
        oq_rtn_quantize_f32(W[0], K, N, N, OQ_UINT4, OQ_GROUP, g, 0, 0, 1.0f, 0, q_ref, s_ref, zp_ref, OQ_LAYOUT_NBITS, ws, wsb, nullptr);
        CK(hipMemset(q, 0xAA, n / 2)); CK(hipMemset(sc, 0xAA, groups * 4)); CK(hipMemset(zp, 0xAA, groups));
        a.W = W[0];
        launch_variant(v, a, nullptr);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hq.data(), q, n / 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hq_ref.data(), q_ref, n / 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hz.data(), zp, groups, hipMemcpyDeviceToHost)); CK(hipMemcpy(hz_ref.data(), zp_ref, groups, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), sc, groups * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs_ref.data(), s_ref, groups * 4, hipMemcpyDeviceToHost));
        const bool ok = hq == hq_ref && hz == hz_ref && memcmp(hs.data(), hs_ref.data(), groups * 4) == 0;
        if (!ok) {
            printf("%-58s MISMATCH (q %d zp %d scale %d)\n", v.name.c_str(), hq == hq_ref, hz == hz_ref,
                   memcmp(hs.data(), hs_ref.data(), groups * 4) == 0);
            continue;
        }
        for (int i = 0; i < 20; ++i) { a.W = W[i % NBUF]; launch_variant(v, a, nullptr); }
        CK(hipEventRecord(e0));
        for (int i = 0; i < REPS; ++i) { a.W = W[i % NBUF]; launch_variant(v, a, nullptr); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t = ms * 1e3 / REPS;
        printf("%-58s %8.2f us  %6.3f TB/s  frac %.3f\n", v.name.c_str(), t, alg / t / 1e6, alg / t / 1e6 / 8.0);
        fflush(stdout);
    }
    t = time_ref();
    printf("%-58s %8.2f us  %6.3f TB/s  frac %.3f\n", "shipped rtn_group_wave<8> (library, again)", t, alg / t / 1e6, alg / t / 1e6 / 8.0);
    return 0;
}
