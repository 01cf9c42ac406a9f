// A1: fused RTN weight quantization for gfx950 (rtn.py:54-109 of the reference, with utils.py:6-79
// folded in).  HBM-bound byte work: W[K, N] fp32 is read exactly once with 16-byte coalesced loads,
// held in registers, reduced down the K axis per column (lane-local, then across waves through a
// small LDS exchange), turned into (scale, zero point) in registers and quantized from the same
// registers.  Nothing is reshaped into a GEMM and no transposed copy (utils.py:24) is ever made.
#include "oq_common.hpp"

#include <cstdlib>
#include <type_traits>

namespace oq {

// Attribution switches (bits 3-6 of OQ_RTN_NT: drop the parameter stores, drop the blob stores, write a wave-contiguous
// -- WRONG -- layout, 8-byte instead of 16-byte blob stores) exist only in builds made with -DOQ_RTN_ATTRIBUTION
// (scripts/sweep_rtn*.sh).  The shipped library compiles them out and masks the environment value to the two
// layout-neutral bits (0: non-temporal W loads / [K,N] stores, 1: non-temporal blob stores), so no environment variable
// can make liboq_hip.so return anything but the reference's bytes (tests/test_library_abi.py).
#ifdef OQ_RTN_ATTRIBUTION
#define OQ_ATTR(nt, bit) (((nt) & (bit)) != 0)
constexpr int kNtMask = 127;
#else
#define OQ_ATTR(nt, bit) false
constexpr int kNtMask = 3;
#endif

// Division of a block id by a launch constant without the ~30-instruction integer-divide expansion: the quotient
// estimate float(n) * fl(1/d) is within 1 of n / d for n < 2^22 and is corrected with the exact remainder.
struct FastDiv {
    uint32_t d;
    float rcp;
};
static inline FastDiv make_fastdiv(uint32_t d) { return FastDiv{d, 1.0f / static_cast<float>(d)}; }
__device__ __forceinline__ uint32_t fast_divmod(uint32_t n, const FastDiv& f, uint32_t& rem) {
    uint32_t q = static_cast<uint32_t>(static_cast<float>(n) * f.rcp);
    int32_t r = static_cast<int32_t>(n - q * f.d);
    if (r < 0) { --q; r += static_cast<int32_t>(f.d); }
    else if (r >= static_cast<int32_t>(f.d)) { ++q; r -= static_cast<int32_t>(f.d); }
    rem = static_cast<uint32_t>(r);
    return q;
}

struct RtnPtrs {   // mirrors oq_rtn_ptrs (include/oq_hip.h): the four device pointers of one matrix of a list
    const float* W;
    uint8_t* q;
    float* scale;
    uint8_t* zp;
};

struct RtnArgs {
    const float* W;
    int64_t K, N, ldw;
    int64_t g;        // rows per group (K for channel / tensor)
    int64_t kgroups;  // K / g
    uint8_t* q;       // may be null (qparams only)
    float* scale;
    uint8_t* zp;
    float* scale_t;   // optional staging [kgroups, N] (coalesced); transposed to `scale` by transpose_qparams
    uint8_t* zp_t;
    // Round 6 (template mode TR): staging in the caller's zeroed STATE, transposed inside this launch by `ncol_tiles` blocks appended to
    // the grid (see transposer_block).  tr_zp: 8 bytes per four columns, each zero point in 16 bits with bit 8 set.
    float* tr_scale;
    uint2* tr_zp;
    QGrid grid;
    int32_t layout;
    int32_t wpg;      // waves per group
    int32_t gpb;      // groups per block
    uint32_t ncol_tiles, nrow_tiles;
    int32_t order;    // 0: K-direction fastest + XCD strips, 1: column tiles fastest (row-sequential DRAM stream)
    int32_t nt;       // non-temporal loads of W
    int32_t stage_q;  // NBITS: transpose the block's packed output through LDS (64/128-byte chunks per column)
    int32_t gk;       // order 2: row tiles per id chunk (see the block order in rtn_group_fused)
    uint32_t xg_log2; // order 2: log2 of the neighbouring column tiles one XCD owns inside a chunk (0 = one)
    FastDiv fd_ncol, fd_band, fd_chunk;  // ncol_tiles, ncol_tiles * gk, 8 * gk (block ids < 2^22, checked by the host)
    int32_t spb_log2;  // wave kernel: log2(strips per block)
    uint32_t pair_owner;  // fused kernel, direct parameter stores: wave (inside its group) that stores pair p, 3 bits each
    // strided batch (oq_rtn_quantize_batched_f32): matrix b lives at base + b * stride (elements / bytes as noted)
    int64_t w_stride, q_stride, p_stride;  // fp32 elements of W; bytes of q; entries of scale / zp (also of the staging)
    const RtnPtrs* table;  // list of matrices (oq_rtn_quantize_ptrs_f32): matrix b = table[b] instead of base + b * stride
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// The [K,N] results leave as WRITE-THROUGH stores (`sc1`, agent scope).  A default-policy or `nt` store leaves its line dirty in the
// XCD's L2, and what is dirty when the kernel ends is written back before the next launch of the stream starts: 45 MB stored by a
// kernel cost 3.3-3.9 us between its last workgroup's end and the next kernel's first start against 1.2-1.4 us for the same bytes
// written through, whenever in the kernel they were stored (scripts/lab_launch_gap.hip).  Beside the W stream a part of that
// comes back as slower stores; what stays (same box, alternating, 4096 x 11008): bytes 42.1-42.4 -> 41.5-41.7 us, packed nibbles
// 40.1-40.3 -> 39.25, 11008 x 4096 45.0 -> 44.5 and 43.7 -> 43.05.  Every store instruction there writes whole 128-byte lines.  The
// MatMulNBits blob does NOT: its 16-byte pieces meet their neighbours (other k-groups: other waves) in the L2, written through they
// are a fabric write each (38.3-38.4 -> 38.5-38.7 us): bit 0 stays off.  OQ_RTN_SC1 (lab builds): bit 0 the 4-bit blob of the
// wave kernel, bit 1 the [K,N] bytes, bit 2 the packed [K,N/2] dwords.
#ifndef OQ_RTN_SC1
#define OQ_RTN_SC1 6
#endif
// K1's fast path is proven row group by row group: OQ_WAVE_GROUP / OQ_FUSED_GROUP rows share ONE decision (a running
// v_maximum3_f32 of the residuals against the narrowest band, then one ballot) where rounds 1-5 spent a compare and a scalar OR
// per element and a ballot + branch per row.  Same box, alternating, 4096 x 11008: blob 38.8-39.0 -> 38.1 us (4096 x 4096 17.35 ->
// 16.2), [K,N] bytes 41.7 -> 40.4, packed nibbles 39.3 -> 37.85.  Four rows per decision cost the wave kernel its fifth wave per
// SIMD in effect (50.3 us) and the fused kernel its fourth (129 registers): two it is.
#ifndef OQ_WAVE_GROUP
#define OQ_WAVE_GROUP 2
#endif
#ifndef OQ_FUSED_GROUP
#define OQ_FUSED_GROUP 2
#endif
__device__ __forceinline__ void store_sc1(uint32_t w, uint32_t* p) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(w) : "memory"); }

constexpr int kDefaultWps = 5;     // wave kernel build used when OQ_RTN_WPS is unset (see Tuning)
constexpr int kColsPerWave = 256;  // 64 lanes x 4 columns: 1 KiB of one fp32 row per wave-instruction
constexpr int kMaxWaves = 8;

// Column owned by slot i of a lane.  VEC4: four neighbouring columns (one 16-byte load per row).
// Scalar fallback (N % 4 != 0 or unaligned base): lane-strided so every load stays coalesced.
template <bool VEC4>
__device__ __forceinline__ int64_t slot_col(int64_t tile_col0, int lane, int i) {
    return VEC4 ? tile_col0 + lane * 4 + i : tile_col0 + i * 64 + lane;
}

// NW consecutive little-endian 32-bit words to a (4*NW)-byte aligned address, widest stores first.
template <int NW>
__device__ __forceinline__ void store_words(uint8_t* dst, const uint32_t (&w)[NW]) {
    if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NW; j += 4) *reinterpret_cast<uint4*>(dst + 4 * j) = make_uint4(w[j], w[j + 1], w[j + 2], w[j + 3]);
    } else if constexpr (NW % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NW; j += 2) *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(w[j], w[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < NW; ++j) *reinterpret_cast<uint32_t*>(dst + 4 * j) = w[j];
    }
}

// Block id -> (row tile, column tile).  Speed only: every order visits every tile exactly once.
__device__ __forceinline__ void tile_of_block(const RtnArgs& a, uint32_t bid, uint32_t nblk, uint32_t& row_tile, uint32_t& col_tile) {
    if (a.order == 0) {
        const uint32_t id = xcd_remap(bid, nblk);
        col_tile = id / a.nrow_tiles;   // K-direction fastest: an XCD owns whole column strips
        row_tile = id - col_tile * a.nrow_tiles;
    } else if (a.order == 1) {
        row_tile = fast_divmod(bid, a.fd_ncol, col_tile);  // column tiles fastest: co-resident blocks stream whole rows
    } else {
        // Column tiles fastest over bands of `gk` row tiles, ids blocked as [8 column tiles] x [gk row tiles] with
        // the column tile in the low 3 bits: blocks of one column tile and neighbouring k-groups are 8 ids apart,
        // i.e. on one XCD at about the same time (observed round-robin placement; speed only).  Pieces of one
        // 128-byte line that different blocks produce -- the two 64-byte halves of a MatMulNBits line, the 4-byte
        // scales of neighbouring k-groups -- then merge in that XCD's L2 before they are written back, while a
        // band still streams whole rows of W.
        uint32_t r;
        const uint32_t b = fast_divmod(bid, a.fd_band, r);
        const uint32_t gk_eff = min(static_cast<uint32_t>(a.gk), a.nrow_tiles - b * a.gk);
        uint32_t cc;
        if (gk_eff == static_cast<uint32_t>(a.gk)) {
            cc = fast_divmod(r, a.fd_chunk, r);
        } else {  // last, shorter band
            cc = r / ((8u << a.xg_log2) * gk_eff);
            r -= cc * (8u << a.xg_log2) * gk_eff;
        }
        // xg_log2 > 0 (lab): an XCD owns 2^xg_log2 NEIGHBOURING column tiles of a chunk instead of one
        const uint32_t cw = 8u << a.xg_log2;
        const uint32_t w = min(cw, a.ncol_tiles - cc * cw);
        if (w == cw) {
            const uint32_t within = r & (cw - 1u);
            row_tile = b * a.gk + (r >> (3 + a.xg_log2));
            col_tile = cc * cw + ((within & 7u) << a.xg_log2) + (within >> 3);
        } else {  // last, narrower chunk of column tiles
            row_tile = b * a.gk + r / w;
            col_tile = cc * cw + r % w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused one-pass kernel.  One block = GPB groups (stacked along K) x 256 columns; one group =
// WPG waves x RPW rows.  Registers per lane: RPW x 4 fp32 of W.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// The [K/g, N] -> [N, K/g] transposition of the staged parameters INSIDE the fused launch (round 6; template mode TR, stateful entry
// point only, packed nibbles of up to 32 k-groups: see the host's rule).  The separate transpose launch is 5 us of which the kernel
// itself needs 1.5 (rocprofv3: 43.0 + 5.0 us against 43.2-44.5 us per call).
// Every main block stores its parameters into the caller's zeroed state with agent-scope (`sc1`) word stores -- four scales and two
// words of zero-point pairs (16 bits each, bit 8 set) per lane -- and waits for nothing: a scale is never zero (utils.py:266-268)
// and a pair word never is, so every WORD says by itself whether it has been written.  `ncol_tiles` blocks are appended to the
// grid; they are dispatched behind the last main block, into slots no main block is waiting for, and each takes one column tile:
// poll ONE word of the tile's last k-group until it is there (the k-groups of a column tile finish roughly in order), read all of
// the tile's words with `sc1` loads -- re-reading the few that are not written yet --, transpose them through the 16 KB of LDS the
// fold uses, store whole 128-byte lines of the result, and put the zeros back (the state is what the next call finds).
// Also built: the transposition by the block of each column tile's LAST k-group after its own stores (no appended blocks) --
// bit-exact, slower than the second launch (packed 44.5 -> 45.0 us, bytes 45.5 -> 48.9): the last round of blocks then carries
// the read-back's round trips on top of its own work.
// Forward progress: a main block never waits; an appended block waits only for main blocks of its own launch, and there are at
// most `ncol_tiles` of them -- a few dozen of the device's workgroup slots -- so the blocks they wait for always find a slot whatever
// the dispatch order.  (Launches of DIFFERENT streams could in principle fill the device with waiting blocks: the host orders them
// through the ticket chain of rtn_resident.hip.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lab_spin_over(uint32_t spins) {
#ifdef OQ_SPIN_LIMIT      /* lab builds: give up instead of hanging the box (the result is garbage then) */
    return spins >= static_cast<uint32_t>(OQ_SPIN_LIMIT);
#else
    (void)spins;
    return false;
#endif
}
// One appended block = COLS columns of the result (all k-groups): 32 x COLS / 4 granules per pass of 32 k-groups, TWO per thread,
// both loads in flight at once (four one after the other cost the wave kernel 6 us of round trips at the tail of the launch).
// NTHREADS = 512, COLS = 128 (fused kernel).  `s_f`: 32 x COLS floats, `s_z`: 32 x COLS bytes.  (Built for the blob's wave kernel as well,
// 256 threads x 64 columns, the wave's 32 scales leaving as ONE line instead of 32: 39.2 -> 41.5 us -- with no second launch to remove, the
// appended blocks' round trips behind the last main block cost more than the scattered stores: docs/LAB_NOTES_r06.md.)
template <int NTHREADS, int COLS>
__device__ __forceinline__ void transposer_block(const RtnArgs& a, uint32_t part, float* s_f, uint8_t* s_z) {
    constexpr int GQ = COLS / 4;                       // granules per row
    static_assert(32 * GQ == 2 * NTHREADS, "two granules per thread");
    const int tid = threadIdx.x;
    const int64_t N = a.N, kgroups = a.kgroups;
    const int64_t cbase = static_cast<int64_t>(part) * COLS;
    if (cbase >= N) return;                            // uniform
    // one poll: the first word of this column range in the LAST k-group
    if (tid == 0) {
        const u32x4* probe = reinterpret_cast<const u32x4*>(a.tr_scale + (kgroups - 1) * N + cbase);
        u32x4 v;
        for (uint32_t spins = 0;; ++spins) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(probe) : "memory");
            if (v[0] != 0u || lab_spin_over(spins)) break;
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
    for (int64_t kb = 0; kb < kgroups; kb += 32) {
        // ---- in: 32 rows x GQ granules of four columns (16 bytes of scales + 8 bytes of zero-point pairs)
        u32x4 sv[2];
        u32x2 zv[2];
        const u32x4* ps[2];
        const u32x2* pz[2];
        bool live[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + NTHREADS * j, r = idx / GQ, gq = idx % GQ;
            const int64_t kg = kb + r, col = cbase + gq * 4;
            live[j] = kg < kgroups && col < N;
            const int64_t o = live[j] ? kg * N + col : (kgroups - 1) * N + cbase;      // a valid address for the lanes outside: loaded, not used
            ps[j] = reinterpret_cast<const u32x4*>(a.tr_scale + o);
            pz[j] = reinterpret_cast<const u32x2*>(a.tr_zp + o / 4);
        }
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx2 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(sv[0]), "=&v"(zv[0]), "=&v"(sv[1]), "=&v"(zv[1]) : "v"(ps[0]), "v"(pz[0]), "v"(ps[1]), "v"(pz[1]) : "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + NTHREADS * j, r = idx / GQ, gq = idx % GQ;
            if (live[j]) {
                for (uint32_t spins = 0; !((sv[j][0] != 0u && sv[j][1] != 0u && sv[j][2] != 0u && sv[j][3] != 0u && zv[j][0] != 0u && zv[j][1] != 0u) ||
                                           lab_spin_over(spins)); ++spins) {      // rare: a granule that was not written yet
                    __builtin_amdgcn_s_sleep(8);
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(sv[j]), "=&v"(zv[j]) : "v"(ps[j]), "v"(pz[j]) : "memory");
                }
                // the zeros go back: the state is what the next call finds
                const u32x4 z4 = {0u, 0u, 0u, 0u};
                const u32x2 z2 = {0u, 0u};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %2, %3, off sc1" : : "v"(ps[j]), "v"(z4), "v"(pz[j]), "v"(z2) : "memory");
            }
            const int at = r * COLS + ((gq + r) % GQ) * 4;      // granule index rotated by the row: the transposed reads spread over the banks
            *reinterpret_cast<u32x4*>(s_f + at) = sv[j];
            *reinterpret_cast<uint32_t*>(s_z + at) = (zv[j][0] & 0xffu) | ((zv[j][0] >> 8) & 0xff00u) | ((zv[j][1] & 0xffu) << 16) | ((zv[j][1] << 8) & 0xff000000u);
        }
        __syncthreads();
        // ---- out: column c x four consecutive k-groups per thread, eight threads = one column's 128 bytes of scales / 32 of zero points
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + NTHREADS * j, kq = idx & 7, c = idx >> 3;
            const int64_t col = cbase + c, kg0 = kb + kq * 4;
            if (col < N && kg0 < kgroups) {
                float o[4];
                uint32_t w = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = kq * 4 + i, at = r * COLS + (((c >> 2) + r) % GQ) * 4 + (c & 3);
                    o[i] = s_f[at];
                    w |= static_cast<uint32_t>(s_z[at]) << (8 * i);
                }
                *reinterpret_cast<float4*>(a.scale + col * kgroups + kg0) = make_float4(o[0], o[1], o[2], o[3]);      // kgroups % 4 == 0 (host)
                *reinterpret_cast<uint32_t*>(a.zp + col * kgroups + kg0) = w;
            }
        }
        __syncthreads();
    }
}

template <int RPW, bool VEC4, bool EMIT_Q, bool NT = false, bool TR = false>      // TR: parameters staged in the caller's state and transposed inside the launch
__global__ __launch_bounds__(kMaxWaves* kWave) void rtn_group_fused(const RtnArgs a_in) {
    __shared__ float4 s_mn[kMaxWaves][kWave];
    __shared__ float4 s_mx[kMaxWaves][kWave];

    if constexpr (TR) {
        if (blockIdx.x >= a_in.ncol_tiles * a_in.nrow_tiles) {      // uniform: one of the blocks appended to the grid: 128 columns each
            __shared__ __attribute__((aligned(16))) float s_tr_f[32 * 128];
            __shared__ __attribute__((aligned(16))) uint8_t s_tr_z[32 * 128];
            transposer_block<kMaxWaves * kWave, 128>(a_in, blockIdx.x - a_in.ncol_tiles * a_in.nrow_tiles, s_tr_f, s_tr_z);
            return;
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wig = wave % a_in.wpg;   // wave inside its group
    const int gib = wave / a_in.wpg;   // group inside the block

    const uint32_t nblk = a_in.ncol_tiles * a_in.nrow_tiles;
    // Batched launch: block ids run over all matrices, so the tail of one matrix overlaps the head of the next.
    const uint32_t mat = blockIdx.y;   // strided batch: x runs fastest, so the tail of one matrix overlaps the head of the next
    const uint32_t bid = blockIdx.x;
    RtnArgs a = a_in;
    if (a.table != nullptr) {   // wave-uniform: four scalar loads
        const RtnPtrs p = a.table[mat];
        a.W = p.W; a.q = p.q; a.scale = p.scale; a.zp = p.zp;
    } else {
        a.W += static_cast<int64_t>(mat) * a.w_stride;
        if (a.q) a.q += static_cast<int64_t>(mat) * a.q_stride;
        a.scale += static_cast<int64_t>(mat) * a.p_stride;
        a.zp += static_cast<int64_t>(mat) * a.p_stride;
    }
    if (a.scale_t) { a.scale_t += static_cast<int64_t>(mat) * a.p_stride; a.zp_t += static_cast<int64_t>(mat) * a.p_stride; }
    uint32_t col_tile, row_tile;
    tile_of_block(a, bid, nblk, row_tile, col_tile);

    const int64_t kg = static_cast<int64_t>(row_tile) * a.gpb + gib;
    const bool group_ok = kg < a.kgroups;
    const int64_t row0 = kg * a.g + static_cast<int64_t>(wig) * RPW;
    const int64_t tile_col0 = static_cast<int64_t>(col_tile) * kColsPerWave;

    bool col_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) col_ok[i] = slot_col<VEC4>(tile_col0, lane, i) < a.N;

    // Loads are unconditional: lanes past the right edge and waves past the last group read a clamped
    // (valid, duplicate) address instead of being predicated off.  A predicated load makes hipcc wrap every
    // load in its own exec branch with a vmcnt(0) behind it (cdna_hip_programming.md section 5, trap (c)),
    // which serialises the 16 row loads of a wave.  Only the stores are masked.
    float v[RPW][4];
    {
        const int64_t lrow0 = group_ok ? row0 : static_cast<int64_t>(wig) * RPW;
        if constexpr (VEC4) {
            int64_t lcol = tile_col0 + lane * 4;
            lcol = lcol < a.N ? lcol : a.N - 4;
            const float* p = a.W + lrow0 * a.ldw + lcol;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float4 t;
                if constexpr (NT) {
                    const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + r * a.ldw));
                    t = make_float4(u[0], u[1], u[2], u[3]);
                } else {
                    t = *reinterpret_cast<const float4*>(p + r * a.ldw);
                }
                v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int64_t c = slot_col<false>(tile_col0, lane, i);
                    c = c < a.N ? c : a.N - 1;
                    v[r][i] = a.W[(lrow0 + r) * a.ldw + c];
                }
        }
    }

    // R1 (utils.py:60-61): lane-local column min / max over this wave's rows.
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mn[i] = mx[i] = v[0][i];
#pragma unroll
    for (int r = 1; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = nmin(mn[i], v[r][i]);
            mx[i] = nmax(mx[i], v[r][i]);
        }

    if (a.wpg > 1) {  // uniform over the block: cross-wave combine of the group's partials in LDS
        s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
        __syncthreads();
        const int w0 = gib * a.wpg;
        for (int w = 0; w < a.wpg; ++w) {
            const float4 tn = s_mn[w0 + w][lane];
            const float4 tx = s_mx[w0 + w][lane];
            mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y);
            mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
            mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y);
            mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
        }
    }
    if (!group_ok) return;

    // Q1 in registers (every wave of the group derives the same parameters).
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    // bias: signed levels are produced as 0..255 (0..15 for packed nibbles) and flipped back with one XOR per word
    const int32_t bias = qmin < 0 ? (((a.layout == OQ_LAYOUT_NBITS || a.layout == OQ_LAYOUT_KN_PACKED4) && a.grid.bits == 4) ? 8 : 128) : 0;
    ColQ cq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const QParam p = qparam_from_minmax(mn[i], mx[i], a.grid);
        cq[i] = make_colq(p, mn[i], mx[i], bias);
    }

    if (wig == 0) {
        if constexpr (TR) {
            // staged in the caller's state for the block that transposes this column tile: agent-scope (`sc1`) word stores, nothing to wait
            // for.  Every word says by itself whether it has been written: a scale is never zero, a zero-point pair carries bits 8 and 24
            if (col_ok[0]) {
                const int64_t o = kg * a.N + tile_col0 + lane * 4;
                // (one 16-byte and one 8-byte `sc1` granule per lane as inline assembly costs this instantiation 130+ registers; six word stores per
                // lane are twice the fabric writes: [K,N] bytes 43.05 us against 42.5 with three 8-byte stores)
                uint64_t* ps = reinterpret_cast<uint64_t*>(a.tr_scale + o);      // 8-byte agent-scope stores: half as many fabric writes as words
                __hip_atomic_store(ps, static_cast<uint64_t>(__float_as_uint(cq[0].scale)) | (static_cast<uint64_t>(__float_as_uint(cq[1].scale)) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ps + 1, static_cast<uint64_t>(__float_as_uint(cq[2].scale)) | (static_cast<uint64_t>(__float_as_uint(cq[3].scale)) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t z01 = 0x01000100u | (static_cast<uint32_t>(cq[0].zp) & 0xffu) | ((static_cast<uint32_t>(cq[1].zp) & 0xffu) << 16);
                const uint32_t z23 = 0x01000100u | (static_cast<uint32_t>(cq[2].zp) & 0xffu) | ((static_cast<uint32_t>(cq[3].zp) & 0xffu) << 16);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(a.tr_zp + o / 4), static_cast<uint64_t>(z01) | (static_cast<uint64_t>(z23) << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (VEC4 && a.scale_t != nullptr) {
            // staged [kg, n]: 16 B + 4 B per lane, fully coalesced (the n-major scatter of 4-byte pieces at a
            // 128-byte stride costs ~7 us on this matrix: partial-line writes)
            if (col_ok[0]) {
                const int64_t o = kg * a.N + tile_col0 + lane * 4;
                *reinterpret_cast<float4*>(a.scale_t + o) = make_float4(cq[0].scale, cq[1].scale, cq[2].scale, cq[3].scale);
                *reinterpret_cast<uint32_t*>(a.zp_t + o) = (static_cast<uint32_t>(cq[0].zp) & 0xffu) | ((static_cast<uint32_t>(cq[1].zp) & 0xffu) << 8) |
                                                           ((static_cast<uint32_t>(cq[2].zp) & 0xffu) << 16) | ((static_cast<uint32_t>(cq[3].zp) & 0xffu) << 24);
            }
        }
    }
    if (!TR && !(VEC4 && a.scale_t != nullptr)) {
        // rtn.py:98-109 result layout: row n*(K/g)+kg of the [N*K/g, 1] arrays.  Every wave of the group holds the same
        // parameters, so the 256 scattered 4-byte + 1-byte stores of a group (128-byte stride) are dealt over its waves:
        // eight (column slot, half wave) pairs, the owner of pair p (= p % wpg, three bits each in `pair_owner`) stores it.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int own_lo = static_cast<int>((a.pair_owner >> (6 * i)) & 7u), own_hi = static_cast<int>((a.pair_owner >> (6 * i + 3)) & 7u);
            if (((lane < 32) ? own_lo : own_hi) == wig && col_ok[i]) {
                const int64_t o = slot_col<VEC4>(tile_col0, lane, i) * a.kgroups + kg;
                a.scale[o] = cq[i].scale;
                a.zp[o] = static_cast<uint8_t>(cq[i].zp);
            }
        }
    }
    if constexpr (!EMIT_Q) return;

    // K1 from registers: v[r][i] <- clamped, biased level (an exact small float).
    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
#if OQ_FUSED_GROUP > 1
    // one decision per OQ_FUSED_GROUP rows on a running NaN-propagating maximum of |t - k| (v_maximum3_f32 with |.| modifiers) against
    // the narrowest band of the lane's four columns, instead of a compare + scalar OR per element and a ballot + branch per row
    {
        constexpr int GROUP = RPW % OQ_FUSED_GROUP == 0 ? OQ_FUSED_GROUP : 1;
        const float thr_min = nmin(nmin(cq[0].thr, cq[1].thr), nmin(cq[2].thr, cq[3].thr));
#pragma unroll
        for (int rg = 0; rg < RPW; rg += GROUP) {
            float f[GROUP][4];
            float far = 0.0f;
#pragma unroll
            for (int r = 0; r < GROUP; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = v[rg + r][i] * cq[i].rinv;
                    const float k = rintf(t);
                    far = nmax(far, fabsf(t - k));
                    f[r][i] = __builtin_amdgcn_fmed3f(k + cq[i].zpb, lo_b, hi_b);
                }
            if (__builtin_amdgcn_ballot_w64(!(far < thr_min)) != 0) {  // wave-uniform, rare: redo these rows with the IEEE divide
#pragma unroll
                for (int r = 0; r < GROUP; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[r][i] = quantize_exact_biased(v[rg + r][i], cq[i], qmin, qmax, bias);
            }
#pragma unroll
            for (int r = 0; r < GROUP; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rg + r][i] = f[r][i];
        }
    }
#else
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float f[4];
        bool unsafe = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = quantize_fast_biased(v[r][i], cq[i], lo_b, hi_b, unsafe);
        if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {  // wave-uniform, rare: redo this row with the IEEE divide
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = quantize_exact_biased(v[r][i], cq[i], qmin, qmax, bias);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[r][i] = f[i];
    }
#endif

    do {      // the layout's stores; `break` instead of `return`: every wave reaches the tail below
    if (a.layout == OQ_LAYOUT_KN) {
        const uint32_t flip = bias ? 0x80808080u : 0u;
        if constexpr (VEC4) {
            if (col_ok[0]) {
                uint8_t* o = a.q + row0 * a.N + tile_col0 + lane * 4;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][0], 0, 0);
                    w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][1], 1, w);
                    w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][2], 2, w);
                    w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][3], 3, w);
                    if constexpr ((OQ_RTN_SC1 & 2) != 0) store_sc1(w ^ flip, reinterpret_cast<uint32_t*>(o + r * a.N));
                    else if constexpr (NT) __builtin_nontemporal_store(w ^ flip, reinterpret_cast<uint32_t*>(o + r * a.N));
                    else *reinterpret_cast<uint32_t*>(o + r * a.N) = w ^ flip;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col_ok[i])
                        a.q[(row0 + r) * a.N + slot_col<false>(tile_col0, lane, i)] =
                            static_cast<uint8_t>((__builtin_amdgcn_cvt_pk_u8_f32(v[r][i], 0, 0) ^ flip) & 0xffu);
        }
    } else if (a.layout == OQ_LAYOUT_KN_PACKED4) {
        // core/_pack.py:8-22 on the [K, N] result: flat order, even index in the low nibble -- with N even, byte j of row k
        // holds columns 2j (low) and 2j + 1 (high); signed levels as two's-complement nibbles.  A lane's four columns of a
        // row are two bytes.  VEC4 only (checked by the host).
        const uint32_t flip = bias ? 0x88888888u : 0u;
        uint32_t t[RPW];    // byte 0 = columns 0 | 1 << 4, byte 2 = columns 2 | 3 << 4 of row r
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][0], 0, 0);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][1], 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][2], 2, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[r][3], 3, w);
            t[r] = (w | (w >> 4)) ^ flip;           // levels < 16: the high nibble of every byte of w is zero
        }
        uint8_t* o = a.q + (row0 * a.N + tile_col0 + lane * 4) / 2;
        const int64_t row_bytes = a.N / 2;
        if constexpr (RPW == 16) {
            if ((a.N & 7) == 0) {
                // Dword stores: lanes 2m and 2m + 1 hold neighbouring byte pairs of every row.  Row j and row j + 8 travel in one
                // dword; after one quad-perm exchange the even lane stores rows 0-7 (its own pair below its partner's), the
                // odd lane rows 8-15: eight 4-byte stores per lane, every instruction two rows x 128 contiguous bytes.
                const bool odd = (lane & 1) != 0;
                const bool pair_ok = tile_col0 + (lane & ~1) * 4 < a.N;     // N % 8 == 0: a lane pair is in or out together
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t mine = __builtin_amdgcn_perm(t[j + 8], t[j], 0x06040200u);   // [row j pair, row j + 8 pair]
                    const uint32_t theirs = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mine), 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
                    const uint32_t out = odd ? ((theirs >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (theirs << 16));
                    uint8_t* dst = a.q + ((row0 + j + (odd ? 8 : 0)) * a.N + tile_col0 + (lane & ~1) * 4) / 2;
                    if (pair_ok) {
                        if constexpr ((OQ_RTN_SC1 & 4) != 0) store_sc1(out, reinterpret_cast<uint32_t*>(dst));
                        else *reinterpret_cast<uint32_t*>(dst) = out;
                    }
                }
                break;
            }
        }
        if (col_ok[0]) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                *reinterpret_cast<uint16_t*>(o + r * row_bytes) = static_cast<uint16_t>(__builtin_amdgcn_perm(0u, t[r], 0x0c0c0200u));
        }
    } else {
        // MatMulNBits blob (qrules/_common.py:72-87): for out-channel n, k-group kg: g*bits/8 bytes,
        // k ascending, even k in the low nibble.  This wave owns bytes [wig*RPW*bits/8, +RPW*bits/8).
        const int64_t blob = a.g * a.grid.bits / 8;
        if constexpr (RPW == 16 && VEC4) {
            if (a.stage_q) {
                // The 8 waves of the block own, for every column, consecutive pieces (8 B of nibbles / 16 B of
                // bytes each) of ONE contiguous chunk of the blob (k-groups stacked in a block are adjacent in
                // [N, K/g, g*bits/8]).  Writing them from registers is 64 lanes x 8 B to 64 different lines per
                // instruction (2 M tiny L2 writes per matrix); instead the chunk is assembled in LDS (XOR-swizzled
                // slots, see below) and written with 16 B per lane, 4 or 8 lanes per column.
                __shared__ uint4 s_out[kColsPerWave * 8];   // 256 columns x 128 B
                uint8_t* so = reinterpret_cast<uint8_t*>(s_out);
                const bool four = a.grid.bits == 4;
                const int chunk = four ? 64 : 128;           // bytes per column and block
                const int sw = lane & 7;                     // swizzle key of this lane's columns ((col/4) & 7)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cl = lane * 4 + i;
                    if (four) {
                        const uint32_t flip = bias ? 0x88888888u : 0u;
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
                        *reinterpret_cast<uint2*>(so + cl * 64 + ((wave ^ sw) * 8)) = make_uint2(words[0], words[1]);
                    } else {
                        const uint32_t flip = bias ? 0x80808080u : 0u;
                        uint32_t words[4];
#pragma unroll
                        for (int wd = 0; wd < 4; ++wd) {
                            uint32_t acc = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 4 + j][i], j, acc);
                            words[wd] = acc ^ flip;
                        }
                        *reinterpret_cast<uint4*>(so + cl * 128 + ((wave ^ sw) * 16)) = make_uint4(words[0], words[1], words[2], words[3]);
                    }
                }
                __syncthreads();
                const int lpc = chunk / 16;                  // lanes per column chunk: 4 or 8
                const int cpi = 64 / lpc;                    // chunks per wave-instruction
                const int64_t kg0 = static_cast<int64_t>(row_tile) * a.gpb;
                for (int it = 0; it < 32 / cpi; ++it) {
                    const int cl = wave * 32 + it * cpi + lane / lpc;
                    const int part = lane % lpc;
                    const int64_t col = tile_col0 + cl;
                    const int key = (cl >> 2) & 7;
                    uint4 t;
                    if (four) {   // 16 B = slots 2*part, 2*part+1 -> swizzled pair (part ^ (key >> 1)), halves swapped if key is odd
                        t = *reinterpret_cast<const uint4*>(so + cl * 64 + ((part ^ (key >> 1)) * 16));
                        if (key & 1) t = make_uint4(t.z, t.w, t.x, t.y);
                    } else {
                        t = *reinterpret_cast<const uint4*>(so + cl * 128 + ((part ^ key) * 16));
                    }
                    if (col < a.N) *reinterpret_cast<uint4*>(a.q + (col * a.kgroups + kg0) * blob + part * 16) = t;
                }
                break;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!col_ok[i]) continue;
            const int64_t col = slot_col<VEC4>(tile_col0, lane, i);
            uint8_t* o = a.q + (col * a.kgroups + kg) * blob;
            if (a.grid.bits == 4) {
                if constexpr (RPW >= 8) {
                    const uint32_t flip = bias ? 0x88888888u : 0u;
                    uint32_t words[RPW / 8];
#pragma unroll
                    for (int wd = 0; wd < RPW / 8; ++wd) {
                        uint32_t ev = 0, od = 0;  // bytes of the even-k / odd-k levels
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ev = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 8 + 2 * j][i], j, ev);
                            od = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 8 + 2 * j + 1][i], j, od);
                        }
                        words[wd] = (ev | (od << 4)) ^ flip;
                    }
                    store_words<RPW / 8>(o + wig * (RPW / 2), words);
                }
            } else {
                if constexpr (RPW >= 4) {
                    const uint32_t flip = bias ? 0x80808080u : 0u;
                    uint32_t words[RPW / 4];
#pragma unroll
                    for (int wd = 0; wd < RPW / 4; ++wd) {
                        uint32_t acc = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 4 + j][i], j, acc);
                        words[wd] = acc ^ flip;
                    }
                    store_words<RPW / 4>(o + wig * RPW, words);
                }
            }
        }
    }
    } while (false);
}

// ---------------------------------------------------------------------------------------------
// Wave-owns-group kernel for the MatMulNBits blob.  One wave = one k-group of G = 16 * (64 / LPR)
// rows x (4 * LPR) columns: LPR lanes span one row piece (LPR x 16 B), the 64 / LPR lane sets `h`
// hold 16 consecutive rows each, so a wave-instruction loads 64 / LPR row pieces.  The group's
// column range is folded lane-locally over 16 rows and then across the lane sets with xor
// butterflies: no LDS, no barrier, waves never wait for each other.  In the blob a lane's 16 rows
// of a column are 8 (4-bit) / 16 (8-bit) consecutive bytes and the lane sets of a column are
// adjacent, so one store instruction writes whole 64 / 128-byte chunks.  G = 128 for LPR = 8.
// ---------------------------------------------------------------------------------------------
template <int LPR, bool EMIT_Q, int WPS = 0>
__global__ __launch_bounds__(WPS ? 256 : kMaxWaves* kWave, WPS ? WPS : 1) void rtn_group_wave(const RtnArgs a_in) {
    constexpr int RS = 64 / LPR;  // lane sets (row sub-ranges) per wave
    constexpr int G = 16 * RS;    // rows per group
    // WPS > 0: built for WPS waves per SIMD (blocks of <= 4 waves).  The four scales of a lane are only needed again by
    // the rare exact-division fallback: they are parked in LDS (4 * LPR floats per wave, written and read by the same wave:
    // no barrier) instead of four registers, which is what separates 100 registers from the 96 that five waves allow.
    __shared__ float s_scale[WPS ? 4 : 1][WPS ? 4 * LPR : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar address math below
    const int h = lane / LPR, cl = lane % LPR;

    const uint32_t nblk = a_in.ncol_tiles * a_in.nrow_tiles;
    const uint32_t mat = blockIdx.y;   // strided batch: x runs fastest, so the tail of one matrix overlaps the head of the next
    const uint32_t bid = blockIdx.x;
    RtnArgs a = a_in;
    if (a.table != nullptr) {   // wave-uniform: four scalar loads
        const RtnPtrs p = a.table[mat];
        a.W = p.W; a.q = p.q; a.scale = p.scale; a.zp = p.zp;
    } else {
        a.W += static_cast<int64_t>(mat) * a.w_stride;
        if (a.q) a.q += static_cast<int64_t>(mat) * a.q_stride;
        a.scale += static_cast<int64_t>(mat) * a.p_stride;
        a.zp += static_cast<int64_t>(mat) * a.p_stride;
    }
    uint32_t col_tile, row_tile;
    tile_of_block(a, bid, nblk, row_tile, col_tile);

    // block = gpb k-groups x spb strips of 4 * LPR columns; neighbouring strips on neighbouring waves
    const int spb = 1 << a.spb_log2;
    const int64_t kg = static_cast<int64_t>(row_tile) * a.gpb + (wave >> a.spb_log2);
    const int64_t strip0 = (static_cast<int64_t>(col_tile) * spb + (wave & (spb - 1))) * (4 * LPR);
    if (kg >= a.kgroups || strip0 >= a.N) return;  // wave-uniform; nothing below synchronises
    // WPS > 0: 32-bit output indices (the host checks N * K/g * G/2 < 2^32): scalar base + 32-bit lane offset addressing
    // instead of 64-bit address pairs held across the kernel
    using idx_t = std::conditional_t<(WPS > 0), uint32_t, int64_t>;
    const idx_t c0 = static_cast<idx_t>(strip0 + cl * 4);
    const idx_t kgroups_i = static_cast<idx_t>(a.kgroups), kg_i = static_cast<idx_t>(kg);
    const bool col_ok = strip0 + cl * 4 < a.N;     // N % 4 == 0: a lane's four columns are in or out together

    float v[16][4];
    {
        // scalar row base + one 32-bit lane offset for all 16 loads (host guarantees 112 * ldw * 4 + 4 * N < 2^32);
        // lanes past the right edge read a clamped address, never a predicated load
        const char* base = reinterpret_cast<const char*>(a.W + kg * G * a.ldw + strip0);
        const uint32_t loff = static_cast<uint32_t>(h * 16 * a.ldw + (col_ok ? cl * 4 : a.N - 4 - strip0)) * 4u;
        const int64_t row_bytes = a.ldw * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + r * row_bytes + loff));
            v[r][0] = u[0]; v[r][1] = u[1]; v[r][2] = u[2]; v[r][3] = u[3];
        }
    }
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mn[i] = mx[i] = v[0][i];
#pragma unroll
    for (int r = 1; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = nmin(mn[i], v[r][i]);
            mx[i] = nmax(mx[i], v[r][i]);
        }
    if constexpr (WPS > 0) {   // butterflies over the lane sets on DPP / permlane swaps instead of ds_bpermute
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (LPR <= 8) { mn[i] = xor_min<8>(mn[i]); mx[i] = xor_max<8>(mx[i]); }
            if constexpr (LPR <= 16) { mn[i] = xor_min<16>(mn[i]); mx[i] = xor_max<16>(mx[i]); }
            mn[i] = xor_min<32>(mn[i]); mx[i] = xor_max<32>(mx[i]);
        }
    } else {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mn[i] = nmin(mn[i], __shfl_xor(mn[i], off, 64));
                mx[i] = nmax(mx[i], __shfl_xor(mx[i], off, 64));
            }
    }

    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? (a.grid.bits == 4 ? 8 : 128) : 0;
    // per column: scale, fl(1/scale), float(zp + bias); one safety band for the four columns (the narrowest).
    // Fewer live registers than four full ColQ records (102 VGPRs, 4 waves per SIMD; forcing 96 for a fifth wave
    // spills 7 dwords and measured 44.7 us against 41.0).
    float sc[4], rinv[4], zpb[4], thr;
    [[maybe_unused]] float own_scale = 0.f;
    constexpr bool ONECOL = WPS > 0 && RS >= 4;
    if constexpr (ONECOL) {
        // Every lane set holds the ranges of all four column slots after the butterflies; deriving all four parameter sets
        // in every lane set is 8 x redundant (three IEEE divisions per column) and its temporaries are the register peak of
        // the kernel.  Lane set h derives column slot h & 3 only, the other three arrive by ds_bpermute from lane sets 0-3.
        const int j = h & 3;
        const float mnj = j == 0 ? mn[0] : j == 1 ? mn[1] : j == 2 ? mn[2] : mn[3];
        const float mxj = j == 0 ? mx[0] : j == 1 ? mx[1] : j == 2 ? mx[2] : mx[3];
        const ColQ c = make_colq(qparam_from_minmax(mnj, mxj, a.grid), mnj, mxj, bias);
        own_scale = c.scale;
        if (h < 4) s_scale[wave][j * LPR + cl] = c.scale;      // for the fallback: slot i of strip column cl at [i * LPR + cl]
        if (h < 4 && col_ok && !OQ_ATTR(a.nt, 8)) {   // rtn.py:98-109 result layout (entry n * K/g + kg); lane set j stores column slot j
            idx_t o = (c0 + j) * kgroups_i + kg_i;
            if (OQ_ATTR(a.nt, 32)) o = static_cast<idx_t>((kg * (a.N / (4 * LPR)) + strip0 / (4 * LPR)) * (4 * LPR) + lane);   // attribution builds only
            a.scale[o] = c.scale;
            a.zp[o] = static_cast<uint8_t>(static_cast<int32_t>(c.zpb) - bias);
        }
        float t = c.thr;
        if constexpr (LPR == 8) { t = xor_min<8>(t); thr = xor_min<16>(t); }
        else { t = xor_min<16>(t); thr = xor_min<32>(t); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rinv[i] = __shfl(c.rinv, i * LPR + cl, 64);
            zpb[i] = __shfl(c.zpb, i * LPR + cl, 64);
            sc[i] = 0.f;
        }
    } else {
        ColQ c = make_colq(qparam_from_minmax(mn[0], mx[0], a.grid), mn[0], mx[0], bias);
        sc[0] = c.scale; rinv[0] = c.rinv; zpb[0] = c.zpb; thr = c.thr;
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            c = make_colq(qparam_from_minmax(mn[i], mx[i], a.grid), mn[i], mx[i], bias);
            sc[i] = c.scale; rinv[i] = c.rinv; zpb[i] = c.zpb; thr = nmin(thr, c.thr);
        }
        // rtn.py:98-109 result layout (entry n * K/g + kg); lane set j stores column slot j
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += RS) {
            const int j = j0 + h;
            if (j < 4 && col_ok) {
                const float s = j == 0 ? sc[0] : j == 1 ? sc[1] : j == 2 ? sc[2] : sc[3];
                const float z = j == 0 ? zpb[0] : j == 1 ? zpb[1] : j == 2 ? zpb[2] : zpb[3];
                const idx_t o = (c0 + j) * kgroups_i + kg_i;
                a.scale[o] = s;
                a.zp[o] = static_cast<uint8_t>(static_cast<int32_t>(z) - bias);
            }
        }
    }
    if constexpr (!EMIT_Q) return;
    if constexpr (WPS > 0 && !ONECOL) {
        if (h == 0) *reinterpret_cast<float4*>(&s_scale[wave][cl * 4]) = make_float4(sc[0], sc[1], sc[2], sc[3]);
    }

    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
    // the scales of the four column slots for the exact-division fallback
    auto fallback_scales = [&](float (&se)[4]) {
        if constexpr (ONECOL) {
#pragma unroll
            for (int i = 0; i < 4; ++i) se[i] = s_scale[wave][i * LPR + cl];
        } else if constexpr (WPS > 0) {
            const float4 t4 = *reinterpret_cast<const float4*>(&s_scale[wave][cl * 4]);
            se[0] = t4.x; se[1] = t4.y; se[2] = t4.z; se[3] = t4.w;
        } else {
            se[0] = sc[0]; se[1] = sc[1]; se[2] = sc[2]; se[3] = sc[3];
        }
    };

    if constexpr (WPS > 0) {
        // K1 in the "magic number" domain, two elements per instruction.  With M = 1.5 * 2^23 the fp32 grid around M + k
        // has spacing 1, so u = fma(x, rinv, zp + bias + M) IS M + rint(x * rinv + zp + bias): the product, the zero point
        // and the rounding in ONE correctly rounded operation (round-half-even on the sum).  The residual
        // r = fma(x, rinv, (zp + bias + M) - u) is the exact distance of x * rinv + zp from that integer (one rounding of
        // <= 2^-25); whenever |r| < thr = 0.5 - B * 2^-21 (the band of oq_common.hpp: B >= max |x / s| of the group bounds
        // the distance between x * fl(1/s) and fl(x / s)) no half-integer separates the two and the integer is the
        // reference's rint(fl(x / s)) + zp; ties and everything inside the band (and NaN / inf) are redone with the IEEE
        // division.  Clamping happens on M + level as well and the level is the low byte of the float's bits:
        // v_pk_fma, v_pk_add, v_pk_fma, 2 compares, 2 v_med3 per PAIR of elements and 7 bit operations per packed word
        // instead of 6 scalar operations per element and 10 per word.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23, bits 0x4B400000
        const f32x2 rv[2] = {{rinv[0], rinv[1]}, {rinv[2], rinv[3]}};
        const f32x2 zm[2] = {{zpb[0] + kMagic, zpb[1] + kMagic}, {zpb[2] + kMagic, zpb[3] + kMagic}};
        const float lo_m = lo_b + kMagic, hi_m = hi_b + kMagic;
#if OQ_WAVE_GROUP > 1
        // one decision per OQ_WAVE_GROUP rows: the residuals fold into a running NaN-propagating maximum (one v_maximum3_f32 with |.|
        // modifiers per pair) instead of two compares and two scalar ORs per pair and a ballot + branch per row
#pragma unroll
        for (int rg = 0; rg < 16; rg += OQ_WAVE_GROUP) {
            float f[OQ_WAVE_GROUP][4];
            float far = 0.0f;
#pragma unroll
            for (int r = 0; r < OQ_WAVE_GROUP; ++r)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const f32x2 x = {v[rg + r][2 * p2], v[rg + r][2 * p2 + 1]};
                    const f32x2 u = __builtin_elementwise_fma(x, rv[p2], zm[p2]);
                    const f32x2 res = __builtin_elementwise_fma(x, rv[p2], zm[p2] - u);
                    far = nmax(far, nmax(fabsf(res.x), fabsf(res.y)));
                    f[r][2 * p2] = __builtin_amdgcn_fmed3f(u.x, lo_m, hi_m);
                    f[r][2 * p2 + 1] = __builtin_amdgcn_fmed3f(u.y, lo_m, hi_m);
                }
            if (__builtin_amdgcn_ballot_w64(!(far < thr)) != 0) {  // wave-uniform, rare: redo these rows with the IEEE divide
                float se[4];
                fallback_scales(se);
#pragma unroll
                for (int r = 0; r < OQ_WAVE_GROUP; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        f[r][i] = __uint_as_float(0x4B400000u + static_cast<uint32_t>(quantize_one(v[rg + r][i], se[i], static_cast<int32_t>(zpb[i]) - bias, qmin, qmax) + bias));
            }
#pragma unroll
            for (int r = 0; r < OQ_WAVE_GROUP; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rg + r][i] = f[r][i];
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float f[4];
            bool unsafe = false;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const f32x2 x = {v[r][2 * p2], v[r][2 * p2 + 1]};
                const f32x2 u = __builtin_elementwise_fma(x, rv[p2], zm[p2]);
                const f32x2 res = __builtin_elementwise_fma(x, rv[p2], zm[p2] - u);
                unsafe = unsafe || !(fabsf(res.x) < thr) || !(fabsf(res.y) < thr);
                f[2 * p2] = __builtin_amdgcn_fmed3f(u.x, lo_m, hi_m);
                f[2 * p2 + 1] = __builtin_amdgcn_fmed3f(u.y, lo_m, hi_m);
            }
            if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {  // wave-uniform, rare: redo this row with the IEEE divide
                float se[4];
                fallback_scales(se);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    f[i] = __uint_as_float(0x4B400000u + static_cast<uint32_t>(quantize_one(v[r][i], se[i], static_cast<int32_t>(zpb[i]) - bias, qmin, qmax) + bias));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[r][i] = f[i];
        }
#endif
        if (!col_ok || OQ_ATTR(a.nt, 16)) return;
        // qrules/_common.py:72-87: out-channel n, k-group kg -> G * bits / 8 bytes, k ascending, even k in the low nibble.
        // v[r][i] holds M + level: the level is byte 0 of its bits (the other three bytes are those of M).
        auto low_bytes = [](uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {   // [b0.0, b1.0, b2.0, b3.0]
            return __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
        };
        if (a.grid.bits == 4 && LPR == 8 && !OQ_ATTR(a.nt, 64)) {
            // 16-byte stores: the lane sets 2m and 2m + 1 of a 16-lane DPP row hold neighbouring 8-byte pieces of every
            // column slot.  The even set takes its partner's pieces of slots 0 and 1, the odd set its partner's pieces of
            // slots 2 and 3 (one row_ror:8 per dword), and every lane issues TWO 16-byte stores instead of four 8-byte ones.
            const uint32_t flip = bias ? 0x88888888u : 0u;
            uint32_t wq[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    uint32_t pr[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)   // byte 0 = odd row << 4 | even row (levels < 16, M's low byte is 0)
                        pr[j] = (__float_as_uint(v[wd * 8 + 2 * j + 1][i]) << 4) | __float_as_uint(v[wd * 8 + 2 * j][i]);
                    wq[i][wd] = low_bytes(pr[0], pr[1], pr[2], pr[3]) ^ flip;
                }
            const bool odd = (h & 1) != 0;
            auto swap8 = [](uint32_t x) { return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false)); };
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // this lane stores slot t (even set) or slot 2 + t (odd set); it sends the pieces of the slot its partner stores
                const uint32_t s0 = odd ? wq[t][0] : wq[2 + t][0], s1 = odd ? wq[t][1] : wq[2 + t][1];
                const uint32_t r0 = swap8(s0), r1 = swap8(s1);
                const uint32_t m0 = odd ? wq[2 + t][0] : wq[t][0], m1 = odd ? wq[2 + t][1] : wq[t][1];
                const u32x4 val = odd ? u32x4{r0, r1, m0, m1} : u32x4{m0, m1, r0, r1};
                const idx_t col = c0 + (odd ? 2 + t : t);
                u32x4* o = reinterpret_cast<u32x4*>(a.q + ((col * kgroups_i + kg_i) * (G / 2) + (h & ~1) * 8));
                if constexpr ((OQ_RTN_SC1 & 1) != 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(o), "v"(val) : "memory");
                else if (a.nt & 2) __builtin_nontemporal_store(val, o);
                else *o = val;
            }
        } else if (a.grid.bits == 4) {
            const uint32_t flip = bias ? 0x88888888u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t words[2];
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    uint32_t pr[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)   // byte 0 = odd row << 4 | even row (levels < 16, M's low byte is 0)
                        pr[j] = (__float_as_uint(v[wd * 8 + 2 * j + 1][i]) << 4) | __float_as_uint(v[wd * 8 + 2 * j][i]);
                    words[wd] = low_bytes(pr[0], pr[1], pr[2], pr[3]) ^ flip;
                }
                u32x2* o = reinterpret_cast<u32x2*>(a.q + (((c0 + i) * kgroups_i + kg_i) * (G / 2) + h * 8));
                if (OQ_ATTR(a.nt, 32))   // attribution builds only: every store instruction writes 512 contiguous bytes
                    o = reinterpret_cast<u32x2*>(a.q + ((kg * (a.N / (4 * LPR)) + strip0 / (4 * LPR)) * (2 * G * LPR) + i * 512 + lane * 8));
                const u32x2 t = {words[0], words[1]};
                if (a.nt & 2) __builtin_nontemporal_store(t, o);
                else *o = t;
            }
        } else {
            const uint32_t flip = bias ? 0x80808080u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t words[4];
#pragma unroll
                for (int wd = 0; wd < 4; ++wd)
                    words[wd] = low_bytes(__float_as_uint(v[wd * 4][i]), __float_as_uint(v[wd * 4 + 1][i]), __float_as_uint(v[wd * 4 + 2][i]),
                                          __float_as_uint(v[wd * 4 + 3][i])) ^ flip;
                u32x4* o = reinterpret_cast<u32x4*>(a.q + (((c0 + i) * kgroups_i + kg_i) * G + h * 16));
                const u32x4 t = {words[0], words[1], words[2], words[3]};
                if (a.nt & 2) __builtin_nontemporal_store(t, o);
                else *o = t;
            }
        }
        return;
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float f[4];
        bool unsafe = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = v[r][i] * rinv[i];
            const float k = rintf(t);
            unsafe = unsafe || !(fabsf(t - k) < thr);
            f[i] = __builtin_amdgcn_fmed3f(k + zpb[i], lo_b, hi_b);
        }
        if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {  // wave-uniform, rare: redo this row with the IEEE divide
            float se[4];
            fallback_scales(se);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                f[i] = static_cast<float>(quantize_one(v[r][i], se[i], static_cast<int32_t>(zpb[i]) - bias, qmin, qmax) + bias);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[r][i] = f[i];
    }
    if (!col_ok) return;

    // qrules/_common.py:72-87: out-channel n, k-group kg -> G * bits / 8 bytes, k ascending, even k in the low nibble
    if (a.grid.bits == 4) {
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
            u32x2* o = reinterpret_cast<u32x2*>(a.q + (((c0 + i) * kgroups_i + kg_i) * (G / 2) + h * 8));
            const u32x2 t = {words[0], words[1]};
            if (a.nt & 2) __builtin_nontemporal_store(t, o);
            else *o = t;
        }
    } else {
        const uint32_t flip = bias ? 0x80808080u : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t words[4];
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_cvt_pk_u8_f32(v[wd * 4 + j][i], j, acc);
                words[wd] = acc ^ flip;
            }
            u32x4* o = reinterpret_cast<u32x4*>(a.q + (((c0 + i) * kgroups_i + kg_i) * G + h * 16));
            const u32x4 t = {words[0], words[1], words[2], words[3]};
            if (a.nt & 2) __builtin_nontemporal_store(t, o);
            else *o = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-pass path (channel, tensor, groups too tall for registers).  Pass 1: per (row chunk, column)
// min / max.  Pass 2: fold the chunks of one group -> qparams.  Pass 3: elementwise K1.
// ---------------------------------------------------------------------------------------------
constexpr int kChunkRows = 128;  // 8 waves x 16 rows

struct RangeArgs {
    const float* W;
    int64_t K, N, ldw;
    int64_t g, kgroups, chunks;  // chunks per group = ceil(g / kChunkRows)
    float* pmin;                 // [kgroups*chunks, N]
    float* pmax;
    uint32_t ncol_tiles, nrow_tiles;  // nrow_tiles = kgroups*chunks
};

template <bool VEC4>
__global__ __launch_bounds__(kMaxWaves* kWave) void col_range_partial(const RangeArgs a) {
    __shared__ float4 s_mn[kMaxWaves][kWave];
    __shared__ float4 s_mx[kMaxWaves][kWave];
    constexpr int RPW = kChunkRows / kMaxWaves;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nblk = a.ncol_tiles * a.nrow_tiles;
    const uint32_t id = xcd_remap(blockIdx.x, nblk);
    const uint32_t col_tile = id / a.nrow_tiles;
    const uint32_t row_tile = id - col_tile * a.nrow_tiles;
    const int64_t kg = row_tile / a.chunks, c = row_tile % a.chunks;
    const int64_t row_end = min(kg * a.g + a.g, a.K);
    const int64_t row0 = kg * a.g + c * kChunkRows + wave * RPW;
    const int64_t tile_col0 = static_cast<int64_t>(col_tile) * kColsPerWave;

    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; }
    if constexpr (VEC4) {
        const bool ok = tile_col0 + lane * 4 < a.N;
        const float* p = a.W + row0 * a.ldw + tile_col0 + lane * 4;
        float4 t[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            t[r] = (ok && row0 + r < row_end) ? *reinterpret_cast<const float4*>(p + r * a.ldw)
                                              : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const bool live = ok && row0 + r < row_end;
            mn[0] = nmin(mn[0], t[r].x); mn[1] = nmin(mn[1], t[r].y);
            mn[2] = nmin(mn[2], t[r].z); mn[3] = nmin(mn[3], t[r].w);
            if (live) {
                mx[0] = nmax(mx[0], t[r].x); mx[1] = nmax(mx[1], t[r].y);
                mx[2] = nmax(mx[2], t[r].z); mx[3] = nmax(mx[3], t[r].w);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t col = slot_col<false>(tile_col0, lane, i);
                if (col < a.N && row0 + r < row_end) {
                    const float x = a.W[(row0 + r) * a.ldw + col];
                    mn[i] = nmin(mn[i], x);
                    mx[i] = nmax(mx[i], x);
                }
            }
    }
    s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
    s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < kMaxWaves; ++w) {
        const float4 tn = s_mn[w][lane], tx = s_mx[w][lane];
        mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y); mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
        mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y); mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t col = slot_col<VEC4>(tile_col0, lane, i);
        if (col < a.N) {
            a.pmin[static_cast<int64_t>(row_tile) * a.N + col] = mn[i];
            a.pmax[static_cast<int64_t>(row_tile) * a.N + col] = mx[i];
        }
    }
}

// Pass 2 (group / channel): thread per (kg, column).
__global__ void col_range_finalize(const float* pmin, const float* pmax, int64_t N, int64_t kgroups,
                                   int64_t chunks, QGrid grid, float* scale, uint8_t* zp) {
    const int64_t col = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t kg = blockIdx.y;
    if (col >= N) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t c = 0; c < chunks; ++c) {
        mn = nmin(mn, pmin[(kg * chunks + c) * N + col]);
        mx = nmax(mx, pmax[(kg * chunks + c) * N + col]);
    }
    const QParam p = qparam_from_minmax(mn, mx, grid);
    scale[col * kgroups + kg] = p.scale;
    zp[col * kgroups + kg] = static_cast<uint8_t>(p.zp);
}

// Pass 2 (tensor), step 1: per-column fold of all chunk partials, in place into row 0.
__global__ void col_fold_kernel(float* pmin, float* pmax, int64_t N, int64_t rows) {
    const int64_t col = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col >= N) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t r = 0; r < rows; ++r) {
        mn = nmin(mn, pmin[r * N + col]);
        mx = nmax(mx, pmax[r * N + col]);
    }
    pmin[col] = mn;
    pmax[col] = mx;
}

// Pass 2 (tensor), step 2: one block folds the N column values into one (scale, zp).
__global__ __launch_bounds__(1024) void tensor_range_finalize(const float* pmin, const float* pmax, int64_t count,
                                                              QGrid grid, float* scale, uint8_t* zp) {
    __shared__ float s_mn[16], s_mx[16];
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
        mn = nmin(mn, pmin[i]);
        mx = nmax(mx, pmax[i]);
    }
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
        const QParam p = qparam_from_minmax(mn, mx, grid);
        scale[0] = p.scale;
        zp[0] = static_cast<uint8_t>(p.zp);
    }
}

// Pass 3: q[k, n] = K1(W[k, n]; params[(k / g) + n * kgroups])  (tensor: one entry).  Same tiling as pass 1
// (128-row chunk x 256 columns, a chunk never straddles a group), parameters fetched once per lane, and the
// same exact-reciprocal fast path as the fused kernel (band from the worst-case |x / scale| of the grid:
// (qmax - qmin) / clip_ratio, resp. levels / clip_ratio).
struct QuantKnArgs {
    const float* W;
    int64_t K, N, ldw, g, kgroups, chunks;
    const float* scale;
    const uint8_t* zp;
    uint8_t* q;
    QGrid grid;
    int32_t zp_signed, tensor;
    uint32_t ncol_tiles, nrow_tiles;
    int32_t layout;   // OQ_LAYOUT_NBITS: groups that are a multiple of 128 rows, VEC4 only (checked by the host)
};

template <bool VEC4>
__global__ __launch_bounds__(kMaxWaves* kWave) void quantize_kn(const QuantKnArgs a) {
    constexpr int RPW = kChunkRows / kMaxWaves;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t row_tile = blockIdx.x / a.ncol_tiles;          // column tiles fastest
    const uint32_t col_tile = blockIdx.x - row_tile * a.ncol_tiles;
    const int64_t kg = row_tile / a.chunks, c = row_tile % a.chunks;
    const int64_t row_end = min(kg * a.g + a.g, a.K);
    const int64_t row0 = kg * a.g + c * kChunkRows + wave * RPW;
    const int64_t tile_col0 = static_cast<int64_t>(col_tile) * kColsPerWave;
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? 128 : 0;
    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
    // x 1.25: the MSE search (rtn_mse.hip) hands over parameters of a range shrunk by p >= 0.81
    const float worst = 1.25f * (a.grid.symmetric ? static_cast<float>(a.grid.levels) : static_cast<float>(qmax - qmin)) / a.grid.clip_ratio;

    ColQ cq[4];
    bool col_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t col = slot_col<VEC4>(tile_col0, lane, i);
        col_ok[i] = col < a.N;
        const int64_t pi = a.tensor ? 0 : (col_ok[i] ? col : a.N - 1) * a.kgroups + kg;
        QParam p;
        p.scale = a.scale[pi];
        p.zp = a.zp_signed ? static_cast<int32_t>(static_cast<int8_t>(a.zp[pi])) : static_cast<int32_t>(a.zp[pi]);
        // raw extrema are not kept by pass 2; |x| <= worst * scale holds for every element of the group
        cq[i] = make_colq(p, -worst * p.scale, worst * p.scale, bias);
    }
    const uint32_t flip = bias ? 0x80808080u : 0u;
    if constexpr (VEC4) {
        int64_t lcol = tile_col0 + lane * 4;
        lcol = lcol < a.N ? lcol : a.N - 4;
        float4 t[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int64_t row = row0 + r < row_end ? row0 + r : row_end - 1;   // clamped, never predicated
            const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.W + row * a.ldw + lcol));
            t[r] = make_float4(u[0], u[1], u[2], u[3]);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float xs[4] = {t[r].x, t[r].y, t[r].z, t[r].w};
            float f[4];
            bool unsafe = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = quantize_fast_biased(xs[i], cq[i], lo_b, hi_b, unsafe);
            if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = quantize_exact_biased(xs[i], cq[i], qmin, qmax, bias);
            }
            if (a.layout == OQ_LAYOUT_NBITS) {   // keep the levels; packed per column below
                t[r] = make_float4(f[0], f[1], f[2], f[3]);
                continue;
            }
            uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(f[0], 0, 0);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[1], 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[2], 2, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[3], 3, w);
            if (col_ok[0] && row0 + r < row_end)
                __builtin_nontemporal_store(w ^ flip, reinterpret_cast<uint32_t*>(a.q + (row0 + r) * a.N + tile_col0 + lane * 4));
        }
        if (a.layout == OQ_LAYOUT_NBITS && col_ok[0]) {
            // qrules/_common.py:72-87: out-channel n, k-group kg -> g * bits / 8 bytes, k ascending, even k in the low nibble;
            // this wave owns rows [c * 128 + wave * 16, +16) of the group
            const int64_t blob = a.g * a.grid.bits / 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* lv = reinterpret_cast<const float*>(t) + i;   // t[r].{x,y,z,w}: stride 4 floats
                uint8_t* o = a.q + ((tile_col0 + lane * 4 + i) * a.kgroups + kg) * blob;
                if (a.grid.bits == 4) {
                    // signed levels are biased by 128 here: the low nibble of (level + 128) already is the two's-complement nibble
                    uint32_t words[2];
#pragma unroll
                    for (int wd = 0; wd < 2; ++wd) {
                        uint32_t ev = 0, od = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ev = __builtin_amdgcn_cvt_pk_u8_f32(lv[(wd * 8 + 2 * j) * 4], j, ev);
                            od = __builtin_amdgcn_cvt_pk_u8_f32(lv[(wd * 8 + 2 * j + 1) * 4], j, od);
                        }
                        words[wd] = (ev & 0x0f0f0f0fu) | ((od & 0x0f0f0f0fu) << 4);
                    }
                    *reinterpret_cast<uint2*>(o + c * (kChunkRows / 2) + wave * (RPW / 2)) = make_uint2(words[0], words[1]);
                } else {
                    uint32_t words[4];
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        uint32_t acc = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_cvt_pk_u8_f32(lv[(wd * 4 + j) * 4], j, acc);
                        words[wd] = acc ^ flip;
                    }
                    *reinterpret_cast<uint4*>(o + c * kChunkRows + wave * RPW) = make_uint4(words[0], words[1], words[2], words[3]);
                }
            }
        }
    } else {
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t col = slot_col<false>(tile_col0, lane, i);
                if (col_ok[i] && row0 + r < row_end)
                    a.q[(row0 + r) * a.N + col] = static_cast<uint8_t>(quantize_one(a.W[(row0 + r) * a.ldw + col], cq[i].scale, cq[i].zp, qmin, qmax));
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Ragged fallback: K % g != 0 but N*K % g == 0, i.e. W.T.reshape(-1, g) lets a group run from the
// tail of one column into the head of the next (utils.py:24).  One wave per group, strided gathers.
// Correctness path only; the reference's product path never produces it (qrules/_common.py:13-29).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rtn_flat_groups(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t g,
                                                       int64_t ngroups, QGrid grid, uint8_t* q, float* scale,
                                                       uint8_t* zp) {
    const int lane = threadIdx.x & 63;
    const int64_t grp = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (grp >= ngroups) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t t = lane; t < g; t += 64) {
        const int64_t f = grp * g + t, n = f / K, k = f - n * K;
        const float x = W[k * ldw + n];
        mn = nmin(mn, x);
        mx = nmax(mx, x);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const QParam p = qparam_from_minmax(mn, mx, grid);
    if (lane == 0) {
        scale[grp] = p.scale;
        zp[grp] = static_cast<uint8_t>(p.zp);
    }
    if (q == nullptr) return;
    for (int64_t t = lane; t < g; t += 64) {
        const int64_t f = grp * g + t, n = f / K, k = f - n * K;
        q[k * N + n] = static_cast<uint8_t>(quantize_one(W[k * ldw + n], p.scale, p.zp, grid.qmin, grid.qmax));
    }
}

// [kgroups, N] staging -> the reference's n-major [N*K/g] arrays (entry n*kgroups + kg), 32x32 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_qparams(const float* scale_t, const uint8_t* zp_t, int64_t kgroups, int64_t N,
                                                         float* scale, uint8_t* zp, const RtnPtrs* table) {
    __shared__ float ts[32][33];
    __shared__ uint8_t tz[32][36];
    {   // blockIdx.z = matrix of the launch: the staging advances by kgroups * N entries; the outputs do so too for a
        // strided batch and come from the entry's own pointers for a list of matrices (oq_rtn_quantize_ptrs_f32)
        const int64_t off = static_cast<int64_t>(blockIdx.z) * kgroups * N;
        scale_t += off; zp_t += off;
        if (table != nullptr) { scale = table[blockIdx.z].scale; zp = table[blockIdx.z].zp; }
        else { scale += off; zp += off; }
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t n0 = static_cast<int64_t>(blockIdx.x) * 32, k0 = static_cast<int64_t>(blockIdx.y) * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t kg = k0 + ty + j, n = n0 + tx;
        if (kg < kgroups && n < N) {
            ts[ty + j][tx] = scale_t[kg * N + n];
            tz[ty + j][tx] = zp_t[kg * N + n];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t n = n0 + ty + j, kg = k0 + tx;
        if (kg < kgroups && n < N) {
            scale[n * kgroups + kg] = ts[tx][ty + j];
            zp[n * kgroups + kg] = tz[tx][ty + j];
        }
    }
}

// Experiment knobs: speed only, every setting produces the same bytes (-1 = the tuned default).
struct Tuning {
    int order = -1, gk = -1, nt = -1, stage = -1, stage_q = -1, wavek = -1, gpb = -1, wpb = -1, wps = -1, resident = -1, xg = -1;
    static int env_int(const char* name) { const char* v = getenv(name); return v ? atoi(v) : -1; }
    static Tuning from_env() {
        Tuning t;
        t.order = env_int("OQ_RTN_ORDER");      // 0 K-fastest + XCD strips, 1 column tiles fastest, 2 L2-merging blocks
        t.gk = env_int("OQ_RTN_GK");            // row tiles per id block of order 2
        t.nt = env_int("OQ_RTN_NT");            // bit 0 non-temporal W loads and [K,N] stores, bit 1 non-temporal blob stores (masked with kNtMask)
        t.stage = env_int("OQ_RTN_STAGE");      // stage (scale, zp) as [K/g, N] + transpose launch
        t.stage_q = env_int("OQ_RTN_STAGE_Q");  // assemble blob chunks in LDS
        t.wavek = env_int("OQ_RTN_WAVEK");      // blob layout: wave-owns-group kernel (1) or the block kernel (0)
        t.gpb = env_int("OQ_RTN_GPB");          // wave kernel: k-groups per block
        t.wpb = env_int("OQ_RTN_WPB");          // wave kernel: waves per block
        t.wps = env_int("OQ_RTN_WPS");          // wave kernel: 0 = the 4-waves-per-SIMD build, 5 = the 92-register build (5 per SIMD)
        t.resident = env_int("OQ_RTN_RESIDENT"); // channel / tensor / tall groups: 0 = the three-launch path that reads W twice
        t.xg = env_int("OQ_RTN_XG");            // order 2: log2 of the neighbouring column tiles one XCD owns (speed only)
        return t;
    }
};

// ------------------------------------------------------------------------------------ dispatch
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Rows per wave / waves per group for the fused kernel; false when the group is too tall.
static bool fused_shape(int64_t g, int* rpw, int* wpg) {
    for (int r : {16, 8, 4, 2, 1}) {
        if (g % r == 0 && g / r <= kMaxWaves) {
            *rpw = r;
            *wpg = static_cast<int>(g / r);
            return true;
        }
    }
    if (g % 32 == 0 && g / 32 <= kMaxWaves) {
        *rpw = 32;
        *wpg = static_cast<int>(g / 32);
        return true;
    }
    return false;
}

// Fills the id-division constants of the chosen block order; ids beyond the fast-division range use order 0.
static void set_block_order(RtnArgs& a) {
    if (a.gk < 1) a.gk = 1;
    const uint64_t nblk = static_cast<uint64_t>(a.ncol_tiles) * a.nrow_tiles;
    if (nblk >= (1u << 22)) a.order = 0;
    a.fd_ncol = make_fastdiv(a.ncol_tiles);
    a.fd_band = make_fastdiv(a.ncol_tiles * static_cast<uint32_t>(a.gk));
    a.fd_chunk = make_fastdiv((8u << a.xg_log2) * static_cast<uint32_t>(a.gk));
}

template <bool VEC4, bool EMIT_Q>
static void launch_fused(int rpw, const RtnArgs& a, dim3 grid, dim3 block, hipStream_t s) {
    switch (rpw) {
        case 1: hipLaunchKernelGGL((rtn_group_fused<1, VEC4, EMIT_Q>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((rtn_group_fused<2, VEC4, EMIT_Q>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((rtn_group_fused<4, VEC4, EMIT_Q>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((rtn_group_fused<8, VEC4, EMIT_Q>), grid, block, 0, s, a); break;
        case 16:
            if constexpr (VEC4 && EMIT_Q) {
                if (a.tr_scale != nullptr) {      // the host sets it only with rpw == 16, eight waves per group, vec4, emit_q
                    if (a.nt) hipLaunchKernelGGL((rtn_group_fused<16, true, true, true, true>), grid, block, 0, s, a);
                    else hipLaunchKernelGGL((rtn_group_fused<16, true, true, false, true>), grid, block, 0, s, a);
                    break;
                }
            }
            if (a.nt) hipLaunchKernelGGL((rtn_group_fused<16, VEC4, EMIT_Q, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((rtn_group_fused<16, VEC4, EMIT_Q, false>), grid, block, 0, s, a);
            break;
        default: hipLaunchKernelGGL((rtn_group_fused<32, VEC4, EMIT_Q>), grid, block, 0, s, a); break;
    }
}

// the zeroed state of the in-launch transposition: [kgroups, N] fp32 scales + one 8-byte {four zero points, marker} granule per four columns
static size_t fused_state_bytes(int64_t K, int64_t N, int64_t g) {
    const int64_t kgroups = K / g;
    return static_cast<size_t>((kgroups * N * 4 + 255) / 256 * 256 + kgroups * N * 2 + 256);
}
static size_t stage_ws(int64_t K, int64_t N, int64_t g) {  // [kgroups, N] fp32 scales + bytes, 256-byte aligned halves
    const int64_t kgroups = K / g;
    return static_cast<size_t>((kgroups * N * 4 + 255) / 256 * 256 + (kgroups * N + 255) / 256 * 256);
}

static size_t twopass_ws(int64_t K, int64_t N, int64_t g) {
    const int64_t kgroups = K / g, chunks = ceil_div(g, kChunkRows);
    return static_cast<size_t>(2 * kgroups * chunks * N) * sizeof(float);
}

static int32_t resolve_group(int32_t strategy, int64_t K, int64_t group_size, int64_t* g) {
    if (strategy == OQ_GROUP) {
        OQ_REQUIRE(group_size > 0 || group_size == -1, OQ_ERR_INVALID_ARGUMENT,
                   "group strategy needs group_size > 0 or -1, got %lld", (long long)group_size);
        int64_t gs = group_size > K ? K : group_size;  // utils.py:19-20
        if (gs == -1) gs = K;                          // utils.py:22
        *g = gs;
    } else if (strategy == OQ_CHANNEL || strategy == OQ_TENSOR) {
        *g = K;
    } else {
        return fail(OQ_ERR_INVALID_ARGUMENT, "unknown strategy %d", strategy);
    }
    return OQ_OK;
}

int32_t rtn_impl(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                 int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio, int32_t mse,
                 void* q_out, float* scale_out, void* zp_out, int32_t layout, void* workspace,
                 size_t workspace_bytes, void* stream, bool emit_q);

}  // namespace oq

// MSE search lives in rtn_mse.hip, the one-read channel / tensor kernels in rtn_resident.hip
namespace oq {
bool rtn_resident_eligible(int64_t K, int64_t N, int64_t ldw, const float* W, const void* q, int32_t strategy, int64_t g, int32_t layout,
                           bool emit_q, size_t workspace_bytes);
bool rtn_stream_is_capturing(hipStream_t s);
int32_t ticket_chain_begin(hipStream_t s);
void ticket_chain_end(hipStream_t s);
int32_t rtn_resident_impl(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy, int64_t g, uint8_t* q,
                          float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s, bool zeroed_state);
size_t rtn_resident_workspace(int64_t K, int64_t N, int32_t strategy, int64_t g);
int32_t rtn_mse_impl(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy,
                     int64_t g, void* q_out, float* scale_out, void* zp_out, int32_t zp_signed, void* workspace,
                     size_t workspace_bytes, hipStream_t s, bool emit_q);
size_t rtn_mse_workspace(int64_t K, int64_t N, int32_t strategy, int64_t g);

// Pass 3 of the two-pass path (also the final pass of the MSE search): K1 with stored parameters.
int32_t launch_quantize_kn(const float* W, int64_t K, int64_t N, int64_t ldw, int64_t g, int64_t kgroups, const float* scale,
                           const uint8_t* zp, uint8_t* q, const QGrid& grid, int32_t zp_signed, bool tensor, hipStream_t s,
                           int32_t layout) {
    QuantKnArgs qa;
    qa.layout = layout;
    qa.W = W; qa.K = K; qa.N = N; qa.ldw = ldw; qa.g = g; qa.kgroups = kgroups;
    qa.chunks = ceil_div(g, kChunkRows);
    qa.scale = scale; qa.zp = zp; qa.q = q;
    qa.grid = grid; qa.zp_signed = zp_signed; qa.tensor = tensor;
    qa.ncol_tiles = static_cast<uint32_t>(ceil_div(N, kColsPerWave));
    qa.nrow_tiles = static_cast<uint32_t>(kgroups * qa.chunks);
    const bool vec4 = (N % 4 == 0) && (ldw % 4 == 0) && aligned16(W) && (reinterpret_cast<uintptr_t>(q) & 3u) == 0;
    OQ_REQUIRE(layout == OQ_LAYOUT_KN || (vec4 && g % kChunkRows == 0 && aligned16(q) && !tensor), OQ_ERR_UNSUPPORTED,
               "NBITS layout with group_size > 256 needs group_size %% 128 == 0, N %% 4 == 0 and 16-byte aligned buffers");
    const dim3 grid_dim(qa.ncol_tiles * qa.nrow_tiles), block(kMaxWaves * kWave);
    if (vec4) hipLaunchKernelGGL(quantize_kn<true>, grid_dim, block, 0, s, qa);
    else hipLaunchKernelGGL(quantize_kn<false>, grid_dim, block, 0, s, qa);
    return check_launch("quantize_kn");
}

// oq_rtn_quantize_ptrs_f32: matrices of one shape per launch.  Measured with a model's worth of weights and DISTINCT outputs
// in HBM (scripts/quick_many3.py; uint4 g128 blob, fraction of the 8 TB/s peak at 1 / best / all matrices per launch):
//   2048 x 2048   0.18 / 0.73 (16) / 0.70      4096 x 4096    0.54 / 0.72 (8-16) / 0.69      8192 x 8192  0.56 / 0.59 / 0.59
//   4096 x 11008  0.62 / 0.63 (2-4) / 0.58     11008 x 4096   0.64 / 0.67 (4-8) / 0.63
// A lone small matrix cannot fill the chip (4096 waves on 5120 slots all start and end together); very long merged launches
// of large matrices lose a little again (dirty output lines evicted between another matrix' reads instead of written back
// in a burst at a kernel's end).  ~1.6e8 parameters per launch sits at or next to the best point of every row.
static int64_t matrices_per_launch(int64_t K, int64_t N, int64_t count) {
    static const int64_t per_launch = [] {      // OQ_RTN_MPL: parameters per launch in millions (lab; speed only)
        const char* v = getenv("OQ_RTN_MPL");
        const long x = v ? atol(v) : 0;
        return x > 0 ? static_cast<int64_t>(x) * 1000000 : static_cast<int64_t>(160000000);
    }();
    int64_t m = per_launch / (K * N);
    if (m > 65535) m = 65535;   // blockIdx.y
    if (m < 1) m = 1;
    return m < count ? m : count;
}

// Strided batch of equally shaped matrices for the fused group kernel (set by oq_rtn_quantize_batched_f32 around
// its call of rtn_impl; 1 matrix otherwise).
struct BatchCtx { int64_t count = 1, w_stride = 0, q_stride = 0; const RtnPtrs* table = nullptr; };
static thread_local BatchCtx g_batch;
// oq_rtn_quantize_stateful_f32: the caller's zeroed, self-cleaning state for the one-read channel / tensor kernels
struct StateCtx { void* state = nullptr; size_t bytes = 0; };
static thread_local StateCtx g_state;

int32_t rtn_impl(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                 int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio, int32_t mse,
                 void* q_out, float* scale_out, void* zp_out, int32_t layout, void* workspace,
                 size_t workspace_bytes, void* stream, bool emit_q) {
    OQ_REQUIRE(W && scale_out && zp_out && (q_out || !emit_q), OQ_ERR_INVALID_ARGUMENT, "rtn: null pointer argument");
    OQ_REQUIRE(K > 0 && N > 0 && ldw >= N, OQ_ERR_INVALID_ARGUMENT, "rtn: bad shape K=%lld N=%lld ldw=%lld",
               (long long)K, (long long)N, (long long)ldw);
    OQ_REQUIRE(matrix_ok(K, N, ldw), OQ_ERR_UNSUPPORTED, "rtn: matrix too large (K=%lld N=%lld ldw=%lld)", (long long)K, (long long)N, (long long)ldw);
    OQ_REQUIRE(clip_ratio > 0.0f && clip_ratio <= 1.0f, OQ_ERR_INVALID_ARGUMENT,
               "clip_ratio must be in (0.0, 1.0], got %g", clip_ratio);
    OQ_REQUIRE(layout == OQ_LAYOUT_KN || layout == OQ_LAYOUT_NBITS || layout == OQ_LAYOUT_KN_PACKED4, OQ_ERR_INVALID_ARGUMENT,
               "rtn: bad layout %d", layout);
    QGrid grid;
    int32_t st = make_grid(qtype, symmetric, reduce_range, clip_ratio, &grid);
    if (st != OQ_OK) return st;
    int64_t g;
    st = resolve_group(strategy, K, group_size, &g);
    if (st != OQ_OK) return st;
    hipStream_t s = as_stream(stream);
    const int32_t zp_signed = (qtype == OQ_INT4 || qtype == OQ_INT8) ? 1 : 0;
    uint8_t* q8 = static_cast<uint8_t*>(q_out);
    uint8_t* zp8 = static_cast<uint8_t*>(zp_out);

    if (layout == OQ_LAYOUT_KN_PACKED4) {
        // two columns per byte, written by the fused group kernel's epilogue; everything else is refused loudly, never routed
        // through an unpacked result + a second launch
        int rpw_p = 0, wpg_p = 0;
        OQ_REQUIRE(emit_q && grid.bits == 4 && strategy == OQ_GROUP && K % g == 0 && fused_shape(g, &rpw_p, &wpg_p) && !mse, OQ_ERR_UNSUPPORTED,
                   "KN_PACKED4 layout needs a 4-bit type, the group strategy with K %% group_size == 0 and group_size <= 256, no mse");
        OQ_REQUIRE(N % 4 == 0 && ldw % 4 == 0 && aligned16(W) && (reinterpret_cast<uintptr_t>(q_out) & 3u) == 0, OQ_ERR_UNSUPPORTED,
                   "KN_PACKED4 layout needs N %% 4 == 0, ldw %% 4 == 0, a 16-byte aligned W and a 4-byte aligned output");
    }
    if (strategy == OQ_GROUP && K % g != 0) {
        OQ_REQUIRE((K * N) % g == 0, OQ_ERR_INVALID_ARGUMENT,
                   "cannot reshape array of size %lld into rows of %lld", (long long)(K * N), (long long)g);
        OQ_REQUIRE(!mse, OQ_ERR_UNSUPPORTED, "mse with groups that straddle columns (K %% group_size != 0) is not supported");
        OQ_REQUIRE(layout == OQ_LAYOUT_KN, OQ_ERR_UNSUPPORTED, "NBITS layout needs K %% group_size == 0");
        const int64_t ngroups = K * N / g;
        hipLaunchKernelGGL(rtn_flat_groups, dim3(static_cast<uint32_t>(ceil_div(ngroups, 4))), dim3(256), 0, s, W, K, N,
                           ldw, g, ngroups, grid, emit_q ? q8 : nullptr, scale_out, zp8);
        return check_launch("rtn_flat_groups");
    }

    if (mse) {
        OQ_REQUIRE(layout == OQ_LAYOUT_KN, OQ_ERR_UNSUPPORTED, "NBITS layout with mse is not supported");
        return rtn_mse_impl(W, K, N, ldw, grid, strategy, g, q_out, scale_out, zp_out, zp_signed, workspace,
                            workspace_bytes, s, emit_q);
    }

    const bool vec4 = (N % 4 == 0) && (ldw % 4 == 0) && aligned16(W) && (!emit_q || (reinterpret_cast<uintptr_t>(q_out) & 3u) == 0);
    const int64_t kgroups = K / g;
    int rpw = 0, wpg = 0;
    const bool fused = strategy == OQ_GROUP && fused_shape(g, &rpw, &wpg);
    if (layout == OQ_LAYOUT_NBITS) {
        OQ_REQUIRE(strategy == OQ_GROUP && emit_q, OQ_ERR_UNSUPPORTED, "NBITS layout needs the group strategy");
        OQ_REQUIRE(g % 16 == 0 && aligned16(q_out), OQ_ERR_UNSUPPORTED,
                   "NBITS layout needs group_size %% 16 == 0 and a 16-byte aligned output");
    }

    // MatMulNBits blob with a group of 32 / 64 / 128 rows: wave-owns-group kernel (no LDS, no barrier)
    {
        static const Tuning tw = Tuning::from_env();
        const bool wave_ok = strategy == OQ_GROUP && layout == OQ_LAYOUT_NBITS && vec4 && (g == 32 || g == 64 || g == 128) && tw.wavek != 0 &&
                             ldw < (1 << 22);   // 32-bit lane offsets
        if (wave_ok) {
            OQ_REQUIRE(emit_q && aligned16(q_out), OQ_ERR_UNSUPPORTED, "NBITS layout needs a 16-byte aligned output");
            RtnArgs a;
            a.W = W; a.K = K; a.N = N; a.ldw = ldw; a.g = g; a.kgroups = kgroups;
            a.q = q8; a.scale = scale_out; a.zp = zp8; a.grid = grid; a.layout = layout;
            a.scale_t = nullptr; a.zp_t = nullptr; a.wpg = 1; a.stage_q = 0; a.pair_owner = 0;
            a.tr_scale = nullptr; a.tr_zp = nullptr;
            const int64_t batch = g_batch.count;
            a.w_stride = g_batch.w_stride; a.q_stride = g_batch.q_stride; a.p_stride = N * kgroups; a.table = g_batch.table;
            const int lpr = static_cast<int>(1024 / g);
            int wpb = tw.wpb > 0 ? tw.wpb : 4;
            if (wpb > kMaxWaves) wpb = kMaxWaves;
            while (wpb & (wpb - 1)) wpb &= wpb - 1;   // power of two
            int gpb = tw.gpb > 0 ? tw.gpb : 1;
            while (gpb & (gpb - 1)) gpb &= gpb - 1;
            while (gpb > 1 && (gpb > wpb || gpb > kgroups)) gpb >>= 1;
            a.gpb = gpb;
            a.spb_log2 = 0;
            while ((1 << (a.spb_log2 + 1)) <= wpb / gpb) ++a.spb_log2;
            const int64_t nstrips = ceil_div(N, 4 * lpr);
            a.ncol_tiles = static_cast<uint32_t>(ceil_div(nstrips, wpb / gpb));
            a.nrow_tiles = static_cast<uint32_t>(ceil_div(kgroups, gpb));
            a.order = tw.order >= 0 ? tw.order : 2;
            a.gk = tw.gk > 0 ? tw.gk : 4;
            a.nt = (tw.nt >= 0 ? tw.nt : 1) & kNtMask;
            // Column tiles (512 B of a row) that one XCD owns side by side inside a chunk of the L2-merging order.  One tile per XCD
            // leaves an XCD a fixed 512-byte residue of every 4 KB of a row band; two neighbours (1 KB) measured 2-5 % faster on
            // every Llama width but the multiples of 16384 (4096: 18.0 -> 17.2 us, 11008: 40.1 -> 38.8, 28672: 100.4 -> 96.3,
            // 32000: 122.3 -> 116.4; 16384: 56.9 -> 57.7), eight (4 KB) on rows of 32 KB (8192: 32.3 -> 30.0, 8192 x 8192:
            // 62.4 -> 56.6): scripts/lab_order_sweep.sh, docs/LAB_NOTES_r05.md.  Speed only.
            // (g = 128; a tile of the g = 64 / 32 builds is 1 / 2 KB wide already: 4096 x 11008 g = 64 43.1 us with one tile, 44.6 with two)
            {
                const int want_log2 = N % 16384 == 0 ? 9 : (N % 8192 == 0 ? 12 : 10);      // bytes of a row one XCD owns side by side
                int tile_log2 = 6;                                                          // 64 * lpr bytes: four strips of 4 * lpr columns
                for (int l = lpr; l > 1; l >>= 1) ++tile_log2;
                const int rule = want_log2 > tile_log2 ? want_log2 - tile_log2 : 0;
                a.xg_log2 = tw.xg >= 0 ? static_cast<uint32_t>(tw.xg > 4 ? 4 : tw.xg) : static_cast<uint32_t>(rule);
            }
            set_block_order(a);
            const dim3 grid_dim(a.ncol_tiles * a.nrow_tiles, static_cast<uint32_t>(batch)), block(static_cast<uint32_t>(wpb * kWave));
            // the 5-waves-per-SIMD build addresses its outputs with 32-bit offsets and needs blocks of <= 4 waves
            const bool small_idx = static_cast<uint64_t>(N) * static_cast<uint64_t>(K) * static_cast<uint64_t>(grid.bits) / 8u < (1ull << 32);
            const int wps = tw.wps >= 0 ? tw.wps : kDefaultWps;
            if (wps > 0 && small_idx && wpb <= 4 && lpr <= 16) {   // g = 32 (two lane sets) does not fit 96 registers: plain build
                if (lpr == 8) hipLaunchKernelGGL((rtn_group_wave<8, true, 5>), grid_dim, block, 0, s, a);
                else hipLaunchKernelGGL((rtn_group_wave<16, true, 5>), grid_dim, block, 0, s, a);
                return check_launch("rtn_group_wave<5 waves per SIMD>");
            }
            switch (lpr) {
                case 8: hipLaunchKernelGGL((rtn_group_wave<8, true>), grid_dim, block, 0, s, a); break;
                case 16: hipLaunchKernelGGL((rtn_group_wave<16, true>), grid_dim, block, 0, s, a); break;
                default: hipLaunchKernelGGL((rtn_group_wave<32, true>), grid_dim, block, 0, s, a); break;
            }
            return check_launch("rtn_group_wave");
        }
    }

    if (fused) {
        RtnArgs a;
        a.W = W; a.K = K; a.N = N; a.ldw = ldw; a.g = g; a.kgroups = kgroups;
        a.q = q8; a.scale = scale_out; a.zp = zp8; a.grid = grid; a.layout = layout;
        a.scale_t = nullptr; a.zp_t = nullptr;
        const int64_t batch = g_batch.count;
        a.w_stride = g_batch.w_stride; a.q_stride = g_batch.q_stride; a.p_stride = N * kgroups; a.table = g_batch.table;
        static const Tuning tune_s = Tuning::from_env();
        // [K,N] layouts: the n-major parameters either leave the kernel as they are (4-byte stores 128 bytes apart) or are
        // staged [K/g, N] and transposed by a second launch (~5 us of kernel + a boundary, whatever the size).  Rounds 1-4
        // staged always -- tuned on 4096 x 11008, the one Llama width where that is right: its 43 column tiles are odd, so
        // with column-fastest ids the k-groups of a column run on eight different XCDs and their 4-byte pieces of a line
        // never meet in one L2.  Round 5 (scripts/lab_knob_sweep.sh, STAGE=0, [K,N] bytes, K = 4096): N = 256 11.6 -> 8.5 us,
        // 2048 15.0 -> 11.6, 4096 22.4 -> 19.3, 5120 25.7 -> 23.7, 8192 35.0 -> 32.2, 11264 45.4 -> 43.6, 14336 54.4 -> 51.9,
        // 28672 101.6 -> 99.5; 11008 44.8 -> 46.6, 27648 98.1 -> 100.1, 32000 112.4 -> 124.3.  Direct when the column tiles
        // are a multiple of eight (every k-group of a column on one XCD), or the parameters are few enough for the fixed
        // cost of the second launch to dominate.  Speed only.
        const int64_t nct = ceil_div(N, kColsPerWave), nparams = kgroups * N;
        const bool direct = nct % 8 == 0 || nparams <= (256 << 10) || (nct % 2 == 0 && nparams <= (512 << 10));
        const bool want_stage = tune_s.stage >= 0 ? tune_s.stage != 0 : (layout != OQ_LAYOUT_NBITS && !direct);
        const bool staged = want_stage && vec4 && kgroups > 1 && workspace != nullptr &&
                            workspace_bytes >= static_cast<size_t>(batch) * stage_ws(K, N, g) && (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0;
        if (staged) {
            a.scale_t = static_cast<float*>(workspace);
            a.zp_t = static_cast<uint8_t*>(workspace) + (batch * kgroups * N * 4 + 255) / 256 * 256;
        }
        // Round 6: with the caller's zeroed state (oq_rtn_quantize_stateful_f32) the staged parameters are transposed inside the
        // launch, by blocks appended to the grid, instead of by a second launch (see transposer_block)
        a.tr_scale = nullptr; a.tr_zp = nullptr;
        // Measured (scripts/lab_kn_inlaunch.py, same box, staged + launch -> in the launch, us; the shipped transposer: one block per 128
        // columns, two granules per thread in flight): packed nibbles 4096 x 11008 44.2-44.3 -> 41.3, 4096 x 11000 48.6 -> 45.3-45.4,
        // 8192 x 11008 (two passes of 32 k-groups) 74.5-75.1 -> 73.4-74.1, 4096 x 27648 91.2-93.1 -> 91.2-91.6, 4096 x 32000 108.0-108.2 ->
        // 106.7-107.5; [K,N] bytes int8 4096 x 11008 45.5-45.6 -> 43.2, 8192 x 11008 79.5-79.9 -> 78.8-79.0, 4096 x 27648 98.2-98.6 ->
        // 97.7-98.3, 4096 x 32000 112.8-113.1 -> 113.7 (the one loss).  More than two passes put their round trips on the tail of the
        // launch: up to 64 k-groups.  Speed only.
        const bool in_launch = staged && tune_s.stage < 0 && emit_q && (layout == OQ_LAYOUT_KN_PACKED4 || layout == OQ_LAYOUT_KN) && kgroups <= 64 &&
                               rpw == 16 && wpg == kMaxWaves && batch == 1 && g_batch.table == nullptr &&
                               kgroups % 4 == 0 && N % 4 == 0 && g_state.state != nullptr && g_state.bytes >= fused_state_bytes(K, N, g) &&
                               (reinterpret_cast<uintptr_t>(scale_out) & 15u) == 0 && (reinterpret_cast<uintptr_t>(zp8) & 3u) == 0 &&
                               !rtn_stream_is_capturing(s);
        if (in_launch) {
            a.tr_scale = static_cast<float*>(g_state.state);
            a.tr_zp = reinterpret_cast<uint2*>(static_cast<uint8_t*>(g_state.state) + (kgroups * N * 4 + 255) / 256 * 256);
            a.scale_t = nullptr; a.zp_t = nullptr;
        }
        a.wpg = wpg;
        a.pair_owner = 0;
        for (uint32_t pr = 0; pr < 8; ++pr) a.pair_owner |= (pr % static_cast<uint32_t>(wpg)) << (3 * pr);
        a.gpb = kMaxWaves / wpg > 0 ? kMaxWaves / wpg : 1;
        if (a.gpb > kgroups) a.gpb = static_cast<int32_t>(kgroups);
        a.ncol_tiles = static_cast<uint32_t>(ceil_div(N, kColsPerWave));
        a.nrow_tiles = static_cast<uint32_t>(ceil_div(kgroups, a.gpb));
        // Tuned on 4096x11008 (profiles/r01_rtn_knob_sweep.txt).  [K,N] bytes: plain column-fastest ids, parameters
        // staged [K/g, N] + transpose launch.  MatMulNBits blob: L2-merging id order (its 64-byte half lines and the
        // 4-byte scales of neighbouring k-groups meet in one L2) + direct n-major parameter stores.  The OQ_RTN_*
        // environment variables override these choices for experiments only.
        static const Tuning tune = Tuning::from_env();
        const bool blob = layout == OQ_LAYOUT_NBITS;
        a.order = tune.order >= 0 ? tune.order : (blob ? 2 : 1);      // KN and KN_PACKED4 alike: plain column-fastest ids
        a.gk = tune.gk > 0 ? tune.gk : 8;
        // 16 column tiles (N = 4096): with column-fastest ids an XCD owns tiles x and x + 8 of EVERY row band; the L2-merging order
        // with two neighbouring tiles per XCD measured 25.1 -> 22.3 us (4096 x 4096 [K,N] bytes), 50.7 -> 48.1 (11008 x 4096),
        // 60.5 -> 57.6 (14336 x 4096); every other width of the sweep keeps the plain order (scripts/lab_order_sweep.sh)
        const bool sixteen = !blob && a.ncol_tiles == 16 && tune.order < 0;
        if (sixteen) a.order = 2;
        a.nt = (tune.nt >= 0 ? tune.nt : 1) & kNtMask;
        a.stage_q = ((tune.stage_q != 0) && blob && vec4 && rpw == 16 && a.wpg * a.gpb == kMaxWaves && kgroups % a.gpb == 0) ? 1 : 0;
        a.spb_log2 = 0;
        a.xg_log2 = tune.xg >= 0 ? static_cast<uint32_t>(tune.xg > 4 ? 4 : tune.xg) : (sixteen ? 1u : 0u);
        set_block_order(a);
        const dim3 grid_dim(a.ncol_tiles * a.nrow_tiles + (in_launch ? static_cast<uint32_t>(ceil_div(N, 128)) : 0u), static_cast<uint32_t>(batch)),
            block(static_cast<uint32_t>(a.wpg * a.gpb * kWave));
        if (in_launch) {      // its appended blocks wait for its main blocks: ordered against every other waiting launch of the device
            st = ticket_chain_begin(s);
            if (st != OQ_OK) return st;
        }
        if (vec4) {
            if (emit_q) launch_fused<true, true>(rpw, a, grid_dim, block, s);
            else launch_fused<true, false>(rpw, a, grid_dim, block, s);
        } else {
            if (emit_q) launch_fused<false, true>(rpw, a, grid_dim, block, s);
            else launch_fused<false, false>(rpw, a, grid_dim, block, s);
        }
        st = check_launch("rtn_group_fused");
        if (in_launch) { ticket_chain_end(s); return st; }
        if (st != OQ_OK || !staged) return st;
        hipLaunchKernelGGL(transpose_qparams,
                           dim3(static_cast<uint32_t>(ceil_div(N, 32)), static_cast<uint32_t>(ceil_div(kgroups, 32)), static_cast<uint32_t>(batch)),
                           dim3(256), 0, s, a.scale_t, a.zp_t, kgroups, N, scale_out, zp8, a.table);
        return check_launch("transpose_qparams");
    }
    OQ_REQUIRE(g_batch.count == 1, OQ_ERR_UNSUPPORTED, "batched call needs the fused group path");

    // channel, tensor, groups taller than the fused kernel holds: W read once, ranges completed across workgroups
    // (rtn_resident.hip); what it does not take (scalar-aligned operands, the blob layout, ranges without integers,
    // columns taller than 16384 rows) runs on the three launches below
    {
        static const Tuning tr = Tuning::from_env();
        // a stream under capture takes the three launches below: the ticketed kernels must never overlap one another, and a
        // replayed graph is ordered against nothing the library sees
        const bool ticketed_ok = tr.resident != 0 && !rtn_stream_is_capturing(s);
        if (ticketed_ok && g_state.state != nullptr && rtn_resident_eligible(K, N, ldw, W, q_out, strategy, g, layout, emit_q, g_state.bytes))
            return rtn_resident_impl(W, K, N, ldw, grid, strategy, g, q8, scale_out, zp8, layout, g_state.state, g_state.bytes, s, true);
        if (ticketed_ok && rtn_resident_eligible(K, N, ldw, W, q_out, strategy, g, layout, emit_q, workspace ? workspace_bytes : 0))
            return rtn_resident_impl(W, K, N, ldw, grid, strategy, g, q8, scale_out, zp8, layout, workspace, workspace_bytes, s, false);
    }

    // two-pass
    const size_t need = twopass_ws(K, N, g);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "rtn: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    RangeArgs r;
    r.W = W; r.K = K; r.N = N; r.ldw = ldw; r.g = g; r.kgroups = kgroups;
    r.chunks = ceil_div(g, kChunkRows);
    r.pmin = static_cast<float*>(workspace);
    r.pmax = r.pmin + kgroups * r.chunks * N;
    r.ncol_tiles = static_cast<uint32_t>(ceil_div(N, kColsPerWave));
    r.nrow_tiles = static_cast<uint32_t>(kgroups * r.chunks);
    const bool vec_in = (N % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    if (vec_in)
        hipLaunchKernelGGL(col_range_partial<true>, dim3(r.ncol_tiles * r.nrow_tiles), dim3(kMaxWaves * kWave), 0, s, r);
    else
        hipLaunchKernelGGL(col_range_partial<false>, dim3(r.ncol_tiles * r.nrow_tiles), dim3(kMaxWaves * kWave), 0, s, r);
    st = check_launch("col_range_partial");
    if (st != OQ_OK) return st;
    if (strategy == OQ_TENSOR) {
        // fold the chunks per column in parallel (into the first row of the partial arrays), then one block folds N values
        hipLaunchKernelGGL(col_fold_kernel, dim3(static_cast<uint32_t>(ceil_div(N, 256))), dim3(256), 0, s, r.pmin, r.pmax, N,
                           kgroups * r.chunks);
        hipLaunchKernelGGL(tensor_range_finalize, dim3(1), dim3(1024), 0, s, r.pmin, r.pmax, N, grid, scale_out, zp8);
    } else {
        hipLaunchKernelGGL(col_range_finalize, dim3(static_cast<uint32_t>(ceil_div(N, 256)), static_cast<uint32_t>(kgroups)),
                           dim3(256), 0, s, r.pmin, r.pmax, N, kgroups, r.chunks, grid, scale_out, zp8);
    }
    st = check_launch("range_finalize");
    if (st != OQ_OK || !emit_q) return st;
    return launch_quantize_kn(W, K, N, ldw, g, kgroups, scale_out, zp8, q8, grid, zp_signed, strategy == OQ_TENSOR, s, layout);
}

}  // namespace oq

extern "C" {

size_t oq_rtn_workspace_bytes(int64_t K, int64_t N, int32_t strategy, int64_t group_size, int32_t mse) {
    if (!oq::matrix_ok(K, N, N)) return 0;
    int64_t g;
    if (oq::resolve_group(strategy, K, group_size, &g) != OQ_OK) return 0;
    if (strategy == OQ_GROUP && K % g != 0) return 0;
    size_t need = oq::twopass_ws(K, N, g);
    const size_t st = oq::stage_ws(K, N, g);
    if (st > need) need = st;
    const size_t rs = oq::rtn_resident_workspace(K, N, strategy, g);
    if (rs > need) need = rs;
    if (mse) need += oq::rtn_mse_workspace(K, N, strategy, g);
    return need + 256;
}

int32_t oq_rtn_quantize_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                            int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio,
                            int32_t mse, void* q_out, float* scale_out, void* zp_out, int32_t layout,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return oq::rtn_impl(W, K, N, ldw, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, mse, q_out,
                        scale_out, zp_out, layout, workspace, workspace_bytes, stream, true);
}

size_t oq_rtn_state_bytes(int64_t K, int64_t N, int32_t strategy, int64_t group_size) {
    if (!oq::matrix_ok(K, N, N)) return 0;
    int64_t g;
    if (oq::resolve_group(strategy, K, group_size, &g) != OQ_OK || K % g != 0) return 0;
    int rpw = 0, wpg = 0;
    if (strategy == OQ_GROUP && oq::fused_shape(g, &rpw, &wpg))      // the fused group kernels: the staging of the in-launch parameter transposition
        return (rpw == 16 && wpg == oq::kMaxWaves && (K / g) % 4 == 0 && N % 4 == 0 && K / g > 1) ? oq::fused_state_bytes(K, N, g) : 0;
    return oq::rtn_resident_workspace(K, N, strategy, g);
}

int32_t oq_rtn_quantize_stateful_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                                     int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio,
                                     int32_t mse, void* q_out, float* scale_out, void* zp_out, int32_t layout,
                                     void* workspace, size_t workspace_bytes, void* state, size_t state_bytes, void* stream) {
    OQ_REQUIRE(state == nullptr || (reinterpret_cast<uintptr_t>(state) & 15u) == 0, OQ_ERR_INVALID_ARGUMENT,
               "oq_rtn_quantize_stateful_f32: state must be 16-byte aligned");
    oq::g_state.state = state;
    oq::g_state.bytes = state ? state_bytes : 0;
    const int32_t st = oq::rtn_impl(W, K, N, ldw, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, mse, q_out, scale_out, zp_out,
                                    layout, workspace, workspace_bytes, stream, true);
    oq::g_state = oq::StateCtx();
    return st;
}

int32_t oq_rtn_qparams_f32(const float* W, int64_t K, int64_t N, int64_t ldw, int32_t qtype, int32_t strategy,
                           int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio,
                           int32_t mse, float* scale_out, void* zp_out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    return oq::rtn_impl(W, K, N, ldw, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, mse, nullptr,
                        scale_out, zp_out, OQ_LAYOUT_KN, workspace, workspace_bytes, stream, false);
}

size_t oq_rtn_batched_workspace_bytes(int64_t batch, int64_t K, int64_t N, int64_t group_size) {
    if (batch <= 0 || batch > 65535 || !oq::matrix_ok(K, N, N)) return 0;
    int64_t g;
    if (oq::resolve_group(OQ_GROUP, K, group_size, &g) != OQ_OK || K % g != 0) return 0;
    return static_cast<size_t>(batch) * oq::stage_ws(K, N, g) + 512;
}

int32_t oq_rtn_quantize_batched_f32(const float* W, int64_t batch, int64_t w_stride, int64_t K, int64_t N, int64_t ldw, int32_t qtype,
                                    int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio, void* q_out,
                                    float* scale_out, void* zp_out, int32_t layout, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    OQ_REQUIRE(oq::matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT, "oq_rtn_quantize_batched_f32: bad shape K=%lld N=%lld ldw=%lld", (long long)K,
               (long long)N, (long long)ldw);
    OQ_REQUIRE(batch >= 1 && batch <= 65535 && w_stride >= K * ldw && w_stride <= oq::kMaxElements, OQ_ERR_INVALID_ARGUMENT,
               "oq_rtn_quantize_batched_f32: bad batch %lld (1 to 65535) / stride %lld", (long long)batch, (long long)w_stride);
    int64_t g;
    int32_t st = oq::resolve_group(OQ_GROUP, K, group_size, &g);
    if (st != OQ_OK) return st;
    OQ_REQUIRE(K % g == 0, OQ_ERR_UNSUPPORTED, "oq_rtn_quantize_batched_f32 needs K %% group_size == 0");
    int rpw = 0, wpg = 0;
    OQ_REQUIRE(oq::fused_shape(g, &rpw, &wpg), OQ_ERR_UNSUPPORTED, "oq_rtn_quantize_batched_f32 needs a group the fused kernel holds (<= 256 rows)");
    const int64_t bits = (qtype == OQ_INT4 || qtype == OQ_UINT4) ? 4 : 8;
    oq::g_batch.count = batch;
    oq::g_batch.w_stride = w_stride;
    oq::g_batch.q_stride = layout == OQ_LAYOUT_KN ? K * N : K * N * bits / 8;      // blob and KN_PACKED4 (4-bit only) alike
    st = oq::rtn_impl(W, K, N, ldw, qtype, OQ_GROUP, group_size, symmetric, reduce_range, clip_ratio, 0, q_out, scale_out, zp_out, layout,
                      workspace, workspace_bytes, stream, true);
    oq::g_batch = oq::BatchCtx();
    return st;
}

int32_t oq_rtn_quantize_ptrs_f32(const oq_rtn_ptrs* table_host, const oq_rtn_ptrs* table_device, int64_t count, int64_t K, int64_t N,
                                 int64_t ldw, int32_t qtype, int64_t group_size, int32_t symmetric, int32_t reduce_range, float clip_ratio,
                                 int32_t layout, void* workspace, size_t workspace_bytes, void* stream) {
    static_assert(sizeof(oq::RtnPtrs) == sizeof(oq_rtn_ptrs), "device view of oq_rtn_ptrs");
    OQ_REQUIRE(K > 0 && N > 0 && count >= 1 && count <= oq::kMaxExtent, OQ_ERR_INVALID_ARGUMENT, "oq_rtn_quantize_ptrs_f32: bad shape K=%lld N=%lld / count %lld",
               (long long)K, (long long)N, (long long)count);
    OQ_REQUIRE(oq::matrix_ok(K, N, ldw), OQ_ERR_INVALID_ARGUMENT, "oq_rtn_quantize_ptrs_f32: bad shape K=%lld N=%lld ldw=%lld", (long long)K, (long long)N,
               (long long)ldw);     // before any arithmetic on K * N (matrices_per_launch divides by it)
    const int64_t per_launch = oq::matrices_per_launch(K, N, count);
    OQ_REQUIRE(table_host && (table_device || per_launch == 1) && count >= 1, OQ_ERR_INVALID_ARGUMENT,
               "oq_rtn_quantize_ptrs_f32: bad table / count %lld", (long long)count);
    int64_t g;
    int32_t st = oq::resolve_group(OQ_GROUP, K, group_size, &g);
    if (st != OQ_OK) return st;
    OQ_REQUIRE(K % g == 0, OQ_ERR_UNSUPPORTED, "oq_rtn_quantize_ptrs_f32 needs K %% group_size == 0");
    int rpw = 0, wpg = 0;
    OQ_REQUIRE(oq::fused_shape(g, &rpw, &wpg), OQ_ERR_UNSUPPORTED, "oq_rtn_quantize_ptrs_f32 needs a group the fused kernel holds (<= 256 rows)");
    // the kernel variant is chosen from the alignment of ONE matrix: every entry has to meet what the first one promises
    for (int64_t i = 0; i < count; ++i) {
        const oq_rtn_ptrs& p = table_host[i];
        OQ_REQUIRE(p.W && p.q_out && p.scale_out && p.zp_out, OQ_ERR_INVALID_ARGUMENT, "oq_rtn_quantize_ptrs_f32: null pointer in entry %lld", (long long)i);
        OQ_REQUIRE(oq::aligned16(p.W) && oq::aligned16(p.q_out) && (reinterpret_cast<uintptr_t>(p.scale_out) & 3u) == 0, OQ_ERR_UNSUPPORTED,
                   "oq_rtn_quantize_ptrs_f32: entry %lld is not 16-byte aligned", (long long)i);
    }
    // How many matrices share a launch (blockIdx.y = entry): oq::matrices_per_launch.
    for (int64_t i = 0; i < count; i += per_launch) {
        const int64_t m = count - i < per_launch ? count - i : per_launch;
        oq::g_batch.count = m;
        oq::g_batch.table = m > 1 ? reinterpret_cast<const oq::RtnPtrs*>(table_device) + i : nullptr;
        st = oq::rtn_impl(table_host[i].W, K, N, ldw, qtype, OQ_GROUP, group_size, symmetric, reduce_range, clip_ratio, 0, table_host[i].q_out,
                          table_host[i].scale_out, table_host[i].zp_out, layout, workspace, workspace_bytes, stream, true);
        oq::g_batch = oq::BatchCtx();
        if (st != OQ_OK) return st;
    }
    return OQ_OK;
}

}  // extern "C"
