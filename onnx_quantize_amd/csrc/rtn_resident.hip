// A1 for the strategies whose range spans more rows than one workgroup holds -- per-channel, per-tensor (the reference's
// DEFAULT QWeightArgs(): core/_qconfig.py:232-268) and groups taller than 256 rows -- with W read from HBM ONCE
// (rtn.py:54-109 = utils.py:42-69 R1 + :242-299 Q1 + :72-79 K1).
//
// Round 1-3 ran these as three launches (range partials, finalize, quantize) that read W twice: 89 / 94 us on 4096 x 11008
// where the group kernel needs 45.  Here a workgroup keeps its 128 x 256 tile in registers while the range it belongs to
// is completed by other workgroups:
//
//   * work is handed out by TICKETS (one agent-scope atomic add per tile), never by blockIdx: a workgroup only ever waits
//     for tickets that have already been taken, i.e. for workgroups that are running.  HIP promises no dispatch order;
//     tickets make the forward-progress argument independent of it (see the two kernels for the argument itself);
//   * partial ranges meet in HBM-side atomics: float -> order-preserving uint32 key, atomic max at agent scope (the minimum
//     as the maximum of the complemented key), NaN as the top key so that it propagates like np.min / np.max.  Every access to a key or a
//     counter is an agent-scope atomic (RMW or load), so nothing of the hand-off ever sits in a non-coherent L1 / L2 line:
//     no release / acquire fence (a release fence writes back the XCD's whole L2: 1.7-6.5 us per workgroup,
//     MI355X_MICROARCH.md) is needed, only the producer's own `s_waitcnt vmcnt(0)` between its key atomics and its
//     counter add;
//   * the integers come out of the same registers with the exact-reciprocal fast path of the group kernels
//     (oq_common.hpp), so they are the reference's bits by construction.
//
// `rtn_resident_groups`: channel and tall groups of up to 4096 rows.  The tiles of one range (one column tile x one
// k-group) have consecutive tickets, are loaded at about the same time by different workgroups, and every one of them waits
// for its siblings' partial ranges before it quantizes: one read of W, no second pass.
// `rtn_resident_stream`: the same for taller ranges with persistent workgroups and two tile slots: a tile is published
// into one slot before the workgroup waits for the range of the tile in the other one, and that tile's rows are stored
// while the next tile's rows are loaded into their place.
// `rtn_tensor_onepass`: per-tensor.  ONE 8-wave workgroup per CU.  Phase A streams all tiles once (running min / max in
// registers, no barrier per tile beyond the ticket exchange) through one register slot; a workgroup KEEPS the last FOUR tiles
// it loaded -- the slot, two tiles parked in its ACCUMULATION registers, one in 128 KB of LDS: 128 MB of the matrix stay on
// the chip.  One returning add per workgroup counts the tiles and hands out an arrival slot for its partial range and the ids
// of the tiles it keeps; the workgroup that completes the count folds the slots and broadcasts {go, keys}; kept tiles are
// quantized from registers / LDS; the rows of the other half tiles are dealt out statically and evenly over all waves (no
// tickets, no barriers), re-read most-recent-first and software-pipelined over the two halves of the slot.  A matrix of up
// to 1024 tiles is read exactly once and skips that phase altogether.
#include "oq_common.hpp"

#include <cstdlib>
#include <mutex>

namespace oq {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

constexpr int kResWaves = 8;
constexpr int kResRows = 16;                      // rows per wave
constexpr int kResTileRows = kResWaves * kResRows;  // 128, the chunk height of the two-pass path (rtn.hip kChunkRows)
constexpr int kResCols = 256;                     // 64 lanes x 4 columns
constexpr int kResHeader = 160;
constexpr int kResMaxTensorTiles = 32768;         // bitmap of kept half tiles + its prefix sums in LDS (8 KB each): sized for this many tiles (the kernel takes up to kT4MaxTiles: 1 G parameters; larger tensors take the three-launch path)
constexpr int kResTensorHeader = 128 + 64 * 32 + 64 * 32;   // tickets / counter, 512 arrival slots of 16 bytes, 64 result replicas (a 128-byte line each)
constexpr int kResGroupTileRows = 128;            // tile height of rtn_resident_stream and of rtn_resident_groups<8> (see groups_tile_rows)
constexpr int kResCtrPad = 32;                   // uint32 words per range counter: a 128-byte line each (hundreds of workgroups poll them)
#ifndef OQ_RES_A_NT
#define OQ_RES_A_NT false   /* default-policy loads in phase A keep the lines in the Infinity Cache for phase B: 75 us against 78 with nt */
#endif
#ifndef OQ_RES_SLEEP
#define OQ_RES_SLEEP 8   /* s_sleep between two polls of a counter */
#endif

struct ResidentArgs {
    const float* W;
    int64_t K, N, ldw;
    int64_t g, kgroups, chunks;   // rows per range, ranges per column, tiles per range and column tile
    uint8_t* q;
    float* scale;
    uint8_t* zp;
    QGrid grid;
    int32_t layout;
    uint32_t ncol_tiles, ntiles;
    uint32_t* key_max;    // groups: [slots] ordered key of the running maximum; tensor: the arrival slots (16 bytes per workgroup)
    uint32_t* key_nmin;   // groups: [slots] complement of the ordered key of the running minimum (kept as a maximum); tensor: the replica lines
    uint32_t* counters;   // groups: one per (column tile, k-group); tensor: [0] = tiles counted + arrivals
    uint32_t* tickets;    // [0]
    uint32_t* held;       // unused since the kept tiles travel in the arrival slots (kept for the layout of the workspace)
    // Self-cleaning (oq_rtn_quantize_stateful_f32: the caller's `state` is zero when the call starts and zero again when it
    // ends, so no clear launch runs in front of the kernel): every workgroup counts itself out at `done`; the cleaner --
    // the one that holds the last ticket / ticket 0 -- waits for all the others and zeroes [clean_base, +clean_words).
    uint32_t* done;         // nullptr: the state is a plain workspace that the host cleared
    uint32_t* clean_base;
    uint32_t clean_words;
};

__device__ __forceinline__ uint32_t okey_plain(float x) {   // monotone float -> uint32 for everything but NaN
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(uint32_t k) {   // 0xFFFFFFFF -> 0x7FFFFFFF and 0 -> 0xFFFFFFFF: both NaN
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// Both running extrema are kept as MAXIMA of a key so that one zero-filled array serves them: the maximum as the key itself
// (NaN = top), the minimum as the COMPLEMENT of the key (NaN = key 0, complement = top), decoded by okey_inv(~stored).
// No floating-point negation anywhere: `-okey_inv(k)` was miscompiled by hipcc 7.2 for two of four unrolled columns (the
// fneg folded into v_cndmask source modifiers was dropped: the first GPU run returned min = +|min| for even columns).
__device__ __forceinline__ uint32_t key_of_max(float x) { return (x != x) ? 0xFFFFFFFFu : okey_plain(x); }
__device__ __forceinline__ uint32_t key_of_min(float x) { return (x != x) ? 0xFFFFFFFFu : ~okey_plain(x); }
__device__ __forceinline__ float max_of_key(uint32_t k) { return okey_inv(k); }
__device__ __forceinline__ float min_of_key(uint32_t k) { return okey_inv(~k); }
__device__ __forceinline__ void agent_max(uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t agent_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t agent_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every earlier vector-memory operation of this wave (the key atomics) has been performed when this returns; inline asm
// so that no compiler pass can drop or move it (MI355X_MICROARCH.md, "Compiler hazard")
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef OQ_SPIN_LIMIT   /* lab builds only (first runs of a new hand-off on a shared GPU box): a spin gives up after OQ_SPIN_LIMIT polls and
                          counts itself; the results are then garbage and oq_lab_spin_timeouts() says so -- but nothing hangs */
__device__ uint32_t g_spin_timeouts;
__device__ __forceinline__ void spin_until(const uint32_t* p, uint32_t target) {
    for (uint32_t i = 0; agent_load(p) < target; ++i) {
        if (i >= static_cast<uint32_t>(OQ_SPIN_LIMIT)) { atomicAdd(&g_spin_timeouts, 1u); return; }
        __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
    }
}
#else
__device__ __forceinline__ void spin_until(const uint32_t* p, uint32_t target) {
    while (agent_load(p) < target) __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
}
#endif

// the cleaner's last act: every other workgroup of the grid has left (`expected` of them), nobody reads the state any more
__device__ __forceinline__ void clean_state(const ResidentArgs& a, uint32_t expected, int nthreads) {
    if (threadIdx.x == 0) spin_until(a.done, expected);
    __syncthreads();
    uint4* p = reinterpret_cast<uint4*>(a.clean_base);
    for (uint32_t i = threadIdx.x; i < (a.clean_words + 3u) / 4u; i += nthreads) p[i] = make_uint4(0u, 0u, 0u, 0u);
}


// 16 rows x 4 columns of a lane, clamped addresses (never a predicated load: rtn.hip).  Rows past `row_end` repeat the
// last row of the range and columns past N repeat the last four: duplicates of valid elements of the SAME range, so they
// cannot change a minimum or a maximum; only the stores are masked.
template <bool NT = true, int ROWS = kResRows>
__device__ __forceinline__ void load_tile(const ResidentArgs& a, int64_t row0, int64_t row_end, int64_t tile_col0, int lane,
                                          float (&v)[ROWS][4]) {
    int64_t lcol = tile_col0 + lane * 4;
    lcol = lcol < a.N ? lcol : a.N - 4;
    const float* p = a.W + lcol;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int64_t row = row0 + r < row_end ? row0 + r : row_end - 1;
        f32x4r u;
        if constexpr (NT) u = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(p + row * a.ldw));
        else u = *reinterpret_cast<const f32x4r*>(p + row * a.ldw);    // default policy: the line may stay in the Infinity Cache for phase B
        v[r][0] = u[0]; v[r][1] = u[1]; v[r][2] = u[2]; v[r][3] = u[3];
    }
}

// One dword of quantized bytes to HBM.  OQ_RES_STORE (lab): 0 = `nt` (streaming, but write-BACK: the lines stay dirty in the XCD's L2
// until they are evicted or the kernel ends), 1 = default policy, 2 = `sc1` (agent scope: written through), 3 = `sc0 sc1`, 4 = `sc1 nt`,
// 5 = `sc0 sc1 nt`.
#ifndef OQ_RES_STORE
#define OQ_RES_STORE 0
#endif
__device__ __forceinline__ void store_q_word(uint32_t w, uint32_t* p) {
#if OQ_RES_STORE == 0
    __builtin_nontemporal_store(w, p);
#elif OQ_RES_STORE == 1
    *p = w;
#elif OQ_RES_STORE == 2
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(w) : "memory");
#elif OQ_RES_STORE == 3
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(w) : "memory");
#elif OQ_RES_STORE == 4
    asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(p), "v"(w) : "memory");
#else
    asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(w) : "memory");
#endif
}

// K1 from registers + [K, N] byte stores (one dword = a lane's four columns of a row).
// GROUP > 1 (the kernels that run one or two waves per SIMD and so have nobody to fill the bubbles): the fast path of GROUP rows
// as independent chains with ONE decision behind them -- |t - k| of every element folded into a running NaN-propagating maximum
// on the vector ALU (v_maximum3_f32 with |.| modifiers: half an instruction per element) and compared once with the narrowest
// band of the lane's four columns, instead of a compare + scalar OR per element and a ballot + branch per row.  (The packed
// "magic number" form of rtn.hip's wave kernel was tried here: 65.1 -> 70.3 us per call, packed fp32 runs at half rate.)
template <int ROWS = kResRows, int GROUP = 1>
__device__ __forceinline__ void quantize_store_tile(const ResidentArgs& a, const ColQ (&cq)[4], float (&v)[ROWS][4], int64_t row0,
                                                    int64_t row_end, int64_t tile_col0, int lane) {
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? 128 : 0;
    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
    const uint32_t flip = bias ? 0x80808080u : 0u;
    const bool col_ok = tile_col0 + lane * 4 < a.N;
    uint8_t* o = a.q + row0 * a.N + tile_col0 + lane * 4;
    if constexpr (GROUP == 1) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float f[4];
            bool unsafe = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = quantize_fast_biased(v[r][i], cq[i], lo_b, hi_b, unsafe);
            if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {   // wave-uniform, rare: redo this row with the IEEE divide
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = quantize_exact_biased(v[r][i], cq[i], qmin, qmax, bias);
            }
            uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(f[0], 0, 0);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[1], 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[2], 2, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(f[3], 3, w);
            if (col_ok && row0 + r < row_end) store_q_word(w ^ flip, reinterpret_cast<uint32_t*>(o + r * a.N));
        }
    } else {
        // In the "magic number" domain of rtn.hip's wave kernel: with M = 1.5 * 2^23 the fp32 grid around M + k has spacing 1, so
        // u = fma(x, rinv, zp + bias + M) IS M + rint(x * rinv + zp + bias) -- product, zero point and rounding in one correctly
        // rounded operation -- and res = fma(x, rinv, (zp + bias + M) - u) is the distance of x * rinv + zp from that integer,
        // exact but for one rounding of <= 2^-25: |res| inside the band proves the integer the reference's, everything else (ties,
        // NaN, inf, sums beyond the grid) is redone with the IEEE division.  The level is byte 0 of the clamped float's bits.
        // fma, sub, fma, half a v_maximum3, med3 and 3/4 of a byte permute per element: 5.25 instructions against 6.5.
        static_assert(ROWS % GROUP == 0, "whole groups");
        constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23, bits 0x4B400000
        const float thr_min = nmin(nmin(cq[0].thr, cq[1].thr), nmin(cq[2].thr, cq[3].thr));   // the narrowest band of the four columns: never less careful
        const float zm[4] = {cq[0].zpb + kMagic, cq[1].zpb + kMagic, cq[2].zpb + kMagic, cq[3].zpb + kMagic};
        const float lo_m = lo_b + kMagic, hi_m = hi_b + kMagic;
#pragma unroll
        for (int rg = 0; rg < ROWS; rg += GROUP) {
            uint32_t f[GROUP][4];
            float far = 0.0f;          // the largest |res| of the group, NaN if any
#pragma unroll
            for (int r = 0; r < GROUP; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = __builtin_fmaf(v[rg + r][i], cq[i].rinv, zm[i]);
                    const float res = __builtin_fmaf(v[rg + r][i], cq[i].rinv, zm[i] - u);
                    far = nmax(far, fabsf(res));
                    f[r][i] = __float_as_uint(__builtin_amdgcn_fmed3f(u, lo_m, hi_m));
                }
            if (__builtin_amdgcn_ballot_w64(!(far < thr_min)) != 0) {   // wave-uniform, rare: redo these rows with the IEEE divide
#pragma unroll
                for (int r = 0; r < GROUP; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[r][i] = static_cast<uint32_t>(quantize_one(v[rg + r][i], cq[i].scale, cq[i].zp, qmin, qmax) + bias);
            }
#pragma unroll
            for (int r = 0; r < GROUP; ++r) {
                const uint32_t w = __builtin_amdgcn_perm(f[r][1], f[r][0], 0x0c0c0400u) | __builtin_amdgcn_perm(f[r][3], f[r][2], 0x04000c0cu);   // byte 0 of each
                if (col_ok && row0 + rg + r < row_end) store_q_word(w ^ flip, reinterpret_cast<uint32_t*>(o + (rg + r) * a.N));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Channel / tall groups.  One workgroup = one ticket = one 128 x 256 tile.  Ticket t -> range t / chunks (column tile
// major, k-group minor), chunk t % chunks.
//
// Forward progress.  A workgroup waits only after it has published its own partial range, and only for the `chunks`
// tickets of its own range.  Let R be the oldest incomplete range.  If one of R's tickets has not been taken, no later
// ticket has been taken either, so every waiting workgroup belongs to R and there are at most chunks - 1 of them; any
// other running workgroup takes the next ticket.  With at least `chunks` workgroups running (the host enforces
// chunks <= 128 against 256 CUs) the missing tickets are always taken, loaded and published without waiting, and R
// completes.  Workgroups that are not resident yet hold no ticket and nobody waits for them.
// ---------------------------------------------------------------------------------------------
#ifndef OQ_GROUPS_GROUP
#define OQ_GROUPS_GROUP 2   /* rows per decision of K1 (quantize_store_tile): 4096 x 4096 int8 per channel 27.7 us with 1, 26.6 with 2, 26.9-27.9 with 4 (<= 128 registers) */
#endif
template <int WAVES, int ROWS, int WPS>
__global__ __launch_bounds__(WAVES* kWave, WPS) void rtn_resident_groups(const ResidentArgs a) {
    constexpr int kTileRows = WAVES * ROWS;
    __shared__ float4 s_mn[WAVES][kWave];
    __shared__ float4 s_mx[WAVES][kWave];
    __shared__ uint32_t s_ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
    __syncthreads();
    const uint32_t t = s_ticket;
    if (t >= a.ntiles) return;   // uniform; cannot happen with grid == ntiles, kept as the exit every wave reaches
    const uint32_t chunks = static_cast<uint32_t>(a.chunks), kgroups = static_cast<uint32_t>(a.kgroups);
    const uint32_t range = t / chunks, c = t - range * chunks;
    const uint32_t col_tile = range / kgroups, kg = range - col_tile * kgroups;
    const int64_t row_end = min(static_cast<int64_t>(kg) * a.g + a.g, a.K);
    const int64_t row0 = static_cast<int64_t>(kg) * a.g + static_cast<int64_t>(c) * kTileRows + wave * ROWS;
    const int64_t tile_col0 = static_cast<int64_t>(col_tile) * kResCols;

    float v[ROWS][4];
    // a wave whose rows all lie past the range's end (last chunk of a ragged range) repeats the range's last row
    load_tile<true, ROWS>(a, row0 < row_end ? row0 : row_end - 1, row_end, tile_col0, lane, v);
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mn[i] = mx[i] = v[0][i];
#pragma unroll
    for (int r = 1; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = nmin(mn[i], v[r][i]);
            mx[i] = nmax(mx[i], v[r][i]);
        }
    // slot of column (tile_col0 + lane * 4 + i) of k-group kg: [kg][column tile][i][lane] -- a wave-instruction of key
    // atomics (fixed i) then covers 256 contiguous bytes = four 64-byte requests at the memory side instead of sixteen
    const int64_t slot0 = (static_cast<int64_t>(kg) * a.ncol_tiles + col_tile) * kResCols + lane;
    const bool col_ok = tile_col0 + lane * 4 < a.N;
    if (chunks > 1) {
        s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const float4 tn = s_mn[w][lane], tx = s_mx[w][lane];
                mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y); mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
                mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y); mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
            }
            if (col_ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    agent_max(a.key_max + slot0 + i * kWave, key_of_max(mx[i]));
                    agent_max(a.key_nmin + slot0 + i * kWave, key_of_min(mn[i]));
                }
            }
            drain_vmem();                                        // this wave's key atomics are performed ...
            if (lane == 0) agent_add(a.counters + range * kResCtrPad, 1u);    // ... before the range counts this tile
        }
        if (threadIdx.x == 0) spin_until(a.counters + range * kResCtrPad, chunks);
        __syncthreads();
        if (col_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mx[i] = max_of_key(agent_load(a.key_max + slot0 + i * kWave));
                mn[i] = min_of_key(agent_load(a.key_nmin + slot0 + i * kWave));
            }
        }
    } else {   // the range is this tile: fold the block's waves and go on
        s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float4 tn = s_mn[w][lane], tx = s_mx[w][lane];
            mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y); mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
            mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y); mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
        }
    }
    const int32_t bias = a.grid.qmin < 0 ? 128 : 0;
    ColQ cq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cq[i] = make_colq(qparam_from_minmax(mn[i], mx[i], a.grid), mn[i], mx[i], bias);
    if (c == 0 && wave == 0 && col_ok) {   // rtn.py:98-109 result layout: entry n * kgroups + kg
        if (kgroups == 1) {
            *reinterpret_cast<float4*>(a.scale + tile_col0 + lane * 4) = make_float4(cq[0].scale, cq[1].scale, cq[2].scale, cq[3].scale);
            *reinterpret_cast<uint32_t*>(a.zp + tile_col0 + lane * 4) =
                (static_cast<uint32_t>(cq[0].zp) & 0xffu) | ((static_cast<uint32_t>(cq[1].zp) & 0xffu) << 8) |
                ((static_cast<uint32_t>(cq[2].zp) & 0xffu) << 16) | ((static_cast<uint32_t>(cq[3].zp) & 0xffu) << 24);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t o = (tile_col0 + lane * 4 + i) * a.kgroups + kg;
                a.scale[o] = cq[i].scale;
                a.zp[o] = static_cast<uint8_t>(cq[i].zp);
            }
        }
    }
    quantize_store_tile<ROWS, OQ_GROUPS_GROUP>(a, cq, v, row0, row_end, tile_col0, lane);
    if (a.done != nullptr) {   // uniform
        // every state access of this workgroup is complete (its key loads returned, its counter add was seen by its own poll)
        if (t + 1u == a.ntiles) clean_state(a, a.ntiles - 1u, WAVES * kWave);
        else if (threadIdx.x == 0) agent_add(a.done, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// Channel / tall groups, streamed: ONE persistent 8-wave workgroup per CU with TWO tile slots in registers (the per-tensor
// kernel's shape).  One slot holds a PUBLISHED tile, the other a tile whose loads are in flight.  A step (stream_step) polls
// the published tile's range once -- it was published a whole tile-load earlier and is normally complete --, lets the key
// loads and the next ticket travel while the other tile lands, is folded and published, then stores the finished tile row
// by row while its registers are refilled with the next one.  The memory-side round trips of the hand-off no longer stand
// between a tile's load and its store with nothing else in flight on the CU (rtn_resident_groups: 65 us on 4096 x 11008 int8).
//
// Forward progress.  A workgroup blocks only in the branch of stream_step that has just published the tile it had in
// flight, and it takes a ticket only behind that branch; the ticket's tile is in flight during the rest of the step and is
// published in the next step before anything blocks.  So a workgroup never waits while it holds an unpublished tile.
// Let R be the oldest incomplete range.  If one of R's tickets has not been taken, no later ticket has been taken either, so
// nobody holds a tile of a younger range and every blocked workgroup waits for R while holding a published tile of R: at
// most chunks - 1 of them.  Any other running workgroup is loading / storing / publishing, or polls an older, i.e. complete
// range: it goes on and takes the next ticket, which is R's.  With at least `chunks` workgroups running (the host checks
// chunks against 3/4 of the workgroups the device holds) R completes.  Workgroups that are not resident yet hold no ticket
// and nobody waits for them.
// ---------------------------------------------------------------------------------------------
#ifdef OQ_TENSOR_STAMPS   // lab build only (scripts/lab_tensor_stamps.py): 100 MHz wall-clock stamps of every workgroup's phases
__device__ uint64_t g_tensor_stamps[512 * 8];
#define OQ_STAMP(i) do { if (threadIdx.x == 0) g_tensor_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
// accumulate the time since *T0 into counter i and restart the clock (stream kernel: per-phase sums over a workgroup's tiles)
#define OQ_LAP(i, T0) do { if (threadIdx.x == 0) { const uint64_t now_ = wall_clock64(); g_tensor_stamps[blockIdx.x * 8 + (i)] += now_ - (T0); (T0) = now_; } } while (0)
#else
#define OQ_STAMP(i) do { } while (0)
#define OQ_LAP(i, T0) do { } while (0)
#endif

struct StreamTile {          // everything uniform over the workgroup except slot0 / col_ok
    uint32_t range, c, kg;
    int64_t row0, row_end, tile_col0, slot0;
    bool col_ok;
};

__device__ __forceinline__ StreamTile stream_tile(const ResidentArgs& a, uint32_t t, int lane, int wave) {
    StreamTile s;
    const uint32_t chunks = static_cast<uint32_t>(a.chunks), kgroups = static_cast<uint32_t>(a.kgroups);
    s.range = t / chunks;
    s.c = t - s.range * chunks;
    const uint32_t col_tile = s.range / kgroups;
    s.kg = s.range - col_tile * kgroups;
    s.row_end = min(static_cast<int64_t>(s.kg) * a.g + a.g, a.K);
    s.row0 = static_cast<int64_t>(s.kg) * a.g + static_cast<int64_t>(s.c) * kResTileRows + wave * kResRows;
    s.tile_col0 = static_cast<int64_t>(col_tile) * kResCols;
    // slot of column (tile_col0 + lane * 4 + i) of k-group kg: [kg][column tile][i][lane] -- a wave-instruction of key
    // atomics (fixed i) then covers 256 contiguous bytes = four 64-byte requests at the memory side instead of sixteen
    s.slot0 = (static_cast<int64_t>(s.kg) * a.ncol_tiles + col_tile) * kResCols + lane;
    s.col_ok = s.tile_col0 + lane * 4 < a.N;
    return s;
}

// lane-local column ranges of a freshly loaded tile -> this wave's row of the LDS exchange
__device__ __forceinline__ void stream_fold_local(int lane, int wave, const float (&v)[kResRows][4], float4 (&s_mn)[kResWaves][kWave],
                                                  float4 (&s_mx)[kResWaves][kWave]) {
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mn[i] = mx[i] = v[0][i];
#pragma unroll
    for (int r = 1; r < kResRows; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = nmin(mn[i], v[r][i]);
            mx[i] = nmax(mx[i], v[r][i]);
        }
    s_mn[wave][lane] = make_float4(mn[0], mn[1], mn[2], mn[3]);
    s_mx[wave][lane] = make_float4(mx[0], mx[1], mx[2], mx[3]);
}

// one wave folds the workgroup's partials of a tile and publishes them: key atomics, drain, counter add.  Never waits for
// another workgroup.
__device__ __forceinline__ void stream_publish_wave(const ResidentArgs& a, const StreamTile& s, int lane, float4 (&s_mn)[kResWaves][kWave],
                                                    float4 (&s_mx)[kResWaves][kWave]) {
    float4 tn = s_mn[0][lane], tx = s_mx[0][lane];
    float mn[4] = {tn.x, tn.y, tn.z, tn.w}, mx[4] = {tx.x, tx.y, tx.z, tx.w};
#pragma unroll
    for (int w = 1; w < kResWaves; ++w) {
        tn = s_mn[w][lane]; tx = s_mx[w][lane];
        mn[0] = nmin(mn[0], tn.x); mn[1] = nmin(mn[1], tn.y); mn[2] = nmin(mn[2], tn.z); mn[3] = nmin(mn[3], tn.w);
        mx[0] = nmax(mx[0], tx.x); mx[1] = nmax(mx[1], tx.y); mx[2] = nmax(mx[2], tx.z); mx[3] = nmax(mx[3], tx.w);
    }
    if (s.col_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            agent_max(a.key_max + s.slot0 + i * kWave, key_of_max(mx[i]));
            agent_max(a.key_nmin + s.slot0 + i * kWave, key_of_min(mn[i]));
        }
    }
    drain_vmem();                                        // this wave's key atomics are performed ...
    if (lane == 0) agent_add(a.counters + s.range * kResCtrPad, 1u);    // ... before the range counts this tile
}

// final keys of a complete range -> this lane's four parameter sets; chunk 0 writes them out (rtn.py:98-109 layout: entry n * kgroups + kg)
__device__ __forceinline__ void stream_keys(const ResidentArgs& a, const StreamTile& s, uint32_t (&kmx)[4], uint32_t (&kmn)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) kmx[i] = kmn[i] = 0x80000000u;     // the key of 0.0f for lanes past the last column
    if (s.col_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kmx[i] = agent_load(a.key_max + s.slot0 + i * kWave);
            kmn[i] = ~agent_load(a.key_nmin + s.slot0 + i * kWave);
        }
    }
}
__device__ __forceinline__ void stream_params(const ResidentArgs& a, const StreamTile& s, int lane, int wave, const uint32_t (&kmx)[4],
                                              const uint32_t (&kmn)[4], ColQ (&cq)[4]) {
    float mn[4], mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        mx[i] = okey_inv(kmx[i]);
        mn[i] = okey_inv(kmn[i]);
    }
    const int32_t bias = a.grid.qmin < 0 ? 128 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) cq[i] = make_colq(qparam_from_minmax(mn[i], mx[i], a.grid), mn[i], mx[i], bias);
    if (s.c == 0 && wave == 0 && s.col_ok) {
        if (a.kgroups == 1) {
            *reinterpret_cast<float4*>(a.scale + s.tile_col0 + lane * 4) = make_float4(cq[0].scale, cq[1].scale, cq[2].scale, cq[3].scale);
            *reinterpret_cast<uint32_t*>(a.zp + s.tile_col0 + lane * 4) =
                (static_cast<uint32_t>(cq[0].zp) & 0xffu) | ((static_cast<uint32_t>(cq[1].zp) & 0xffu) << 8) |
                ((static_cast<uint32_t>(cq[2].zp) & 0xffu) << 16) | ((static_cast<uint32_t>(cq[3].zp) & 0xffu) << 24);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t o = (s.tile_col0 + lane * 4 + i) * a.kgroups + s.kg;
                a.scale[o] = cq[i].scale;
                a.zp[o] = static_cast<uint8_t>(cq[i].zp);
            }
        }
    }
}

// Quantize and store the tile in `v` in groups of OQ_STREAM_GROUP rows, each group's registers refilled at once with the same rows
// of tile `n` (when `refill`): the CU's loads run beside its stores instead of after them.  K1 as in quantize_store_tile's grouped
// form: the magic-number domain, one decision per group on a running v_maximum3 of the residuals (per row with a compare and a
// scalar OR per element until round 6: 62.1-64.5 -> 60.9-61.9 us per call on 4096 x 11008).
#ifndef OQ_STREAM_GROUP
#define OQ_STREAM_GROUP 4
#endif
__device__ __forceinline__ void stream_rows(const ResidentArgs& a, const StreamTile& s, const ColQ (&cq)[4], bool refill, const StreamTile& n,
                                            int lane, float (&v)[kResRows][4]) {
    constexpr int GROUP = OQ_STREAM_GROUP;
    static_assert(kResRows % GROUP == 0, "whole groups");
    constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? 128 : 0;
    int64_t lcol = n.tile_col0 + lane * 4;
    lcol = lcol < a.N ? lcol : a.N - 4;
    const float* src = a.W + lcol;
    const int64_t nrow0 = n.row0 < n.row_end ? n.row0 : n.row_end - 1;   // a wave past the range's end repeats its last row
    const float lo_m = static_cast<float>(qmin + bias) + kMagic, hi_m = static_cast<float>(qmax + bias) + kMagic;
    const uint32_t flip = bias ? 0x80808080u : 0u;
    const float thr_min = nmin(nmin(cq[0].thr, cq[1].thr), nmin(cq[2].thr, cq[3].thr));
    const float zm[4] = {cq[0].zpb + kMagic, cq[1].zpb + kMagic, cq[2].zpb + kMagic, cq[3].zpb + kMagic};
    uint8_t* o = a.q + s.row0 * a.N + s.tile_col0 + lane * 4;
#pragma unroll
    for (int rg = 0; rg < kResRows; rg += GROUP) {
        uint32_t f[GROUP][4];
        float far = 0.0f;          // the largest |residual| of the group, NaN if any
#pragma unroll
        for (int r = 0; r < GROUP; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = __builtin_fmaf(v[rg + r][i], cq[i].rinv, zm[i]);
                const float res = __builtin_fmaf(v[rg + r][i], cq[i].rinv, zm[i] - u);
                far = nmax(far, fabsf(res));
                f[r][i] = __float_as_uint(__builtin_amdgcn_fmed3f(u, lo_m, hi_m));
            }
        if (__builtin_amdgcn_ballot_w64(!(far < thr_min)) != 0) {   // wave-uniform, rare: redo these rows with the IEEE divide
#pragma unroll
            for (int r = 0; r < GROUP; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) f[r][i] = static_cast<uint32_t>(quantize_one(v[rg + r][i], cq[i].scale, cq[i].zp, qmin, qmax) + bias);
        }
#pragma unroll
        for (int r = 0; r < GROUP; ++r) {
            const uint32_t w = __builtin_amdgcn_perm(f[r][1], f[r][0], 0x0c0c0400u) | __builtin_amdgcn_perm(f[r][3], f[r][2], 0x04000c0cu);   // byte 0 of each
            if (s.col_ok && s.row0 + rg + r < s.row_end) store_q_word(w ^ flip, reinterpret_cast<uint32_t*>(o + (rg + r) * a.N));
        }
        if (refill) {
#pragma unroll
            for (int r = 0; r < GROUP; ++r) {
                const int64_t row = nrow0 + rg + r < n.row_end ? nrow0 + rg + r : n.row_end - 1;
                const f32x4r u = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(src + row * a.ldw));
                v[rg + r][0] = u[0]; v[rg + r][1] = u[1]; v[rg + r][2] = u[2]; v[rg + r][3] = u[3];
            }
        }
    }
}

// One step of the streamed kernel.  On entry slot X (`tx`, `vx`) holds a PUBLISHED tile whose range may or may not be
// complete, slot Y (`ty`, `vy`, when `y_valid`) a tile whose loads are in flight and which is NOT published yet.  The
// range of X is polled once without blocking: normally it is complete (X was published a whole tile-load ago), and then
// X's key loads and the next ticket travel while Y lands, is folded and published; only if the range is incomplete, Y is
// published FIRST and the workgroup blocks -- it never waits while it holds an unpublished tile, and the ticket is taken
// behind the wait.  X is then stored row by row while its registers are refilled with the next tile, which leaves X in
// flight and unpublished: the roles swap.  Returns X's new ticket (>= ntiles: the slot is empty).
__device__ __forceinline__ uint32_t stream_step(const ResidentArgs& a, uint32_t tx, float (&vx)[kResRows][4], bool y_valid, uint32_t ty,
                                                float (&vy)[kResRows][4], int lane, int wave, float4 (&s_mn)[kResWaves][kWave],
                                                float4 (&s_mx)[kResWaves][kWave], uint32_t& s_ticket, uint32_t& s_flag, uint64_t& lap0) {
    const StreamTile sx = stream_tile(a, tx, lane, wave);
    const StreamTile sy = stream_tile(a, y_valid ? ty : tx, lane, wave);
    const uint32_t chunks = static_cast<uint32_t>(a.chunks);
    // the poller sits in wave 1: wave 0 may still be draining the key atomics of the tile it published last
    if (threadIdx.x == kWave) s_flag = agent_load(a.counters + sx.range * kResCtrPad) >= chunks ? 1u : 0u;
    __syncthreads();
    bool y_published = !y_valid;
    if (s_flag == 0u) {   // uniform, rare
        if (y_valid) {
            stream_fold_local(lane, wave, vy, s_mn, s_mx);
            __syncthreads();
            if (wave == 0) stream_publish_wave(a, sy, lane, s_mn, s_mx);
            y_published = true;
        }
        if (threadIdx.x == kWave) spin_until(a.counters + sx.range * kResCtrPad, chunks);
        __syncthreads();
    }
    OQ_LAP(1, lap0);
    uint32_t pending = 0;
    if (threadIdx.x == 0) pending = agent_add(a.tickets, 1u);     // behind every wait of this step; its tile is published before the next one
    uint32_t kmx[4], kmn[4];
    stream_keys(a, sx, kmx, kmn);
    if (!y_published) {   // uniform
        stream_fold_local(lane, wave, vy, s_mn, s_mx);
        __syncthreads();
        if (wave == 0) stream_publish_wave(a, sy, lane, s_mn, s_mx);
    }
    OQ_LAP(4, lap0);
    ColQ cq[4];
    stream_params(a, sx, lane, wave, kmx, kmn, cq);
    if (threadIdx.x == 0) s_ticket = pending;
    __syncthreads();
    const uint32_t tn = s_ticket;     // the next write of s_ticket lies behind the first barrier of the next step
    OQ_LAP(2, lap0);
    const bool refill = tn < a.ntiles;                         // uniform
    const StreamTile n = stream_tile(a, refill ? tn : tx, lane, wave);
    stream_rows(a, sx, cq, refill, n, lane, vx);
    OQ_LAP(3, lap0);
#ifdef OQ_TENSOR_STAMPS
    if (threadIdx.x == 0) g_tensor_stamps[blockIdx.x * 8 + 6] += 1;
#endif
    return tn;
}

__global__ __launch_bounds__(kResWaves* kWave, 2) void rtn_resident_stream(const ResidentArgs a) {
    __shared__ float4 s_mn[kResWaves][kWave];
    __shared__ float4 s_mx[kResWaves][kWave];
    __shared__ uint32_t s_ticket, s_flag;
    // the wave index as a scalar: row numbers and row pointers of the tile descriptors stay out of the vector registers
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const uint32_t ntiles = a.ntiles;
    float v0[kResRows][4], v1[kResRows][4];
    uint64_t lap0 = 0;
#ifdef OQ_TENSOR_STAMPS
    if (threadIdx.x < 8) g_tensor_stamps[blockIdx.x * 8 + threadIdx.x] = 0;
    __syncthreads();
    lap0 = wall_clock64();
    const uint64_t kernel_t0 = lap0;
#endif
    if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
    __syncthreads();
    uint32_t ta = s_ticket, tb = ntiles;
    __syncthreads();
    const bool first = ta == 0;
    if (ta < ntiles) {   // uniform
        {
            const StreamTile s = stream_tile(a, ta, lane, wave);
            // a wave whose rows all lie past the range's end (last chunk of a ragged range) repeats the range's last row
            load_tile<true>(a, s.row0 < s.row_end ? s.row0 : s.row_end - 1, s.row_end, s.tile_col0, lane, v0);
            stream_fold_local(lane, wave, v0, s_mn, s_mx);
            __syncthreads();
            if (wave == 0) stream_publish_wave(a, s, lane, s_mn, s_mx);
        }
        if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);   // behind the publish
        __syncthreads();
        tb = s_ticket;
        __syncthreads();
        if (tb < ntiles) {   // slot 1: in flight, published by the first step
            const StreamTile s = stream_tile(a, tb, lane, wave);
            load_tile<true>(a, s.row0 < s.row_end ? s.row0 : s.row_end - 1, s.row_end, s.tile_col0, lane, v1);
        }
        OQ_LAP(0, lap0);
        while (ta < ntiles || tb < ntiles) {
            if (ta < ntiles) ta = stream_step(a, ta, v0, tb < ntiles, tb, v1, lane, wave, s_mn, s_mx, s_ticket, s_flag, lap0);
            if (tb < ntiles) tb = stream_step(a, tb, v1, ta < ntiles, ta, v0, lane, wave, s_mn, s_mx, s_ticket, s_flag, lap0);
        }
    }
#ifdef OQ_TENSOR_STAMPS
    if (threadIdx.x == 0) g_tensor_stamps[blockIdx.x * 8 + 5] = wall_clock64() - kernel_t0;
#endif
    if (a.done != nullptr) {   // uniform; `first` = the workgroup that took ticket 0 (it always exists)
        if (first) clean_state(a, gridDim.x - 1u, kResWaves * kWave);
        else if (threadIdx.x == 0) agent_add(a.done, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-tensor.  Persistent workgroups; tile index = ticket (column tiles fastest: co-resident workgroups stream whole rows).
//
// Forward progress.  Phase A never waits: a workgroup takes tickets, loads, folds the tile into a running min / max
// held in registers, and goes on until the tickets run out (the NEXT ticket is taken before the current tile is
// processed, so a workgroup knows that a tile is its last one while it still holds it).  Then it publishes its range
// once, adds the number of tiles it processed to counters[0] and waits for counters[0] == ntiles.  Every tile was
// taken by a running workgroup that reaches its add without waiting for anybody, so the wait ends whatever the number of
// resident workgroups is; a workgroup that starts late finds no ticket, adds nothing and waits like the others.  Phase B
// never waits either.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// rtn_tensor_onepass: per-tensor, W read once where the chip can hold it and 1.26 times on 4096 x 11008.
//
// ONE 8-wave workgroup per CU (two waves per SIMD: 256 registers per lane each, which a kernel that names accumulation registers
// gets as 128 architectural + 128 accumulation registers).  A tile is 128 x 256 = 16 rows per wave = 64 registers per lane.
// Phase A streams all tiles once (tickets; running min / max in registers) through ONE register slot V; in front of every load
// but the first the tile in V moves to a PARK, the three parks in turn: two in the ACCUMULATION REGISTERS (v_accvgpr_write_b32 /
// v_accvgpr_read_b32: one VALU instruction per dword, no LDS, no memory -- the matrix cores these registers exist for are idle
// here) and one in 128 KB of LDS.  So V and the parks hold the last FOUR tiles a workgroup loaded: 512 KB per CU, 128 MB of the
// matrix stay on the chip across the hand-off; a matrix of up to 1024 tiles is read exactly once, 4096 x 11008 re-reads 46 MB.
// (Rounds 4-6 kept THREE tiles -- two register slots written alternately + LDS -- and re-read 80 MB: 73.5 us against 65 on
// 4096 x 11008, 100 against 96 on 8192 x 8192, same box; a third REGISTER slot was tried in round 4 and spilled to scratch in
// every formulation, a fourth tile in the XCD's L2 in round 6: docs/LAB_NOTES_r06.md.)
// One returning add per workgroup counts the tiles and hands out an arrival slot for its partial range and the ids of the tiles
// it keeps; the workgroup that completes the count folds the slots and broadcasts {go, keys}; kept tiles are quantized from
// registers / LDS; the rows of the other half tiles are dealt out statically and evenly over all waves (no tickets, no
// barriers), re-read most-recent-first and software-pipelined over the two halves of V.
//
// The same with ONE wave per SIMD (4-wave workgroups, 32 rows per wave, 256 + 256 registers) streams phase A just as fast but
// needs 6.8 us instead of 2.3 to quantize and store a tile -- a single wave per SIMD has nobody to fill its bubbles --: 80.4-87.5 us.
// The template parameter is kept for that measurement; only <8> is instantiated.
// ---------------------------------------------------------------------------------------------
constexpr int kHalfRows = kResTileRows / 2;                      // phase-B unit: half a tile, 64 rows x 256 columns
constexpr int kParkBytes = kResWaves * kResRows * kWave * 16;    // one tile: 128 KB
constexpr uint32_t kNoTile = 0xFFFFFFFFu;
#ifndef OQ_T4_GROUP
#define OQ_T4_GROUP 4   /* rows per decision of the quantize + store code (quantize_store_tile) */
#endif
constexpr uint32_t kT4MaxTiles = 32766;               // the fourth kept tile travels as tile + 1 in 15 bits of its arrival slot

// The parks are PHYSICAL accumulation registers named in the assembly text: a[0..63] and a[64..127] (a[0..127] and a[128..255] in
// the one-wave-per-SIMD form).  As operands of the `a` register class they went through the register allocator, which copied them
// at every join of the control flow: 738 v_accvgpr_mov and 1.2 KB of scratch per lane.  Named like this the compiler does not
// know that they hold anything -- and it does place values of its own in a0, a1, ... once a kernel that names accumulation
// registers runs out of architectural ones (128 here).  So the kernel is written to stay below that: V is loaded at ONE place of
// the code, row addresses are scalar, nothing is scheduled across a park's move; tests/test_kernel_resources.py counts the
// v_accvgpr instructions of the code object (exactly the ones written here, no other `a` operand) and fails the build otherwise.
// The clobber statement at the top of the kernel makes the kernel descriptor reserve the registers.
template <int BASE, int H, int HR>
__device__ __forceinline__ void acc_put_half(const float (&v)[HR][4]) {
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v[r][0]), "n"(BASE + (H * HR + r) * 4 + 0));
        asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v[r][1]), "n"(BASE + (H * HR + r) * 4 + 1));
        asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v[r][2]), "n"(BASE + (H * HR + r) * 4 + 2));
        asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v[r][3]), "n"(BASE + (H * HR + r) * 4 + 3));
    }
}
template <int BASE, int H, int HR>
__device__ __forceinline__ void acc_get_half(float (&v)[HR][4]) {
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v[r][0]) : "n"(BASE + (H * HR + r) * 4 + 0));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v[r][1]) : "n"(BASE + (H * HR + r) * 4 + 1));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v[r][2]) : "n"(BASE + (H * HR + r) * 4 + 2));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v[r][3]) : "n"(BASE + (H * HR + r) * 4 + 3));
    }
}

template <int ROWS>
__device__ __forceinline__ void tile4_origin(uint32_t tile, uint32_t ncol, int wave, int64_t& row0, int64_t& col0) {
    const uint32_t row_tile = tile / ncol, col_tile = tile - row_tile * ncol;
    row0 = static_cast<int64_t>(row_tile) * kResTileRows + wave * ROWS;
    col0 = static_cast<int64_t>(col_tile) * kResCols;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES* kWave, WAVES / 4) void rtn_tensor_onepass(const ResidentArgs a) {
    constexpr int ROWS = kResTileRows / WAVES;        // rows per wave: 32 (one wave per SIMD) or 16 (two)
    constexpr int HR = ROWS / 2;                      // V is two halves
    constexpr int THREADS = WAVES * kWave;
    constexpr int A0 = 0, A1 = ROWS * 4;              // first accumulation register of the two parks
    constexpr int GROUP = OQ_T4_GROUP;
    extern __shared__ __attribute__((aligned(16))) unsigned char park_lds[];   // kParkBytes: [wave][row of the wave][lane 64] x 16 B
    __shared__ float s_mn[WAVES], s_mx[WAVES];
    __shared__ uint32_t s_ticket, s_keys[3], s_wsum[WAVES];
    __shared__ uint32_t s_held[kResMaxTensorTiles / 16];          // two bits per tile: its halves
    __shared__ uint32_t s_pref[kResMaxTensorTiles / 16], s_total;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR: row bases are scalar
    const uint32_t ntiles = a.ntiles, ncol = a.ncol_tiles, nunits = 2u * ntiles;
    float4* park = reinterpret_cast<float4*>(park_lds) + (wave * ROWS) * kWave + lane;   // this lane's slot of row 0

    float v[2][HR][4];                          // V: two halves of 16 rows
    if constexpr (WAVES == 4) asm volatile("" : : : "a0", "a255");   // the descriptor reserves the two parks (see acc_put_half): a[0..255]
    else asm volatile("" : : : "a0", "a127");                        // a[0..127]
    float rmn = INFINITY, rmx = -INFINITY;
    uint32_t processed = 0, idV = kNoTile, idA0 = kNoTile, idA1 = kNoTile, idL = kNoTile;
#ifdef OQ_TENSOR_STAMPS
    if (threadIdx.x == 0) g_tensor_stamps[blockIdx.x * 8 + 7] = g_tensor_stamps[blockIdx.x * 8 + 5];   // the end of the previous call
#endif
    OQ_STAMP(0);
    if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
    __syncthreads();
    uint32_t t = s_ticket;
    __syncthreads();
    const bool first = t == 0;

    // one phase-A step: the next ticket travels, tile `t` is loaded into V and folded
    auto load_fold = [&]() {
        __builtin_amdgcn_sched_barrier(0);            // no load of this tile in front of the park of the previous one: it would take a second set of 128 registers
        if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
        idV = t;
        int64_t row0, col0;
        tile4_origin<ROWS>(t, ncol, wave, row0, col0);
        const int64_t r_hi = row0 + HR;
        load_tile<OQ_RES_A_NT, HR>(a, row0 < a.K ? row0 : a.K - 1, a.K, col0, lane, v[0]);
        load_tile<OQ_RES_A_NT, HR>(a, r_hi < a.K ? r_hi : a.K - 1, a.K, col0, lane, v[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < HR; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rmn = nmin(rmn, v[h][r][i]);
                    rmx = nmax(rmx, v[h][r][i]);
                }
        ++processed;
        __syncthreads();
        const uint32_t nxt = s_ticket;
        __syncthreads();
        t = nxt;
    };
    // The tile in V moves to a park in front of every load but the first, the parks in turn: V and the three parks hold the
    // last FOUR tiles a workgroup loaded.  ONE loop body, so that V is loaded at one place of the code (three unrolled copies
    // met at the loop's exits with V in three different register assignments: the joins were made through scratch); the parks
    // are chosen by a uniform branch and define nothing the compiler sees.
    for (uint32_t turn = 0; t < ntiles; ++turn) {
        if (turn != 0u) {
            const uint32_t where = (turn - 1u) % 3u;
            if (where == 0u) {
                acc_put_half<A0, 0>(v[0]);
                acc_put_half<A0, 1>(v[1]);
                idA0 = idV;
            } else if (where == 1u) {
                acc_put_half<A1, 0>(v[0]);
                acc_put_half<A1, 1>(v[1]);
                idA1 = idV;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < HR; ++r) park[(h * HR + r) * kWave] = make_float4(v[h][r][0], v[h][r][1], v[h][r][2], v[h][r][3]);
                idL = idV;
            }
        }
        load_fold();
    }
    // Publish: as rtn_tensor_onepass (one returning add per workgroup, one 16-byte arrival slot, the last finisher folds the
    // slots and broadcasts {go, keys} to 64 replica lines).  The slot names up to four kept tiles.
    uint32_t* replica = a.key_nmin + (blockIdx.x & 63u) * 32u;
    OQ_STAMP(1);
#ifdef OQ_TENSOR_STAMPS
    if (threadIdx.x == 0) g_tensor_stamps[blockIdx.x * 8 + 6] = processed;
#endif
    if (processed) {
        rmn = wave_min(rmn);
        rmx = wave_max(rmx);
        if (lane == 0) { s_mn[wave] = rmn; s_mx[wave] = rmx; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < WAVES; ++w) { rmn = nmin(rmn, s_mn[w]); rmx = nmax(rmx, s_mx[w]); }
            const uint32_t before = agent_add(a.counters, (1u << 16) | processed);
            const uint32_t arrival = before >> 16;
            // kept tiles as tile + 1 (0: none; V always holds one): V | A0 << 16, A1 | LDS << 16 (15 bits); bit 31: this
            // workgroup dropped a tile (it loaded more than the four it keeps)
            const u32x4r slot = {key_of_max(rmx), key_of_min(rmn), (idV + 1u) | ((idA0 + 1u) << 16),
                                 (idA1 + 1u) | ((idL + 1u) << 16) | (processed > 4u ? 0x80000000u : 0u)};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(a.key_max + arrival * 4u), "v"(slot) : "memory");
            s_ticket = ((before & 0xffffu) + processed == ntiles) ? arrival + 1u : 0u;   // the last finisher learns how many arrived
        }
        __syncthreads();
        const uint32_t arrivals = s_ticket;
        if (arrivals != 0u) {                      // the last finisher (uniform over the workgroup): one slot per thread
            uint32_t kmx = 0u, kmn = 0u, dropped = 0u;
            if (threadIdx.x < arrivals) {
                u32x4r line;
                do {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(line) : "v"(a.key_max + threadIdx.x * 4u) : "memory");
                } while (line[2] == 0u);
                kmx = line[0];
                kmn = line[1];
                dropped = line[3] & 0x80000000u;
            }
            dropped = __syncthreads_or(static_cast<int>(dropped != 0u)) ? 0x80000000u : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmx = max(kmx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmx), off, 64)));
                kmn = max(kmn, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmn), off, 64)));
            }
            if (lane == 0) { s_mn[wave] = __uint_as_float(kmx); s_mx[wave] = __uint_as_float(kmn); }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < WAVES; ++w) { kmx = max(kmx, __float_as_uint(s_mn[w])); kmn = max(kmn, __float_as_uint(s_mx[w])); }
                uint32_t* rep = a.key_nmin + lane * 32;
                const u32x4r line = {1u, kmx, kmn, arrivals | dropped};   // bit 31: somebody dropped a tile, i.e. phase B has work
                asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(rep), "v"(line) : "memory");
            }
        }
    }
    if (threadIdx.x == 0) {
        u32x4r line;
        do {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(line) : "v"(replica) : "memory");
            if (line[0] == 0u) __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
        } while (line[0] == 0u);
        s_keys[0] = line[1];
        s_keys[1] = line[2];
        s_keys[2] = line[3];
    }
    __syncthreads();
    OQ_STAMP(2);
    const bool any_dropped = (s_keys[2] & 0x80000000u) != 0u;      // uniform over the grid
    u32x4r my_slot = {0u, 0u, 0u, 0u};
    if (any_dropped && threadIdx.x < (s_keys[2] & 0x7fffffffu))
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(my_slot) : "v"(a.key_max + threadIdx.x * 4u) : "memory");
    const uint32_t nwords = (nunits + 31u) / 32u;
    if (any_dropped)
        for (uint32_t i = threadIdx.x; i < nwords; i += THREADS) s_held[i] = 0u;
    const float gmx = max_of_key(s_keys[0]), gmn = min_of_key(s_keys[1]);
    const int32_t bias = a.grid.qmin < 0 ? 128 : 0;
    ColQ cq[4];
    cq[0] = make_colq(qparam_from_minmax(gmn, gmx, a.grid), gmn, gmx, bias);
    cq[1] = cq[2] = cq[3] = cq[0];
    if (first && threadIdx.x == 0) {
        a.scale[0] = cq[0].scale;
        a.zp[0] = static_cast<uint8_t>(cq[0].zp);
    }
    // The kept tiles go into the bitmap BEFORE V's stores leave: a wait for the slot behind the stores would wait for the stores
    // too (one counter, in order: 2.6 us from here to the bitmap); the rest of the bitmap work runs while they travel.
    uint32_t total = 0, pos = 0, stop = 0;
    if (any_dropped) {
        __syncthreads();                                   // s_held zeroed
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(my_slot) : : "memory");
        {
            const uint32_t ids[4] = {my_slot[2] & 0xffffu, my_slot[2] >> 16, my_slot[3] & 0xffffu, (my_slot[3] >> 16) & 0x7fffu};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ids[i] != 0u) atomicOr(&s_held[(ids[i] - 1u) >> 4], 3u << ((2u * (ids[i] - 1u)) & 31u));   // both halves of the tile
        }
    }
    if (idV != kNoTile) {   // uniform
        int64_t row0, col0;
        tile4_origin<ROWS>(idV, ncol, wave, row0, col0);
        quantize_store_tile<HR, GROUP>(a, cq, v[0], row0, a.K, col0, lane);
        quantize_store_tile<HR, GROUP>(a, cq, v[1], row0 + HR, a.K, col0, lane);
    }
    OQ_STAMP(3);
    // Phase B: as rtn_tensor_onepass (the rows of the unheld halves, most recently read first, dealt out statically and evenly
    // over all waves of the grid), pipelined over the two halves of V; the first step's loads are issued before the parked
    // tiles are stored.
    // Once every wave has its slot this workgroup has read the state for the last time: it counts itself out HERE, not at its
    // end, so that the cleaner finds the count complete when it gets there and the zeroing hides behind the others' phase B.
    if (any_dropped) {
        __syncthreads();                                   // s_held complete: every wave's slot has landed
        if (a.done != nullptr && !first && threadIdx.x == 0) agent_add(a.done, 1u);
        // s_pref[w] = unheld halves in words [0, w): 2048 / THREADS words per thread, a wave scan, the waves' sums through LDS
        {
            constexpr int WPT = 2048 / THREADS;
            uint32_t c[WPT], cnt = 0;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const uint32_t w = threadIdx.x * WPT + i;
                uint32_t bits = w < nwords ? ~s_held[w] : 0u;
                if (w == nwords - 1u && (nunits & 31u)) bits &= (1u << (nunits & 31u)) - 1u;
                c[i] = __popc(bits);
                cnt += c[i];
            }
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), off, 64));
                if (lane >= off) incl += up;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t run = incl - cnt;
#pragma unroll
            for (int w = 0; w < WAVES; ++w)
                if (w < wave) run += s_wsum[w];
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const uint32_t w = threadIdx.x * WPT + i;
                if (w < nwords) s_pref[w] = run;
                run += c[i];
            }
            if (threadIdx.x == THREADS - 1) s_total = run;
        }
        __syncthreads();
        total = s_total;
        const uint32_t total_rows = total * kHalfRows;
        const uint32_t nwv = gridDim.x * WAVES, per = (total_rows + nwv - 1u) / nwv;
        pos = (blockIdx.x * WAVES + wave) * per;
        stop = pos + per < total_rows ? pos + per : total_rows;
    } else if (a.done != nullptr && !first && threadIdx.x == 0) {
        agent_add(a.done, 1u);                             // the replica line was this workgroup's last read of the state
    }
    OQ_STAMP(4);
    int64_t a_row0 = 0, a_end = 0, a_col0 = 0, b_row0 = 0, b_end = 0, b_col0 = 0;
    auto next_step = [&](int64_t& row0, int64_t& row_end, int64_t& col0) -> bool {
        while (pos < stop) {
            const uint32_t j = pos / kHalfRows, r_in = pos - j * kHalfRows;
            uint32_t n = kHalfRows - r_in;
            n = n < static_cast<uint32_t>(HR) ? n : static_cast<uint32_t>(HR);
            n = n < stop - pos ? n : stop - pos;
            pos += n;
            const uint32_t want = total - 1u - j;          // rank of the half among the unheld ones, ascending
            uint32_t lo = 0, hi = nwords - 1u;             // last word whose prefix is <= want
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (s_pref[mid] <= want) lo = mid; else hi = mid - 1u;
            }
            uint32_t bits = ~s_held[lo];
            if (lo == nwords - 1u && (nunits & 31u)) bits &= (1u << (nunits & 31u)) - 1u;
            for (uint32_t skip = want - s_pref[lo]; skip > 0; --skip) bits &= bits - 1u;   // drop the lowest set bits
            const uint32_t unit = lo * 32u + static_cast<uint32_t>(__builtin_ctz(bits));
            const uint32_t tile = unit >> 1, row_tile = tile / ncol, col_tile = tile - row_tile * ncol;
            row0 = static_cast<int64_t>(row_tile) * kResTileRows + (unit & 1u) * kHalfRows + r_in;
            if (row0 >= a.K) continue;                     // uniform over the wave: rows past the matrix
            row_end = row0 + n < a.K ? row0 + n : a.K;
            col0 = static_cast<int64_t>(col_tile) * kResCols;
            return true;
        }
        return false;
    };
    bool a_live = next_step(a_row0, a_end, a_col0);
    if (a_live) load_tile<true, HR>(a, a_row0, a_end, a_col0, lane, v[0]);         // in flight while the parked tiles are stored
    if (idA0 != kNoTile) {   // uniform
        int64_t row0, col0;
        tile4_origin<ROWS>(idA0, ncol, wave, row0, col0);
        acc_get_half<A0, 0>(v[1]);
        quantize_store_tile<HR, GROUP>(a, cq, v[1], row0, a.K, col0, lane);
        acc_get_half<A0, 1>(v[1]);
        quantize_store_tile<HR, GROUP>(a, cq, v[1], row0 + HR, a.K, col0, lane);
    }
    if (idA1 != kNoTile) {   // uniform
        int64_t row0, col0;
        tile4_origin<ROWS>(idA1, ncol, wave, row0, col0);
        acc_get_half<A1, 0>(v[1]);
        quantize_store_tile<HR, GROUP>(a, cq, v[1], row0, a.K, col0, lane);
        acc_get_half<A1, 1>(v[1]);
        quantize_store_tile<HR, GROUP>(a, cq, v[1], row0 + HR, a.K, col0, lane);
    }
    if (idL != kNoTile) {   // uniform: back from LDS into the same lanes' registers
        int64_t row0, col0;
        tile4_origin<ROWS>(idL, ncol, wave, row0, col0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const float4 x = park[(h * HR + r) * kWave];
                v[1][r][0] = x.x; v[1][r][1] = x.y; v[1][r][2] = x.z; v[1][r][3] = x.w;
            }
            quantize_store_tile<HR, GROUP>(a, cq, v[1], row0 + h * HR, a.K, col0, lane);
        }
    }
    bool b_live = next_step(b_row0, b_end, b_col0);
    if (b_live) load_tile<true, HR>(a, b_row0, b_end, b_col0, lane, v[1]);
    while (a_live || b_live) {
        if (a_live) {
            quantize_store_tile<HR, GROUP>(a, cq, v[0], a_row0, a_end, a_col0, lane);
            a_live = next_step(a_row0, a_end, a_col0);
            if (a_live) load_tile<true, HR>(a, a_row0, a_end, a_col0, lane, v[0]);
        }
        if (b_live) {
            quantize_store_tile<HR, GROUP>(a, cq, v[1], b_row0, b_end, b_col0, lane);
            b_live = next_step(b_row0, b_end, b_col0);
            if (b_live) load_tile<true, HR>(a, b_row0, b_end, b_col0, lane, v[1]);
        }
    }
    OQ_STAMP(5);
    if (a.done != nullptr && first) clean_state(a, gridDim.x - 1u, THREADS);   // uniform; `first` = the workgroup that took ticket 0 (it always exists)
}

// Tickets, counters and keys start from zero.  hipMemsetAsync's fill kernel took 4.6 us for these ~100 KB (rocprofv3,
// profiles/r04_strategies_*), a tenth of the whole call; this one is a 16-byte store per lane.
__global__ __launch_bounds__(256) void clear_words_kernel(uint4* p, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------ host side
static int resident_blocks(const void* kernel, size_t dynamic_lds, int threads = kResWaves * kWave) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dynamic_lds) != hipSuccess) return 0;
    return cus * per_cu;
}

// Which kernel takes a channel / tall-group call (speed only, same bytes).  Measured, int8 per channel, same box
// (scripts/quick_strategies.py --lib, round 4):
//                                   4096x11008  4096x4096  8192x8192  11008x4096  16384x4096
//   rtn_resident_groups, 256 rows      65.0        27.7      100        83.7        125.9 us
//   rtn_resident_groups, 128 rows      70          30          -        84            -
//   rtn_resident_stream (128 rows)     62.2        31.8       85.5      61.5         90
// The streamed kernel wins wherever a workgroup refills its slots a few times: taller ranges than 4096 rows, or at least
// four 128-row tiles per workgroup of a 256-CU device (4096 x 11008: 5.4).  With two tiles per workgroup (4096 x 4096) they
// are all it ever loads, one after the other, and the one-tile-per-workgroup kernel keeps the call.
// 64- and 32-row tiles (4 / 7 workgroups per CU) took 92 / 154 us on the first.  OQ_RTN_RES_TILE = 256 | 128 forces
// rtn_resident_groups with that tile height, OQ_RTN_RES_TILE = 1 the streamed kernel (lab switch).
constexpr int64_t kResStreamAbove = 4096;
constexpr int64_t kResStreamTiles = 1024;
static int forced_tile_rows() {
    static const int forced = [] {
        const char* v = getenv("OQ_RTN_RES_TILE");
        const int r = v ? atoi(v) : 0;
        return (r == 256 || r == 128 || r == 1) ? r : 0;
    }();
    return forced;
}
// `ranges`: column tiles x k-groups of the call
static bool groups_streamed(int64_t g, int64_t ranges) {
    if (forced_tile_rows()) return forced_tile_rows() == 1;
    return g > kResStreamAbove || ranges * ceil_div(g, kResGroupTileRows) >= kResStreamTiles;
}
static int groups_tile_rows(int64_t g, int64_t ranges) {
    if (forced_tile_rows() > 1) return forced_tile_rows();
    if (groups_streamed(g, ranges)) return kResGroupTileRows;
    // a call of fewer 256-row tiles than the device has CUs leaves CUs idle: 128-row tiles then (4096 x 2048 int8 per channel:
    // 128 tiles 20.1 us, 256 tiles of 128 rows 17.0; at 256 tiles -- 4096 x 4096, 2048 x 8192 -- the taller tiles win: 27.3 / 24.9 against 30 / 28.0)
    return ranges * ceil_div(g, 256) <= 160 ? kResGroupTileRows : 256;
}
static int64_t ranges_of(int64_t K, int64_t N, int64_t g) { return ceil_div(N, kResCols) * (K / g); }

// workgroups of the channel / tall-group kernel chosen for `g` that the current device runs at once (cached per device)
static int groups_resident(int64_t g, int64_t ranges) {
    static int cache[3][64];
    static bool filled[3][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    const int which = groups_streamed(g, ranges) ? 0 : (groups_tile_rows(g, ranges) == 256 ? 1 : 2);
    if (!filled[which][dev]) {   // benign race: every thread computes the same value
        int cus = 0, per_cu = 0;
        const void* k = which == 0 ? reinterpret_cast<const void*>(rtn_resident_stream)
                      : which == 1 ? reinterpret_cast<const void*>(rtn_resident_groups<16, 16, 4>) : reinterpret_cast<const void*>(rtn_resident_groups<8, 16, 4>);
        const int threads = which == 1 ? 16 * kWave : kResWaves * kWave;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, 0) != hipSuccess) return 0;
        cache[which][dev] = cus * per_cu;
        filled[which][dev] = true;
    }
    return cache[which][dev];
}

size_t rtn_resident_workspace(int64_t K, int64_t N, int32_t strategy, int64_t g) {
    const int64_t kgroups = K / g, chunks = ceil_div(g, groups_tile_rows(g, ranges_of(K, N, g)));
    const int64_t ncol_tiles = ceil_div(N, kResCols);
    if (strategy == OQ_TENSOR) return static_cast<size_t>(kResTensorHeader + (2 * ncol_tiles * ceil_div(K, kResTileRows) + 31) / 32 + 1) * 4 + 256;
    (void)chunks;
    return static_cast<size_t>(2 * kgroups * ncol_tiles * kResCols + ncol_tiles * kgroups * kResCtrPad + kResHeader) * 4 + 256;
}

// true when this path takes the call (the caller falls back to the three-launch path otherwise)
bool rtn_resident_eligible(int64_t K, int64_t N, int64_t ldw, const float* W, const void* q, int32_t strategy, int64_t g, int32_t layout,
                           bool emit_q, size_t workspace_bytes) {
    if (!emit_q || layout != OQ_LAYOUT_KN) return false;
    if ((N % 4) || (ldw % 4) || (reinterpret_cast<uintptr_t>(W) & 15u) || (reinterpret_cast<uintptr_t>(q) & 3u)) return false;
    if (K % g) return false;
    const int64_t chunks = ceil_div(g, strategy == OQ_TENSOR ? kResTileRows : groups_tile_rows(g, ranges_of(K, N, g)));
    const int64_t ntiles = ceil_div(N, kResCols) * (K / g) * chunks;
    if (ntiles >= (1LL << 31) || (strategy == OQ_TENSOR && ntiles > static_cast<int64_t>(kT4MaxTiles))) return false;
    // forward progress needs `chunks` running workgroups (see the kernels): at most 3/4 of what THIS device holds of the kernel
    // that would run (192 of one workgroup on each of 256 CUs; a 32-CU partition takes ranges of up to 24 chunks)
    if (strategy != OQ_TENSOR && chunks > 1 && chunks * 4 > static_cast<int64_t>(groups_resident(g, ranges_of(K, N, g))) * 3) return false;
    return workspace_bytes >= rtn_resident_workspace(K, N, strategy, g);
}

// Ticketed kernels of one device never overlap.  Their forward-progress arguments count on the workgroups the occupancy
// query promised for an otherwise free device: two of them launched from different streams could each hold half of the CUs
// with workgroups that wait for siblings which the other kernel keeps from ever being dispatched (the per-tensor kernel's
// `go` needs every workgroup of its grid to arrive) -- a hang of the whole GPU, not an error.  So the launches of one device
// form a chain.  The chain never keeps a caller's stream HANDLE (a destroyed stream's handle is a dangling
// pointer); it keeps the handle's VALUE as an opaque key that is compared and never handed to HIP again, and the only stream
// the chain ever touches is the one of the call in progress.  (`hipStreamGetId` would be the cleaner key; the libamdhip64 that
// torch 2.10+rocm7.0 loads does not export it.  A new stream at a destroyed stream's address compares equal: the runtime frees
// a queue object only after the queue has drained, so the address cannot come back while the old stream's kernel still runs.)
//   * as long as every ticketed call of the device came from ONE stream, the stream itself orders them: a call pays a
//     capture query and a mutex, no event (recorded behind every launch it cost 2.5-3.5 us per call: 62.7 ->
//     66.2 us per channel call on 4096 x 11008);
//   * the first call from a SECOND stream blocks the host once (`hipDeviceSynchronize`: whatever the first stream queued has
//     ended, whether or not that stream still exists) and switches the device to the eager form for good;
//   * eager form: every call makes its stream wait (device side) for the event of the previous ticketed launch and records
//     the event behind its own launch -- on its own, live stream.
// A capturing stream never gets here (rtn.hip takes the three-launch path, which has no tickets: a replayed graph is ordered
// against nothing the library can see).  Kernels of OTHER processes on the same GPU are out of reach: include/oq_hip.h says so.
struct TicketChain {
    std::mutex m;
    hipEvent_t ev = nullptr;
    uintptr_t last_id = 0;
    bool any = false, eager = false, recorded = false;
};
static TicketChain g_chain[64];

bool rtn_stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cap != hipStreamCaptureStatusNone;
}

static int32_t rtn_resident_launch(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy, int64_t g, uint8_t* q,
                                   float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s, bool zeroed_state);

// The chain as two calls around a launch (also used by rtn.hip for the fused group launch whose appended blocks wait for its
// main blocks).  `ticket_chain_begin` takes the device's mutex, makes `s` wait as described above and returns OQ_OK holding the
// mutex; `ticket_chain_end` records the event (eager form) and releases it.  A failing begin holds nothing.
static thread_local TicketChain* t_chain_held = nullptr;
int32_t ticket_chain_begin(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(OQ_ERR_LAUNCH, "rtn: no current device");
#ifdef OQ_NO_TICKET_CHAIN      /* lab: what the chain costs a single stream */
    return OQ_OK;
#endif
    OQ_REQUIRE(!rtn_stream_is_capturing(s), OQ_ERR_UNSUPPORTED, "rtn: a ticketed kernel cannot be captured into a graph (its replays would be ordered against nothing)");
    const uintptr_t id = reinterpret_cast<uintptr_t>(s);      // an opaque key from here on
    TicketChain& c = g_chain[dev];
    c.m.lock();                                                // wait + launch + record are one step of the chain
    if (c.any && !c.eager && c.last_id != id) {
        if (hipDeviceSynchronize() != hipSuccess) { c.m.unlock(); return fail(OQ_ERR_LAUNCH, "rtn: cannot order the launch behind the previous ticketed kernel"); }
        c.eager = true;
    }
    if (c.eager) {
        if (c.ev == nullptr && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) {
            c.ev = nullptr;
            c.m.unlock();
            return fail(OQ_ERR_LAUNCH, "rtn: cannot create the event that orders ticketed launches");
        }
        if (c.recorded && c.last_id != id && hipStreamWaitEvent(s, c.ev, 0) != hipSuccess) {
            c.m.unlock();
            return fail(OQ_ERR_LAUNCH, "rtn: cannot order the launch behind the previous ticketed kernel");
        }
    }
    t_chain_held = &c;
    return OQ_OK;
}
void ticket_chain_end(hipStream_t s) {
    TicketChain* c = t_chain_held;
    if (c == nullptr) return;                                  // OQ_NO_TICKET_CHAIN
    t_chain_held = nullptr;
    if (c->eager) {      // also after a launch that failed half way (its clear launch may be in the stream)
        if (hipEventRecord(c->ev, s) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();       // nothing of this call may still run when the next one starts
            c->recorded = false;
        } else {
            c->recorded = true;
        }
    }
    c->last_id = reinterpret_cast<uintptr_t>(s);
    c->any = true;
    c->m.unlock();
}

int32_t rtn_resident_impl(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy, int64_t g, uint8_t* q,
                          float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s, bool zeroed_state) {
    const int32_t pre = ticket_chain_begin(s);
    if (pre != OQ_OK) return pre;
    const int32_t st = rtn_resident_launch(W, K, N, ldw, grid, strategy, g, q, scale, zp, layout, workspace, workspace_bytes, s, zeroed_state);
    ticket_chain_end(s);
    return st;
}

static int32_t rtn_resident_launch(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy, int64_t g, uint8_t* q,
                                   float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s, bool zeroed_state) {
    ResidentArgs a;
    const int tile_rows = strategy == OQ_TENSOR ? kResTileRows : groups_tile_rows(g, ranges_of(K, N, g));
    a.W = W; a.K = K; a.N = N; a.ldw = ldw; a.g = g; a.kgroups = K / g; a.chunks = ceil_div(g, strategy == OQ_TENSOR ? kResTileRows : tile_rows);
    a.q = q; a.scale = scale; a.zp = zp; a.grid = grid; a.layout = layout;
    a.ncol_tiles = static_cast<uint32_t>(ceil_div(N, kResCols));
    const size_t need = rtn_resident_workspace(K, N, strategy, g);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "rtn: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    uint32_t* base = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    a.done = nullptr; a.clean_base = base; a.clean_words = static_cast<uint32_t>((need - 256) / 4);
    if (zeroed_state) {
        a.done = base + 40;           // a line of its own between the tickets and the counters
    } else {
        const uint32_t n16 = static_cast<uint32_t>((need - 256 + 15) / 16);   // `need` counts whole words past the aligned base; the +256 slack covers the round-up
        hipLaunchKernelGGL(clear_words_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint4*>(base), n16);
    }
    a.tickets = base;                 // [0] phase A; phase B's at [32] (see the kernels: tickets + 32)
    a.counters = base + kResHeader;   // groups: one per range
    if (strategy == OQ_TENSOR) {
        a.ntiles = a.ncol_tiles * static_cast<uint32_t>(ceil_div(K, kResTileRows));
        a.counters = base + 64;                  // tiles counted
        a.key_max = base + 128;                  // one 16-byte slot {1, max key, complemented min key} per arrival (<= 512 workgroups)
        a.key_nmin = base + 128 + 64 * 32;       // 64 replicas x 32 words: {go, final max key, final complemented min key}
        a.held = base + kResTensorHeader;
        // once per device: the attribute belongs to that device's copy of the kernel; the occupancy does not change either
        // (two host calls of several microseconds each: a 256 x 512 call is host-bound)
        static int resident_of[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(OQ_ERR_LAUNCH, "rtn: no current device");
        const void* kernel = reinterpret_cast<const void*>(rtn_tensor_onepass<kResWaves>);
        if (resident_of[dev] == 0) {   // benign race: every thread computes the same value
            if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kParkBytes) != hipSuccess)
                return fail(OQ_ERR_LAUNCH, "rtn: %d bytes of LDS refused", kParkBytes);
            resident_of[dev] = resident_blocks(kernel, kParkBytes);
        }
        const int resident = resident_of[dev];
        OQ_REQUIRE(resident > 0, OQ_ERR_LAUNCH, "rtn: occupancy query failed");
        uint32_t blocks = a.ntiles < static_cast<uint32_t>(resident) ? a.ntiles : static_cast<uint32_t>(resident);
        if (blocks > static_cast<uint32_t>(kResWaves * kWave)) blocks = kResWaves * kWave;   // the arrival slots: one per thread of the last finisher
        hipLaunchKernelGGL(rtn_tensor_onepass<kResWaves>, dim3(blocks), dim3(kResWaves * kWave), kParkBytes, s, a);
        return check_launch("rtn_tensor_onepass");
    }
    const int64_t ranges = static_cast<int64_t>(a.ncol_tiles) * a.kgroups;
    a.ntiles = static_cast<uint32_t>(ranges * a.chunks);
    a.held = nullptr;
    a.key_max = a.counters + ranges * kResCtrPad;
    a.key_nmin = a.key_max + a.kgroups * static_cast<int64_t>(a.ncol_tiles) * kResCols;
    if (!groups_streamed(g, ranges)) {
        if (tile_rows == 256) hipLaunchKernelGGL((rtn_resident_groups<16, 16, 4>), dim3(a.ntiles), dim3(16 * kWave), 0, s, a);
        else hipLaunchKernelGGL((rtn_resident_groups<8, 16, 4>), dim3(a.ntiles), dim3(8 * kWave), 0, s, a);
    } else {
        const int resident = groups_resident(g, ranges);
        // forward progress needs at least `chunks` running workgroups (see the kernel): rtn_resident_eligible has checked it
        OQ_REQUIRE(resident > 0 && (a.chunks <= resident || a.ntiles <= static_cast<uint32_t>(resident)), OQ_ERR_UNSUPPORTED,
                   "rtn: %lld chunks per range need as many resident workgroups, the device holds %d", (long long)a.chunks, resident);
        const uint32_t blocks = a.ntiles < static_cast<uint32_t>(resident) ? a.ntiles : static_cast<uint32_t>(resident);
        hipLaunchKernelGGL(rtn_resident_stream, dim3(blocks), dim3(kResWaves * kWave), 0, s, a);
    }
    return check_launch("rtn_resident_groups");
}

}  // namespace oq

#ifdef OQ_SPIN_LIMIT
extern "C" int32_t oq_lab_spin_timeouts(uint32_t* host_out, int32_t reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(oq::g_spin_timeouts), sizeof(uint32_t)) != hipSuccess) return -1;
    const uint32_t zero = 0;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(oq::g_spin_timeouts), &zero, sizeof(uint32_t)) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef OQ_TENSOR_STAMPS
extern "C" int32_t oq_lab_tensor_stamps(uint64_t* host_out /* [512 * 8] */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(oq::g_tensor_stamps), sizeof(uint64_t) * 512 * 8) == hipSuccess ? 0 : -1;
}
#endif
