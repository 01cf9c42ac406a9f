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
// `rtn_resident_groups`: channel and tall groups.  The tiles of one range (one column tile x one k-group) have
// consecutive tickets, are loaded at about the same time by different workgroups, and every one of them waits for its
// siblings' partial ranges before it quantizes: one read of W, no second pass.
// `rtn_tensor_onepass`: per-tensor.  Phase A streams all tiles once (running min / max in registers, no barrier per tile);
// every workgroup KEEPS the last tile it loaded in registers and the first half of the one before in LDS (96 MB of the
// matrix stay on the chip); one returning add per workgroup counts the tiles and hands out an arrival slot for its partial
// range and the ids of the tiles it keeps; the workgroup that completes the count folds the slots and broadcasts {go, keys};
// kept tiles are quantized from registers / LDS, the other half tiles are dealt out statically (no tickets, no barriers)
// and re-read most-recent-first: the Infinity Cache still holds them.  A matrix of up to 768 tiles is read exactly once.
#include "oq_common.hpp"

#include <cstdlib>

namespace oq {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

constexpr int kResWaves = 8;
constexpr int kResRows = 16;                      // rows per wave
constexpr int kResTileRows = kResWaves * kResRows;  // 128, the chunk height of the two-pass path (rtn.hip kChunkRows)
constexpr int kResCols = 256;                     // 64 lanes x 4 columns
constexpr int kResHeader = 160;
constexpr int kResMaxTensorTiles = 32768;         // bitmap of kept half tiles in LDS (8 KB): 1 G parameters; larger tensors take the three-launch path
constexpr int kResTensorHeader = 128 + 64 * 32 + 64 * 32;   // tickets / counter, 64 key shards, 64 result replicas (a 128-byte line each)
constexpr int kResGroupTileRows = 128;            // default tile height of rtn_resident_groups (see groups_tile_rows)
constexpr int kResCtrPad = 32;                   // uint32 words per range counter: a 128-byte line each (hundreds of workgroups poll them)
#ifndef OQ_RES_A_NT
#define OQ_RES_A_NT false   /* default-policy loads in phase A keep the lines in the Infinity Cache for phase B: 75 us against 78 with nt */
#endif
#ifndef OQ_RES_SLEEP
#define OQ_RES_SLEEP 8
#endif                   // uint32 words in front of the arrays: tickets, keys, counter on 128-byte lines of their own

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
    uint32_t* key_max;    // [slots] ordered key of the running maximum
    uint32_t* key_nmin;   // [slots] complement of the ordered key of the running minimum (kept as a maximum)
    uint32_t* counters;   // groups: one per (column tile, k-group); tensor: [0] = tiles counted
    uint32_t* tickets;    // [0] phase A, [32] phase B
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

__device__ __forceinline__ void spin_until(const uint32_t* p, uint32_t target) {
    while (agent_load(p) < target) __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
}

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

// K1 from registers + [K, N] byte stores (one dword = a lane's four columns of a row).
template <int ROWS = kResRows>
__device__ __forceinline__ void quantize_store_tile(const ResidentArgs& a, const ColQ (&cq)[4], float (&v)[ROWS][4], int64_t row0,
                                                    int64_t row_end, int64_t tile_col0, int lane) {
    const int32_t qmin = a.grid.qmin, qmax = a.grid.qmax;
    const int32_t bias = qmin < 0 ? 128 : 0;
    const float lo_b = static_cast<float>(qmin + bias), hi_b = static_cast<float>(qmax + bias);
    const uint32_t flip = bias ? 0x80808080u : 0u;
    const bool col_ok = tile_col0 + lane * 4 < a.N;
    uint8_t* o = a.q + row0 * a.N + tile_col0 + lane * 4;
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
        if (col_ok && row0 + r < row_end) __builtin_nontemporal_store(w ^ flip, reinterpret_cast<uint32_t*>(o + r * a.N));
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
    quantize_store_tile<ROWS>(a, cq, v, row0, row_end, tile_col0, lane);
    if (a.done != nullptr) {   // uniform
        // every state access of this workgroup is complete (its key loads returned, its counter add was seen by its own poll)
        if (t + 1u == a.ntiles) clean_state(a, a.ntiles - 1u, WAVES * kWave);
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
constexpr int kParkWaves = kResWaves / 2;                        // the waves whose rows (the first half of a tile) are parked in LDS
constexpr int kParkBytes = kParkWaves * kResRows * kWave * 16;   // 64 KB
constexpr int kHalfRows = kResTileRows / 2;                      // phase-B unit: half a tile, 8 rows per wave

__global__ __launch_bounds__(kResWaves* kWave, 4) void rtn_tensor_onepass(const ResidentArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char park_lds[];   // kParkBytes: [wave < 4][row 16][lane 64] x 16 B
    __shared__ float s_mn[kResWaves], s_mx[kResWaves];
    __shared__ uint32_t s_ticket, s_keys[3];
    __shared__ uint32_t s_held[kResMaxTensorTiles / 16];          // two bits per tile: its halves
    __shared__ uint32_t s_pref[kResMaxTensorTiles / 16], s_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ntiles = a.ntiles, ncol = a.ncol_tiles, nunits = 2u * ntiles;
    float4* park = reinterpret_cast<float4*>(park_lds) + (wave * kResRows) * kWave + lane;   // this lane's slot of row 0 (waves < 4 only)

    float v[kResRows][4];
    float rmn = INFINITY, rmx = -INFINITY;
    uint32_t processed = 0, mine = 0xFFFFFFFFu, parked = 0xFFFFFFFFu;
    bool first = false;
    if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
    __syncthreads();
    uint32_t t = s_ticket;
    __syncthreads();
    first = t == 0;
    while (t < ntiles) {
        if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);   // the next ticket travels while this tile loads
        if (processed > 0 && wave < kParkWaves) {
            // The tile in `v` is about to be overwritten: its first half (the rows of waves 0-3) is parked in LDS.  If the
            // tile that is loaded now turns out to be this workgroup's last one, the parked half needs no second read either:
            // 64 KB of registers + 64 KB of LDS per workgroup = 96 MB of the matrix stay on the chip across the hand-off.
#pragma unroll
            for (int r = 0; r < kResRows; ++r) park[r * kWave] = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
        }
        if (processed > 0) parked = mine;
        mine = t;
        const uint32_t row_tile = t / ncol, col_tile = t - row_tile * ncol;
        const int64_t row0 = static_cast<int64_t>(row_tile) * kResTileRows + wave * kResRows;
        load_tile<OQ_RES_A_NT>(a, row0 < a.K ? row0 : a.K - 1, a.K, static_cast<int64_t>(col_tile) * kResCols, lane, v);
#pragma unroll
        for (int r = 0; r < kResRows; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rmn = nmin(rmn, v[r][i]);
                rmx = nmax(rmx, v[r][i]);
            }
        ++processed;
        __syncthreads();
        const uint32_t nxt = s_ticket;
        __syncthreads();
        if (nxt >= ntiles) break;   // uniform: the tile in `v` (`mine`) is this workgroup's last one and stays
        t = nxt;
    }
    // Publish.  512 workgroups adding to ONE key pair and polling ONE counter serialise at the memory side (an atomic on a
    // contended line takes 11-13 ns: MI355X_MICROARCH.md "fanin"; the first build of this kernel spent 40 us here).  So: ONE
    // returning add per workgroup counts its tiles (low 16 bits) and hands out an arrival number (high bits); the partial
    // range goes, as one 16-byte store {1, max key, min key}, into the slot of that number -- no atomics on shared keys, no
    // drain in front of the add; the workgroup whose add completes the count reads the slots of all arrivals (re-reading the
    // rare one whose store is still in flight), folds them and broadcasts {go, keys} to 64 replica lines; everybody polls
    // its own replica (8 pollers per line).
    uint32_t* replica = a.key_nmin + (blockIdx.x & 63u) * 32u;      // {go, final max key, final complemented min key}
    if (processed) {
        rmn = wave_min(rmn);
        rmx = wave_max(rmx);
        if (lane == 0) { s_mn[wave] = rmn; s_mx[wave] = rmx; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kResWaves; ++w) { rmn = nmin(rmn, s_mn[w]); rmx = nmax(rmx, s_mx[w]); }
            const uint32_t before = agent_add(a.counters, (1u << 16) | processed);
            const uint32_t arrival = before >> 16;
            // the slot also says which tiles this workgroup keeps (registers: `mine`, both halves; LDS: the first half of
            // `parked`): everybody builds the bitmap of kept halves from the slots, no shared bitmap, no atomics, no drain
            const u32x4r slot = {key_of_max(rmx), key_of_min(rmn), mine + 1u, parked + 1u /* 0: none */};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(a.key_max + arrival * 4u), "v"(slot) : "memory");
            s_ticket = ((before & 0xffffu) + processed == ntiles) ? arrival + 1u : 0u;   // the last finisher learns how many arrived
        }
        __syncthreads();
        const uint32_t arrivals = s_ticket;
        if (arrivals != 0u) {                      // the last finisher (uniform over the workgroup): one slot per thread
            uint32_t kmx = 0u, kmn = 0u;
            if (threadIdx.x < arrivals) {
                u32x4r line;
                do {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(line) : "v"(a.key_max + threadIdx.x * 4u) : "memory");
                } while (line[2] == 0u);
                kmx = line[0];
                kmn = line[1];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmx = max(kmx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmx), off, 64)));
                kmn = max(kmn, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmn), off, 64)));
            }
            if (lane == 0) { s_mn[wave] = __uint_as_float(kmx); s_mx[wave] = __uint_as_float(kmn); }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < kResWaves; ++w) { kmx = max(kmx, __float_as_uint(s_mn[w])); kmn = max(kmn, __float_as_uint(s_mx[w])); }
                // {go, max key, min key} as ONE 16-byte agent-scope store per replica: a 16-byte piece of a line is written by
                // one request (MI355X_MICROARCH.md: 16-byte sc1 granules are observed untorn), so no drain between keys and `go`
                uint32_t* rep = a.key_nmin + lane * 32;
                const u32x4r line = {1u, kmx, kmn, arrivals};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(rep), "v"(line) : "memory");
            }
        }
    }
    // One 16-byte agent-scope load per poll returns {go, max key, min key} together (the keys were performed before `go`
    // was stored, and a 16-byte piece of a line is read in one request), so no second round trip for the keys.
    if (threadIdx.x == 0) {
        u32x4r line;
        do {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(line) : "v"(replica) : "memory");
            if (line[0] == 0u) __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
        } while (line[0] == 0u);
        s_keys[0] = line[1];
        s_keys[1] = line[2];
        s_keys[2] = line[3];      // how many workgroups arrived = how many slots describe kept tiles
    }
    // Every slot is complete once `go` is up (the last finisher read them all).  One slot per thread: its load is issued here
    // and lands while the kept tiles are quantized below; the bitmap of kept halves is then built in LDS from the slots.
    __syncthreads();
    u32x4r my_slot = {0u, 0u, 0u, 0u};
    if (threadIdx.x < s_keys[2])
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(my_slot) : "v"(a.key_max + threadIdx.x * 4u) : "memory");
    for (uint32_t i = threadIdx.x; i < (nunits + 31u) / 32u; i += kResWaves * kWave) s_held[i] = 0u;
    const float gmx = max_of_key(s_keys[0]), gmn = min_of_key(s_keys[1]);
    const int32_t bias = a.grid.qmin < 0 ? 128 : 0;
    ColQ cq[4];
    cq[0] = make_colq(qparam_from_minmax(gmn, gmx, a.grid), gmn, gmx, bias);
    cq[1] = cq[2] = cq[3] = cq[0];
    if (first && threadIdx.x == 0) {
        a.scale[0] = cq[0].scale;
        a.zp[0] = static_cast<uint8_t>(cq[0].zp);
    }
    if (processed) {
        const uint32_t row_tile = mine / ncol, col_tile = mine - row_tile * ncol;
        quantize_store_tile(a, cq, v, static_cast<int64_t>(row_tile) * kResTileRows + wave * kResRows, a.K,
                            static_cast<int64_t>(col_tile) * kResCols, lane);
        if (parked != 0xFFFFFFFFu && wave < kParkWaves) {   // the parked half: back from LDS into the same lanes' registers
#pragma unroll
            for (int r = 0; r < kResRows; ++r) {
                const float4 x = park[r * kWave];
                v[r][0] = x.x; v[r][1] = x.y; v[r][2] = x.z; v[r][3] = x.w;
            }
            const uint32_t prow = parked / ncol, pcol = parked - prow * ncol;
            quantize_store_tile(a, cq, v, static_cast<int64_t>(prow) * kResTileRows + wave * kResRows, a.K,
                                static_cast<int64_t>(pcol) * kResCols, lane);
        }
    }
    // Phase B: the half tiles nobody kept.  No tickets and no barriers any more: the bitmap is the same for everybody, so
    // the unheld halves are dealt out statically -- the j-th unheld half (counted from the END: most recently read first, the
    // Infinity Cache still holds them) goes to team j mod (2 x workgroups), a team = four waves = 64 rows x 256 columns with
    // 16 loads per lane in flight.  (With whole tiles per ticket and a barrier per tile this phase took two rounds of 12 us
    // for 1.7 tiles per workgroup; half tiles of 8 rows per wave were slower still: half the loads in flight.)
    // Safe without residency assumptions: nobody waits in this phase; a workgroup that starts late does its share late.
    const uint32_t nwords = (nunits + 31u) / 32u;
    __syncthreads();                                   // s_held zeroed
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(my_slot) : : "memory");
    if (my_slot[2] != 0u) {
        const uint32_t u = 2u * (my_slot[2] - 1u);
        atomicOr(&s_held[u >> 5], 3u << (u & 31u));
        if (my_slot[3] != 0u) {
            const uint32_t up = 2u * (my_slot[3] - 1u);
            atomicOr(&s_held[up >> 5], 1u << (up & 31u));
        }
    }
    __syncthreads();                                   // s_held complete
    if (wave == 0) {                                   // s_pref[w] = unheld halves in words [0, w); 32 words per lane
        uint32_t cnt = 0;
        const uint32_t w0 = lane * (kResMaxTensorTiles / 16 / kWave), w1 = w0 + kResMaxTensorTiles / 16 / kWave;
        for (uint32_t w = w0; w < w1 && w < nwords; ++w) {
            uint32_t bits = ~s_held[w];
            if (w == nwords - 1u && (nunits & 31u)) bits &= (1u << (nunits & 31u)) - 1u;
            cnt += __popc(bits);
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), off, 64));
            if (lane >= off) incl += up;
        }
        uint32_t run = incl - cnt;
        for (uint32_t w = w0; w < w1 && w < nwords; ++w) {
            s_pref[w] = run;
            uint32_t bits = ~s_held[w];
            if (w == nwords - 1u && (nunits & 31u)) bits &= (1u << (nunits & 31u)) - 1u;
            run += __popc(bits);
        }
        if (lane == 63) s_total = incl;
    }
    __syncthreads();
    const uint32_t total = s_total;
    const uint32_t team = blockIdx.x * 2u + (wave >> 2), nteams = gridDim.x * 2u;
    for (uint32_t j = team; j < total; j += nteams) {
        const uint32_t want = total - 1u - j;          // rank of the unit among the unheld ones, ascending
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
        const int64_t row0 = static_cast<int64_t>(row_tile) * kResTileRows + (unit & 1u) * kHalfRows + (wave & 3) * kResRows;
        load_tile(a, row0 < a.K ? row0 : a.K - 1, a.K, static_cast<int64_t>(col_tile) * kResCols, lane, v);
        quantize_store_tile(a, cq, v, row0, a.K, static_cast<int64_t>(col_tile) * kResCols, lane);
    }
    if (a.done != nullptr) {   // uniform; `first` = the workgroup that took ticket 0 (it always exists)
        if (first) clean_state(a, gridDim.x - 1u, kResWaves * kWave);
        else if (threadIdx.x == 0) agent_add(a.done, 1u);
    }
}

// Tickets, counters and keys start from zero.  hipMemsetAsync's fill kernel took 4.6 us for these ~100 KB (rocprofv3,
// profiles/r04_strategies_*), a tenth of the whole call; this one is a 16-byte store per lane.
__global__ __launch_bounds__(256) void clear_words_kernel(uint4* p, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------ host side
static int resident_blocks(const void* kernel, size_t dynamic_lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kResWaves * kWave, dynamic_lds) != hipSuccess) return 0;
    return cus * per_cu;
}

// Tile of the channel / tall-group kernel: rows per workgroup.  Speed only (OQ_RTN_RES_TILE = 256 | 128 for experiments).
// 64- and 32-row tiles (4 / 7 workgroups per CU) were built and measured SLOWER (92 / 154 us against 70 on 4096 x 11008): more
// siblings per range mean more key atomics and a longer wait for the last of them.
static int groups_tile_rows(int64_t g) {
    static const int forced = [] {
        const char* v = getenv("OQ_RTN_RES_TILE");
        const int r = v ? atoi(v) : 0;
        return (r == 256 || r == 128) ? r : 0;
    }();
    if (forced) return forced;
    // measured on 4096 x 11008 / 4096 x 4096 / 11008 x 4096 (int8 channel): 256-row tiles 65 / 29 / 95 us, 128-row tiles
    // 70 / 30 / 84 us: fewer, larger tiles win while a range has few of them (less skew, fewer atomics), smaller ones when
    // a column is tall (one 16-wave workgroup per CU waits too long for 42 siblings)
    return g <= 8192 ? 256 : kResGroupTileRows;
}

size_t rtn_resident_workspace(int64_t K, int64_t N, int32_t strategy, int64_t g) {
    const int64_t kgroups = K / g, chunks = ceil_div(g, groups_tile_rows(g));
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
    const int64_t chunks = ceil_div(g, strategy == OQ_TENSOR ? kResTileRows : groups_tile_rows(g));
    const int64_t ntiles = ceil_div(N, kResCols) * (K / g) * chunks;
    if (ntiles >= (1LL << 31) || (strategy == OQ_TENSOR && ntiles > kResMaxTensorTiles)) return false;
    if (strategy != OQ_TENSOR && chunks > 192) return false;   // forward progress needs `chunks` running workgroups (256 CUs)
    return workspace_bytes >= rtn_resident_workspace(K, N, strategy, g);
}

int32_t rtn_resident_impl(const float* W, int64_t K, int64_t N, int64_t ldw, const QGrid& grid, int32_t strategy, int64_t g, uint8_t* q,
                          float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s, bool zeroed_state) {
    ResidentArgs a;
    const int tile_rows = groups_tile_rows(g);
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
        // per launch, not once: the attribute belongs to the current device's copy of the kernel
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(rtn_tensor_onepass), hipFuncAttributeMaxDynamicSharedMemorySize, kParkBytes) != hipSuccess)
            return fail(OQ_ERR_LAUNCH, "rtn: %d bytes of LDS refused", kParkBytes);
        const int resident = resident_blocks(reinterpret_cast<const void*>(rtn_tensor_onepass), kParkBytes);
        OQ_REQUIRE(resident > 0, OQ_ERR_LAUNCH, "rtn: occupancy query failed");
        uint32_t blocks = a.ntiles < static_cast<uint32_t>(resident) ? a.ntiles : static_cast<uint32_t>(resident);
        if (blocks > 512u) blocks = 512u;        // the arrival slots (and one slot per thread of the last finisher)
        hipLaunchKernelGGL(rtn_tensor_onepass, dim3(blocks), dim3(kResWaves * kWave), kParkBytes, s, a);
        return check_launch("rtn_tensor_onepass");
    }
    const int64_t ranges = static_cast<int64_t>(a.ncol_tiles) * a.kgroups;
    a.ntiles = static_cast<uint32_t>(ranges * a.chunks);
    a.held = nullptr;
    a.key_max = a.counters + ranges * kResCtrPad;
    a.key_nmin = a.key_max + a.kgroups * static_cast<int64_t>(a.ncol_tiles) * kResCols;
    if (tile_rows == 256) hipLaunchKernelGGL((rtn_resident_groups<16, 16, 4>), dim3(a.ntiles), dim3(16 * kWave), 0, s, a);
    else hipLaunchKernelGGL((rtn_resident_groups<8, 16, 4>), dim3(a.ntiles), dim3(8 * kWave), 0, s, a);
    return check_launch("rtn_resident_groups");
}

}  // namespace oq
