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
// every workgroup KEEPS the last tile it loaded; when all tiles are counted the kept tiles are quantized from registers
// and the others are re-read in reverse order (most recently read first: the Infinity Cache still holds them).  A matrix
// of up to (resident workgroups) tiles -- 64 MB on this chip -- is read exactly once.
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
constexpr int kResMaxTensorTiles = 65536;         // bitmap of kept tiles in LDS (8 KB): 2 G parameters; larger tensors take the three-launch path
constexpr int kResTensorHeader = 128 + 64 * 32 + 64 * 32;   // tickets / counter, 64 key shards, 64 result replicas (a 128-byte line each)
constexpr int kResGroupTileRows = 128;            // default tile height of rtn_resident_groups (see groups_tile_rows)
constexpr int kResCtrPad = 32;                   // uint32 words per range counter: a 128-byte line each (hundreds of workgroups poll them)
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
    uint32_t* held;       // tensor: bitmap over tiles, 1 = quantized from its owner's registers
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
__global__ __launch_bounds__(kResWaves* kWave, 4) void rtn_tensor_onepass(const ResidentArgs a) {
    __shared__ float s_mn[kResWaves], s_mx[kResWaves];
    __shared__ uint32_t s_ticket, s_keys[2], s_first_b;
    __shared__ uint32_t s_held[kResMaxTensorTiles / 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ntiles = a.ntiles, ncol = a.ncol_tiles;

    float v[kResRows][4];
    float rmn = INFINITY, rmx = -INFINITY;
    uint32_t processed = 0, mine = 0xFFFFFFFFu;
    bool first = false;
    if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);
    __syncthreads();
    uint32_t t = s_ticket;
    __syncthreads();
    first = t == 0;
    while (t < ntiles) {
        if (threadIdx.x == 0) s_ticket = agent_add(a.tickets, 1u);   // the next ticket travels while this tile loads
        const uint32_t row_tile = t / ncol, col_tile = t - row_tile * ncol;
        const int64_t row0 = static_cast<int64_t>(row_tile) * kResTileRows + wave * kResRows;
        load_tile<false>(a, row0 < a.K ? row0 : a.K - 1, a.K, static_cast<int64_t>(col_tile) * kResCols, lane, v);
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
        if (nxt >= ntiles) { mine = t; break; }   // uniform: the tile in `v` is this workgroup's last one and stays
        t = nxt;
    }
    // Publish: 512 workgroups adding to ONE key pair and polling ONE counter serialise at the memory side (an atomic on a
    // contended line takes 11-13 ns: MI355X_MICROARCH.md "fanin"; the first build of this kernel spent 40 us here).  So the
    // partial ranges go to 64 shards, the tiles are counted on one word by ONE returning add per workgroup, and the
    // workgroup whose add completes the count folds the shards and broadcasts the result with a "go" word to 64 replica
    // lines; everybody polls its own replica (8 pollers per line).
    uint32_t* shard = a.key_max + (blockIdx.x & 63u) * 32u;          // {max key, complemented min key} of this shard
    uint32_t* replica = a.key_nmin + (blockIdx.x & 63u) * 32u;      // {go, final max key, final complemented min key}
    if (processed) {
        if (threadIdx.x == 0) __hip_atomic_fetch_or(a.held + (mine >> 5), 1u << (mine & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rmn = wave_min(rmn);
        rmx = wave_max(rmx);
        if (lane == 0) { s_mn[wave] = rmn; s_mx[wave] = rmx; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kResWaves; ++w) { rmn = nmin(rmn, s_mn[w]); rmx = nmax(rmx, s_mx[w]); }
            agent_max(shard, key_of_max(rmx));
            agent_max(shard + 1, key_of_min(rmn));
            drain_vmem();                          // range and `held` flag are performed before the tiles are counted
            const uint32_t before = agent_add(a.counters, processed);
            s_ticket = (before + processed == ntiles) ? 1u : 0u;
        }
        __syncthreads();
        if (s_ticket != 0u && wave == 0) {         // the last finisher: every other workgroup's shard update precedes its add
            uint32_t kmx = agent_load(a.key_max + lane * 32), kmn = agent_load(a.key_max + lane * 32 + 1);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmx = max(kmx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmx), off, 64)));
                kmn = max(kmn, static_cast<uint32_t>(__shfl_xor(static_cast<int>(kmn), off, 64)));
            }
            uint32_t* rep = a.key_nmin + lane * 32;
            __hip_atomic_store(rep + 1, kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rep + 2, kmn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            drain_vmem();
            __hip_atomic_store(rep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // One 16-byte agent-scope load per poll returns {go, max key, min key} together (the keys were performed before `go`
    // was stored, and a 16-byte piece of a line is read in one request), so no second round trip for the keys.
    if (threadIdx.x == 0) {
        const uint32_t first_b = agent_add(a.tickets + 32, 1u);   // returns while this thread polls
        u32x4r line;
        do {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(line) : "v"(replica) : "memory");
            if (line[0] == 0u) __builtin_amdgcn_s_sleep(OQ_RES_SLEEP);
        } while (line[0] == 0u);
        s_keys[0] = line[1];
        s_keys[1] = line[2];
        s_first_b = first_b;
    }
    __syncthreads();
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
    }
    // Phase B: what nobody kept, most recently read first.  The bitmap of kept tiles is complete (every owner set its bit
    // before it counted its tiles) and is read once into LDS: a flag load per tile would put a memory round trip in front
    // of every tile's loads.
    for (uint32_t i = threadIdx.x; i < (ntiles + 31u) / 32u; i += kResWaves * kWave) s_held[i] = agent_load(a.held + i);
    __syncthreads();
    uint32_t tb = s_first_b;      // taken before the wait (below the publish): phase B's order does not matter, its latency does
    __syncthreads();
    while (tb < ntiles) {
        if (threadIdx.x == 0) s_ticket = agent_add(a.tickets + 32, 1u);
        const uint32_t tile = ntiles - 1u - tb;
        if (((s_held[tile >> 5] >> (tile & 31u)) & 1u) == 0u) {   // block-uniform
            const uint32_t row_tile = tile / ncol, col_tile = tile - row_tile * ncol;
            const int64_t row0 = static_cast<int64_t>(row_tile) * kResTileRows + wave * kResRows;
            load_tile(a, row0 < a.K ? row0 : a.K - 1, a.K, static_cast<int64_t>(col_tile) * kResCols, lane, v);
            quantize_store_tile(a, cq, v, row0, a.K, static_cast<int64_t>(col_tile) * kResCols, lane);
        }
        __syncthreads();
        tb = s_ticket;
        __syncthreads();
    }
}

// Tickets, counters and keys start from zero.  hipMemsetAsync's fill kernel took 4.6 us for these ~100 KB (rocprofv3,
// profiles/r04_strategies_*), a tenth of the whole call; this one is a 16-byte store per lane.
__global__ __launch_bounds__(256) void clear_words_kernel(uint4* p, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------ host side
static int resident_blocks(const void* kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kResWaves * kWave, 0) != hipSuccess) return 0;
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
    if (strategy == OQ_TENSOR) return static_cast<size_t>(kResTensorHeader + ncol_tiles * ceil_div(K, kResTileRows)) * 4 + 256;
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
                          float* scale, uint8_t* zp, int32_t layout, void* workspace, size_t workspace_bytes, hipStream_t s) {
    ResidentArgs a;
    const int tile_rows = groups_tile_rows(g);
    a.W = W; a.K = K; a.N = N; a.ldw = ldw; a.g = g; a.kgroups = K / g; a.chunks = ceil_div(g, strategy == OQ_TENSOR ? kResTileRows : tile_rows);
    a.q = q; a.scale = scale; a.zp = zp; a.grid = grid; a.layout = layout;
    a.ncol_tiles = static_cast<uint32_t>(ceil_div(N, kResCols));
    const size_t need = rtn_resident_workspace(K, N, strategy, g);
    OQ_REQUIRE(workspace && workspace_bytes >= need, OQ_ERR_WORKSPACE, "rtn: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    uint32_t* base = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    {
        const uint32_t n16 = static_cast<uint32_t>((need - 256 + 15) / 16);   // `need` counts whole words past the aligned base; the +256 slack covers the round-up
        hipLaunchKernelGGL(clear_words_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint4*>(base), n16);
    }
    a.tickets = base;                 // [0] phase A; phase B's at [32] (see the kernels: tickets + 32)
    a.counters = base + kResHeader;   // groups: one per range
    if (strategy == OQ_TENSOR) {
        a.ntiles = a.ncol_tiles * static_cast<uint32_t>(ceil_div(K, kResTileRows));
        a.counters = base + 64;                  // tiles counted
        a.key_max = base + 128;                  // 64 shards x 32 words: {max key, complemented min key}
        a.key_nmin = base + 128 + 64 * 32;       // 64 replicas x 32 words: {go, final max key, final complemented min key}
        a.held = base + kResTensorHeader;
        static const int resident = resident_blocks(reinterpret_cast<const void*>(rtn_tensor_onepass));
        OQ_REQUIRE(resident > 0, OQ_ERR_LAUNCH, "rtn: occupancy query failed");
        const uint32_t blocks = a.ntiles < static_cast<uint32_t>(resident) ? a.ntiles : static_cast<uint32_t>(resident);
        hipLaunchKernelGGL(rtn_tensor_onepass, dim3(blocks), dim3(kResWaves * kWave), 0, s, a);
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
