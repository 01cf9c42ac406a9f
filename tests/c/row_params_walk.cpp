// Host-side guard for elementwise.hip::tile_kernel's parameter walk (row_params.hpp), built by
// tests/test_row_params_guard.py with g++ -fsanitize=address,undefined.  scale / zp live in heap blocks of EXACTLY the
// size the caller of oq_quantize_f32 / oq_dequantize_f32 has to provide, so a load one entry behind the last parameter row
// (the round-2 GPU page fault) is a heap-buffer-overflow here.  The tile loop below is the kernel's: rows in steps of 8,
// a thread = 4 neighbouring columns, parameters through TileParamCursor.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "row_params.hpp"

using oq::ParamIndex;

static constexpr int kTileRows = 32;   // elementwise.hip

static int walk(int64_t R, int64_t C, const ParamIndex& pi) {
    // entries addressed by any (r, c): the exact allocation
    int64_t entries = 0;
    for (int64_t r : {int64_t(0), R - 1})
        for (int64_t c : {int64_t(0), C - 1}) entries = entries > pi(r, c) + 1 ? entries : pi(r, c) + 1;
    float* scale = static_cast<float*>(malloc(entries * sizeof(float)));
    int32_t* zp = static_cast<int32_t*>(malloc(entries * sizeof(int32_t)));
    for (int64_t i = 0; i < entries; ++i) { scale[i] = static_cast<float>(i); zp[i] = static_cast<int32_t>(i); }
    int bad = 0;
    for (int64_t r0 = 0; r0 < R; r0 += kTileRows) {              // blockIdx.y
        const int64_t r1 = r0 + kTileRows < R ? r0 + kTileRows : R;
        for (int64_t c = 0; c < C; c += 4) {                      // one thread
            float s[4];
            int32_t z[4];
            auto load_params = [&](int64_t base) {
                for (int u = 0; u < 4; ++u) {
                    const int64_t p = base + (c + u) * pi.col_stride;
                    s[u] = scale[p];      // ASan: out of bounds -> abort
                    z[u] = zp[p];
                }
            };
            oq::TileParamCursor<decltype(load_params)> params(r0, r1, pi, load_params);
            params.start();
            for (int64_t r = r0; r < r1; r += 8)
                for (int u = 0; u < 8; ++u)
                    if (r + u < r1) {
                        for (int e = 0; e < 4; ++e) {            // the parameters in force must be those of (r + u, c + e)
                            const int64_t want = pi(r + u, c + e);
                            if (s[e] != static_cast<float>(want) || z[e] != static_cast<int32_t>(want)) ++bad;
                        }
                        params.row_done(r + u);
                    }
        }
    }
    free(scale);
    free(zp);
    return bad;
}

int main() {
    struct Case { int64_t R, C, row_div, row_stride, col_stride; };
    std::vector<Case> cases;
    for (int64_t R : {1, 7, 31, 32, 33, 64, 96, 100, 128, 257})
        for (int64_t C : {4, 8, 36}) {
            cases.push_back({R, C, 1, 0, 0});                     // tensor
            cases.push_back({R, C, 1, 1, 0});                     // row
            cases.push_back({R, C, 1, 0, 1});                     // column
            for (int64_t g : {1, 2, 4, 8, 16, 32, 64, 128})       // groups of g rows on [K, N]: entry n * (K / g) + kg
                if (R % g == 0) cases.push_back({R, C, g, 1, R / g});
        }
    int bad = 0;
    for (const Case& k : cases) {
        const int b = walk(k.R, k.C, ParamIndex{k.row_div, k.row_stride, k.col_stride});
        if (b) printf("MISMATCH R=%lld C=%lld div=%lld rs=%lld cs=%lld: %d\n", (long long)k.R, (long long)k.C, (long long)k.row_div,
                      (long long)k.row_stride, (long long)k.col_stride, b);
        bad += b;
    }
    printf("cases=%zu bad=%d\n", cases.size(), bad);
    return bad ? 1 : 0;
}
