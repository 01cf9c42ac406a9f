// Where the (scale, zero point) of a row live, and the walk of one tile's rows over them -- shared by
// elementwise.hip::tile_kernel and by the host-compiled guard test (tests/c/row_params_walk.cpp, built with
// g++ -fsanitize=address), so the index arithmetic that once read one entry past the end of scale / zp
// (round 2, gpurun_out/full1.log) is the SAME code in the kernel and under the sanitizer.  No HIP dependency.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define OQ_HD __host__ __device__ __forceinline__
#else
#define OQ_HD inline
#endif

namespace oq {

// Parameter entry of element (r, c): (r / row_div) * row_stride + c * col_stride  (oq_quantize_f32 / oq_dequantize_f32).
struct ParamIndex {
    int64_t row_div, row_stride, col_stride;
    OQ_HD int64_t operator()(int64_t r, int64_t c) const { return (r / row_div) * row_stride + c * col_stride; }
};

struct RowParams {   // where the parameters of row r live; next() steps to r + 1 without dividing
    int64_t base, left, row_stride, row_div;
    OQ_HD RowParams(int64_t r, const ParamIndex& pi)
        : base((r / pi.row_div) * pi.row_stride), left(pi.row_div - r % pi.row_div), row_stride(pi.row_stride), row_div(pi.row_div) {}
    OQ_HD bool next() {   // true when the parameter row changed
        if (--left > 0) return false;
        left = row_div;
        base += row_stride;
        return row_stride != 0;
    }
};

// The parameter loads of one tile: rows [r0, r_end).  `load(base)` fetches the thread's parameters of the parameter row
// that starts at entry `base`.  It is called once before the first row and then only when the parameter row changes AND
// another row of the tile follows: behind the tile's last row the next parameter row may not exist (for the matrix' last
// rows it lies one entry past the end of scale / zp: a page fault when the array ends its mapping).
template <class Load>
struct TileParamCursor {
    RowParams rp;
    int64_t r_end;
    Load load;
    OQ_HD TileParamCursor(int64_t r0, int64_t r_end_, const ParamIndex& pi, Load load_) : rp(r0, pi), r_end(r_end_), load(load_) {}
    OQ_HD void start() { load(rp.base); }
    OQ_HD void row_done(int64_t r) {
        if (rp.next() && r + 1 < r_end) load(rp.base);
    }
};

}  // namespace oq
