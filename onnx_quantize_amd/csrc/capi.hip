// liboq_hip.so: status / error plumbing and the host-evaluated quantization grid (T1).
#include "oq_common.hpp"

#include <cmath>
#include <cstring>

namespace oq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t fail(int32_t status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

int32_t check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(OQ_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return OQ_OK;
}

// core/_dtypes.py:8-30 as three tables indexed by oq_qtype; :61-70 for the lookup order.
bool qrange_host(int32_t qtype, int32_t symmetric, int32_t reduce_range, int64_t* qmin, int64_t* qmax) {
    static const int64_t full[6][2] = {{-8, 7}, {0, 15}, {-128, 127}, {0, 255},
                                       {-(1LL << 31), (1LL << 31) - 1}, {0, (1LL << 32) - 1}};
    static const int64_t reduced[6][2] = {{-4, 3}, {0, 7}, {-64, 64}, {0, 127},
                                          {-(1LL << 30), (1LL << 30)}, {0, (1LL << 31) - 1}};
    if (qtype < 0 || qtype > 5) return false;
    if (reduce_range) {
        *qmin = reduced[qtype][0];
        *qmax = reduced[qtype][1];
    } else if (symmetric && (qtype == OQ_INT4 || qtype == OQ_INT8 || qtype == OQ_INT32)) {
        *qmax = full[qtype][1];
        *qmin = -full[qtype][1];
    } else {
        *qmin = full[qtype][0];
        *qmax = full[qtype][1];
    }
    return true;
}

int32_t make_grid(int32_t qtype, int32_t symmetric, int32_t reduce_range, float clip_ratio, QGrid* out) {
    int64_t lo, hi;
    OQ_REQUIRE(qrange_host(qtype, symmetric, reduce_range, &lo, &hi), OQ_ERR_INVALID_ARGUMENT,
               "unknown quantization type %d", qtype);
    OQ_REQUIRE(qtype <= OQ_UINT8, OQ_ERR_UNSUPPORTED,
               "32-bit quantization types are only supported by oq_quantize_bias_f32");
    out->qmin = static_cast<int32_t>(lo);
    out->qmax = static_cast<int32_t>(hi);
    out->symmetric = symmetric ? 1 : 0;
    // utils.py:277 zero = np.round((qmax + qmin) / 2.0): half-to-even (nearbyint in the default mode)
    const double zero = std::nearbyint((static_cast<double>(hi) + static_cast<double>(lo)) / 2.0);
    out->zero = static_cast<int32_t>(zero);
    // utils.py:282-284
    out->levels = std::fmin(static_cast<double>(hi) - zero, zero - static_cast<double>(lo));
    out->clip_ratio = clip_ratio;
    out->bits = (qtype == OQ_INT4 || qtype == OQ_UINT4) ? 4 : 8;
    return OQ_OK;
}

}  // namespace oq

extern "C" {

int32_t oq_abi_version(void) { return OQ_ABI_VERSION; }
const char* oq_last_error(void) { return oq::g_err; }
const char* oq_target_arch(void) { return "gfx950"; }

const char* oq_status_string(int32_t status) {
    switch (status) {
        case OQ_OK: return "ok";
        case OQ_ERR_INVALID_ARGUMENT: return "invalid argument";
        case OQ_ERR_UNSUPPORTED: return "unsupported";
        case OQ_ERR_WORKSPACE: return "workspace too small";
        case OQ_ERR_LAUNCH: return "HIP launch failure";
        case OQ_ERR_NOT_SPD: return "matrix not positive definite";
        default: return "unknown status";
    }
}

int32_t oq_qrange(int32_t qtype, int32_t symmetric, int32_t reduce_range, int64_t* qmin, int64_t* qmax) {
    OQ_REQUIRE(qmin && qmax, OQ_ERR_INVALID_ARGUMENT, "oq_qrange: null output");
    OQ_REQUIRE(oq::qrange_host(qtype, symmetric, reduce_range, qmin, qmax), OQ_ERR_INVALID_ARGUMENT,
               "oq_qrange: unknown quantization type %d", qtype);
    return OQ_OK;
}

}  // extern "C"
