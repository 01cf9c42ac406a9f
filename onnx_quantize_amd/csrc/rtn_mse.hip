// M1: MSE range search (utils.py:140-239).  Placeholder until the kernel lands: fails loudly.
#include "oq_common.hpp"

namespace oq {

size_t rtn_mse_workspace(int64_t, int64_t, int32_t, int64_t) { return 0; }

int32_t rtn_mse_impl(const float*, int64_t, int64_t, int64_t, const QGrid&, int32_t, int64_t, void*, float*, void*,
                     int32_t, void*, size_t, hipStream_t, bool) {
    return fail(OQ_ERR_UNSUPPORTED, "mse=True is not implemented by this build of liboq_hip");
}

}  // namespace oq
