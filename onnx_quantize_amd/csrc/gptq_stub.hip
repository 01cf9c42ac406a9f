// TEMPORARY: G1-G3 entry points until hessian.hip / factor.hip / gptq_loop.hip land.  Fail loudly.
#include "oq_common.hpp"
extern "C" {
using namespace oq;
int32_t oq_hessian_accumulate_f32(const float*, int64_t, int64_t, int64_t, int64_t, int64_t, float*, void*) { return fail(OQ_ERR_UNSUPPORTED, "not built yet"); }
size_t oq_gptq_prepare_workspace_bytes(int64_t, int64_t, int32_t) { return 0; }
int32_t oq_gptq_prepare_f32(float*, int64_t, int64_t, float*, int32_t, int32_t*, void*, size_t, void*) { return fail(OQ_ERR_UNSUPPORTED, "not built yet"); }
size_t oq_gptq_factor_workspace_bytes(int64_t) { return 0; }
int32_t oq_gptq_factor_f32(float*, int64_t, float, float*, int32_t*, void*, size_t, void*) { return fail(OQ_ERR_UNSUPPORTED, "not built yet"); }
size_t oq_gptq_loop_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
int32_t oq_gptq_loop_f32(float*, int64_t, int64_t, const float*, int32_t, int32_t, int64_t, int32_t, int32_t, float, int32_t, int64_t, int32_t, void*, float*, float*, int32_t*, void*, size_t, void*) { return fail(OQ_ERR_UNSUPPORTED, "not built yet"); }
}
