/* C consumer of include/oq_hip.h (compiled by tests/test_library_abi.py with gcc -std=c99): the header is plain C, every
 * declared entry point links against liboq_hip.so, and the host-only calls behave.  No device is touched. */
#include <stdio.h>
#include <string.h>

#include "oq_hip.h"

#define TAKE(fn) do { if ((void*)(fn) == NULL) { printf("missing %s\n", #fn); return 2; } ++count; } while (0)

int main(void) {
    int count = 0;
    TAKE(oq_abi_version); TAKE(oq_last_error); TAKE(oq_status_string); TAKE(oq_target_arch); TAKE(oq_qrange);
    TAKE(oq_rtn_workspace_bytes); TAKE(oq_rtn_quantize_f32); TAKE(oq_rtn_batched_workspace_bytes);
    TAKE(oq_rtn_quantize_batched_f32);
    TAKE(oq_rtn_quantize_ptrs_f32); TAKE(oq_rtn_state_bytes); TAKE(oq_rtn_quantize_stateful_f32);
    TAKE(oq_hessian_many_workspace_bytes); TAKE(oq_hessian_accumulate_many_f32);
    TAKE(oq_hessian_pieces_bytes); TAKE(oq_hessian_slab_bytes); TAKE(oq_hessian_prepare_f32); TAKE(oq_hessian_accumulate_prepared_f32);
    TAKE(oq_matmul_pieces_bytes); TAKE(oq_matmul_prepare_f32); TAKE(oq_matmul_pieces_f32);
    TAKE(oq_awq_stats_workspace_bytes); TAKE(oq_awq_scale_search_stats_f32); TAKE(oq_awq_clip_search_stats_f32);
    TAKE(oq_abs_sum_cols_workspace_bytes); TAKE(oq_abs_sum_cols_f32); TAKE(oq_fingerprint64);
    TAKE(oq_awq_workspace_bytes); TAKE(oq_awq_scale_search_f32); TAKE(oq_awq_clip_search_f32);
    TAKE(oq_smooth_quant_workspace_bytes); TAKE(oq_smooth_quant_scale_f32); TAKE(oq_rtn_qparams_f32); TAKE(oq_qparams_f32); TAKE(oq_qparams_f64);
    TAKE(oq_minmax_rows_f32); TAKE(oq_quantize_f32); TAKE(oq_dequantize_f32); TAKE(oq_dequantize_fzp_f32); TAKE(oq_quantize_bias_f32);
    TAKE(oq_minmax_workspace_bytes); TAKE(oq_minmax_collect_f32); TAKE(oq_minmax_collect_f64);
    TAKE(oq_minmax_many_workspace_bytes); TAKE(oq_minmax_collect_many_f32);
    TAKE(oq_absmax_workspace_bytes); TAKE(oq_absmax_f32); TAKE(oq_hessian_workspace_bytes); TAKE(oq_hessian_accumulate_f32);
    TAKE(oq_rtn_tensor_many_workspace_bytes); TAKE(oq_rtn_tensor_many_f32); TAKE(oq_pack_matmul_nbits);
    TAKE(oq_gptq_prepare_workspace_bytes); TAKE(oq_gptq_prepare_f32); TAKE(oq_gptq_factor_workspace_bytes);
    TAKE(oq_gptq_factor_f32); TAKE(oq_gptq_factor_batched_workspace_bytes); TAKE(oq_gptq_factor_batched_f32); TAKE(oq_gptq_loop_workspace_bytes); TAKE(oq_gptq_loop_f32);
    TAKE(oq_hqq_workspace_bytes); TAKE(oq_hqq_optimize_f32); TAKE(oq_pack_zero_points_u4); TAKE(oq_pack_nibbles);

    int64_t lo = 0, hi = 0;
    if (oq_qrange(OQ_INT4, 1, 0, &lo, &hi) != OQ_OK || lo != -7 || hi != 7) return 3;              /* _dtypes.py:17-21 */
    if (oq_qrange(OQ_UINT8, 0, 1, &lo, &hi) != OQ_OK || lo != 0 || hi != 127) return 4;            /* _dtypes.py:23-30 */
    if (strcmp(oq_target_arch(), "gfx950") != 0 || oq_abi_version() != 2) return 5;
    if (oq_rtn_quantize_f32(NULL, 4, 4, 4, OQ_INT8, OQ_TENSOR, -1, 0, 0, 1.0f, 0, NULL, NULL, NULL, OQ_LAYOUT_KN, NULL, 0, NULL)
        != OQ_ERR_INVALID_ARGUMENT) return 6;
    if (strstr(oq_last_error(), "null pointer") == NULL) return 7;
    if (oq_rtn_workspace_bytes(4096, 11008, OQ_GROUP, 128, 0) == 0 || oq_hqq_workspace_bytes(4096, 11008, 128) == 0) return 8;
    if (oq_rtn_state_bytes(4096, 11008, OQ_CHANNEL, -1) == 0) return 13;
    if (sizeof(oq_minmax_desc) != 24) return 9;
    if (sizeof(oq_rtn_tensor_desc) != 40) return 10;
    printf("ok %d entry points\n", count);
    return 0;
}
