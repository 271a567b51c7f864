"""ctypes description of the C ABI declared in include/i2p_ops.h.

One table, used by the product loader (`_lib.py`, device library, every entry takes a trailing
`void* stream`) and by the test-only oracle loader (`oracle/oracle.py`, same names + `_cpu`,
no stream).  Argument kinds:

    'i'  int   'l' long long   'f'  float   'd' double   'p'  device/host pointer (void*)
    'pp' pointer to an array of pointers (const float* const*)
"""
import ctypes as C

# name -> argument kinds (without the trailing stream)
SIGNATURES = {
    "i2p_fused_conv_select_k": ["i"] * 8 + ["f", "i", "i"] + ["p"] * 10 + ["i", "i"],
    "i2p_furthest_point_sampling": ["i", "i", "i", "p", "p", "p"],
    "i2p_gather_points": ["i", "i", "i", "i", "p", "p", "p"],
    "i2p_gather_points_grad": ["i", "i", "i", "i", "p", "p", "p"],
    "i2p_ball_query": ["i", "i", "i", "f", "i", "p", "p", "p"],
    "i2p_group_points": ["i", "i", "i", "i", "i", "p", "p", "p"],
    "i2p_group_points_grad": ["i", "i", "i", "i", "i", "p", "p", "p"],
    "i2p_three_nn": ["i", "i", "i", "p", "p", "p", "p"],
    "i2p_three_interpolate": ["i", "i", "i", "i", "p", "p", "p", "p"],
    "i2p_three_interpolate_grad": ["i", "i", "i", "i", "p", "p", "p", "p"],
    "i2p_project_seq": ["i", "i", "i", "i", "f", "f", "p", "i", "pp", "p", "p", "pp", "p"],
    "i2p_gather_rows": ["i", "i", "i", "i", "i", "p", "p", "p", "p"],
    "i2p_gather_rows_grad": ["i", "i", "i", "i", "i", "p", "p", "p", "p"],
    "i2p_knn": ["i", "i", "i", "i", "p", "p", "p"],
    "i2p_bn_stats": ["l", "i", "p", "p"],
    "i2p_bn_act_fwd": ["l", "i", "p", "p", "p", "p", "f", "f", "p", "p"],
    "i2p_bn_act_bwd_stats": ["l", "i", "p", "p", "p", "p", "p", "f", "p"],
    "i2p_bn_act_bwd": ["l", "i", "p", "p", "p", "p", "p", "f", "p", "p", "p", "p"],
    "i2p_lin_fwd": ["l", "i", "i", "p", "p", "f", "p", "p", "p"],
    "i2p_quat_mul": ["i", "i", "i", "i", "i", "p", "p", "p"],
    "i2p_quat_unit_fwd": ["i", "l", "p", "p"],
    "i2p_quat_unit_bwd": ["i", "l", "p", "p", "p"],
    "i2p_row_unitvar_fwd": ["i", "i", "p", "p", "p"],
    "i2p_row_unitvar_bwd": ["i", "i", "p", "p", "p", "p"],
    "i2p_img_bn_pool_fwd": ["i", "i", "i", "i", "i", "p", "p", "p", "p", "f", "f", "f", "p", "p", "p", "p", "p", "p"],
    "i2p_img_bn_pool_bwd": ["i", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "f", "p", "p", "p", "p"],
    "i2p_bn_finalize": ["l", "i", "p", "p", "p", "f", "p", "p"],
    "i2p_lin_bwd": ["l", "i", "i"] + ["p"] * 8 + ["f"] + ["p"] * 5 + ["f"],
    "i2p_pair_lin_fwd": ["i"] * 5 + ["p"] * 7,
    "i2p_pair_lin_bwd": ["i"] * 5 + ["p"] * 14,
    "i2p_lin_fwd_2src": ["l", "i", "i", "i", "p", "p", "f", "p", "p", "f", "p", "p", "p"],
    "i2p_lin_bwd_2src": ["l", "i", "i", "i"] + ["p"] * 8 + ["f"] + ["p"] * 3 + ["f"] + ["p"] * 8,
    "i2p_bn_act_maxk_fwd": ["l", "i", "i", "p", "p", "f", "p", "p"],
    "i2p_unpool_k": ["l", "i", "i", "p", "p", "p"],
    "i2p_pose_loss": ["i", "i"] + ["p"] * 10,
    "i2p_pair_bias_bn_bwd": ["i"] * 4 + ["p"] * 10,
    "i2p_cv_softmax_wsum_fwd": ["i"] * 4 + ["p", "p", "f", "p", "p", "f", "p", "p"],
    "i2p_cv_softmax_wsum_bwd": ["i"] * 4 + ["p"] * 6 + ["f", "p", "p", "f", "p", "p", "p"],
}

# entries that exist only in the device library (bf16 storage mode, deterministic scatter, fused grouping): the CPU
# oracle restates the reference's algorithms, not our storage formats — these are checked against fp32 results
DEVICE_ONLY = {
    "i2p_intrinsic_inverse": ["i", "p", "f", "f", "p"],
    "i2p_kitti_points_build": ["i", "i", "p", "p", "p", "p", "p"],
    "i2p_kitti_image_build": ["i", "i", "i", "p", "p"],
    "i2p_lin_fwd_bf16": ["l", "i", "i", "p", "i", "p", "f", "p", "p", "p"],
    "i2p_lin_fwd_2src_bf16": ["l", "i", "i", "i", "p", "p", "f", "p", "p", "f", "p", "p", "p"],
    "i2p_pair_lin_fwd_bf16": ["i"] * 5 + ["p"] * 7,
    "i2p_lin_bwd_bf16": ["l", "i", "i"] + ["p"] * 6 + ["i", "p", "p", "f", "p", "p", "i", "p", "p", "p", "f"],
    "i2p_lin_bwd_2src_bf16": ["l", "i", "i", "i"] + ["p"] * 8 + ["f"] + ["p"] * 3 + ["f"] + ["p"] * 8,
    "i2p_pair_lin_bwd_bf16": ["i"] * 5 + ["p"] * 14,
    "i2p_outer_sum_bf16": ["i"] * 4 + ["p"] * 4,
    "i2p_outer_sum": ["i"] * 4 + ["p"] * 4,
    "i2p_to_bf16": ["l", "p", "p"],
    "i2p_bn_act_fwd_bf16": ["l", "i", "p", "p", "f", "p"],
    "i2p_bn_act_maxk_fwd_bf16": ["l", "i", "i", "p", "p", "f", "p", "p"],
    "i2p_unpool_k_bf16": ["l", "i", "i", "p", "p", "p"],
    "i2p_bn_act_bwd_stats_bf16": ["l", "i", "p", "p", "p", "p", "f", "p"],
    "i2p_cv_softmax_wsum_fwd_bf16": ["i"] * 4 + ["p", "p", "f", "p", "p", "f", "p", "p"],
    "i2p_cv_softmax_wsum_bwd_bf16": ["i"] * 4 + ["p"] * 6 + ["f", "p", "p", "f", "p", "p", "p"],
    "i2p_pair_bias_bn_bwd_bf16": ["i"] * 4 + ["p"] * 10,
    "i2p_pair_bias_bn_finish": ["i"] * 4 + ["p"] * 9,
    "i2p_pair_bias_bn_bwd_det": ["i"] * 4 + ["p"] * 9,
    "i2p_gather_rows_grad_fx": ["i", "i", "i", "i", "i", "p", "p", "p", "p", "p"],
    "i2p_sa_l1_group": ["i"] * 10 + ["f", "p", "p", "p"],
    "i2p_gather_rows_grad_fx_ld": ["i", "i", "i", "i", "i", "p", "i", "i", "p", "p", "p", "p"],
    "i2p_sa_rows": ["i"] * 9 + ["p"] * 6,
    "i2p_knn_rows_fwd": ["i"] * 6 + ["p"] * 6,
    "i2p_knn_rows_bwd": ["i"] * 6 + ["p"] * 7,
    "i2p_gemm_tn": ["l", "i", "i", "p", "i", "p", "i", "p", "p"],
    "i2p_lin_fwd_fin": ["l", "i", "i", "p", "p", "f", "p", "p", "p", "p", "p", "f", "p", "p", "p"],
    "i2p_lin_fwd_2src_fin": ["l", "i", "i", "i", "p", "p", "f", "p", "p", "f", "p", "p", "p", "p", "p", "f", "p", "p", "p"],
    "i2p_pair_lin_fwd_fin": ["i"] * 5 + ["p"] * 7 + ["p", "p", "f", "p", "p", "p"],
    "i2p_chain_fwd": ["l", "i", "p", "p", "p", "pp", "pp", "pp", "p", "f", "pp", "pp", "pp", "p", "i", "p", "p", "p", "p"],
    "i2p_chain_bwd": ["l", "i", "p", "p", "p", "pp", "pp", "pp", "pp", "p", "p", "p", "i", "p", "p", "p", "pp", "pp", "p", "p"],
    "i2p_pose_compose_fwd": ["i", "p", "p", "p", "p", "p"],
    "i2p_pose_compose_bwd": ["i"] + ["p"] * 8,
    "i2p_clip_adam": ["l"] + ["p"] * 8 + ["d", "d"] + ["f"] * 4 + ["p", "p"],
    "i2p_defer_flush": [],
    "i2p_img_block_fwd": ["i"] * 7 + ["p", "p", "p", "p", "f", "f", "f"] + ["p"] * 6,
    "i2p_img_block_bwd": ["i"] * 7 + ["p"] * 6 + ["f"] + ["p"] * 4,
    "i2p_img_block_pool": ["i"] * 7 + ["p", "p", "p", "p", "f", "f", "f"] + ["p"] * 6,
    "i2p_img_conv_fwd": ["i"] * 6 + ["p", "p", "p", "p", "p"],
    "i2p_img_conv_bwd_data": ["i"] * 6 + ["p", "p", "p", "p"],
    "i2p_img_conv_wgrad": ["i"] * 6 + ["p", "p", "p", "p", "p"],
    "i2p_img_block_bwd_dx": ["i"] * 7 + ["p"] * 6 + ["f"] + ["p"] * 4,
    "i2p_img_block_bwd_stats": ["i"] * 7 + ["p"] * 6 + ["f", "p"],
    "i2p_img_conv_tail_bwd": ["i"] * 3 + ["p"] * 6 + ["f"] + ["p"] * 7,
    "i2p_img_first_fwd": ["i"] * 4 + ["p"] + ["l"] * 4 + ["p", "p", "p", "p", "f", "f", "f"] + ["p"] * 5 + ["i", "p", "p", "p", "i"],
    "i2p_img_first_bwd": ["i"] * 4 + ["p"] + ["l"] * 4 + ["p", "p", "p", "p", "f", "p", "p", "i"] + ["p"] * 6,
    "i2p_pc_rows_fwd": ["i"] * 6 + ["p"] * 8,
    "i2p_pc_rows_bwd": ["i"] * 6 + ["p"] * 9,
    "i2p_pose_head_fwd": ["i"] * 3 + ["p"] * 12,
    "i2p_pose_head_bwd": ["i"] * 3 + ["p"] * 16,
    "i2p_warp_split_fwd": ["i", "i"] + ["p"] * 7,
    "i2p_warp_split_bwd": ["i", "i"] + ["p"] * 9,
    "i2p_unpool_k_stats": ["l", "i", "i", "p", "p", "p", "p", "p", "p", "f", "p", "p"],
    "i2p_row_valid": ["l", "i", "p", "p"],
    "i2p_max_response_fwd": ["i"] * 4 + ["p"] * 7,
    "i2p_max_response_bwd": ["i"] * 4 + ["p"] * 7,
    "i2p_mask_fill": ["l", "i", "p", "p", "f", "p"],
    "i2p_pad_cols": ["i", "i", "i", "p", "p"],
    "i2p_strided_pick2": ["i"] * 7 + ["p"] * 4,
}
# plain `int f(...)` helpers without a stream argument
HELPERS = {
    "i2p_lin_bwd_bf16_grid": ["l"],
    "i2p_pair_lin_bwd_bf16_grid": ["i", "i", "i"],
    "i2p_pair_lin_bwd_scratch": ["i", "i", "i", "i", "i"],          # returns long long
    "i2p_pair_bias_bn_bwd_scratch": ["i", "i", "i", "i"],           # returns long long
    "i2p_gather_rows_grad_fx_scratch": ["i", "i", "i"],             # returns long long (bytes)
    "i2p_gemm_tn_scratch": ["l", "i", "i"],                         # returns long long (bytes)
    "i2p_chain_fwd_ok": ["l", "i", "p", "i"],
    "i2p_chain_bwd_ok": ["l", "i", "p", "i"],
    "i2p_chain_bwd_slab": ["i", "p", "p"],                          # returns long long (floats)
    "i2p_chain_sums_len": ["i", "i"],                               # returns long long (doubles)
    "i2p_chain_set_error_words": ["p", "p"],
    "i2p_img_first_bwd_rows": ["i", "i", "i", "i"],
    "i2p_img_conv_wgrad_rows": ["i", "i", "i"],
    "i2p_chain_resident_blocks": ["i", "l"],
    "i2p_chain_sync_words": [],                                     # returns long long (uint32 words)
    "i2p_defer_begin": [],
    "i2p_defer_pause": ["i"],
    "i2p_defer_compact_next": ["i", "i", "i"],
    "i2p_defer_pending": [],
    "i2p_defer_end": [],
    "i2p_ktime_enable": ["i"],
    "i2p_ktime_last_us": [],                                        # returns float (microseconds)
}
LONG_HELPERS = {"i2p_pair_lin_bwd_scratch", "i2p_pair_bias_bn_bwd_scratch", "i2p_gather_rows_grad_fx_scratch", "i2p_gemm_tn_scratch", "i2p_chain_sums_len", "i2p_chain_sync_words", "i2p_chain_bwd_slab"}

FLOAT_HELPERS = {"i2p_ktime_last_us"}

_CT = {"l": C.c_longlong, "i": C.c_int, "f": C.c_float, "d": C.c_double, "p": C.c_void_p, "pp": C.c_void_p}


def bind(lib, name, symbol, with_stream):
    """Attach argtypes/restype to `lib.symbol` for table entry `name` and return it."""
    fn = getattr(lib, symbol)
    kinds = SIGNATURES[name] if name in SIGNATURES else DEVICE_ONLY[name] if name in DEVICE_ONLY else HELPERS[name]
    fn.argtypes = [_CT[k] for k in kinds] + ([C.c_void_p] if with_stream else [])
    fn.restype = C.c_longlong if name in LONG_HELPERS else (C.c_float if name in FLOAT_HELPERS else C.c_int)
    return fn


def ptr_array(tensors):
    """Host-side array of data pointers for 'pp' arguments (kept alive by the caller)."""
    arr = (C.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
