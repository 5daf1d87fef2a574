"""ctypes binding of libmrec_hip.so (C-ABI declared in include/mrec.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  torch is used only as the owner of device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MREC_HIP_LIB selects an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("MREC_HIP_LIB") or os.path.join(_HERE, "csrc", "libmrec_hip.so")

MREC_OK = 0
_ERR_NAMES = {-1: "EINVAL", -2: "EWORKSPACE", -3: "EUNSUPPORTED", -4: "EHIP", -5: "ENODEVICE"}


class MrecError(RuntimeError):
    """A libmrec_hip.so entry point returned a non-zero code."""

    def __init__(self, fn, code, detail=""):
        self.code = code
        super().__init__(f"{fn} failed: MREC_{_ERR_NAMES.get(code, code)} {detail}".rstrip())


_lib = None

_vp, _i64, _i32, _u64, _f32, _int, _sz = (C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_float, C.c_int,
                                         C.c_size_t)
_szp = C.POINTER(C.c_size_t)

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGS = {
    "mrec_last_hip_error": [],
    "mrec_version": [],
    "mrec_device_ok": [],
    "mrec_fill_normal_f32": [_vp, _i64, _i32, _i64, _u64, _i64, _i64, _f32, _vp],
    "mrec_dedup_workspace_bytes": [_i64, _szp],
    "mrec_dedup_i32": [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_dedup_i64": [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_group_workspace_bytes": [_i64, _szp],
    "mrec_group_by_inverse": [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_sparse_plan_workspace_bytes": [_i64, _szp],
    "mrec_sparse_plan_i32": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_sparse_plan_i64": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_sparse_plan_ex_i32": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.c_uint32, _vp],
    "mrec_sparse_plan_ex_i64": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.c_uint32, _vp],
    "mrec_gather_rows_f32_i32": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_f32_i64": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_bf16_i32": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_bf16_i64": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_f16_i32": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_f16_i64": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_gather_rows_wide": [_vp, _i64, _i64, _i32, _vp, _i32, _i64, _vp, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _i32, _vp],
    "mrec_sparse_lazy_adam_wide": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i64, _vp,
                                   _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _sz, _vp, _vp, _vp],
    "mrec_gather_rows_wide_ex": [_vp, _i64, _i64, _i32, _vp, _i32, _i64, _i64, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _i32,
                                 C.c_uint32, _vp, _vp],
    "mrec_shard_route_slots_workspace_bytes": [_i64, _i32, _szp],
    "mrec_shard_route_slots_i32": [_vp, _vp, _i64, _i32, _i64, _int, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_slots_i64": [_vp, _vp, _i64, _i32, _i64, _int, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_slots_nv_i32": [_vp, _vp, _i64, _vp, _i32, _i64, _int, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_slots_nv_i64": [_vp, _vp, _i64, _vp, _i32, _i64, _int, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_unpack_req": [_vp, _i32, _i64, _vp, _vp, _vp],
    "mrec_shard_unroute_slots": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp],
    "mrec_shard_route_grads": [_vp, _i64, _vp, _i32, _vp, _i64, _i32, _vp, _i64, _vp],
    "mrec_step_state_init": [_vp, _f32, _f32, _i64, _vp],
    "mrec_step_advance": [_vp, _f32, _f32, _f32, _vp],
    "mrec_wall_clock_khz": [_vp],
    "mrec_head_fwd_bwd_wide": [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_wide_sum_f32_i32": [_vp, _i64, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "mrec_wide_sum_f32_i64": [_vp, _i64, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "mrec_sparse_apply_workspace_bytes": [_i64, _i32, _szp],
    "mrec_const_cols_detect": [_vp, _i32, _i64, _i32, _i64, _i64, _vp, _vp],
    "mrec_sparse_apply_next_const_cols": [_vp, _vp, _i32, _i64],
    "mrec_sparse_apply_window": [_i32, _int],
    "mrec_segment_sum_f32": [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _f32, _i32, _vp, _vp, _sz, _vp],
    "mrec_segment_sum_g16": [_vp, _vp, _vp, _i64, _vp, _i32, _i64, _vp, _f32, _i32, _vp, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_f32_i32": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                      _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_f32_i64": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                      _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_bf16g_i32": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                        _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_bf16g_i64": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                        _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_f16g_i32": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                       _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_lazy_adam_f16g_i64": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                       _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _sz, _vp],
    "mrec_sparse_ftrl_f32_i32": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                 _f32, _f32, _f32, _f32, _f32, _vp, _sz, _vp],
    "mrec_sparse_ftrl_f32_i64": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                 _f32, _f32, _f32, _f32, _f32, _vp, _sz, _vp],
    "mrec_dense_adam_f32": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp],
    "mrec_dense_adam_ex_f32": [_vp, _vp, _vp, _vp, _int, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp],
    "mrec_sparse_lazy_adam_wide_defer": [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i64, _vp,
                                         _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _sz,
                                         _vp, _vp, _vp, _vp],
    "mrec_dense_adam_slabs_finish_f32": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32,
                                         _f32, _f32, _int, _vp, _vp, _vp, _vp],
    "mrec_x3_parts_elems": [_i64, _i64, _vp],
    "mrec_x3_split": [_vp, _i64, _i64, _i32, _vp, _vp],
    "mrec_x3_gemm": [_int, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp],
    "mrec_x3_bias_relu": [_vp, _i64, _i64, _i32, _vp, _int, _vp, _vp],
    "mrec_x3_wgrad_slabs": [_i64, _i32, _i32, _vp],
    "mrec_x3_gemm_fwd": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _vp, _vp],
    "mrec_x3_gemm_dgrad_workspace_bytes": [_i64, _i32, _i32, _vp],
    "mrec_x3_gemm_dgrad": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _f32, _vp, _vp, _vp, _sz, _vp],
    "mrec_x3_mask_colsum": [_vp, _i64, _i64, _i32, _vp, _i64, _f32, _vp, _vp, _vp],
    "mrec_dense_adam_l2_workspace_bytes": [_i64, _vp],
    "mrec_dense_adam_l2_f32": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _f32, _vp, _int, _vp, _sz, _vp],
    "mrec_dense_adam_one_ftrl_f32": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp, _vp],
    "mrec_dense_adam_slabs_one_ftrl_f32": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32,
                                           _f32, _f32, _int, _vp, _vp, _vp],
    "mrec_dense_sum_slab_segments_f32": [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp],
    "mrec_dense_adam_slabs_f32": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32,
                                  _f32, _int, _vp, _vp],
    "mrec_dense_fwd_bf16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _int, _vp, _i64, _vp, _vp],
    "mrec_dense_fwd_f16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _int, _vp, _i64, _vp, _vp],
    "mrec_dense_fwd_wt_bf16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _int, _vp, _i64, _vp, _vp],
    "mrec_dense_fwd_wt_f16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _int, _vp, _i64, _vp, _vp],
    "mrec_dense_operand_copies": [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "mrec_dense_bwd_input_workspace_bytes": [_i64, _i32, _szp],
    "mrec_dense_bwd_bias_slabs": [_i64, _i32, _i32, _int, C.POINTER(C.c_int32)],
    "mrec_dense_bwd_input_bf16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp, _vp],
    "mrec_dense_bwd_input_f16": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp, _vp],
    "mrec_dense_bwd_weight_slabs": [_i64, _i32, _i32, C.POINTER(C.c_int32)],
    "mrec_dense_bwd_weight_bf16": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp],
    "mrec_dense_bwd_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _sz, _i32, _vp, _vp, _vp, _i32, _vp],
    "mrec_dense_bwd_f16": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _sz, _i32, _vp, _vp, _vp, _i32, _vp],
    "mrec_tail_supported": [_i64, _i32, _i32, _i32],
    "mrec_tail_workspace_bytes": [_i64, _szp],
    "mrec_tail_packed_elems": [_i32, _i32, _i32, C.POINTER(C.c_int64)],
    "mrec_tail_pack_weights": [_vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "mrec_tail_fwd_bwd": [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _f32,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp],
    "mrec_dropout": [_vp, _i64, _vp, _i64, _i32, _i64, _i32, _vp, _vp],
    "mrec_dropout_mask_f32": [_vp, _i64, _i64, _i32, _vp, _vp],
    "mrec_dense_sum_slabs_f32": [_vp, _i32, _i64, _vp, _vp],
    "mrec_dense_bwd_weight_f16": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp],
    "mrec_dense_ftrl_f32": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp],
    "mrec_head_workspace_bytes": [_i64, _i32, _szp],
    "mrec_head_fwd_bwd_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_head_fwd_bwd_f16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_head_fwd_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_map_bytes": [_i64, _szp],
    "mrec_map_create": [C.POINTER(_vp), _vp, _sz, _i64, _vp],
    "mrec_map_destroy": [_vp],
    "mrec_map_counters_dev": [_vp],
    "mrec_map_row_keys_dev": [_vp],
    "mrec_map_workspace_bytes": [_i64, _szp],
    "mrec_map_tracking_dev": [_vp, _vp, _vp, _vp],
    "mrec_map_lookup_workspace_bytes": [_i64, _szp],
    "mrec_map_lookup": [_vp, _vp, _i32, _i64, _vp, C.c_uint32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp],
    "mrec_map_lookup_out": [_vp, _vp, _i32, _i64, _vp, C.c_uint32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _sz, _vp],
    "mrec_gather_rows_f32_skip_i32": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp],
    "mrec_map_fill_missing": [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp],
    "mrec_map_evict": [_vp, _i64, _i64, _vp, _vp, _sz, _vp],
    "mrec_map_export_dirty": [_vp, _vp, _vp, _vp, _vp, _int, _vp, _sz, _vp],
    "mrec_map_mark_dirty": [_vp, _vp, _i64, _vp],
    "mrec_put_rows_last_f32": [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mrec_map_find_or_insert": [_vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _sz, _vp],
    "mrec_map_erase": [_vp, _vp, _i64, _vp, _sz, _vp],
    "mrec_map_export": [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_init_rows_f32": [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _u64, _f32, _f32, _vp],
    "mrec_copy3": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp],
    "mrec_copy_many": [_i32, _vp, _vp, _vp, _vp],
    "mrec_compose_i32": [_vp, _vp, _i64, _vp, _vp],
    "mrec_widen_i32_i64": [_vp, _i64, _vp, _vp],
    "mrec_scatter_rows_f32": [_vp, _i64, _i32, _vp, _i64, _vp, _vp],
    "mrec_move_rows_f32": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _vp],
    "mrec_cross_layers_f32": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp],
    "mrec_cross_layers_bwd_workspace_bytes": [_i32, _i64, _i32, _szp],
    "mrec_cross_layers_bwd_f32": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_cross_layers_bwd_acc_f32": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_dcn_head_workspace_bytes": [_i64, _i32, _i32, _szp],
    "mrec_dcn_head_fwd_bwd": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp,
                              _vp, _sz, _vp],
    "mrec_dense32_fwd": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _int, _vp, _i64, _vp],
    "mrec_dense32_bwd_input": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp],
    "mrec_dense32_bwd_weight": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp],
    "mrec_dense32_bwd_weight_slabs": [_i64, _i32, _i32, C.POINTER(C.c_int32)],
    "mrec_fm_fwd_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "mrec_fm_bwd_f32": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "mrec_fm_fwd_add_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "mrec_fm_fwd_add16_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "mrec_fm_bwd_mix_f32": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp],
    "mrec_scatter_add_rows_f32": [_vp, _i64, _i32, _vp, _i64, _vp, _vp],
    "mrec_shard_route_workspace_bytes": [_i64, _i32, _szp],
    "mrec_shard_route_i32": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_i64": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_hash_i32": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_route_hash_i64": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp],
    "mrec_shard_unroute_f32": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "mrec_shard_route_rows_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp],
    "mrec_shard_unroute_ld_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp],
    "mrec_shard_route_rows_ld_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp],
    "mrec_shard_pack_iw_i32": [_vp, _vp, _vp, _i64, _vp, _vp],
    "mrec_shard_unpack_iw_i32": [_vp, _i64, _vp, _vp, _vp],
    "mrec_event_create": [C.POINTER(_vp)],
    "mrec_event_destroy": [_vp],
    "mrec_event_elapsed_ms": [_vp, _vp, C.POINTER(C.c_float)],
    "mrec_profile_next_apply": [_vp, _vp],
    "mrec_dense_adam_rows_l2_workspace_bytes": [_i64, _i32, _vp],
    "mrec_dense_adam_rows_l2_f32": [_vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _int, _f32, _vp, _int,
                                    _vp, _vp, _sz, _vp],
    "mrec_crc32c_host": [C.c_char_p, _sz, _vp],
}
_RESTYPES = {"mrec_strerror": C.c_char_p, "mrec_map_counters_dev": _vp, "mrec_map_row_keys_dev": _vp}

EXPORTED = sorted(list(_SIGS) + ["mrec_strerror"])


def lib():
    """Loads libmrec_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(mindrec_amd has no CPU fallback)")
        # torch first: it ships its own copy of the HIP runtime, and this library must bind to THAT copy (the one whose streams
        # and allocations it is handed) -- loaded before torch it binds to /opt/rocm's, and the process ends up with two
        # runtimes (seen as "no usable gfx950 device" from mrec_device_ok)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            f = getattr(l, name)
            f.argtypes = args
            f.restype = _RESTYPES.get(name, C.c_int)
        l.mrec_strerror.argtypes = [C.c_int]
        l.mrec_strerror.restype = C.c_char_p
        _lib = l
    return _lib


def check(fn, code):
    if code != MREC_OK:
        detail = ""
        if code == -4:
            detail = f"(hipError {lib().mrec_last_hip_error()})"
        raise MrecError(fn, code, detail)


def call(name, *args):
    check(name, getattr(lib(), name)(*args))


def query_bytes(name, *args):
    out = C.c_size_t(0)
    check(name, getattr(lib(), name)(*args, C.byref(out)))
    return int(out.value)
