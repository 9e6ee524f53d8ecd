"""ctypes binding of libcoldbrew_hip.so (the C ABI declared in include/coldbrew_hip.h).

There is no CPU fallback: if the shared library is missing or a tensor is not on an
MI355X device, the product path raises.  Build with `python __graft_entry__.py` or
`make -C gnn-tail-generalization_amd/csrc`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libcoldbrew_hip.so')

_c_i32p = ctypes.c_void_p
_c_f32p = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int32
_SZ = ctypes.c_size_t
_P = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/coldbrew_hip.h one to one
SIGNATURES = {
    'cb_version': (ctypes.c_int, []),
    'cb_last_error': (ctypes.c_char_p, []),
    'cb_device_status': (ctypes.c_int, []),
    'cb_agg_gemm_handover_selftest': (ctypes.c_int, [_P]),
    'cb_rows_zero_outside_mask_f32': (ctypes.c_int, [_P, _I64, _I64, _I64, _P, _P, _P]),
    'cb_csr_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_csr_from_coo_i64': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'cb_deg_norm_f32': (ctypes.c_int, [_P, _I64, _P, _P]),
    'cb_csr64_from_coo_i64': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'cb_csr_rebase_i64': (ctypes.c_int, [_P, _I64, _I64, _P, _P]),
    'cb_deg_norm_i64ptr_f32': (ctypes.c_int, [_P, _I64, _P, _P]),
    'cb_spmm_hub_count': (ctypes.c_int, [_P, _I64, _I32, _P, _P]),
    'cb_spmm_hub_fill_scratch_ints': (_I64, [_I64]),
    'cb_spmm_hub_fill': (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _P, _P, _P]),
    'cb_spmm_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_spmm_csr_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64,
                                       _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_colscale_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64,
                                                _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_dropout_f32': (ctypes.c_int, [_P, _P, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P]),
    'cb_axpby_f32': (ctypes.c_int, [ctypes.c_float, _P, ctypes.c_float, _P, _P, _I64, _P]),
    'cb_colsum_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_act_bwd_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _SZ, _P]),
    'cb_reduce_workspace_bytes': (_SZ, []),
    'cb_frobenius_norm_f32': (ctypes.c_int, [_P, _I64, _P, _P, _SZ, _P]),
    'cb_nll_logsoftmax_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _SZ, _P]),
    'cb_adam_step_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, _I64, _P, _P]),
    'cb_adam_multi_f32': (ctypes.c_int, [_I32, _P, _P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, _I64, _P, _P, _P]),
    'cb_gemm_nn_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_gemm_nn_splitk_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
    'cb_gemm_nn_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, ctypes.c_int, _P, _SZ, _P]),
    'cb_gemm_nn_drop2_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _SZ, _P]),
    'cb_gemm_tn_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
    'cb_gemm_tn_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _SZ, _P]),
    'cb_spmm_csr_fused_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P,
                                             _P, _SZ, _P]),
    'cb_spmm_csr_fused_rows_f32': (ctypes.c_int, [_P, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P,
                                                  _P, _SZ, _P]),
    'cb_trunk_layer_bwd_f32': (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, _P, ctypes.c_int, _I64, _I64, ctypes.c_float, ctypes.c_uint64,
                                              _P, _I64, ctypes.c_float, ctypes.c_float, _P, ctypes.c_uint64, ctypes.c_float, _P, _P, _P, _SZ, _P]),
    'cb_gemm_nn_bf16out_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, ctypes.c_int, _P, _SZ, _P]),
    'cb_spmm_csr_bf16_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64,
                                            _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_fused_bf16_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P,
                                                  _P, _P, _SZ, _P]),
    'cb_node_norm_fwd_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
    'cb_node_norm_bwd_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64, ctypes.c_float, ctypes.c_float, _P]),
    'cb_colstats_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_colstats_f32': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    'cb_col_affine_f32': (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_float, _P, _I64, _I64, _P]),
    'cb_col_bwd_combine_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _I64, _I64, _P]),
    'cb_topk_replace_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
    'cb_topk_replace_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _SZ, _P]),
    'cb_gather_rows_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _P, _P]),
    'cb_spmm_csr_acc_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64, _P, _I64,
                                           _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_fused_acc_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32,
                                                 _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_acc_bf16_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64, _P, _I64,
                                                _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_fused_acc_bf16_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float,
                                                      ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32,
                                                      _I32, _I32, _P, _P, _P, _SZ, _P]),
    'cb_gather_rows_bf16_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _P, _P]),
    'cb_adam_norm_workspace_bytes': (ctypes.c_size_t, [_I32]),
    'cb_adam_multi_norm_f32': (ctypes.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, _I64, _P, _P, _P, _SZ, _P]),
    'cb_expand_rows_f32': (ctypes.c_int, [_P, _P, _I64, _I64, ctypes.c_float, _P, _P]),
    'cb_gemm_nn_store_rows_supported': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64]),
    'cb_gemm_nn_store_rows_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _P, _P, _I64, _P, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, ctypes.c_int, _P, _I64, _P, _SZ, _P]),
    'cb_trunk_store_rows_f32': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _I64, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P,
                                               ctypes.c_int, _P, _P, _P]),
    'cb_agg_gemm_image_bytes': (_SZ, [_I64, _I64]),
    'cb_agg_gemm_image_f32': (ctypes.c_int, [_P, _I64, _I64, _I64, ctypes.c_int, _P, _SZ, _P]),
    'cb_spmm_gemm_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P,
                                        _P, _SZ, _P, _P, _P, _I64, _P, _I64, _P]),
    'cb_spmm_gemm_fused_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ,
                                              _P, _P, _P, _I64, _P, _I64, _P]),
    'cb_spmm_gemm_fused_eval_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ,
                                                   _P, _P, _P, _I64, _P, _I64, _P]),
    'cb_agg_gemm_head_image_bytes': (_SZ, [_I64, _I64]),
    'cb_agg_gemm_head_image_f32': (ctypes.c_int, [_P, _I64, _I64, _I64, ctypes.c_int, _P, _SZ, _P]),
    'cb_spmm_gemm_fused_head_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ,
                                                   _P, _P, _I64, _P, _I64, _P]),
    'cb_spmm_gemm_fused_head_eval_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ,
                                                   _P, _P, _I64, _P, _I64, _P]),
    'cb_spmm_gemm_trunkbwd_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ, _P, _P,
                                                 _P, _I64, _P, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _P, _I64, _P, _P, _SZ, _P]),
    'cb_spmm_csr_weighted_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, ctypes.c_int, _P, _I64, _P]),
    'cb_spmm_edge_dot_f32': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _I64, _P, _I64, _I64, _P, _P]),
    'cb_gemm_nn_indrop_supported': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64]),
    'cb_gemm_nn_indrop_drop2_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, ctypes.c_int, ctypes.c_float,
                                                   ctypes.c_uint64, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _P]),
    'cb_gemm_nn_indrop_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_uint64, _P, _I64, _P, _P]),
    'cb_gemm_tn_adrop_supported': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I64]),
    'cb_gemm_tn_adrop_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _SZ, _P]),
    'cb_front_image_bytes': (_SZ, [_I64]),
    'cb_front_image_f32': (ctypes.c_int, [_P, _I64, _I64, ctypes.c_int, _P, _SZ, _P]),
    'cb_trunk_front_f32': (ctypes.c_int, [_P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, ctypes.c_float,
                                          ctypes.c_uint64, ctypes.c_uint64, _P, _I64, _P]),
    'cb_gemm_tn_gdrop_supported': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I64]),
    'cb_gemm_tn_gdrop_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _SZ, _P]),
    'cb_spmm_gemm_trunkbwd_workspace_bytes': (_SZ, []),
    'cb_spmm_csr_lp_f32': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _I64, ctypes.c_float, _P, _P, _I64, _I32, _I32, _I32, _P, _P,
                                          _P, _SZ, _P]),
    'cb_trunk_input_bwd_multi_f32': (ctypes.c_int, [_P, ctypes.c_uint64, _I32, _P, _P, ctypes.c_float, _P, _P, _I64, _I64, ctypes.c_float,
                                                    _P, _I64, _P, _P, _SZ, _P, _P, _P]),
    'cb_trunk_input_bwd_multi_cs_f32': (ctypes.c_int, [_P, ctypes.c_uint64, _I32, _P, _P, ctypes.c_float, _P, _P, _I64, _I64, ctypes.c_float,
                                                       _P, _I64, _P, _P, _SZ, _P, _P, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    'cb_spmm_csr_store_bwd_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64,
                                                 _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ, _P, _P]),
    'cb_spmm_store_bwd_mix_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
    'cb_spmm_csr_store_bwd_mix_f32': (ctypes.c_int, [_P, _P, _I32, _I64, _I64, _P, _I64, _I64, _P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _P, _I64,
                                                     _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ, _I32, _P, _P, _P, ctypes.c_float, _P, _P, _SZ, _P]),
    'cb_gemm_tn_instage_supported': (ctypes.c_int, [_P, _P, _P, _I64, _I64, _I64]),
    'cb_gemm_tn_instage_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_gemm_tn_instage_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _I64, ctypes.c_float, ctypes.c_uint64, ctypes.c_float, ctypes.c_uint64, _P, _I64,
                                              _P, _SZ, _P]),
    'cb_trunk_layer_bwd_fold_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, ctypes.c_float, ctypes.c_float, _I32, _P, _P, _P,
                                                   _P, _P, _P, _SZ, _I32, _P, ctypes.c_float, _P, _P, _SZ, _P]),
    'cb_trunk_layer_bwd_rows_f32': (ctypes.c_int, [_P, _P, _I64, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, ctypes.c_float,
                                                   _P, ctypes.c_uint64, ctypes.c_float, _P, _P, _P, _SZ, _P]),
    'cb_id_count_i64': (ctypes.c_int, [_P, _I64, _I64, _P, _P, _P]),
    'cb_value_hist_i32': (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P]),
    'cb_compact_workspace_bytes': (_SZ, [_I64]),
    'cb_select_range_i32': (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _P, _P, _P, _SZ, _P]),
    'cb_craft_isolation_i64': (ctypes.c_int, [_P, _P, _I64, _P, _I64, _P, _P, _P, _P, _SZ, _P]),
    'cb_symmetrize_workspace_bytes': (_SZ, [_I64, _I64]),
    'cb_symmetrize_i64': (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _SZ, _P]),
    'cb_trunk_input_bwd_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64, ctypes.c_float, ctypes.c_uint64, _P, _I64, _P, _P, _SZ, _P]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Loads the shared library once; raises HipExtensionError (never falls back) if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipExtensionError(
            f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
            'Build it with `python __graft_entry__.py` or `make -C gnn-tail-generalization_amd/csrc`.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cb_last_error()
        raise HipExtensionError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')


_guards = {}      # device index -> int32 [1] device tensor (grad_guard)
_guard_listeners = None      # weak set of optimisers whose step counts follow the guard (register_guard_listener)


def register_guard_listener(opt):
    """An optimiser that reads grad_guard in its launch: device_status() calls its roll_back_skipped() when it reports a failed gradient check,
    so that the skipped launches do not count as steps (optim.Adam)."""
    global _guard_listeners
    if _guard_listeners is None:
        import weakref
        _guard_listeners = weakref.WeakSet()
    _guard_listeners.add(opt)


def _guard_failed():
    for opt in list(_guard_listeners or ()):
        opt.roll_back_skipped()
    for g in _guards.values():      # the error is in the caller's hands now: the next step may update again
        g.zero_()


def grad_guard(device, create=True):
    """The device word a failed gradient check sets (cb_rows_zero_outside_mask_f32, `guard`) and the fused Adam reads
    (cb_adam_multi_norm_f32, `guard`): while it is non-zero the optimiser launch writes nothing, so the truncated gradients of a
    row-sparse backward whose claim did not hold never reach the parameters or the moments — without a host synchronisation between
    backward() and step().  One word per device, allocated at the first check; cleared by device_status() when it reports the error (the skipped launches are then
    taken back off the optimiser's step counts too: register_guard_listener)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    g = _guards.get(idx)
    if g is None and create:
        g = _guards[idx] = torch.zeros(1, dtype=torch.int32, device=f'cuda:{idx}')
    return g


def device_status():
    """Raises HipExtensionError if a kernel recorded a device-side error since the last call (a tile hand-over of the aggregation + GEMM
    kernel that timed out: the results of that launch are invalid; a gradient row outside the loss rows that was not zero: the optimiser
    step of that backward was skipped on the device, see grad_guard).  Does not synchronise: the trainer calls it after the
    synchronisation that ends a step (where it reads the loss)."""
    lib = load()
    rc = lib.cb_device_status()
    if rc != 0:
        msg = lib.cb_last_error()
        _guard_failed()
        raise HipExtensionError(f'device-side error (rc={rc}): {msg.decode() if msg else ""}')
    for g in _guards.values():
        # node-sharded: the guard word is all-reduced with the gradients (dist.allreduce_grads), so a check that failed on ANOTHER rank
        # shows here — every rank raises, none trains on alone
        if int(g.item()) != 0:
            _guard_failed()
            raise HipExtensionError('device-side error: the guard word of the gradient-row check is set but this process holds no error report '
                                    '(the check failed on another rank, or the report was taken through cb_device_status() directly); '
                                    'the optimiser step of that backward was skipped')


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipExtensionError('Cold Brew HIP path needs tensors on an MI355X device (got a CPU tensor); '
                                    'there is no CPU fallback')
