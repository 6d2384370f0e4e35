"""ctypes binding of libpolarahip.so (the C ABI declared in include/polara_hip.h).

Fails loudly: there is NO CPU fallback behind this module.  If the shared library is missing the
import error says how to build it; compute calls on a machine without a HIP device raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# POLARA_HIP_LIB: a kernel-tuning build of the same library (tools/probes: diagnostic instances of the sweep); still no CPU path
LIB_PATH = os.environ.get('POLARA_HIP_LIB') or os.path.join(_HERE, 'libpolarahip.so')

PK_VAL_F32, PK_VAL_F64 = 0, 1

_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); mirrors include/polara_hip.h one to one
PROTOTYPES = {
    'pk_last_error': (C.c_char_p, []),
    'pk_version': (C.c_int, []),
    'pk_warm_up': (C.c_int, []),
    'pk_device_info': (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64)]),
    'pk_spmm_csr_f64': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                  _vp, _i64, _i32, _vp, _i64, _vp]),
    'pk_spmm_csr_x': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                _vp, C.c_int, _i64, _i32, _vp, _i64, _vp]),
    'pk_spmm_csr_ex': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                 _vp, C.c_int, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _i64]),
    'pk_scan_work_bytes': (_i64, [_i64]),
    'pk_exclusive_scan_i32': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'pk_radix_work_bytes': (_i64, [_i64]),
    'pk_radix_sort_pairs': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, C.POINTER(_i32)]),
    'pk_coo_to_csr_work_bytes': (_i64, [_i64]),
    'pk_coo_to_csr': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pk_csr_transpose_work_bytes': (_i64, [_i64]),
    'pk_csr_transpose': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp]),
    'pk_csr_relabel_work_bytes': (_i64, [_i64]),
    'pk_csr_relabel_sorted': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    'pk_csr_rows_by_length_work_bytes': (_i64, [_i64]),
    'pk_csr_rows_by_length': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'pk_count_i32': (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    'pk_csr_scale_f64': (C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    'pk_row_plan_work_bytes': (_i64, [_i64]),
    'pk_row_plan_count': (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    'pk_row_plan_fill': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pk_gram_work_bytes': (_i64, [_i64, _i32, _i32]),
    'pk_gram_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    'pk_tsmm_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64]),
    'pk_tsmm_sub_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64]),
    'pk_tsmm_axpby_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _f64, _f64, _vp, _i64, _f64, _vp, _i64, _vp, _i64]),
    'pk_orth_check_f64': (C.c_int, [_vp, _i32, _vp, _i64, _vp, _i32, _vp]),
    'pk_eigh_psd_f64': (C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _f64, _vp]),
    'pk_eigh_psd_rounds_f64': (C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _f64, _vp]),
    'pk_eigh_top_supported': (C.c_int, [_i32, _i32]),
    'pk_eigh_top_work_bytes': (_i64, [_i32]),
    'pk_eigh_top_f64': (C.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp]),
    'pk_chol_work_bytes': (_i64, [_i32]),
    'pk_chol_rinv_f64': (C.c_int, [_vp, _i32, _vp, _i64, _f64, _vp, _i64, _vp, _vp]),
    'pk_chol_rinv_scaled_f64': (C.c_int, [_vp, _i32, _vp, _i64, _f64, _vp, _i64, _vp, _vp]),
    'pk_axpbypcz_f64': (C.c_int, [_vp, _i64, _f64, _vp, _f64, _vp, _f64, _vp, _vp]),
    'pk_resid_blocks': (_i32, [_i64]),
    'pk_resid_colnorm2_f64': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp]),
    'pk_dgemm_small_f64': (C.c_int, [_vp, C.c_int, C.c_int, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64]),
    'pk_scale_cols_f64': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    'pk_pack_elems': (_i64, [_i64, _i32]),
    'pk_pack_kq': (_i32, [_i32]),
    'pk_pack_frag_f32': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    'pk_candidate_capacity': (_i32, [_i32]),
    'pk_score_state_bytes': (_i64, [_i64, _i32]),
    'pk_score_splits': (_i32, [_i64, _i32]),
    'pk_seen_tiles_build': (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _vp, _vp]),
    'pk_seen_tiles_max_unsorted_row': (_i32, []),
    'pk_score_candidates_f32': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp,
                                          _i32, _vp, _vp, _vp, _vp, _i32]),
    'pk_score_candidates_rows_f32': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp, _i32, _i32,
                                               _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32]),
    'pk_score_two_phase_rows_f32': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp, _i32, _i32, _i32,
                                              _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32]),
    'pk_score_two_phase_plan': (C.c_int, [_i64, _i64, _i32, _vp, _vp]),
    'pk_score_two_phase_f32': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                                         _i32, _vp, _vp, _vp, _vp, _i32]),
    'pk_seen_dense_bytes': (_i64, [_i64, _i32]),
    'pk_seen_dense_build': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    'pk_score_chunk_launches': (_i32, [_i64, _i32, _i32, _i32, _i32]),
    'pk_pack_frag_bound_f32': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _f64]),
    'pk_row_norm_bound_f32': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    'pk_tile_norm_bound_f32': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    'pk_rescore_topk_f64': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _vp,
                                      _i32, _f64, _vp, _vp, _vp]),
    'pk_flag_compact': (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    'pk_fold_rows_f64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, C.c_int, _vp, _i64, _i32, _vp, _i64]),
    'pk_spmm_csr_flagged_f64': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _i64, _i32,
                                          _vp, _i64, _vp, _i64, _vp, _i32]),
    'pk_spmm_csr_rows_list_f64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                            _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i32]),
    'pk_q20_lanes': (_i32, [_i32]),
    'pk_q20_kappa': (_f64, [_i32]),
    'pk_q20_image_bytes': (_i64, [_i64, _i32]),
    'pk_q20_encode_f64': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    'pk_q20_decode_f64': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i64]),
    'pk_fold_q20': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _i64, _i32, _i32,
                              _vp, _i64, _vp]),
    'pk_rescore_topk_rows_f64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _i32,
                                           _vp, _vp, _i32, _f64, _vp, _vp, _vp]),
    'pk_rescore_topk_rows_list_f64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _i32,
                                           _vp, _vp, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _i32]),
    'pk_rescore_topk_rows_norms_f64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _i32,
                                           _vp, _vp, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    'pk_zero_i32': (C.c_int, [_vp, _vp, _i32]),
    'pk_scatter_rows_i64': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    'pk_map_ids_i64': (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp]),
    'pk_exact_work_bytes': (_i64, [_i32, _i64]),
    'pk_score_exact_rows_f64': (C.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32,
                                          _vp, _vp, _vp]),
    'pk_score_exact_list_f64': (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32,
                                          _vp, _vp, _vp]),
    'pk_eval_ranks': (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    'pk_eval_cols': (_i32, []),
    'pk_eval_user_metrics': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _i32, _vp]),
    'pk_eval_reduce_work_bytes': (_i64, [_i64]),
    'pk_eval_reduce': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'pk_unique_count_i64': (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'pk_dense_scores_f64': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64]),
    'pk_topk_rows_f64': (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp]),
    'pk_ctx_create': (C.c_int, [_i32, C.POINTER(_vp)]),
    'pk_ctx_destroy': (None, [_vp]),
    'pk_ctx_error': (C.c_char_p, [_vp]),
    'pk_mat_from_csr': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    'pk_mat_from_coo': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i32, C.POINTER(_vp)]),
    'pk_mat_free': (None, [_vp, _vp]),
    'pk_mat_nnz': (_i64, [_vp]),
    'pk_svd_build': (C.c_int, [_vp, _vp, _i32, _i32, _f64, _i32, C.c_uint64, _vp, _vp, _vp, _vp]),
    'pk_sym_eig_topk_f64': (C.c_int, [_vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _f64, _i32, C.c_uint64,
                                      _vp, _i64, _vp, _vp, _vp, _vp, _f64]),
    'pk_svd_build_sharded': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f64, _i32, C.c_uint64, _vp, _vp, _vp, _vp]),
    'pk_ctx_stream': (_vp, [_vp]),
    'pk_ctx_set_option': (C.c_int, [_vp, C.c_char_p, _i32]),
    'pk_set_option': (C.c_int, [C.c_char_p, _i32, _i32]),
    'pk_copy_to_host_async': (C.c_int, [_vp, _vp, _vp, _i64]),
    'pk_v32_image_f32': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    'pk_row_norm_order_work_bytes': (_i64, [_i64]),
    'pk_row_norm_order_f64': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    'pk_sweep_takes_rows': (C.c_int, []),
    'pk_ctx_spmm_timings': (_i64, [_vp, _vp, _vp, _i64]),
    'pk_mat_wrap_device': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    'pk_lanczos_steps': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32]),
    'pk_gramian_apply_f64': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64]),
    'pk_lanczos_products': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _i32]),
    'pk_lanczos_orth': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32]),
    'pk_score_topk': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'pk_serving_create': (C.c_int, [_vp, _i64, _i32, _vp, _vp, C.POINTER(_vp)]),
    'pk_serving_score': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    'pk_serving_free': (None, [_vp, _vp]),
    'pk_hooi': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pk_ttm_f64': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp]),
    'pk_tucker_predict_f64': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
}


class PolaraHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the library once and sets prototypes.  Raises PolaraHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must own the process' HIP runtime: it ships its own libamdhip64, and if ours
    # (/opt/rocm, via RUNPATH) were loaded first the two runtimes would not share the device context
    # ("no ROCm-capable device is detected" on the first launch).  Importing torch first makes the
    # dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PolaraHipError(
            'libpolarahip.so is not built (%s). Run `python -m polara_amd.build_native` '
            '(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what='', lib=None, ctx=None):
    """raises on a non-zero return code; `ctx`: the call belonged to a coarse-ABI context, whose own message says why"""
    if rc != 0:
        lib = lib or load()
        msg = lib.pk_ctx_error(ctx) if ctx is not None else None
        msg = msg or lib.pk_last_error()
        raise PolaraHipError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def device_info(device=0):
    lib = load()
    name = C.create_string_buffer(64)
    cu = C.c_int(0)
    mem = C.c_int64(0)
    n = lib.pk_device_info(device, name, 64, C.byref(cu), C.byref(mem))
    if n <= 0:
        check(n, 'pk_device_info')
    return dict(n_devices=n, arch=name.value.decode(), cu_count=cu.value, hbm_bytes=mem.value)
