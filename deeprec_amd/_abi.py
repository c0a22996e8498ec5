"""ctypes binding of libdeeprec_amd.so (the C ABI declared in include/deeprec_amd.h).

There is no CPU fallback: if the shared library is missing the import fails loudly, and
every wrapper in :mod:`deeprec_amd.kernels` raises on a non-zero status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEEPREC_AMD_LIB: another build of the SAME library (tools/ A/B runs: `make experiments`, ablation builds); a path that
# does not exist fails loudly like the default one
LIB_PATH = os.environ.get("DEEPREC_AMD_LIB") or os.path.join(_HERE, "libdeeprec_amd.so")


class OptParams(C.Structure):
    """dr_opt_params (include/deeprec_amd.h)."""
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("alpha", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("rho", C.c_float), ("momentum", C.c_float),
                ("reg", C.c_float), ("reg_mode", C.c_int32), ("alpha_dev", C.c_void_p),
                ("reg_dev", C.c_void_p)]


_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f = C.c_float
_optp = C.POINTER(OptParams)

# name -> (restype, argtypes); mirrors include/deeprec_amd.h one to one
SIGNATURES = {
    "dr_last_error": (C.c_char_p, []),
    "dr_version": (_i32, []),
    "dr_init": (_i32, [_i32]),
    "dr_gather_f32": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _i32, _p, _p]),
    "dr_scale_rows_f32": (_i32, [_p, _i64, _i32, _p, _p, _i64, _p, _p]),
    "dr_segment_plan_bytes": (_i64, [_i64, _i32]),
    "dr_segment_plan": (_i32, [_p, _i64, _i32, _i64, _p, _i64, _p]),
    "dr_segment_plan_view": (_i32, [_p, _i64, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "dr_scatter_add_sorted_f32": (_i32, [_p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _i64, _p]),
    "dr_sparse_row_update_f32": (_i32, [_p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _i64, _optp, _p]),
    "dr_dense_update_f32": (_i32, [_p, _p, _p, _p, _i64, _optp, _p]),
    "dr_dense_update_masked_f32": (_i32, [_p, _p, _p, _i64, _i32, _p, _optp, _p]),
    "dr_mark_rows_u8": (_i32, [_p, _i64, _p, _i64, _i32, _p]),
    "dr_rownorm_clip_f32": (_i32, [_p, _i64, _i32, _f, _p]),
    "dr_pairwise_step_bytes": (_i64, [_i64, _i32, _i64, _i64, _i32]),
    "dr_bpr_step_f32": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p, _i64, _optp, _p, _p,
                               _i64, _p]),
    "dr_bpr_plan_f32": (_i32, [_i64, _i64, _i32, _p, _p, _p, _i64, _i32, _p, _i64, _p]),
    "dr_bpr_step_planned_f32": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p, _i64, _optp, _p, _p,
                                       _i64, _p]),
    "dr_bpr_step_static_f32": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p, _i64, _optp, _p, _p]),
    "dr_bpr_forward_f32": (_i32, [_p, _p, _i64, _i64, _i32, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "dr_pointwise_forward_f32": (_i32, [_p, _p, _p, _p, _f, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _p,
                                        _p, _p, _p]),
    "dr_pointwise_step_f32": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _f,
                                     _i32, _p, _p, _p, _i64, _optp, _p, _p, _i64, _p]),
    "dr_hinge_step_f32": (_i32, [_p, _p, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _f, _f, _f, _i64,
                                 _optp, _p, _p, _i64, _p]),
    "dr_pair_dot_f32": (_i32, [_p, _i64, _p, _i64, _i32, _p, _p, _i64, _p, _p]),
    "dr_bi_interaction_fwd_f32": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p]),
    "dr_bi_interaction_bwd_f32": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _p]),
    "dr_score_topk_bytes": (_i64, [_i64, _i64, _i32, _i32, _i64, _i32]),
    "dr_score_topk_f32": (_i32, [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _f, _p, _p, _p, _p, _i64, _i32,
                                 _p, _p, _p, _p, _i64, _p]),
    "dr_score_rank_f32": (_i32, [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _f, _p, _p, _p, _p, _i64, _p, _p, _i64, _p]),
    "dr_score_rank_exact_bytes": (_i64, [_i64, _i64, _i32, _i64, _i32]),
    "dr_score_rank_exact_f32": (_i32, [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _f, _f, _p, _p, _p, _p, _i64, _p, _p,
                                       _i64, _p]),
    "dr_topk_rank_f32": (_i32, [_p, _i64, _i64, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "dr_score_matrix_f32": (_i32, [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _f, _i32, _p, _p, _i64, _p]),
    "dr_init_normal_f32": (_i32, [_p, _i64, C.c_uint64, _f, C.c_uint64, _p]),
    "dr_adam_tick_f32": (_i32, [_p, _f, _f, _f, _p]),
    "dr_adam_tick_many_f32": (_i32, [_p, _f, _f, _f, _p, _i32, _p]),
    "dr_sum_f32": (_i32, [_p, _i64, _p, _p]),
    "dr_sample_negatives_i32": (_i32, [_p, _p, _p, _i64, _i32, _i64, _i64, _i64, C.c_uint64, C.c_uint64, _p, _p]),
    "dr_sample_negatives_excl_i32": (_i32, [_p, _i64, _i32, _i32, _i64, _i64, C.c_uint64, C.c_uint64, _p, _p]),
    "dr_bucket_order_bytes": (_i64, [_i64]),
    "dr_bucket_order_i32": (_i32, [_p, _i64, _i32, _p, _p, _p, _i64, _p]),
    "dr_step_route_keys_i32": (_i32, [_p, _i32, _i32, _i64, _p, _i32, _p, _p, _p, _p]),
    "dr_rows_pack_f32": (_i32, [_p, _i64, _i32, _p, _i32, _i64, _p, _p, _i32, _p, _i64, _p]),
    "dr_rows_unpack_f32": (_i32, [_p, _i64, _i32, _p, _i32, _i64, _p, _p, _i32, _p, _i64, _p]),
    "dr_shard_bpr_bytes": (_i64, [_i32, _i64, _i32, _i64, _i64]),
    "dr_shard_bpr_list": (_i32, [_i32, _i64, _i32, _i64, _i64, _p, _i32, C.POINTER(_p)]),
    "dr_shard_bpr_route": (_i32, [_p, _p, _p, _p, _i32, _i32, _i64, _i64, _i64, _i32, _i64, _i64, _p, _i64, _p, _p]),
    "dr_shard_bpr_forward": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _i64, _i64, _i32, _i64, _i64, _p, _p, _optp,
                                    _p, _i64, _p]),
    "dr_shard_bpr_finish": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _i64, _i64, _i32, _i64, _i64, _p, _optp, _p,
                                   _p, _i64, _p]),
    "dr_mlp_forward_f32": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "dr_mlp_backward_f32": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "dr_tower_catalog_scores_f32": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p]),
    "dr_attn_pool_forward_f32": (_i32, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "dr_attn_pool_backward_f32": (_i32, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "dr_caser_vconv_forward_f32": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "dr_caser_vconv_backward_f32": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "dr_caser_hconv_direct_f32": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "dr_caser_hconv_pool_f32": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "dr_caser_hconv_backward_input_f32": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "dr_caser_hconv_backward_filter_f32": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "dr_seq_dist_loss_f32": (_i32, [_p, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _i32, _f, _f, _i32, _f, _p, _p,
                                    _p, _p, _p, _p]),
    "dr_seq_dot_loss_f32": (_i32, [_p, _p, _i64, _i32, _p, _p, _p, _i64, _i32, _i32, _f, _p, _p, _p, _p, _p]),
    "dr_profile_enable": (_i32, [_i32]),
    "dr_profile_collect": (_i32, [C.c_char_p, _i64, C.POINTER(C.c_float), _i32]),
    "dr_profile_collect_timeline": (_i32, [C.c_char_p, _i64, _p, _p, _i32]),
}


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            "deeprec_amd: %s is missing -- build it with `make -C deeprec_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback" % path)
    # torch first: its wheel bundles its own HIP runtime, and a process must end up with ONE runtime.  Loading
    # this library before torch binds it to /opt/rocm's copy and its launches then see no device.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


class DeepRecError(RuntimeError):
    pass


def check(status, what):
    if status is not None and status < 0:
        msg = lib().dr_last_error()
        raise DeepRecError("%s failed (status %d): %s" % (what, status, msg.decode() if msg else ""))
    return status
