"""ctypes binding of libktup_hip.so (declared in include/ktup_hip.h).

The library is the product: there is NO CPU or eager-torch fallback.  If it is missing or a call fails
the error is raised, never swallowed.  torch is imported first on purpose: the library's NEEDED
libamdhip64.so.7 then resolves to the HIP runtime torch already loaded, so torch's streams and
device pointers are valid inside the library.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KTUP_HIP_LIB', os.path.normpath(os.path.join(_HERE, '..', '..', 'libktup_hip.so')))

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_u = ctypes.c_uint64
c_f = ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    'ktup_version': [],
    'ktup_last_error': [],
    'ktup_set_option': [ctypes.c_char_p, c_i],
    'ktup_get_option': [ctypes.c_char_p, c_p],
    'ktup_score_bprmf_fwd': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p],
    'ktup_score_bprmf_bwd': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_p],
    'ktup_score_transe_fwd': [c_p, c_l, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p],
    'ktup_score_transe_bwd': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    'ktup_score_transh_fwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p],
    'ktup_score_transh_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    'ktup_score_transr_workspace_bytes': [c_l, c_l],
    'ktup_score_transr_fwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p],
    'ktup_score_transr_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    'ktup_score_transr_bwd_workspace_bytes': [c_l, c_i, c_l, c_l],
    'ktup_score_transr_bwd_ws': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    'ktup_score_kg_bwd_workspace_bytes': [c_l, c_i, c_l],
    'ktup_score_transe_bwd_ws': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    'ktup_score_transh_bwd_ws': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    'ktup_score_bprmf_bwd_workspace_bytes': [c_l, c_i, c_l, c_l],
    'ktup_score_bprmf_bwd_ws': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    'ktup_pref_workspace_bytes': [c_i, c_i],
    'ktup_pref_prepare': [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p],
    'ktup_score_tup_fwd': [c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_p],
    'ktup_score_ktup_fwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_p],
    'ktup_score_tup_bwd': [c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_score_ktup_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u,
                            c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_score_pref_bwd_workspace_bytes': [c_l, c_i, c_l, c_l],
    'ktup_score_tup_bwd_ws': [c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_p, c_p, c_p, c_p, c_l, c_l,
                              c_p, c_p],
    'ktup_score_ktup_bwd_ws': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u,
                               c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    'ktup_segment_workspace_bytes': [c_l, c_l],
    'ktup_segment_reduce_rows': [c_p, c_l, c_i, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p],
    'ktup_loss_bpr_fwd': [c_p, c_p, c_l, c_f, c_p, c_p],
    'ktup_loss_bpr_bwd': [c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p],
    'ktup_loss_margin_fwd': [c_p, c_p, c_l, c_f, c_p, c_p],
    'ktup_loss_margin_bwd': [c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p],
    'ktup_loss_bpr_fused': [c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p, c_p],
    'ktup_loss_margin_fused': [c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p, c_p],
    'ktup_reg_norm_fused': [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_p],
    'ktup_reg_orth_fused': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p],
    'ktup_reg_norm_fwd': [c_p, c_l, c_i, c_p, c_l, c_p, c_p],
    'ktup_reg_norm_bwd': [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p],
    'ktup_reg_orth_fwd': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_p],
    'ktup_reg_orth_bwd': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_p],
    'ktup_eval_bprmf_scores': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_l, c_p, c_l, c_p],
    'ktup_eval_kg_workspace_bytes': [c_i, c_l],
    'ktup_eval_transe_scores': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_l, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_p],
    'ktup_eval_transh_scores': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_l, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_p],
    'ktup_eval_transr_workspace_bytes': [c_i, c_l, c_l, c_i],
    'ktup_eval_transr_scores': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_l, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_p, c_p],
    'ktup_eval_transr_entities_workspace_bytes': [c_i, c_l, c_i],
    'ktup_eval_transr_prepare': [c_p, c_l, c_p, c_l, c_i, c_l, c_i, c_i, c_p, c_p],
    'ktup_eval_pref_workspace_bytes': [c_i, c_i, c_l, c_l],
    'ktup_eval_pref_scores': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_p, c_l, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_l,
                              c_p, c_p],
    'ktup_eval_pref_items_workspace_bytes': [c_i, c_i, c_l],
    'ktup_eval_pref_items_prepare': [c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_l, c_p, c_p],
    'ktup_eval_pref_scores_prepared': [c_p, c_l, c_p, c_i, c_i, c_p, c_l, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_l, c_p, c_p, c_p],
    'ktup_eval_pref_topk_workspace_bytes': [c_i, c_i, c_l, c_l, c_i],
    'ktup_eval_pref_topk': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_p, c_l, c_l, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    'ktup_eval_pref_topk_hard_workspace_bytes': [c_i, c_i, c_l, c_l, c_i],
    'ktup_eval_pref_topk_hard': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_p, c_l, c_l, c_i, c_i, c_p, c_u, c_u, c_p, c_p, c_i, c_p,
                                 c_p, c_p, c_p],
    'ktup_eval_topk_filtered': [c_p, c_l, c_l, c_l, c_i, c_p, c_p, c_i, c_p, c_p, c_p],
    'ktup_eval_gold_rank_counts': [c_p, c_l, c_l, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_eval_gold_rank_counts_strided': [c_p, c_l, c_l, c_l, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_eval_gold_ranks': [c_p, c_l, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_eval_kg_ranks_workspace_bytes': [c_i, c_l, c_l],
    'ktup_eval_kg_ranks': [c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_l, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_l,
                           c_p, c_p],
    'ktup_eval_kg_ranks_fused_supported': [c_i, c_i, c_i, c_l],
    'ktup_eval_kg_ranks_fused_workspace_bytes': [c_i, c_i, c_l, c_l, c_l, c_l, c_l],
    'ktup_eval_kg_ranks_fused': [c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_p, c_l, c_l, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_p,
                                 c_l, c_l, c_p, c_p, c_p],
    'ktup_eval_kg_ranks_transr_workspace_bytes': [c_i, c_l, c_i, c_l],
    'ktup_eval_kg_ranks_transr': [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p,
                                  c_l, c_p, c_p],
    'ktup_shard_dedupe_workspace_bytes': [c_l],
    'ktup_shard_dedupe': [c_p, c_l, c_p, c_p, c_p, c_p, c_p],
    'ktup_shard_pack_rows': [c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p],
    'ktup_shard_unpack_rows_add': [c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p],
    'ktup_shard_route_workspace_bytes': [c_l],
    'ktup_shard_route_sort_bytes': [c_l, c_l],
    'ktup_shard_route': [c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    'ktup_shard_route_ktup': [c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p],
    'ktup_shard_reduce_rows': [c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p],
    'ktup_shard_ktup_entries': [c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p],
    'ktup_shard_pack_wire': [c_i, c_p, c_p, c_p, c_i, c_p, c_l, c_p, c_l, c_p],
    'ktup_shard_apply': [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f,
                         c_p, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    'ktup_shard_route_kg': [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p],
    'ktup_shard_kg_rel_order': [c_p, c_l, c_l, c_p, c_p],
    'ktup_train_kg_step_rows': [c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_l, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    'ktup_shard_bucket': [c_i, c_i, c_p, c_l, c_p, c_p, c_i, c_p, c_p, ctypes.c_double, c_p],
    'ktup_zero_async': [c_p, c_l, c_p],
    'ktup_shard_reduce_list_len': [c_l, c_i],
    'ktup_shard_zero_shared_rows': [c_p, c_l, c_l, c_p, c_p, c_l, c_i, c_p],
    'ktup_shard_reduce_store': [c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p],
    'ktup_shard_reduce_store_fold': [c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_i, c_l, c_p, c_p],
    'ktup_shard_step_count': [c_p, c_p, c_p, c_f, c_f, c_p],
    'ktup_shard_adam_flush': [c_p, c_l, c_p, c_l, c_i, c_l, c_f, c_f, c_p, c_p],
    'ktup_shard_adam_catchup': [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p],
    'ktup_shard_reduce_norm': [c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_i, c_p, c_l, c_f, c_p, c_i, c_i, c_p, c_i, c_p, c_p],
    'ktup_shard_reduce_norm_fold': [c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_i, c_p, c_l, c_f, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_p],
    'ktup_shard_reduce_apply': [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_i,
                                c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    'ktup_optim_gradnorm_acc': [c_i, c_p, c_p, c_p, c_i, c_p],
    'ktup_train_rec_step_rows': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_l, c_i,
                                 c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_p, c_p],
    'ktup_train_rec_step_rows_ws': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_l, c_i,
                                    c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_p, c_p, ctypes.c_size_t, c_p],
    'ktup_train_rec_step_rows_ws_bytes': [c_l, c_i, c_i],
    'ktup_train_rec_reg_rows': [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_p, c_f, c_f, c_p, c_p],
    'ktup_negsample_rec_workspace_bytes': [c_l],
    'ktup_negsample_rec': [c_p, c_p, c_l, c_l, c_p, c_l, c_u, c_u, c_i, c_p, c_p, c_p, c_p],
    'ktup_optim_gradnorm': [c_i, c_p, c_p, c_p, c_p],
    'ktup_optim_clip_step_capacity': [c_i],
    'ktup_optim_clip_step': [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_f, c_i, c_p, c_i, c_f,
                             c_p, c_p, c_p],
    'ktup_train_step_supported': [c_i, c_i, c_i],
    'ktup_train_rec_step': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_u, c_u,
                            c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_train_kg_step': [c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_l, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    'ktup_optim_step': [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_i, c_p],
    'ktup_eval_rec_metrics': [c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    'ktup_shard_sparse_step': [c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_f, c_f, c_p, c_f, c_p],
    'ktup_negsample_kg': [c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_l, c_u, c_u, c_p, c_p, c_p, c_p],
    'ktup_feed_rec': [c_p, c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_l, c_u, c_i, c_p, c_p, c_p, c_p, c_p],
    'ktup_feed_kg': [c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_l, c_l, c_p, c_l, c_u, c_p, c_p, c_p, c_p, c_p],
}
_RESTYPE = {'ktup_last_error': ctypes.c_char_p, 'ktup_shard_reduce_list_len': ctypes.c_int64, 'ktup_shard_route_workspace_bytes': ctypes.c_size_t,
            'ktup_shard_route_sort_bytes': ctypes.c_size_t, 'ktup_pref_workspace_bytes': ctypes.c_size_t, 'ktup_eval_kg_ranks_workspace_bytes': ctypes.c_size_t, 'ktup_eval_kg_ranks_transr_workspace_bytes': ctypes.c_size_t, 'ktup_eval_kg_ranks_fused_workspace_bytes': ctypes.c_size_t,
            'ktup_eval_kg_workspace_bytes': ctypes.c_size_t, 'ktup_eval_transr_workspace_bytes': ctypes.c_size_t, 'ktup_eval_transr_entities_workspace_bytes': ctypes.c_size_t, 'ktup_score_transr_workspace_bytes': ctypes.c_size_t, 'ktup_score_transr_bwd_workspace_bytes': ctypes.c_size_t,
            'ktup_eval_pref_workspace_bytes': ctypes.c_size_t, 'ktup_eval_pref_items_workspace_bytes': ctypes.c_size_t, 'ktup_negsample_rec_workspace_bytes': ctypes.c_size_t,
            'ktup_score_pref_bwd_workspace_bytes': ctypes.c_size_t, 'ktup_segment_workspace_bytes': ctypes.c_size_t, 'ktup_shard_dedupe_workspace_bytes': ctypes.c_size_t,
            'ktup_eval_pref_topk_workspace_bytes': ctypes.c_size_t, 'ktup_eval_pref_topk_hard_workspace_bytes': ctypes.c_size_t,
            'ktup_score_kg_bwd_workspace_bytes': ctypes.c_size_t, 'ktup_score_bprmf_bwd_workspace_bytes': ctypes.c_size_t,
            'ktup_train_rec_step_rows_ws_bytes': ctypes.c_size_t}

_lib = None


class KtupError(RuntimeError):
    code = None                    # the library's status (KTUP_ERR_*) when the error came from an entry point


ERR_UNSUPPORTED = -3               # KTUP_ERR_UNSUPPORTED (include/ktup_hip.h): the entry point does not cover this shape


def load():
    """Load the shared library once; raise (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KtupError('libktup_hip.so not found at %s -- build it with '
                        '`python joint-kg-recommender_amd/build_hip.py` (there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point; a non-zero status raises with the library's message."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        err = KtupError('%s failed (%d): %s' % (name, rc, lib.ktup_last_error().decode('utf-8', 'replace')))
        err.code = rc
        raise err


def get_option(name):
    lib = load()
    val = ctypes.c_int(0)
    if lib.ktup_get_option(name.encode(), ctypes.byref(val)) != 0:
        raise KtupError(lib.ktup_last_error().decode('utf-8', 'replace'))
    return val.value


def set_option(name, value):
    """Process-wide kernel-selection knob (include/ktup_hip.h ktup_set_option); returns the previous value."""
    lib = load()
    old = ctypes.c_int(0)
    if lib.ktup_get_option(name.encode(), ctypes.byref(old)) != 0 or lib.ktup_set_option(name.encode(), int(value)) != 0:
        raise KtupError(lib.ktup_last_error().decode('utf-8', 'replace'))
    return old.value


def bind(name, *args):
    """Pre-convert the arguments of an int-returning entry point once and return a zero-argument launcher: for loops that
    issue the SAME launch many times (fixed buffers), where Python-side argument marshalling would otherwise starve the GPU."""
    lib = load()
    fn = getattr(lib, name)
    conv = [a if a is None else t(a) for t, a in zip(fn.argtypes, args)]
    if len(conv) != len(args):
        raise KtupError('%s takes %d arguments, got %d' % (name, len(fn.argtypes), len(args)))

    def run():
        rc = fn(*conv)
        if rc != 0:
            raise KtupError('%s failed (%d): %s' % (name, rc, lib.ktup_last_error().decode('utf-8', 'replace')))
    return run


import contextlib as _contextlib
import gc as _gc


@_contextlib.contextmanager
def capture(graph, **kw):
    """`with torch.cuda.graph(graph)` with Python's cyclic garbage collector held off.  A collection that happens to run INSIDE a
    capture (binding a step's launches allocates thousands of ctypes objects) may finalise garbage of earlier steppers -- an older
    CUDAGraph, tensors whose blocks go back to the driver -- and the HIP runtime refuses those calls while a stream is capturing:
    the process aborts (seen once in three runs of the GPU suite, in whichever test captured when the collector's turn came).
    Holding the collector off is all that needs: a full `gc.collect()` in front of every capture (rounds 4-5) walked the whole heap --
    120 ms with an ml1m-size dataset's dicts on it, charged to the evaluation pass that captures (tools/cli_throughput.py: 20.4 k ->
    11.5 k steps/s over the interval that holds it)."""
    was = _gc.isenabled()
    _gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            _gc.enable()
