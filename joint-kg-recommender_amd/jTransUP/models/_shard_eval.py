"""Candidate-shard scoring for evaluation with the catalogue split over the ranks (`-shard_eval_candidates`).

The reference's evaluate* methods score every query against the WHOLE catalogue (bprmf.py:51-54, transUP.py:84-102,
jTransUP.py:163-247, transE.py:65-105, transH.py:73-121).  Here a rank scores its contiguous candidate slice [lo, hi) only --
the same kernels on a row slice of the candidate table -- and the ranking is completed across ranks by
parallel.sharded_topk / parallel.sharded_gold_ranks.  Returns None for models without a sliceable candidate table (TransR
projects candidates per relation; shared-embedding KTUP re-maps ids): the drivers then deal whole batches to the ranks."""
import torch

from jTransUP.hip import ops


def rec_shard_fn(model):
    """-> (n_candidates, f(u_ids, lo, hi) -> (len(u), hi - lo) scores) or None."""
    name = type(model).__name__
    if name == 'BPRMF':
        U, I = model.user_embeddings.weight, model.item_embeddings.weight
        return I.shape[0], lambda u, lo, hi: ops.eval_bprmf(U, I[lo:hi], u)
    if name == 'TransUPModel':
        U, I, P, Pn = model._tables()

        def f(u, lo, hi):
            mode, uni, seed, off = model._gumbel.mode_and_stream(model.use_st_gumbel, None, u.numel() * (hi - lo) * P.shape[0])
            return ops.eval_tup(U, I[lo:hi], P, Pn, u, model.L1_flag, mode, uni, seed, off)
        return I.shape[0], f
    if name == 'jTransUPModel' and not model.is_share:
        U, I, E, P, Pn, R, Rn = model._rec_tables()
        item2ent = model._eval_item2ent

        def f(u, lo, hi):
            mode, uni, seed, off = model._gumbel.mode_and_stream(model.use_st_gumbel, None, u.numel() * (hi - lo) * P.shape[0])
            return ops.eval_ktup(U, I[lo:hi], E, P, Pn, R, Rn, item2ent[lo:hi].contiguous(), u, model.L1_flag, mode, uni, seed, off)
        return I.shape[0], f
    return None


def kg_shard_fn(model, head):
    """-> (n_candidates, f(q_ids, r_ids, lo, hi) -> (len(q), hi - lo) scores) or None."""
    name = type(model).__name__
    if name == 'TransEModel':
        E, R = model.ent_embeddings.weight, model.rel_embeddings.weight
        return E.shape[0], lambda q, r, lo, hi: ops.eval_transe(E, R, q, r, model.L1_flag, head, candidates=E[lo:hi])
    if name == 'TransHModel' or (name == 'jTransUPModel' and not model.is_share):
        E, R, N = model.ent_embeddings.weight, model.rel_embeddings.weight, model.norm_embeddings.weight
        return E.shape[0], lambda q, r, lo, hi: ops.eval_transh(E, R, N, q, r, model.L1_flag, head, candidates=E[lo:hi])
    return None
