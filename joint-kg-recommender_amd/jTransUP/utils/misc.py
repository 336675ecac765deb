"""Host-side helpers mirrored from the reference's jTransUP/utils/misc.py (device placement only here;
the ranking entry points live in jTransUP/utils/ranking.py and are re-exported below)."""
import torch

USE_CUDA = torch.cuda.is_available()   # "cuda" is the ROCm/HIP device in PyTorch-ROCm


def to_gpu(var):
    """utils/misc.py:13-16."""
    if USE_CUDA:
        return var.cuda()
    return var


# ---- the projection helpers the reference's models call (utils/misc.py:18-37).  Plain tensor expressions for callers that use them
# directly; the models of this build do not come through here (their projections are fused into the HIP kernels).
def projection_transH_pytorch(original, norm):
    """utils/misc.py:18-19: the component of `original` inside the hyperplane with normal `norm` (last dimension)."""
    return original - torch.sum(original * norm, dim=original.dim() - 1, keepdim=True) * norm


def projection_transR_pytorch(original, proj_matrix):
    """utils/misc.py:21-26: rows (B, d_e) through their own (d_r x d_e) matrices given flat as (B, d_r * d_e) -> (B, d_r)."""
    d_e = original.shape[1]
    d_r = proj_matrix.shape[1] // d_e
    return torch.matmul(proj_matrix.view(-1, d_r, d_e), original.view(-1, d_e, 1)).view(-1, d_r)


def projection_transR_pytorch_batch(original, proj_matrix):
    """utils/misc.py:29-33: every row of `original` (E, d_e) through each of B matrices (B, d_r * d_e) -> (B, E, d_r)."""
    d_e = original.shape[1]
    d_r = proj_matrix.shape[1] // d_e
    return torch.matmul(proj_matrix.view(-1, d_r, d_e), original.transpose(0, 1)).transpose(1, 2)


def projection_transD_pytorch_samesize(entity_embedding, entity_projection, relation_projection):
    """utils/misc.py:36-37 (TransD itself is out of scope here: `init_model` refuses it; the helper is kept for callers)."""
    return entity_embedding + torch.sum(entity_embedding * entity_projection, dim=entity_embedding.dim() - 1, keepdim=True) * relation_projection


def recursively_set_device(inp, gpu=USE_CUDA):
    """utils/misc.py:250-263, including its quirks: dict values are replaced in place, a tuple comes back as a generator, and the
    `gpu` argument is ignored in favour of USE_CUDA."""
    if hasattr(inp, 'keys'):
        for k in list(inp.keys()):
            inp[k] = recursively_set_device(inp[k], USE_CUDA)
    elif isinstance(inp, list):
        return [recursively_set_device(ii, USE_CUDA) for ii in inp]
    elif isinstance(inp, tuple):
        return (recursively_set_device(ii, USE_CUDA) for ii in inp)
    elif hasattr(inp, 'cpu'):
        inp = inp.cuda() if USE_CUDA else inp.cpu()
    return inp


class Accumulator(object):
    """utils/misc.py:39-59 -- trailing statistics."""

    def __init__(self, maxlen=None):
        from collections import deque
        self._deque, self.maxlen, self.cache = deque, maxlen, dict()

    def add(self, key, val):
        self.cache.setdefault(key, self._deque(maxlen=self.maxlen)).append(val)

    def get(self, key, clear=True):
        ret = self.cache.get(key, [])
        if clear:
            self.cache.pop(key, None)
        return ret

    def get_avg(self, key, clear=True):
        import numpy as np
        return np.array(self.get(key, clear)).mean()


# ---- ranking entry points (utils/misc.py:61-248 of the reference), device-backed: see jTransUP/utils/ranking.py
def evalRecProcess(pred_scores, eval_dict, all_dicts=None, descending=True, num_processes=None, topn=10, queue_limit=10, **kwargs):
    """utils/misc.py:186-210 (`num_processes` / `queue_limit` accepted and ignored: no process fan-out)."""
    from jTransUP.utils.ranking import evalRecProcess as f
    return f(pred_scores, eval_dict, all_dicts=all_dicts, descending=descending, num_processes=num_processes, topn=topn,
             queue_limit=queue_limit, **kwargs)


def evalKGProcess(pred_scores, eval_dict, all_dicts=None, descending=True, num_processes=None, topn=10, queue_limit=10, **kwargs):
    """utils/misc.py:98-122."""
    from jTransUP.utils.ranking import evalKGProcess as f
    return f(pred_scores, eval_dict, all_dicts=all_dicts, descending=descending, num_processes=num_processes, topn=topn,
             queue_limit=queue_limit, **kwargs)


def getRecPerformance(pred, gold, fliter_samples=None, topn=10):
    """utils/misc.py:213-248 for one (already sign-adjusted, lower = better) score row."""
    rows = evalRecProcess([(0, pred)], {0: gold}, all_dicts=None if fliter_samples is None else [{0: fliter_samples}],
                          descending=False, topn=topn)
    f1, p, r, hit, ndcg, (_, top_ids, _) = rows[0]
    return f1, p, r, hit, ndcg, top_ids


def getKGPerformance(pred, gold, fliter_samples=None, topn=10):
    """utils/misc.py:125-146 for one (already sign-adjusted) score row."""
    rows = evalKGProcess([(0, pred)], {0: gold}, all_dicts=None if fliter_samples is None else [{0: fliter_samples}],
                         descending=False, topn=topn)
    return [h for h, _, _, _ in rows], [rk for _, rk, _, _ in rows], [g for _, _, _, g in rows]
