"""Host-side helpers mirrored from the reference's jTransUP/utils/misc.py (device placement only here;
the ranking entry points live in jTransUP/utils/ranking.py and are re-exported below)."""
import torch

USE_CUDA = torch.cuda.is_available()   # "cuda" is the ROCm/HIP device in PyTorch-ROCm


def to_gpu(var):
    """utils/misc.py:13-16."""
    if USE_CUDA:
        return var.cuda()
    return var


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
def evalRecProcess(*args, **kwargs):
    from jTransUP.utils.ranking import evalRecProcess as f
    return f(*args, **kwargs)


def evalKGProcess(*args, **kwargs):
    from jTransUP.utils.ranking import evalKGProcess as f
    return f(*args, **kwargs)


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
