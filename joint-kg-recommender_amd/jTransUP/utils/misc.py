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
