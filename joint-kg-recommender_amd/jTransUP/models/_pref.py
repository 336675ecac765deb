"""Cold-path pieces shared by TUP and KTUP: the reporting-only getPreferences (transUP.py:105-115,
jTransUP.py:250-260) and the ST-Gumbel estimator used by it, as plain device tensor ops.  The training /
evaluation hot path never comes through here: it is fused inside the HIP kernels (jTransUP/hip/ops.py)."""
import torch
import torch.nn.functional as F

from jTransUP.hip import ops


def st_gumbel_softmax(logits, temperature=1.0, uniform=None):
    """transUP.py:143-170: one-hot forward value, softmax backward."""
    eps = 1e-20
    u = torch.rand_like(logits) if uniform is None else uniform
    y = F.softmax((logits - torch.log(-torch.log(u + eps) + eps)) / temperature, dim=logits.dim() - 1)
    y_hard = torch.zeros_like(y).scatter_(y.dim() - 1, y.max(y.dim() - 1)[1].unsqueeze(-1), 1.0)
    return (y_hard - y).detach() + y


class GateHelpers(object):
    """The gate's helper methods the reference's preference models expose (transUP.py:118-170, jTransUP.py:262-314), for callers
    that use them directly; cold path like getPreferences."""

    def convert_to_one_hot(self, indices, num_classes):
        """transUP.py:118-135: (...,) integer indices -> (..., num_classes) one-hot of the indices' dtype."""
        shape = tuple(indices.shape) + (int(num_classes),)
        return indices.new_zeros(shape).scatter_(indices.dim(), indices.unsqueeze(indices.dim()), 1)

    def masked_softmax(self, logits):
        """transUP.py:138-141: a softmax over the last dimension (the reference's version masks nothing either)."""
        return F.softmax(logits, dim=logits.dim() - 1)

    def st_gumbel_softmax(self, logits, temperature=1.0):
        """transUP.py:143-170: one-hot forward value, softmax backward; noise from torch's generator like the reference's."""
        return st_gumbel_softmax(logits, temperature)


class GumbelState(object):
    """Production ST-Gumbel draws come from Philox4x32-10 on the device; (seed, offset) advance per call so
    forward and backward of one call see the same noise while successive calls are independent."""

    def __init__(self):
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())

    def mode_and_stream(self, use_st_gumbel, uniform, count):
        if not use_st_gumbel:
            return ops.GUMBEL_OFF, None, 0, 0
        if uniform is not None:
            return ops.GUMBEL_INPUT, uniform, 0, 0
        return ops.GUMBEL_PHILOX, None, self.seed, ops.next_philox_offset(count)
