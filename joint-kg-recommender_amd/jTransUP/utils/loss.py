"""Losses and regularisers with the reference's names and call shapes (jTransUP/utils/loss.py), computed by
HIP kernels (jTransUP/hip/ops.py -> ktup_loss_* / ktup_reg_*).

The reference passes already-gathered embeddings to normLoss / orthogonalLoss (a second nn.Embedding lookup of
the batch's rows).  Those call shapes still work; the drivers in this package use the fused form
`normLoss(table.weight, ids=...)` which never materialises the gathered rows."""
import torch.nn as nn

from jTransUP.hip import ops


class marginLoss(nn.Module):
    """utils/loss.py:8-16 : sum(max(pos - neg + margin, 0))."""

    def __init__(self):
        super(marginLoss, self).__init__()

    def forward(self, pos, neg, margin):
        return ops.margin_loss(pos, neg, margin)


def orthogonalLoss(rel_embeddings, norm_embeddings, ids=None):
    """utils/loss.py:18-19."""
    return ops.orthogonal_loss(rel_embeddings, norm_embeddings, ids)


def normLoss(embeddings, dim=1, ids=None):
    """utils/loss.py:21-23 (only dim=1, the one the reference ever uses)."""
    if dim != 1:
        raise NotImplementedError('normLoss: only dim=1 is used by the reference drivers')
    return ops.norm_loss(embeddings, ids)


def bprLoss(pos, neg, target=1.0):
    """utils/loss.py:29-31."""
    return ops.bpr_loss(pos, neg, target)


def pNormLoss(emb1, emb2, L1_flag=False):
    """utils/loss.py:33-38 -- only used by the cofm/cfkg alignment term (out of scope models); plain tensor ops."""
    import torch
    distance = torch.sum(torch.abs(emb1 - emb2), 1) if L1_flag else torch.sum((emb1 - emb2) ** 2, 1)
    return distance.mean()
