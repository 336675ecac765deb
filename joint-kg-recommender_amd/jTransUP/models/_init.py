"""Table construction shared by the model classes: xavier-uniform then row-L2 normalisation, drawn from
torch's global generator in the same order as the reference ctors (e.g. transUP.py:37-62), so that
torch.manual_seed(s) yields the same initial tables."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def xavier_table(rows, cols):
    w = torch.empty(rows, cols, dtype=torch.float32)
    nn.init.xavier_uniform_(w)
    return w


def make_embedding(weight, normalize=True, padding_idx=None):
    rows, cols = weight.shape
    emb = nn.Embedding(rows, cols, padding_idx=padding_idx)   # consumes RNG exactly like the reference ctor
    if normalize:
        weight = F.normalize(weight, p=2, dim=1)
    emb.weight = nn.Parameter(weight.contiguous())
    return emb


class GradToggle(object):
    """disable_grad / enable_grad of every reference model (e.g. transE.py:107-113)."""

    def disable_grad(self):
        for _, param in self.named_parameters():
            param.requires_grad = False

    def enable_grad(self):
        for _, param in self.named_parameters():
            param.requires_grad = True
