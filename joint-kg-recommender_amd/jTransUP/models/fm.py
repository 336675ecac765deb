"""FM with the reference's class surface (jTransUP/models/fm.py): global bias + user bias + item bias + u . i.  The pair term is
K1 (the BPRMF row-dot kernel) and the all-item form K11 (the fp32 MFMA GEMM); the three bias terms are broadcast adds."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return FM(FLAGS.embedding_size, user_total, item_total)


class BiasTable(nn.Module):
    """The reference keeps its biases as nn.Embedding(n, 1) whose weight is replaced by a 1-D Parameter of n zeros (fm.py:31-44):
    state-dict key `<name>.weight`, shape (n,).  Same key, same shape; a lookup is an index."""

    def __init__(self, n):
        super(BiasTable, self).__init__()
        self.weight = nn.Parameter(torch.zeros(n, dtype=torch.float32))

    def forward(self, ids):
        return self.weight[ids]


class FM(nn.Module, GradToggle):
    def __init__(self, embedding_size, user_total, item_total):
        super(FM, self).__init__()
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.is_pretrained = False
        user_weight = xavier_table(user_total, embedding_size)
        item_weight = xavier_table(item_total, embedding_size)
        self.user_embeddings = to_gpu(make_embedding(user_weight))
        self.item_embeddings = to_gpu(make_embedding(item_weight))
        self.user_bias = to_gpu(BiasTable(user_total))
        self.item_bias = to_gpu(BiasTable(item_total))
        self.bias = nn.Parameter(to_gpu(torch.zeros(1, dtype=torch.float32)))

    def forward(self, u_ids, i_ids):
        """fm.py:58-67: y = b + b_u + b_i + u . i  (the row dot is K1)."""
        dot = ops.score_bprmf(self.user_embeddings.weight, self.item_embeddings.weight, u_ids, i_ids)
        return self.bias + self.user_bias(u_ids) + self.item_bias(i_ids) + dot

    def evaluate(self, u_ids):
        """fm.py:69-80: (len(u), item_total); the GEMM is K11."""
        gemm = ops.eval_bprmf(self.user_embeddings.weight, self.item_embeddings.weight, u_ids)
        return gemm + (self.bias + self.user_bias(u_ids))[:, None] + self.item_bias.weight[None, :]
