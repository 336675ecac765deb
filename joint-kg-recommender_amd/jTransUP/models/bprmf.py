"""BPRMF with the reference's class surface (jTransUP/models/bprmf.py), scored by HIP kernels."""
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return BPRMF(FLAGS.embedding_size, user_total, item_total)


class BPRMF(nn.Module, GradToggle):
    def __init__(self, embedding_size, user_total, item_total):
        super(BPRMF, self).__init__()
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.is_pretrained = False
        user_weight = xavier_table(user_total, embedding_size)
        item_weight = xavier_table(item_total, embedding_size)
        self.user_embeddings = to_gpu(make_embedding(user_weight))
        self.item_embeddings = to_gpu(make_embedding(item_weight))

    def forward(self, u_ids, i_ids):
        """K1: score[b] = U[u_b] . I[i_b]   (bprmf.py:46-49)."""
        return ops.score_bprmf(self.user_embeddings.weight, self.item_embeddings.weight, u_ids, i_ids)

    def evaluate(self, u_ids):
        """K11: all-item scores U[u] . I^T (bprmf.py:51-54)."""
        return ops.eval_bprmf(self.user_embeddings.weight, self.item_embeddings.weight, u_ids)
