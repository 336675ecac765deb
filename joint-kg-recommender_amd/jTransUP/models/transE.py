"""TransE with the reference's class surface (jTransUP/models/transE.py), scored by HIP kernels."""
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return TransEModel(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, ent_total=entity_total,
                       rel_total=relation_total)


class TransEModel(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, ent_total, rel_total):
        super(TransEModel, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.ent_total = ent_total
        self.rel_total = rel_total
        self.is_pretrained = False
        ent_weight = xavier_table(ent_total, embedding_size)
        rel_weight = xavier_table(rel_total, embedding_size)
        self.ent_embeddings = to_gpu(make_embedding(ent_weight))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))

    def forward(self, h, t, r):
        """K2: sum|h + r - t| or sum (h + r - t)^2  (transE.py:51-63)."""
        return ops.score_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, h, t, r, self.L1_flag)

    def evaluateHead(self, t, r):
        """K12: distance of t - r to every entity (transE.py:65-84)."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, t, r, self.L1_flag, head=True)

    def evaluateTail(self, h, r):
        """K12: distance of h + r to every entity (transE.py:86-105)."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, h, r, self.L1_flag, head=False)

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None):
        """A whole evaluateHead / evaluateTail pass + the filtered gold ranks of utils/misc.py:125-146 in one call (K12 + K18 per
        chunk of 512 keys under the C ABI): int32 device vector, one rank per gold entry of the pass's CSR index."""
        return ops.eval_kg_ranks(self.ent_embeddings.weight, self.rel_embeddings.weight, None, q, r, self.L1_flag, head, descending,
                                 gold_off, gold_ids, filt_off, filt_ids)
