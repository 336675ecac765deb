"""TransH with the reference's class surface (jTransUP/models/transH.py), scored by HIP kernels."""
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return TransHModel(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, ent_total=entity_total,
                       rel_total=relation_total)


class TransHModel(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, ent_total, rel_total):
        super(TransHModel, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.ent_total = ent_total
        self.rel_total = rel_total
        self.is_pretrained = False
        ent_weight = xavier_table(ent_total, embedding_size)
        rel_weight = xavier_table(rel_total, embedding_size)
        norm_weight = xavier_table(rel_total, embedding_size)
        self.ent_embeddings = to_gpu(make_embedding(ent_weight))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))
        self.norm_embeddings = to_gpu(make_embedding(norm_weight))

    def _tables(self):
        return self.ent_embeddings.weight, self.rel_embeddings.weight, self.norm_embeddings.weight

    def forward(self, h, t, r):
        """K3: hyperplane projection of h and t on w_r, then the translation distance (transH.py:58-71)."""
        E, R, N = self._tables()
        return ops.score_transh(E, R, N, h, t, r, self.L1_flag)

    def evaluateHead(self, t, r):
        """K13 (transH.py:73-96): every entity is projected on the QUERY's hyperplane."""
        E, R, N = self._tables()
        return ops.eval_transh(E, R, N, t, r, self.L1_flag, head=True)

    def evaluateTail(self, h, r):
        """K13 (transH.py:98-121)."""
        E, R, N = self._tables()
        return ops.eval_transh(E, R, N, h, r, self.L1_flag, head=False)

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None):
        """A whole evaluateHead / evaluateTail pass + the filtered gold ranks of utils/misc.py:125-146 in one call (K13 + K18 per
        chunk of 512 keys under the C ABI): int32 device vector, one rank per gold entry of the pass's CSR index."""
        E, R, N = self._tables()
        return ops.eval_kg_ranks(E, R, N, q, r, self.L1_flag, head, descending, gold_off, gold_ids, filt_off, filt_ids)
