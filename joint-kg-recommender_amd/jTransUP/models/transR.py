"""TransR with the reference's class surface (jTransUP/models/transR.py), scored by HIP kernels."""
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return TransRModel(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, ent_total=entity_total,
                       rel_total=relation_total)


class TransRModel(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, ent_total, rel_total):
        super(TransRModel, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.ent_total = ent_total
        self.rel_total = rel_total
        self.is_pretrained = False
        self.max_entity_batch = 10   # declared but unused by the reference too (transR.py:30)
        ent_weight = xavier_table(ent_total, embedding_size)
        rel_weight = xavier_table(rel_total, embedding_size)
        proj_weight = xavier_table(rel_total, embedding_size * embedding_size)
        self.ent_embeddings = to_gpu(make_embedding(ent_weight))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))
        self.proj_embeddings = to_gpu(make_embedding(proj_weight, normalize=False))   # transR.py:52,56

    def _tables(self):
        return self.ent_embeddings.weight, self.rel_embeddings.weight, self.proj_embeddings.weight

    def forward(self, h, t, r):
        """K4: M_r h + r - M_r t (transR.py:65-78)."""
        E, R, M = self._tables()
        return ops.score_transr(E, R, M, h, t, r, self.L1_flag)

    def prepare_entities(self):
        """Entity side of evaluateHead / evaluateTail (every entity under every relation's M_r), to share between the batches of a
        pass: pass it as `ents=`."""
        E, R, M = self._tables()
        return ops.eval_transr_entities(E, M, R.shape[0], self.L1_flag)

    def evaluateHead(self, t, r, ents=None):
        """K14 (transR.py:80-103): every entity projected by the query's M_r."""
        E, R, M = self._tables()
        return ops.eval_transr(E, R, M, t, r, self.L1_flag, head=True, ents=ents)

    def evaluateTail(self, h, r, ents=None):
        """K14 (transR.py:105-128)."""
        E, R, M = self._tables()
        return ops.eval_transr(E, R, M, h, r, self.L1_flag, head=False, ents=ents)

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, ents=None):
        """A whole evaluateHead / evaluateTail pass + the filtered gold ranks of utils/misc.py:125-146 in one call (K14 + K18 per
        chunk of 512 keys under the C ABI, the rank kernel of a chunk beside the score kernel of the next)."""
        E, R, M = self._tables()
        return ops.eval_kg_ranks_transr(E, R, M, q, r, self.L1_flag, head, descending, gold_off, gold_ids, filt_off, filt_ids, ents=ents)
