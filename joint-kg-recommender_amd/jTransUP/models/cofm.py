"""coFM with the reference's class surface (jTransUP/models/cofm.py): FM on the rating side (bias terms + u . i: K1 / K11), TransE on
the triple side (K2 / K12), the item table optionally BEING the entity table (-share_embeddings)."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.models.fm import BiasTable
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return coFM(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, user_total=user_total, item_total=item_total,
                entity_total=entity_total, relation_total=relation_total, isShare=FLAGS.share_embeddings)


class coFM(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, user_total, item_total, entity_total, relation_total, isShare):
        super(coFM, self).__init__()
        self.L1_flag = L1_flag
        self.is_share = isShare
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.ent_total = entity_total
        self.rel_total = relation_total
        self.is_pretrained = False
        # construction order = the reference's (cofm.py:42-95), so torch.manual_seed(s) yields the same tables
        self.user_embeddings = to_gpu(make_embedding(xavier_table(user_total, embedding_size)))
        self.user_bias = to_gpu(BiasTable(user_total))
        self.item_bias = to_gpu(BiasTable(item_total))
        self.bias = nn.Parameter(to_gpu(torch.zeros(1, dtype=torch.float32)))
        self.rel_embeddings = to_gpu(make_embedding(xavier_table(relation_total, embedding_size)))
        self.ent_embeddings = to_gpu(make_embedding(xavier_table(entity_total, embedding_size)))
        if isShare:
            assert item_total == entity_total, "item numbers didn't match entities!"
            self.item_embeddings = self.ent_embeddings
        else:
            self.item_embeddings = to_gpu(make_embedding(xavier_table(item_total, embedding_size)))

    def forward(self, ratings, triples, is_rec=True):
        if is_rec and ratings is not None:                    # cofm.py:99-108
            u_ids, i_ids = ratings
            dot = ops.score_bprmf(self.user_embeddings.weight, self.item_embeddings.weight, u_ids, i_ids)
            return self.bias + self.user_bias(u_ids) + self.item_bias(i_ids) + dot
        if not is_rec and triples is not None:                # cofm.py:110-122: TransE on the shared entity table
            h, t, r = triples
            return ops.score_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, h, t, r, self.L1_flag)
        raise NotImplementedError

    def evaluateRec(self, u_ids, all_i_ids=None):
        """cofm.py:127-141; all_i_ids only matters with shared tables (the item ids' rows of the entity table)."""
        if all_i_ids is not None and self.is_share:
            all_i = self.item_embeddings(all_i_ids).detach().contiguous()
            all_b = self.item_bias(all_i_ids)
        else:
            all_i, all_b = self.item_embeddings.weight, self.item_bias.weight
        gemm = ops.eval_bprmf(self.user_embeddings.weight, all_i, u_ids)
        return gemm + (self.bias + self.user_bias(u_ids))[:, None] + all_b[None, :]

    def _cand(self, all_e_ids):
        return self.ent_embeddings(all_e_ids).detach().contiguous() if (all_e_ids is not None and self.is_share) else None

    def evaluateHead(self, t, r, all_e_ids=None):
        """cofm.py:143-168."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, t, r, self.L1_flag, head=True,
                               candidates=self._cand(all_e_ids))

    def evaluateTail(self, h, r, all_e_ids=None):
        """cofm.py:170-193."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, h, r, self.L1_flag, head=False,
                               candidates=self._cand(all_e_ids))

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, all_e_ids=None):
        """A whole evaluateHead / evaluateTail pass + the filtered gold ranks (utils/misc.py:125-146) in one call."""
        return ops.eval_kg_ranks(self.ent_embeddings.weight, self.rel_embeddings.weight, None, q, r, self.L1_flag, head, descending,
                                 gold_off, gold_ids, filt_off, filt_ids, candidates=self._cand(all_e_ids))
