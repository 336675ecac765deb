"""CFKG with the reference's class surface (jTransUP/models/CFKG.py): TransE over users, item-entities and relations, with one
extra "buy" relation (the last row of the relation table) between a user and an item -- a thin module over the K2 / K12 kernels.

Ratings are triples (user, buy, item) whose head lives in the USER table and whose tail in the shared item / entity table
(CFKG.py:62, 66-80).  K2 gathers heads and tails from one table, so the rec branch scores on the row-stack [users ; entities]
(torch.cat -- plumbing; autograd splits the gradient back) with item ids offset by the user count; the all-item evaluation needs
no stacking: K12 takes the query table (users) and the candidate table (entities) separately."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return CFKG(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, user_total=user_total, item_total=item_total,
                entity_total=entity_total, relation_total=relation_total)


class CFKG(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, user_total, item_total, entity_total, relation_total):
        super(CFKG, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.ent_total = entity_total
        self.rel_total = relation_total + 1           # + the buy relation (CFKG.py:34)
        self.is_pretrained = False
        self.user_embeddings = to_gpu(make_embedding(xavier_table(user_total, embedding_size)))
        ent_weight = xavier_table(entity_total, embedding_size)
        rel_weight = xavier_table(self.rel_total, embedding_size)
        self.ent_embeddings = to_gpu(make_embedding(ent_weight))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))
        self.item_embeddings = self.ent_embeddings    # shared table (CFKG.py:62)

    def forward(self, ratings, triples, is_rec=True):
        E, R = self.ent_embeddings.weight, self.rel_embeddings.weight
        if is_rec and ratings is not None:
            u_ids, i_ids = ratings
            # CFKG.py:66-80 : user + buy - item-entity, on the batch's rows only (heads = the gathered user rows, tails = the
            # gathered entity rows of one 2B-row table) instead of concatenating the two whole tables per step
            n = u_ids.numel()
            both = torch.cat([self.user_embeddings(u_ids), self.ent_embeddings(i_ids)])
            at = torch.arange(n, device=u_ids.device)
            buy = torch.full_like(u_ids, self.rel_total - 1)
            return ops.score_transe(both, R, at, at + n, buy, self.L1_flag)
        if not is_rec and triples is not None:
            h, t, r = triples
            return ops.score_transe(E, R, h, t, r, self.L1_flag)                                             # CFKG.py:81-92
        raise NotImplementedError

    def evaluateRec(self, u_ids, all_i_ids=None):
        """CFKG.py:100-118: (user + buy) against every (or the given) item-entity row."""
        cand = self.item_embeddings(all_i_ids) if all_i_ids is not None else self.item_embeddings.weight
        buy = torch.full_like(u_ids, self.rel_total - 1)
        return ops.eval_transe(self.user_embeddings.weight, self.rel_embeddings.weight, u_ids, buy, self.L1_flag, head=False,
                               candidates=cand.detach().contiguous())

    def _cand(self, all_e_ids):
        return self.ent_embeddings(all_e_ids).detach().contiguous() if all_e_ids is not None else None

    def evaluateHead(self, t, r, all_e_ids=None):
        """CFKG.py:120-138."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, t, r, self.L1_flag, head=True,
                               candidates=self._cand(all_e_ids))

    def evaluateTail(self, h, r, all_e_ids=None):
        """CFKG.py:140-158."""
        return ops.eval_transe(self.ent_embeddings.weight, self.rel_embeddings.weight, h, r, self.L1_flag, head=False,
                               candidates=self._cand(all_e_ids))

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, all_e_ids=None):
        """A whole evaluateHead / evaluateTail pass (CFKG.py:120-158) + the filtered gold ranks of utils/misc.py:125-146 in one call
        (K12 + K18 per chunk of 512 keys under the C ABI)."""
        return ops.eval_kg_ranks(self.ent_embeddings.weight, self.rel_embeddings.weight, None, q, r, self.L1_flag, head, descending,
                                 gold_off, gold_ids, filt_off, filt_ids, candidates=self._cand(all_e_ids))
