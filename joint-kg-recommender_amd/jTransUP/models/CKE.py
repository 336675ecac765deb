"""CKE with the reference's class surface (jTransUP/models/CKE.py): BPRMF over (item + aligned entity) rows for ratings,
TransR for triples -- a thin module over the K1 / K4 / K11 / K14 kernels.

The item side `I[i] + E[item2ent[i]]` (CKE.py:126-133; the per-item dict walk `paddingItems` :106-112 becomes an int32 device
table built once) is formed as one (items x d) table per call by torch's gather + add -- plumbing; autograd routes its gradient to
both tables and nn.Embedding(padding_idx) keeps the pad entity row gradient-free like the reference -- and the scores come from
the same HIP kernels BPRMF and TransR use."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return CKE(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, user_total=user_total, item_total=item_total,
               entity_total=entity_total, relation_total=relation_total, i_map=i_map, new_map=new_map)


class CKE(nn.Module, GradToggle):
    def __init__(self, L1_flag, embedding_size, user_total, item_total, entity_total, relation_total, i_map, new_map):
        super(CKE, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.ent_total = entity_total + 1            # + the zero pad row for items without an entity (CKE.py:37)
        self.rel_total = relation_total
        self.is_pretrained = False
        self.i_map, self.new_map = i_map, new_map
        self.is_share = False
        user_weight = xavier_table(user_total, embedding_size)
        item_weight = xavier_table(item_total, embedding_size)
        self.user_embeddings = to_gpu(make_embedding(user_weight))
        self.item_embeddings = to_gpu(make_embedding(item_weight))
        ent_weight = torch.nn.functional.normalize(xavier_table(entity_total, embedding_size), p=2, dim=1)
        rel_weight = xavier_table(relation_total, embedding_size)
        proj_weight = xavier_table(relation_total, embedding_size * embedding_size)
        self.ent_embeddings = to_gpu(make_embedding(torch.cat([ent_weight, torch.zeros(1, embedding_size)]), normalize=False,
                                                    padding_idx=self.ent_total - 1))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))
        self.proj_embeddings = to_gpu(make_embedding(proj_weight, normalize=False))
        table = self.paddingItems(range(item_total), self.ent_total - 1)
        self.register_buffer('_item2ent', to_gpu(torch.tensor(table, dtype=torch.int64)), persistent=False)

    def paddingItems(self, i_ids, pad_index):
        """CKE.py:106-112."""
        out = []
        for i_id in i_ids:
            ent_id = self.new_map[self.i_map[int(i_id)]][0]
            out.append(ent_id if ent_id != -1 else pad_index)
        return out

    def _item_side(self):
        return self.item_embeddings.weight + self.ent_embeddings(self._item2ent)      # (items x d); pad rows add zero

    def forward(self, ratings, triples, is_rec=True):
        if is_rec and ratings is not None:
            u_ids, i_ids = ratings
            # CKE.py:122-135 on the BATCH's rows only: item row + aligned entity row (pad row = zero) gathered per pair, so a
            # step costs O(batch) memory traffic, not a pass over the item table (the full table is built for evaluateRec only)
            iv = self.item_embeddings(i_ids) + self.ent_embeddings(self._item2ent[i_ids])
            return ops.score_bprmf(self.user_embeddings.weight, iv, u_ids, torch.arange(i_ids.numel(), device=i_ids.device))
        if not is_rec and triples is not None:
            h, t, r = triples
            return ops.score_transr(self.ent_embeddings.weight, self.rel_embeddings.weight, self.proj_embeddings.weight, h, t, r,
                                    self.L1_flag)                                                            # CKE.py:136-149
        raise NotImplementedError

    def evaluateRec(self, u_ids, all_i_ids=None):
        """CKE.py:142-153 (all_i_ids is ignored by the reference too)."""
        with torch.no_grad():
            return ops.eval_bprmf(self.user_embeddings.weight, self._item_side().contiguous(), u_ids)

    def prepare_entities(self):
        """Entity side of evaluateHead / evaluateTail, once per evaluation pass (`ents=`)."""
        return ops.eval_transr_entities(self.ent_embeddings.weight, self.proj_embeddings.weight, self.rel_embeddings.weight.shape[0], self.L1_flag)

    def evaluateHead(self, t, r, all_e_ids=None, ents=None):
        """CKE.py:155-178 over the entity table INCLUDING the pad row."""
        return ops.eval_transr(self.ent_embeddings.weight, self.rel_embeddings.weight, self.proj_embeddings.weight, t, r, self.L1_flag,
                               head=True, ents=ents)

    def evaluateTail(self, h, r, all_e_ids=None, ents=None):
        """CKE.py:180-203."""
        return ops.eval_transr(self.ent_embeddings.weight, self.rel_embeddings.weight, self.proj_embeddings.weight, h, r, self.L1_flag,
                               head=False, ents=ents)
