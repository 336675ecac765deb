"""KTUP with the reference's class surface (jTransUP/models/jTransUP.py), scored by HIP kernels."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.models._pref import GateHelpers, GumbelState, st_gumbel_softmax
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return jTransUPModel(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, user_total=user_total,
                         item_total=item_total, entity_total=entity_total, relation_total=relation_total, i_map=i_map,
                         new_map=new_map, isShare=FLAGS.share_embeddings, use_st_gumbel=FLAGS.use_st_gumbel)


class jTransUPModel(nn.Module, GradToggle, GateHelpers):
    def __init__(self, L1_flag, embedding_size, user_total, item_total, entity_total, relation_total, i_map, new_map,
                 isShare, use_st_gumbel):
        super(jTransUPModel, self).__init__()
        self.L1_flag = L1_flag
        self.is_share = isShare
        self.use_st_gumbel = use_st_gumbel
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.ent_total = entity_total + 1          # + zero pad row for items aligned with no entity (jTransUP.py:46)
        self.rel_total = relation_total
        self.is_pretrained = False
        self.i_map = i_map                         # item id -> index into new_map
        self.new_map = new_map                     # index -> (entity id | -1, item id | -1)
        # TUP side (draw order as jTransUP.py:55-62)
        user_weight = xavier_table(user_total, embedding_size)
        item_weight = xavier_table(item_total, embedding_size)
        pref_weight = xavier_table(relation_total, embedding_size)
        pref_norm_weight = xavier_table(relation_total, embedding_size)
        self.user_embeddings = to_gpu(make_embedding(user_weight))
        self.item_embeddings = to_gpu(make_embedding(item_weight))
        self.pref_embeddings = to_gpu(make_embedding(pref_weight))
        self.pref_norm_embeddings = to_gpu(make_embedding(pref_norm_weight))
        # TransH side (jTransUP.py:88-107): normalised entity rows followed by the all-zero pad row
        ent_weight = xavier_table(self.ent_total - 1, embedding_size)
        rel_weight = xavier_table(relation_total, embedding_size)
        norm_weight = xavier_table(relation_total, embedding_size)
        ent_full = torch.cat([torch.nn.functional.normalize(ent_weight, p=2, dim=1), torch.zeros(1, embedding_size)], dim=0)
        self.ent_embeddings = to_gpu(make_embedding(ent_full, normalize=False, padding_idx=self.ent_total - 1))
        self.rel_embeddings = to_gpu(make_embedding(rel_weight))
        self.norm_embeddings = to_gpu(make_embedding(norm_weight))
        self._gumbel = GumbelState()
        # paddingItems (jTransUP.py:114-120) is a per-item python dict walk inside forward; here it is one
        # int32 device table built once.  `_item2ent[i]` serves forward, `_eval_item2ent[j]` serves evaluateRec,
        # which pairs item row j with the entity of the j-th KEY of i_map (jTransUP.py:174).
        pad = self.ent_total - 1
        table = [pad] * item_total
        self._unmapped = []
        for i_id in range(item_total):
            try:
                ent_id = new_map[i_map[i_id]][0]
                table[i_id] = ent_id if ent_id != -1 else pad
            except KeyError:
                self._unmapped.append(i_id)    # the reference raises KeyError when such an item is scored
        key_order = []
        for i_id in i_map:
            ent_id = new_map[i_map[i_id]][0]
            key_order.append(ent_id if ent_id != -1 else pad)
        self.register_buffer('_item2ent', to_gpu(torch.tensor(table, dtype=torch.int32)), persistent=False)
        self.register_buffer('_eval_item2ent', to_gpu(torch.tensor(key_order, dtype=torch.int32)), persistent=False)

    def paddingItems(self, i_ids, pad_index):
        """jTransUP.py:114-120 (kept for callers; the kernels use the precomputed table)."""
        padded_e_ids = []
        for i_id in i_ids:
            ent_id = self.new_map[self.i_map[int(i_id)]][0]
            padded_e_ids.append(ent_id if ent_id != -1 else pad_index)
        return padded_e_ids

    def _rec_tables(self):
        return (self.user_embeddings.weight, self.item_embeddings.weight, self.ent_embeddings.weight,
                self.pref_embeddings.weight, self.pref_norm_embeddings.weight, self.rel_embeddings.weight,
                self.norm_embeddings.weight)

    def forward(self, ratings, triples, is_rec=True, uniform=None):
        if is_rec and ratings is not None:
            # K6 (+K7): jTransUP.py:122-143
            u_ids, i_ids = ratings
            U, I, E, P, Pn, R, Rn = self._rec_tables()
            mode, uni, seed, off = self._gumbel.mode_and_stream(self.use_st_gumbel, uniform, u_ids.numel() * P.shape[0])
            return ops.score_ktup(U, I, E, P, Pn, R, Rn, self._item2ent, u_ids, i_ids, self.L1_flag, mode, uni, seed, off,
                                  ent_pad=self.ent_total - 1)
        elif not is_rec and triples is not None:
            # K3 on the shared tables: jTransUP.py:144-157
            h, t, r = triples
            return ops.score_transh(self.ent_embeddings.weight, self.rel_embeddings.weight, self.norm_embeddings.weight,
                                    h, t, r, self.L1_flag)
        else:
            raise NotImplementedError

    def _eval_items(self, all_i_ids):
        I = self.item_embeddings.weight
        if all_i_ids is not None and self.is_share:
            I = self.item_embeddings(all_i_ids)
            return I, to_gpu(torch.tensor(self.paddingItems(all_i_ids.tolist(), self.ent_total - 1), dtype=torch.int32))
        item2ent = self._eval_item2ent
        if item2ent.numel() != I.shape[0]:
            raise ValueError('evaluateRec: i_map has %d items but the item table has %d rows' % (item2ent.numel(), I.shape[0]))
        return I, item2ent

    def evaluateRec(self, u_ids, all_i_ids=None, uniform=None, items=None):
        """K16: jTransUP.py:163-191.  `items` (this build): `prepare_items(all_i_ids)` taken once per evaluation pass."""
        U, _, E, P, Pn, R, Rn = self._rec_tables()
        I, item2ent = self._eval_items(all_i_ids)
        mode, uni, seed, off = self._gumbel.mode_and_stream(self.use_st_gumbel, uniform,
                                                            u_ids.numel() * I.shape[0] * P.shape[0])
        return ops.eval_ktup(U, I, E, P, Pn, R, Rn, item2ent, u_ids, self.L1_flag, mode, uni, seed, off, items=items)

    def evaluate_topk(self, u_ids, items, topn, filt_off=None, filt_ids=None):
        """K16 + K17 for a whole evaluation pass in one sweep (this build): filtered top-n item ids of every user of `u_ids`
        without the (users x items) matrix.  The ST-Gumbel gate takes the hard gate's sweep (fresh Philox noise per pass, as `evaluate`
        draws it per batch); None when no fused pass applies (soft gate with L1 or an unsupported width)."""
        if self.use_st_gumbel:
            mode, uni, seed, off = self._gumbel.mode_and_stream(True, None, u_ids.numel() * items.n_items * items.P)
            return ops.eval_pref_topk_hard(self.user_embeddings.weight, u_ids, items, self.L1_flag, topn, mode, uni, seed, off,
                                           filt_off, filt_ids)
        return ops.eval_pref_topk(self.user_embeddings.weight, u_ids, items, self.L1_flag, topn, filt_off, filt_ids)

    def prepare_items(self, all_i_ids=None):
        """Item side of `evaluateRec` (item + entity rows through the preference gate), to share between the batches of a pass."""
        _, _, E, P, Pn, R, Rn = self._rec_tables()
        I, item2ent = self._eval_items(all_i_ids)
        return ops.eval_pref_items(I, E, P, Pn, R, Rn, item2ent)

    def _all_entities(self, all_e_ids):
        return self.ent_embeddings(all_e_ids) if all_e_ids is not None and self.is_share else self.ent_embeddings.weight

    def evaluateHead(self, t, r, all_e_ids=None):
        """K13 over ent_total rows INCLUDING the pad row (jTransUP.py:193-219)."""
        return ops.eval_transh(self.ent_embeddings.weight, self.rel_embeddings.weight, self.norm_embeddings.weight, t, r,
                               self.L1_flag, head=True, candidates=self._all_entities(all_e_ids))

    def evaluateTail(self, h, r, all_e_ids=None):
        """K13 (jTransUP.py:221-247)."""
        return ops.eval_transh(self.ent_embeddings.weight, self.rel_embeddings.weight, self.norm_embeddings.weight, h, r,
                               self.L1_flag, head=False, candidates=self._all_entities(all_e_ids))

    def rank_entities(self, q, r, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, all_e_ids=None):
        """A whole evaluateHead / evaluateTail pass (jTransUP.py:193-247) + the filtered gold ranks of utils/misc.py:125-146 in one
        call (K13 + K18 per chunk of 512 keys under the C ABI): int32 device vector, one rank per gold entry of the pass's index."""
        return ops.eval_kg_ranks(self.ent_embeddings.weight, self.rel_embeddings.weight, self.norm_embeddings.weight, q, r, self.L1_flag,
                                 head, descending, gold_off, gold_ids, filt_off, filt_ids, candidates=self._all_entities(all_e_ids))

    def getPreferences(self, u_e, i_e, use_st_gumbel=False):
        """jTransUP.py:250-260 on already-gathered embeddings (reporting path only)."""
        A = self.pref_embeddings.weight + self.rel_embeddings.weight
        C = self.pref_norm_embeddings.weight + self.norm_embeddings.weight
        pre_probs = torch.matmul(u_e + i_e, torch.t(A)) / 2
        if use_st_gumbel:
            pre_probs = st_gumbel_softmax(pre_probs)
        return pre_probs, torch.matmul(pre_probs, A) / 2, torch.matmul(pre_probs, C) / 2

    def reportPreference(self, u_id, i_ids):
        """jTransUP.py:317-328."""
        item_num = len(i_ids)
        u_e = self.user_embeddings(u_id.expand(item_num))
        ie_e = self.item_embeddings(i_ids) + self.ent_embeddings(self._item2ent[i_ids].long())
        return self.getPreferences(u_e, ie_e, use_st_gumbel=self.use_st_gumbel)
