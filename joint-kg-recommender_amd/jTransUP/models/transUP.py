"""TUP with the reference's class surface (jTransUP/models/transUP.py), scored by HIP kernels."""
import torch
import torch.nn as nn

from jTransUP.hip import ops
from jTransUP.models._init import GradToggle, make_embedding, xavier_table
from jTransUP.models._pref import GateHelpers, GumbelState, st_gumbel_softmax
from jTransUP.utils.misc import to_gpu


def build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=None, e_map=None, new_map=None):
    return TransUPModel(L1_flag=FLAGS.L1_flag, embedding_size=FLAGS.embedding_size, user_total=user_total,
                        item_total=item_total, preference_total=FLAGS.num_preferences, use_st_gumbel=FLAGS.use_st_gumbel)


class TransUPModel(nn.Module, GradToggle, GateHelpers):
    def __init__(self, L1_flag, embedding_size, user_total, item_total, preference_total, use_st_gumbel):
        super(TransUPModel, self).__init__()
        self.L1_flag = L1_flag
        self.embedding_size = embedding_size
        self.user_total = user_total
        self.item_total = item_total
        self.preference_total = preference_total
        self.is_pretrained = False
        self.use_st_gumbel = use_st_gumbel
        user_weight = xavier_table(user_total, embedding_size)
        item_weight = xavier_table(item_total, embedding_size)
        pref_weight = xavier_table(preference_total, embedding_size)
        norm_weight = xavier_table(preference_total, embedding_size)
        self.user_embeddings = to_gpu(make_embedding(user_weight))
        self.item_embeddings = to_gpu(make_embedding(item_weight))
        self.pref_embeddings = to_gpu(make_embedding(pref_weight))
        self.pref_norm_embeddings = to_gpu(make_embedding(norm_weight))
        self._gumbel = GumbelState()

    def _tables(self):
        return (self.user_embeddings.weight, self.item_embeddings.weight, self.pref_embeddings.weight,
                self.pref_norm_embeddings.weight)

    def forward(self, u_ids, i_ids, uniform=None):
        """K5 (+K7): transUP.py:69-82.  `uniform` (B x P) pins the ST-Gumbel draw (tests); otherwise Philox."""
        U, I, P, Pn = self._tables()
        mode, uni, seed, off = self._gumbel.mode_and_stream(self.use_st_gumbel, uniform, u_ids.numel() * P.shape[0])
        return ops.score_tup(U, I, P, Pn, u_ids, i_ids, self.L1_flag, mode, uni, seed, off)

    def evaluate(self, u_ids, uniform=None, items=None):
        """K15: TUP score of every (user, item) pair (transUP.py:84-102); stochastic under ST-Gumbel like the reference.
        `items` (this build): `prepare_items()` taken once per evaluation pass, while the weights are frozen."""
        U, I, P, Pn = self._tables()
        mode, uni, seed, off = self._gumbel.mode_and_stream(self.use_st_gumbel, uniform,
                                                            u_ids.numel() * I.shape[0] * P.shape[0])
        return ops.eval_tup(U, I, P, Pn, u_ids, self.L1_flag, mode, uni, seed, off, items=items)

    def evaluate_topk(self, u_ids, items, topn, filt_off=None, filt_ids=None):
        """K15 + K17 for a whole evaluation pass in one sweep (this build): filtered top-n item ids of every user of `u_ids`
        without the (users x items) matrix.  The ST-Gumbel gate takes the hard gate's sweep (fresh Philox noise per pass, as `evaluate`
        draws it per batch); None when no fused pass applies (soft gate with L1 or an unsupported width)."""
        if self.use_st_gumbel:
            mode, uni, seed, off = self._gumbel.mode_and_stream(True, None, u_ids.numel() * items.n_items * items.P)
            return ops.eval_pref_topk_hard(self.user_embeddings.weight, u_ids, items, self.L1_flag, topn, mode, uni, seed, off,
                                           filt_off, filt_ids)
        return ops.eval_pref_topk(self.user_embeddings.weight, u_ids, items, self.L1_flag, topn, filt_off, filt_ids)

    def prepare_items(self):
        """Item side of `evaluate` (the item projections of the preference gate), to share between the batches of a pass."""
        U, I, P, Pn = self._tables()
        return ops.eval_pref_items(I, None, P, Pn, None, None, None)

    def getPreferences(self, u_e, i_e, use_st_gumbel=False):
        """transUP.py:105-115 on already-gathered embeddings (reporting path only)."""
        pre_probs = torch.matmul(u_e + i_e, torch.t(self.pref_embeddings.weight)) / 2
        if use_st_gumbel:
            pre_probs = st_gumbel_softmax(pre_probs)
        r_e = torch.matmul(pre_probs, self.pref_embeddings.weight)
        norm = torch.matmul(pre_probs, self.pref_norm_embeddings.weight)
        return pre_probs, r_e, norm

    def reportPreference(self, u_id, i_ids):
        """transUP.py:172-180."""
        item_num = len(i_ids)
        u_e = self.user_embeddings(u_id.expand(item_num))
        i_e = self.item_embeddings(i_ids)
        return self.getPreferences(u_e, i_e, use_st_gumbel=self.use_st_gumbel)
