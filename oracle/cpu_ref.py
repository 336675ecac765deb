"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the joint KG + recommender scoring path.

This file is a CPU restatement (modern torch, fp32, CPU tensors; numpy for the integer
ranking walk) of the reference's hot path.  It is NOT the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there
only as the checker / the timed CPU baseline.  The product path (``joint-kg-recommender_amd``)
never imports anything under ``oracle/`` and fails loudly when the HIP library is missing.

Parity pin: every function below is checked against golden vectors produced by importing the
reference's own modules in the build container (``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``,
see ``tests/test_oracle_golden.py``).  The reference holds no tests / known-answer vectors of its own
for this path except the ``ndcg_at_k`` doc examples (jTransUP/utils/evaluation.py:88-96), which are
checked too.  The goldens pin torch-2.10-CPU numbers (the reference is un-versioned torch code).

All ``file:line`` citations are relative to the reference checkout (/root/reference).
Functional style: tables are passed explicitly so the same function serves forward checks,
autograd (gradient oracle) and timing.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS_GUMBEL = 1e-20  # jTransUP/models/transUP.py:159


# ----------------------------------------------------------------------------- projections
def projection_transH(original, norm):
    """jTransUP/utils/misc.py:18-19."""
    return original - torch.sum(original * norm, dim=original.dim() - 1, keepdim=True) * norm


def projection_transR(original, proj_matrix):
    """jTransUP/utils/misc.py:21-26 : (B, d) x (B, d_r*d) -> (B, d_r)."""
    d_e = original.shape[1]
    d_r = proj_matrix.shape[1] // d_e
    return torch.matmul(proj_matrix.view(-1, d_r, d_e), original.view(-1, d_e, 1)).view(-1, d_r)


def projection_transR_batch(original, proj_matrix):
    """jTransUP/utils/misc.py:29-33 : (E, d) x (B, d_r*d) -> (B, E, d_r)."""
    d_e = original.shape[1]
    d_r = proj_matrix.shape[1] // d_e
    return torch.matmul(proj_matrix.view(-1, d_r, d_e), original.transpose(0, 1)).transpose(1, 2)


def _dist(x, l1, dim):
    """'L2' is the SQUARED L2 (no sqrt): transE.py:57-61, transUP.py:78-81."""
    return torch.sum(torch.abs(x), dim) if l1 else torch.sum(x ** 2, dim)


# ----------------------------------------------------------------------------- ST-Gumbel
def gumbel_noise(uniform):
    """transUP.py:159-162 : g = -log(-log(u + eps) + eps)."""
    return -torch.log(-torch.log(uniform + EPS_GUMBEL) + EPS_GUMBEL)


def st_gumbel_softmax(logits, uniform):
    """transUP.py:143-170 / jTransUP.py:288-315 with the uniform draw made an input.

    Forward value is the one-hot of argmax softmax(logits + g); backward flows through the softmax.
    """
    y = F.softmax(logits + gumbel_noise(uniform), dim=logits.dim() - 1)
    y_argmax = y.max(y.dim() - 1)[1]
    y_hard = torch.zeros_like(y).scatter_(y.dim() - 1, y_argmax.unsqueeze(-1), 1.0)
    return (y_hard - y).detach() + y


# ----------------------------------------------------------------------------- score functions (train path)
def score_bprmf(U, I, u, i):
    """bprmf.py:46-49 : row dot product (bmm of 1xd . dx1)."""
    return torch.bmm(U[u].unsqueeze(1), I[i].unsqueeze(2)).reshape(-1)


def score_fm(U, I, ub, ib, bias, u, i):
    """fm.py:58-67 / cofm.py:99-108 : global bias + user bias + item bias + u . i."""
    return bias + ub[u] + ib[i] + (U[u] * I[i]).sum(1)


def eval_fm(U, I, ub, ib, bias, u):
    """fm.py:69-80 / cofm.py:127-141 -> (len(u), n_items)."""
    return bias + ub[u][:, None] + ib[None, :] + U[u] @ I.t()


def score_transe(E, R, h, t, r, l1):
    """transE.py:51-63."""
    return _dist(E[h] + R[r] - E[t], l1, 1)


def score_transh(E, R, N, h, t, r, l1):
    """transH.py:58-71."""
    n_e = N[r]
    return _dist(projection_transH(E[h], n_e) + R[r] - projection_transH(E[t], n_e), l1, 1)


def score_transr(E, R, M, h, t, r, l1):
    """transR.py:65-78."""
    m = M[r]
    return _dist(projection_transR(E[h], m) + R[r] - projection_transR(E[t], m), l1, 1)


def tup_preferences(u_e, i_e, P, Pn, uniform=None):
    """transUP.py:105-115 : raw logits are the mixture weights unless ST-Gumbel is on."""
    pre = torch.matmul(u_e + i_e, P.t()) / 2
    if uniform is not None:
        pre = st_gumbel_softmax(pre, uniform)
    return pre, torch.matmul(pre, P), torch.matmul(pre, Pn)


def ktup_preferences(u_e, ie_e, P, Pn, R, Rn, uniform=None):
    """jTransUP.py:250-260 : summed tables, and the extra /2 on r_e and norm."""
    pre = torch.matmul(u_e + ie_e, (P + R).t()) / 2
    if uniform is not None:
        pre = st_gumbel_softmax(pre, uniform)
    return pre, torch.matmul(pre, P + R) / 2, torch.matmul(pre, Pn + Rn) / 2


def _tup_tail(u_e, i_e, r_e, norm, l1):
    proj_u = projection_transH(u_e, norm)
    proj_i = projection_transH(i_e, norm)
    return _dist(proj_u + r_e - proj_i, l1, u_e.dim() - 1)


def score_tup(U, I, P, Pn, u, i, l1, uniform=None):
    """transUP.py:69-82.  ``uniform`` (B, P) switches ST-Gumbel on."""
    u_e, i_e = U[u], I[i]
    _, r_e, norm = tup_preferences(u_e, i_e, P, Pn, uniform)
    return _tup_tail(u_e, i_e, r_e, norm, l1)


def score_ktup_rec(U, I, E, P, Pn, R, Rn, item2ent, u, i, l1, uniform=None):
    """jTransUP.py:122-143.  ``E`` has the zero pad row last; ``item2ent`` is the int64 table the
    per-item dict walk ``paddingItems`` (jTransUP.py:114-120) evaluates to (pad = rows(E)-1)."""
    u_e = U[u]
    ie_e = I[i] + E[item2ent[i]]
    _, r_e, norm = ktup_preferences(u_e, ie_e, P, Pn, R, Rn, uniform)
    return _tup_tail(u_e, ie_e, r_e, norm, l1)


def score_cke_rec(U, I, E, item2ent, u, i):
    """CKE.py:122-135 : BPRMF on the user row and (item row + aligned entity row); E has the zero pad row last."""
    ie = I[i] + E[item2ent[i]]
    return torch.bmm(U[u].unsqueeze(1), ie.unsqueeze(2)).reshape(-1)


def eval_cke_rec(U, I, E, item2ent, u):
    """CKE.py:142-153 : every item in id order."""
    return torch.matmul(U[u], (I + E[item2ent]).t())


def score_cfkg_rec(U, E, R, u, i, l1):
    """CFKG.py:66-80 : TransE with the extra last relation ("buy") between a user and the item's entity row."""
    return _dist(U[u] + R[R.shape[0] - 1].unsqueeze(0) - E[i], l1, 1)


def eval_cfkg_rec(U, E, R, u, l1):
    """CFKG.py:100-118 : (user + buy) against every row of the shared item / entity table."""
    c = U[u] + R[R.shape[0] - 1].unsqueeze(0)
    return _dist(c.unsqueeze(1) - E.unsqueeze(0), l1, 2)


def score_ktup_kg(E, R, Rn, h, t, r, l1):
    """jTransUP.py:144-157 == TransH on the shared tables."""
    return score_transh(E, R, Rn, h, t, r, l1)


# ----------------------------------------------------------------------------- all-candidate evaluation
def eval_bprmf(U, I, u):
    """bprmf.py:51-54."""
    return torch.matmul(U[u], I.t())


def eval_transe(E, R, q, r, l1, head):
    """transE.py:65-105.  head=True: q is the tail, candidate c = t - r; else c = h + r."""
    c = E[q] - R[r] if head else E[q] + R[r]
    return _dist(c.unsqueeze(1) - E.unsqueeze(0), l1, 2)


def eval_transh(E, R, N, q, r, l1, head):
    """transH.py:73-121 and jTransUP.py:193-247 (there E includes the pad row, which is ranked)."""
    n_e = N[r]
    pq = projection_transH(E[q], n_e)
    c = pq - R[r] if head else pq + R[r]
    proj_ent = projection_transH(E.unsqueeze(0).expand(len(q), -1, -1), n_e.unsqueeze(1).expand(-1, E.shape[0], -1))
    return _dist(c.unsqueeze(1) - proj_ent, l1, 2)


def eval_transr(E, R, M, q, r, l1, head):
    """transR.py:80-128."""
    m = M[r]
    pq = projection_transR(E[q], m)
    c = pq - R[r] if head else pq + R[r]
    return _dist(c.unsqueeze(1) - projection_transR_batch(E, m), l1, 2)


def eval_tup(U, I, P, Pn, u, l1, uniform=None):
    """transUP.py:84-102.  ``uniform`` is (B, n_items, P) when ST-Gumbel is on (eval is stochastic too)."""
    B, N, d = len(u), I.shape[0], I.shape[1]
    u_e = U[u].unsqueeze(1).expand(B, N, d)
    i_e = I.unsqueeze(0).expand(B, N, d)
    _, r_e, norm = tup_preferences(u_e, i_e, P, Pn, uniform)
    return _tup_tail(u_e, i_e, r_e, norm, l1)


def eval_ktup_rec(U, I, E, P, Pn, R, Rn, item2ent, u, l1, uniform=None):
    """jTransUP.py:163-191 with all_i_ids=None (not sharing): every item id 0..n_items-1 in order."""
    B, N, d = len(u), I.shape[0], I.shape[1]
    u_e = U[u].unsqueeze(1).expand(B, N, d)
    ie = (I + E[item2ent]).unsqueeze(0).expand(B, N, d)
    _, r_e, norm = ktup_preferences(u_e, ie, P, Pn, R, Rn, uniform)
    return _tup_tail(u_e, ie, r_e, norm, l1)


# ----------------------------------------------------------------------------- losses / regularisers
def bpr_loss(pos, neg, target=1.0):
    """utils/loss.py:29-31 (a MEAN)."""
    return (-F.logsigmoid(target * (pos - neg))).mean()


def margin_loss(pos, neg, margin):
    """utils/loss.py:8-16 (a SUM)."""
    return torch.sum(torch.clamp(pos - neg + margin, min=0.0))


def orthogonal_loss(rel, norm):
    """utils/loss.py:18-19."""
    return torch.sum(torch.sum(norm * rel, dim=1, keepdim=True) ** 2 / torch.sum(rel ** 2, dim=1, keepdim=True))


def norm_loss(emb, dim=1):
    """utils/loss.py:21-23."""
    return torch.sum(torch.clamp(torch.sum(emb ** 2, dim=dim, keepdim=True) - 1.0, min=0.0))


# ----------------------------------------------------------------------------- the drivers' step bodies (loss assembly + one step)
def ktup_rec_step_loss(U, I, E, P, Pn, R, Rn, item2ent, u, pi, ni, l1=False, target=-1.0, uni_pos=None, uni_neg=None):
    """models/knowledgable_recommendation.py:335-344 : bprLoss(pos, neg, target) + orthogonalLoss(pref, pref_norm)."""
    pos = score_ktup_rec(U, I, E, P, Pn, R, Rn, item2ent, u, pi, l1, uni_pos)
    neg = score_ktup_rec(U, I, E, P, Pn, R, Rn, item2ent, u, ni, l1, uni_neg)
    return bpr_loss(pos, neg, target) + orthogonal_loss(P, Pn)


def kg_step_loss(E, R, N, ph, pt, pr, nh, nt, nr, l1=False, margin=1.0, kg_lambda=1.0):
    """models/knowledge_representation.py:189-204 (TransE: N is None; TransH) and knowledgable_recommendation.py:368-383 (KTUP's
    kg branch = TransH on its own tables, times kg_lambda): marginLoss + orthogonalLoss(rel rows, norm rows) [TransH] +
    normLoss(ent rows of ph, pt, nh, nt) + normLoss(rel rows of pr, nr)."""
    if N is None:
        pos, neg = score_transe(E, R, ph, pt, pr, l1), score_transe(E, R, nh, nt, nr, l1)
    else:
        pos, neg = score_transh(E, R, N, ph, pt, pr, l1), score_transh(E, R, N, nh, nt, nr, l1)
    loss = margin_loss(pos, neg, margin)
    rel_ids = torch.cat([pr, nr])
    if N is not None:
        loss = loss + orthogonal_loss(R[rel_ids], N[rel_ids])
    loss = loss + norm_loss(E[torch.cat([ph, pt, nh, nt])]) + norm_loss(R[rel_ids])
    return kg_lambda * loss


def tup_rec_step_loss(U, I, P, Pn, u, pi, ni, l1=False, target=-1.0, uni_pos=None, uni_neg=None):
    """models/item_recommendation.py:171-180 (transup): bprLoss + orthogonalLoss(pref, pref_norm) + normLoss(user rows) +
    normLoss(item rows of pos and neg) + normLoss(pref)."""
    pos, neg = score_tup(U, I, P, Pn, u, pi, l1, uni_pos), score_tup(U, I, P, Pn, u, ni, l1, uni_neg)
    return bpr_loss(pos, neg, target) + orthogonal_loss(P, Pn) + norm_loss(U[u]) + norm_loss(I[torch.cat([pi, ni])]) + norm_loss(P)


def make_optimizer(params, optimizer_type, lr, l2_lambda, momentum=0.9):
    """utils/trainer.py:63-77 : torch.optim with weight_decay = l2_lambda (dense: every row of every table moves)."""
    if optimizer_type == 'Adam':
        return torch.optim.Adam(params, lr=lr, weight_decay=l2_lambda)
    if optimizer_type == 'SGD':
        return torch.optim.SGD(params, lr=lr, weight_decay=l2_lambda, momentum=momentum)
    if optimizer_type == 'Adagrad':
        return torch.optim.Adagrad(params, lr=lr, weight_decay=l2_lambda)
    if optimizer_type == 'Rmsprop':
        return torch.optim.RMSprop(params, lr=lr, weight_decay=l2_lambda, momentum=momentum)
    raise ValueError(optimizer_type)


def train_step(params, optimizer, loss_fn, clip_max, pad_row_of=None):
    """One step as every driver ends it (e.g. knowledgable_recommendation.py:394-403): zero_grad, loss, backward,
    clip_grad_norm over ALL parameters, optimizer.step -- with the zero-filling zero_grad of the torch 0.3 the reference targets
    (the goldens are generated with the same shim, tests/golden/make_goldens.py #5).  pad_row_of: the entity table whose last row is nn.Embedding's
    padding_idx (jTransUP.py:96) -- autograd on a plain tensor would give it a gradient, the reference's Embedding does not.
    -> (loss value, pre-clip global gradient norm)."""
    optimizer.zero_grad(set_to_none=False)     # torch 0.3's zero_grad zero-FILLS: a table that has had a gradient once keeps being
    loss = loss_fn()                            # updated (weight decay, Adam moments) on steps that do not touch it
    loss.backward()
    if pad_row_of is not None and pad_row_of.grad is not None:
        pad_row_of.grad[-1].zero_()
    norm = torch.nn.utils.clip_grad_norm_(params, clip_max)
    optimizer.step()
    return float(loss.detach()), float(norm)


# ----------------------------------------------------------------------------- ranking walk (integer work -> numpy)
def dcg_at_k(r, k, method=1):
    """utils/evaluation.py:41-77 (np.asfarray replaced by its definition)."""
    r = np.asarray(r, dtype=np.float64)[:k]
    if r.size:
        if method == 0:
            return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
        elif method == 1:
            return np.sum(r / np.log2(np.arange(2, r.size + 2)))
        raise ValueError('method must be 0 or 1.')
    return 0.


def ndcg_at_k(r, k, method=0):
    """utils/evaluation.py:80-110."""
    dcg_max = dcg_at_k(sorted(r, reverse=True), k, method)
    if not dcg_max:
        return 0.
    return dcg_at_k(r, k, method) / dcg_max


def argsort_tiebreak(pred):
    """np.argsort(pred) as in utils/misc.py:127,215 with the tie rule made explicit:
    ascending score, then ascending id (kind='stable').  The reference's default sort kind is
    not stable, so its order on exact ties is unspecified; this is the declared rule."""
    return np.argsort(np.asarray(pred), kind='stable')


def rec_performance(pred, gold, filter_samples=None, topn=10):
    """utils/misc.py:213-248 (getRecPerformance).  Lower ``pred`` = better."""
    hits, top_ids, current_rank = [], [], 0
    for rank_id in argsort_tiebreak(pred):
        rank_id = int(rank_id)
        if filter_samples is not None and rank_id in filter_samples:
            continue
        hits.append(1 if rank_id in gold else 0)
        top_ids.append(rank_id)
        current_rank += 1
        if current_rank >= topn:
            break
    hits_count = sum(hits)
    k, k_gold = len(hits), len(gold)
    f1 = p = r = ndcg = 0.0
    hit = 1 if hits_count > 0 else 0
    if hits_count > 0:
        p = float(hits_count) / k
        r = float(hits_count) / k_gold
        f1 = 2 * p * r / (p + r)
        ndcg = ndcg_at_k(hits, k)
    return f1, p, r, hit, ndcg, top_ids


def kg_performance(pred, gold, filter_samples=None, topn=10):
    """utils/misc.py:125-146 (getKGPerformance): 0-based filtered rank per gold id; other golds do
    not advance the rank; stops once every gold is found."""
    gold_ranks, hits, gold_ids, current_rank = [], [], [], 0
    for rank_id in argsort_tiebreak(pred):
        rank_id = int(rank_id)
        if filter_samples is not None and rank_id in filter_samples:
            continue
        if rank_id in gold:
            gold_ranks.append(current_rank)
            gold_ids.append(rank_id)
            hits.append(1 if current_rank < topn else 0)
            if len(gold_ranks) == len(gold):
                break
        else:
            current_rank += 1
    return hits, gold_ranks, gold_ids


def mrr_from_ranks(gold_ranks):
    """Mean reciprocal rank over the 0-based filtered ranks of utils/misc.py:134-144: mean of 1 / (rank + 1).  The reference
    never computes MRR (its KG summary is hit@n and mean rank, knowledge_representation.py:49-63); BASELINE.json's north_star
    asks for it, so it is DEFINED here on the reference's own rank lists."""
    ranks = np.asarray(list(gold_ranks), dtype=np.float64)
    return float((1.0 / (ranks + 1.0)).mean()) if ranks.size else 0.0


def _filter_union(key, all_dicts):
    """utils/misc.py:83-89,168-174."""
    if all_dicts is None:
        return None
    s = set()
    for dic in all_dicts:
        if key in dic:
            s.update(dic[key])
    return s


def eval_rec_rows(pred_scores, eval_dict, all_dicts=None, descending=True, topn=10):
    """Serial equivalent of utils/misc.py:186-210 (evalRecProcess) -- same rows, deterministic order."""
    out = []
    for key, row in pred_scores:
        if key not in eval_dict:
            continue
        gold = eval_dict[key]
        per = np.asarray(row) if not descending else -np.asarray(row)
        f1, p, r, hit, ndcg, top_ids = rec_performance(per, gold, _filter_union(key, all_dicts), topn)
        out.append([f1, p, r, hit, ndcg, (key, top_ids, gold)])
    return out


def eval_kg_rows(pred_scores, eval_dict, all_dicts=None, descending=True, topn=10):
    """Serial equivalent of utils/misc.py:98-122 (evalKGProcess)."""
    out = []
    for key, row in pred_scores:
        if key not in eval_dict:
            continue
        gold = eval_dict[key]
        per = np.asarray(row) if not descending else -np.asarray(row)
        hits, ranks, ids = kg_performance(per, gold, _filter_union(key, all_dicts), topn)
        out.extend(list(zip(hits, ranks, [key] * len(hits), ids)))
    return out


# ----------------------------------------------------------------------------- joint schedule / alignment
def is_rec_step(step, joint_ratio):
    """models/knowledgable_recommendation.py:209,320 : rec step iff step % 10 < 10 * joint_ratio."""
    return step % 10 < 10 * joint_ratio


def rebuild_entity_item_vocab(map1, map2, links):
    """data/load_kg_rating_data.py:21-48."""
    new_map, index, has_map2, remap1 = {}, 0, {}, {}
    for org1 in map1:
        mapped2 = -1
        if org1 in links:
            org2 = links[org1]
            if org2 in map2:
                mapped2 = map2[org2]
                has_map2[org2] = index
        new_map[index] = (map1[org1], mapped2)
        remap1[map1[org1]] = index
        index += 1
    remap2 = {}
    for org2 in map2:
        if org2 in has_map2:
            remap2[map2[org2]] = has_map2[org2]
            continue
        new_map[index] = (-1, map2[org2])
        remap2[map2[org2]] = index
        index += 1
    return new_map, remap1, remap2, len(has_map2)


def item2ent_table(item_total, i_map, new_map, pad_index):
    """The table jTransUP.py:114-120 (paddingItems) evaluates to, for item ids 0..item_total-1."""
    out = []
    for i_id in range(item_total):
        ent_id = new_map[i_map[i_id]][0] if i_id in i_map else -1
        out.append(ent_id if ent_id != -1 else pad_index)
    return torch.tensor(out, dtype=torch.int64)


# ----------------------------------------------------------------------------- table construction (ctor semantics)
def make_table(rows, d, gen, normalize=True):
    """Ctor recipe (e.g. transUP.py:37-62): xavier_uniform then row-L2 normalise."""
    bound = math.sqrt(6.0 / (rows + d))
    w = (torch.rand(rows, d, generator=gen, dtype=torch.float32) * 2 - 1) * bound
    return F.normalize(w, p=2, dim=1) if normalize else w
