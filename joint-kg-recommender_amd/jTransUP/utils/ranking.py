"""Evaluation ranking with the reference's entry points (jTransUP/utils/misc.py:61-248), run on the device.

The reference copies every (B x N) score matrix to the host, pickles the rows to `num_processes` freshly spawned
worker processes and walks a full np.argsort per row in python.  Here the filtered top-n ids (K17) and the filtered
gold ranks (K18) come from HIP kernels (jTransUP/hip/ops.py); only ids / ranks (a few KB) cross to the host, where
the metric arithmetic (hit, precision, recall, F1, NDCG -- float64, exactly the reference's formulas) stays.

Tie rule (declared; the reference's np.argsort default kind is not stable): ascending score, then ascending id.
"""
import numpy as np
import torch

from jTransUP.hip import ops
from jTransUP.utils.evaluation import ndcg_at_k


def _union(key, all_dicts):
    """utils/misc.py:83-89 / 168-174."""
    if all_dicts is None:
        return None
    s = set()
    for dic in all_dicts:
        if key in dic:
            s.update(dic[key])
    return s


class RankIndex(object):
    """CSR filter / gold sets for an ordered list of evaluation keys, resident on the device.

    Built once per evaluation pass; a batch that is a contiguous slice of `keys` (what MakeEvalIterator yields,
    utils/data.py:112-133) is served by zero-copy views."""

    def __init__(self, keys, eval_dict, all_dicts, device):
        self.keys = [k if not isinstance(k, list) else tuple(k) for k in keys]
        self.pos = {k: i for i, k in enumerate(self.keys)}
        f_off, f_ids, g_off, g_ids = [0], [], [0], []
        self.gold_sets, self.present = [], []
        for k in self.keys:
            filt = _union(k, all_dicts)
            if filt:
                f_ids.extend(sorted(filt))
            f_off.append(len(f_ids))
            gold = eval_dict.get(k)
            self.present.append(gold is not None)
            gold = gold if gold is not None else set()
            self.gold_sets.append(gold)
            g_ids.extend(sorted(gold))
            g_off.append(len(g_ids))
        self.has_filter = all_dicts is not None
        self.f_off_h = np.asarray(f_off, dtype=np.int64)
        self.g_off_h = np.asarray(g_off, dtype=np.int64)
        self.g_ids_h = np.asarray(g_ids, dtype=np.int32)
        self.f_off = torch.from_numpy(self.f_off_h).to(device)
        self.g_off = torch.from_numpy(self.g_off_h).to(device)
        self.f_ids = torch.from_numpy(np.asarray(f_ids if f_ids else [0], dtype=np.int32)).to(device)
        self.g_ids = torch.from_numpy(self.g_ids_h if len(g_ids) else np.zeros(1, np.int32)).to(device)
        # sorted (row << 32 | id) keys of the gold pairs: membership of a whole batch of top-n lists is one searchsorted
        rows = np.repeat(np.arange(len(self.keys), dtype=np.int64), np.diff(self.g_off_h))
        self.g_keys_h = (rows << 32) | self.g_ids_h.astype(np.int64)
        self.present_h = np.asarray(self.present, dtype=bool)
        self.all_present = bool(self.present_h.all())
        self._memo, self._fslice, self._gslice = {}, {}, {}

    # The batches of an evaluation iterator are the same objects in every pass, so what is derived from one (its row range,
    # its rebased CSR offsets) is memoised: per batch the host then does a couple of dict look-ups, not 512 of them plus two
    # small device kernels -- at ml1m size the device part of a pass is ~1 ms and this bookkeeping used to be as much again.
    def rows_of(self, batch_keys):
        memo = self._memo.get(id(batch_keys))
        if memo is not None and memo[0] is batch_keys:
            return memo[1]
        rows = [self.pos[k if not isinstance(k, list) else tuple(k)] for k in batch_keys]
        if rows != list(range(rows[0], rows[0] + len(rows))):
            raise KeyError('batch is not a contiguous slice of the indexed keys')
        span = (rows[0], rows[0] + len(rows))
        if len(self._memo) >= 4096:     # the batches of a pass are a handful of objects; never grow without bound
            self._memo.clear()
        self._memo[id(batch_keys)] = (batch_keys, span)
        return span

    def filter_slice(self, s, e):
        if not self.has_filter:
            return None, None
        hit = self._fslice.get((s, e))
        if hit is None:
            lo = int(self.f_off_h[s])
            hit = self._fslice[(s, e)] = (self.f_off[s:e + 1] - lo, self.f_ids[lo:])
        return hit

    def gold_slice(self, s, e):
        hit = self._gslice.get((s, e))
        if hit is None:
            lo = int(self.g_off_h[s])
            hit = self._gslice[(s, e)] = (self.g_off[s:e + 1] - lo, self.g_ids[lo:], self.g_off_h[s:e + 1] - lo,
                                          self.g_ids_h[lo:int(self.g_off_h[e])])
        return hit


def rec_metrics(top_ids, gold):
    """utils/misc.py:232-248: (f1, p, r, hit, ndcg) from the ranked unfiltered ids and the gold set."""
    hits = [1 if i in gold else 0 for i in top_ids]
    hits_count = sum(hits)
    k, k_gold = len(hits), len(gold)
    f1 = p = r = ndcg = 0.0
    hit = 1 if hits_count > 0 else 0
    if hits_count > 0:
        p = float(hits_count) / k
        r = float(hits_count) / k_gold
        f1 = 2 * p * r / (p + r)
        ndcg = ndcg_at_k(hits, k)
    return f1, p, r, hit, ndcg


def _as_device_rows(pred_scores, device):
    """Accept the reference's list of (key, numpy row) as well as (keys, device matrix)."""
    if isinstance(pred_scores, tuple) and len(pred_scores) == 2 and torch.is_tensor(pred_scores[1]):
        keys, mat = pred_scores
        return (keys if isinstance(keys, list) else list(keys)), mat    # the SAME list object: RankIndex.rows_of memoises on it
    keys = [k for k, _ in pred_scores]
    rows = [r if torch.is_tensor(r) else torch.from_numpy(np.ascontiguousarray(r, dtype=np.float32)) for _, r in pred_scores]
    return keys, torch.stack([r.to(device) for r in rows]).contiguous()


def _device():
    return torch.device('cuda', torch.cuda.current_device())


def evalRecProcess(pred_scores, eval_dict, all_dicts=None, descending=True, num_processes=None, topn=10, queue_limit=10,
                   index=None, as_array=False):
    """utils/misc.py:186-210.  Returns [[f1, p, r, hit, ndcg, (key, top_ids, gold)], ...] for the keys found in
    eval_dict.  `num_processes` / `queue_limit` are accepted and ignored (there is no process fan-out);
    `index` (a RankIndex over the pass's keys) avoids rebuilding the CSR sets per batch.  `as_array=True` (this build) returns
    only the (n x 5) float64 metric columns -- what the training-time evaluation needs -- without building per-user rows;
    `as_array='device'` leaves them on the device, one row per key of the batch (no host sync)."""
    keys, mat = _as_device_rows(pred_scores, _device())
    if len(keys) == 0:
        return []
    if index is None:
        index = RankIndex(keys, eval_dict, all_dicts, mat.device)
    s, e = index.rows_of(keys)
    f_off, f_ids = index.filter_slice(s, e)
    top = ops.topk_filtered(mat, descending, topn, f_off, f_ids)
    if as_array:                                  # metrics on the device too (K18b): nothing but 5 doubles per key leaves it
        g_off, g_ids = index.gold_slice(s, e)[:2]
        cols = ops.rec_metrics(top, g_off, g_ids)
        if as_array == 'device':                  # no host sync; the caller drops keys absent from eval_dict (index.present_h)
            return cols
        return cols.cpu().numpy()[index.present_h[s:e]]
    top = top.cpu().numpy().astype(np.int64)
    f1, p, r, hit, ndcg = rec_metrics_batch(top, index, s)
    out = []
    cols = zip(f1.tolist(), p.tolist(), r.tolist(), hit.tolist(), ndcg.tolist(), top.tolist(), keys, index.gold_sets[s:e],
               index.present_h[s:e].tolist())
    for a, b, c, d, g, ids, key, gold, here in cols:
        if here:
            out.append([a, b, c, d, g, (key, [i for i in ids if i >= 0], gold)])
    return out


def rec_metrics_batch(top, index, s):
    """utils/misc.py:232-248 + evaluation.py:41-110 (NDCG method 0, ideal from the OBSERVED hits) for a whole batch of ranked
    lists at once: float64 arrays (f1, p, r, hit, ndcg).  `top` is (B x topn) int64, -1 padded; rows are index rows s..s+B-1."""
    B, topn = top.shape
    valid = top >= 0
    tk = ((np.arange(s, s + B, dtype=np.int64)[:, None]) << 32) | np.where(valid, top, 0)
    pos = np.searchsorted(index.g_keys_h, tk)
    found = index.g_keys_h[np.minimum(pos, max(len(index.g_keys_h) - 1, 0))] == tk if len(index.g_keys_h) else np.zeros_like(valid)
    hits = (found & valid).astype(np.float64)
    k = valid.sum(1).astype(np.float64)
    hc = hits.sum(1)
    k_gold = (index.g_off_h[s + 1:s + B + 1] - index.g_off_h[s:s + B]).astype(np.float64)
    some = hc > 0
    with np.errstate(divide='ignore', invalid='ignore'):
        p = np.where(some, hc / k, 0.0)
        r = np.where(some, hc / k_gold, 0.0)
        f1 = np.where(some, 2 * p * r / (p + r), 0.0)
        w = np.ones(topn, dtype=np.float64)
        if topn > 1:
            w[1:] = 1.0 / np.log2(np.arange(2, topn + 1))
        dcg = hits[:, 0] + (hits[:, 1:] * w[1:]).sum(1)
        ideal_by_count = np.concatenate([[0.0], w[0] + np.concatenate([[0.0], np.cumsum(w[1:])])])   # ideal DCG of c hits, c = 0..topn
        ideal = ideal_by_count[hc.astype(np.int64)]
        ndcg = np.where(some, dcg / ideal, 0.0)
    return f1, p, r, some.astype(np.int64), ndcg


def evalKGProcess(pred_scores, eval_dict, all_dicts=None, descending=True, num_processes=None, topn=10, queue_limit=10,
                  index=None, as_array=False):
    """utils/misc.py:98-122.  Returns [(hit, rank, key, gold_id), ...] (0-based filtered ranks).  `as_array='device'` (this
    build) returns the int32 device vector of ranks instead, one per gold entry of the batch and -1 where the walk never
    reaches the gold id -- what the training-time evaluation needs for its two means, without a host sync per batch."""
    keys, mat = _as_device_rows(pred_scores, _device())
    if len(keys) == 0:
        return []
    if index is None:
        index = RankIndex(keys, eval_dict, all_dicts, mat.device)
    s, e = index.rows_of(keys)
    f_off, f_ids = index.filter_slice(s, e)
    g_off, g_ids, g_off_h, g_ids_h = index.gold_slice(s, e)
    if len(g_ids_h) == 0:
        return []
    ranks = ops.gold_ranks(mat, descending, g_off, g_ids, f_off, f_ids)
    if as_array == 'device':
        return ranks[:int(g_off_h[-1])]          # g_ids is an open-ended view of the index: only this batch's entries were ranked
    ranks = ranks.cpu().numpy()
    out = []
    for b, key in enumerate(keys):
        lo, hi = int(g_off_h[b]), int(g_off_h[b + 1])
        rows = sorted((int(ranks[i]), int(g_ids_h[i])) for i in range(lo, hi) if ranks[i] >= 0)
        out.extend((1 if rk < topn else 0, rk, key, gid) for rk, gid in rows)
    return out
