"""Host metric arithmetic with the reference's names (jTransUP/utils/evaluation.py:41-110): float64, a handful of
values per user -- it stays on the host; the ranking that feeds it runs on the device (utils/ranking.py)."""
import numpy as np


def _discounts(n, method):
    if method == 0:      # weights 1, 1, 1/log2(3), 1/log2(4), ...   (evaluation.py:70-71)
        w = np.ones(n, dtype=np.float64)
        w[1:] = 1.0 / np.log2(np.arange(2, n + 1))
        return w
    if method == 1:      # weights 1, 1/log2(3), ...                 (evaluation.py:72-73)
        return 1.0 / np.log2(np.arange(2, n + 2))
    raise ValueError('method must be 0 or 1.')


def dcg_at_k(r, k, method=1):
    """Discounted cumulative gain of relevance list `r` cut at k (default method=1, evaluation.py:41)."""
    rel = np.asarray(r, dtype=np.float64)[:k]
    if rel.size == 0:
        return 0.
    if method == 0:      # keep the reference's summation order: r[0] + sum(rest)
        return rel[0] + np.sum(rel[1:] / np.log2(np.arange(2, rel.size + 1)))
    return np.sum(rel * _discounts(rel.size, method))


def ndcg_at_k(r, k, method=0):
    """DCG normalised by the DCG of the same relevances sorted descending (evaluation.py:80-110; default method=0).
    The ideal is built from the OBSERVED hits, not from the size of the gold set (misc.py:246)."""
    best = dcg_at_k(sorted(r, reverse=True), k, method)
    return dcg_at_k(r, k, method) / best if best else 0.


def get_performance(recommend_list, purchased_list):
    """evaluation.py:8-39: (f1, precision, recall, hit, ndcg) of one ranked list."""
    rank_list = [1 if i in purchased_list else 0 for i in recommend_list]
    hit_number, k = sum(rank_list), len(rank_list)
    if hit_number == 0:
        return 0.0, 0.0, 0.0, 0, 0.0
    p = float(hit_number) / k
    r = float(hit_number) / len(purchased_list)
    return 2 * p * r / (p + r), p, r, 1, ndcg_at_k(rank_list, k)


def evalAll(recommend_list, purchased_list):
    """evaluation.py:112-128: mean of get_performance over users."""
    assert len(recommend_list) == len(purchased_list), "Eval user number not match!"
    perf = np.array([get_performance(a, b) for a, b in zip(recommend_list, purchased_list)]).mean(axis=0)
    return perf[0], perf[1], perf[2], perf[3], perf[4]
