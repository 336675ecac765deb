"""Negative samplers and batch iterators with the reference's names and contracts (jTransUP/utils/data.py).

These are the HOST samplers (python `random`, dict/set filters) used by the drop-in drivers so that a seeded run draws
the same negatives as the reference.  The on-device samplers (K19, same constraints, Philox draws) live in
jTransUP/hip/ops.py and are used by the GPU-resident training loop."""
import random

import numpy as np


def _known(key, member, dicts):
    return dicts is not None and any(key in d and member in d[key] for d in dicts)


def corrupt_head_filter(triple, entityTotal, headDicts=None):
    """data.py:20-36: uniform head != original that is not a known true head of (t, r)."""
    h, t, r = triple
    while True:
        new_head = random.randrange(entityTotal)
        if new_head != h and not _known((t, r), new_head, headDicts):
            return (new_head, t, r)


def corrupt_tail_filter(triple, entityTotal, tailDicts=None):
    """data.py:40-56."""
    h, t, r = triple
    while True:
        new_tail = random.randrange(entityTotal)
        if new_tail != t and not _known((h, r), new_tail, tailDicts):
            return (h, new_tail, r)


def getTripleElements(tripleList):
    return [x[0] for x in tripleList], [x[1] for x in tripleList], [x[2] for x in tripleList]


def getTrainTripleBatch(triple_batch, entity_total, all_head_dicts=None, all_tail_dicts=None):
    """data.py:12-18: per triple a fair coin picks head or tail corruption (the coin is drawn BEFORE the corruption,
    in list order, which fixes the sequence of `random` draws)."""
    neg = [corrupt_head_filter(tr, entity_total, headDicts=all_head_dicts) if random.random() < 0.5
           else corrupt_tail_filter(tr, entity_total, tailDicts=all_tail_dicts) for tr in triple_batch]
    ph, pt, pr = getTripleElements(triple_batch)
    nh, nt, nr = getTripleElements(neg)
    return ph, pt, pr, nh, nt, nr


_SEEN = []          # [(the dicts themselves, {user: union of the user's items over them})], most recent first, at most 4


def _seen_sets(all_dicts):
    """The per-user union of the rating dicts, built once per user and kept while the caller keeps handing over the SAME dict
    objects (the drivers do: one list for the whole run).  The reference rebuilds the union for every rating of every batch
    (data.py:70-75) -- 5.6 ms of a 512-rating batch; the membership tests and hence the draws are the same."""
    for k, (dicts, seen) in enumerate(_SEEN):
        if len(dicts) == len(all_dicts) and all(a is b for a, b in zip(dicts, all_dicts)):
            if k:
                _SEEN.insert(0, _SEEN.pop(k))
            return seen
    _SEEN.insert(0, (list(all_dicts), {}))
    del _SEEN[4:]
    return _SEEN[0][1]


def getNegRatings(ratingList, itemTotal, all_dicts=None):
    """data.py:64-85: one negative per rating: != the positive, not rated by the user in any split, and not already
    used as a negative in this batch.  (Like the reference it requires all_dicts; `None` raises TypeError.)"""
    ni, used = [], set()
    cache = _seen_sets(all_dicts) if all_dicts is not None else None
    for rating in ratingList:
        user, old_item = rating[0], rating[1]
        seen = None
        if cache is not None:
            seen = cache.get(user)
            if seen is None:
                seen = set()
                for dic in all_dicts:
                    if user in dic:
                        seen.update(dic[user])
                cache[user] = seen
        while True:
            cand = random.randrange(itemTotal)
            if cand != old_item and cand not in seen and cand not in used:
                break
        ni.append(cand)
        used.add(cand)
    return [r[0] for r in ratingList], [r[1] for r in ratingList], ni


def getTrainRatingBatch(rating_batch, item_total, all_dicts=None):
    """data.py:5-10: (users, positive items, one sampled negative each) of a batch of ratings.  The reference's body calls an
    `addNegRatings` that its module does not define (the drivers call getNegRatings themselves); this one returns what that
    call was meant to: getNegRatings of the batch."""
    batch = rating_batch.tolist() if hasattr(rating_batch, 'tolist') else list(rating_batch)
    return getNegRatings(batch, item_total, all_dicts=all_dicts)


def MakeTrainIterator(train_data, batch_size, negtive_samples=1):
    """data.py:87-110: endless iterator; shuffles an index list each epoch and DROPS the tail partial batch."""
    train_list = np.array(train_data)

    def data_iter():
        n = len(train_list)
        order = list(range(n)) * negtive_samples
        random.shuffle(order)                                   # (python's shuffle of a list, like the reference: same permutation)
        order_np = np.asarray(order)                            # a batch's rows are then one array slice, not a list -> array copy
        start = -batch_size
        while True:
            start += batch_size
            if start > n - batch_size:
                start = 0
                random.shuffle(order)
                order_np = np.asarray(order)
            yield train_list[order_np[start:start + batch_size]].tolist()

    return data_iter()


def MakeEvalIterator(eval_data, data_type, batch_size):
    """data.py:112-133: list of batches in order; the tail partial batch is KEPT."""
    eval_list = np.asarray(eval_data, data_type)
    return [eval_list[s:s + batch_size].tolist() for s in range(0, len(eval_list), batch_size)]
