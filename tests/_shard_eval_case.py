"""Shared body of the sharded-candidate evaluation tests (third row): every rank holds the scores of one contiguous
slice of the catalogue; parallel.sharded_topk / sharded_gold_ranks must reproduce the single-process results -- the golden
ranked id lists / ranks the reference produced (tests/golden/ranking.*) and a larger seeded case with ties -- on every rank.
Runs on CPU with numpy stand-ins for the two local kernels (gloo, world 2) and on the GPU with the HIP kernels."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _key(score, descending):
    s = -score if descending else score
    return s + 0.0                                         # -0.0 -> +0.0, like the kernels' keys


def np_local_topk(local_scores, descending, topn, f_off, f_ids_local):
    """Stand-in of ops.topk_filtered on CPU tensors: filtered (score, id)-ordered top-n of each row, -1 padded."""
    sc = local_scores.numpy()
    nq, n = sc.shape
    ids = np.full((nq, topn), -1, np.int32)
    out = np.zeros((nq, topn), np.float32)
    for b in range(nq):
        filt = set() if f_off is None else set(f_ids_local[int(f_off[b]):int(f_off[b + 1])].tolist())
        cand = [j for j in range(n) if j not in filt]
        cand.sort(key=lambda j: (_key(sc[b, j], descending), j))
        for k, j in enumerate(cand[:topn]):
            ids[b, k] = j
            out[b, k] = sc[b, j]
    return torch.from_numpy(ids), torch.from_numpy(out)


def np_local_counts(local_scores, lo, descending, g_off, g_ids, gold_scores, f_off, f_ids, stride=1):
    """Stand-in of ops.gold_rank_counts: per gold entry the shard's unfiltered non-gold candidates ordered before it (local candidate j
    has the global id lo + stride * j)."""
    sc = local_scores.numpy()
    nq, n = sc.shape
    counts = np.zeros(len(g_ids), np.int32)
    for b in range(nq):
        filt = set() if f_off is None else set(f_ids[int(f_off[b]):int(f_off[b + 1])].tolist())
        golds = g_ids[int(g_off[b]):int(g_off[b + 1])].tolist()
        for e in range(int(g_off[b]), int(g_off[b + 1])):
            g = int(g_ids[e])
            if g in filt:
                counts[e] = -(1 << 20)
                continue
            gk = (_key(float(gold_scores[e]), descending), g)
            counts[e] = sum(1 for j in range(n) if (lo + stride * j) not in filt and (lo + stride * j) not in golds
                            and (_key(sc[b, j], descending), lo + stride * j) < gk)
    return torch.from_numpy(counts)


def _csr(lists, dtype=np.int32):
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    ids = np.concatenate([np.asarray(sorted(x), dtype=dtype) for x in lists]) if lists else np.zeros(0, dtype)
    return torch.from_numpy(off), torch.from_numpy(ids.astype(dtype))


def run(rank, world, device, local_topk=None, local_counts=None, group=None, layout='block'):
    """layout 'block': a rank holds the contiguous slice shard_bounds gives it (-shard_eval_candidates: whole tables on every rank);
    'lattice': candidates rank, rank + world, ... (-shard_tables: the rows of a table sharded by row % world are a rank's candidates)."""
    from jTransUP.parallel import shard_bounds, sharded_gold_ranks, sharded_topk

    def part(mat):                                         # -> (this rank's columns, global id of local column 0, extra keyword arguments)
        if layout == 'lattice':
            return mat[:, rank::world].copy(), rank, {'stride': world}
        lo, hi = shard_bounds(mat.shape[1], rank, world)
        return mat[:, lo:hi].copy(), lo, {}
    g = dict(np.load(os.path.join(GOLDEN, 'ranking.npz')))
    J = json.load(open(os.path.join(GOLDEN, 'ranking.json')))
    dev = torch.device(device)
    # ---- rec: the reference's ranked id lists (utils/misc.py:213-248), candidates split over the ranks
    rows, filts, want, descs = [], [], [], []
    nrec = g['rec.rows'].shape[0]
    for b, c in enumerate(J['rec']):
        desc = c.get('descending', False)
        rows.append(g['rec.bprmf_rows'][b - nrec] if desc else g['rec.rows'][b])
        filts.append(c['filter'] or [])
        want.append(c['top_ids']); descs.append(desc)
    for desc in (False, True):
        sel = [i for i, d in enumerate(descs) if d == desc]
        if not sel:
            continue
        mat = np.stack([rows[i] for i in sel]).astype(np.float32)
        loc, lo, kw = part(mat)
        f_off, f_ids = _csr([filts[i] for i in sel])
        ids, _ = sharded_topk(torch.from_numpy(loc).to(dev), lo, 10, desc, f_off.to(dev), f_ids.to(dev), group=group,
                              local_topk=local_topk, **kw)
        for k, i in enumerate(sel):
            got = [x for x in ids[k].tolist() if x >= 0]
            assert got == want[i], (rank, i, got, want[i])
    # ---- kg: the reference's 0-based filtered ranks (utils/misc.py:125-146)
    mat = g['kg.rows'].astype(np.float32)
    loc, lo, kw = part(mat)
    golds = [c['gold'] for c in J['kg']]
    f_off, f_ids = _csr([c['filter'] for c in J['kg']])
    g_off, g_ids = _csr(golds)
    g_rows = torch.from_numpy(np.repeat(np.arange(len(golds)), [len(x) for x in golds]).astype(np.int64))
    ranks = sharded_gold_ranks(torch.from_numpy(loc).to(dev), lo, False, g_off.to(dev), g_ids.to(dev), g_rows.to(dev),
                               f_off.to(dev), f_ids.to(dev), group=group, local_counts=local_counts, **kw).cpu().numpy()
    for b, c in enumerate(J['kg']):
        seg = ranks[int(g_off[b]):int(g_off[b + 1])]
        ids_sorted = sorted(c['gold'])
        got = sorted((int(r), i) for r, i in zip(seg, ids_sorted) if r >= 0)
        assert got == sorted(zip(c['ranks'], c['ids'])), (rank, b, got, c['ranks'], c['ids'])
    # ---- a larger seeded case with exact ties across the shard boundary, gold ids inside the filter, empty shards' corner
    rng = np.random.RandomState(5)
    nq, nc = 11, 1001
    sc = (rng.randint(0, 40, size=(nq, nc)) / 8.0).astype(np.float32)
    filt = [sorted(rng.choice(nc, size=rng.randint(0, 200), replace=False).tolist()) for _ in range(nq)]
    gold = [sorted(rng.choice(nc, size=rng.randint(1, 9), replace=False).tolist()) for _ in range(nq)]
    gold[3] = sorted(set(gold[3]) | {filt[3][0]} if filt[3] else gold[3])        # a gold that is itself filtered -> -1
    loc, lo, kw = part(sc)
    f_off, f_ids = _csr(filt); g_off, g_ids = _csr(gold)
    g_rows = torch.from_numpy(np.repeat(np.arange(nq), [len(x) for x in gold]).astype(np.int64))
    for desc in (False, True):
        ids, scs = sharded_topk(torch.from_numpy(loc).to(dev), lo, 10, desc, f_off.to(dev), f_ids.to(dev), group=group,
                                local_topk=local_topk, **kw)
        ranks = sharded_gold_ranks(torch.from_numpy(loc).to(dev), lo, desc, g_off.to(dev), g_ids.to(dev), g_rows.to(dev),
                                   f_off.to(dev), f_ids.to(dev), group=group, local_counts=local_counts, **kw).cpu().numpy()
        for b in range(nq):
            fs, gs = set(filt[b]), set(gold[b])
            order = sorted((j for j in range(nc) if j not in fs), key=lambda j: (_key(sc[b, j], desc), j))
            assert ids[b].tolist() == order[:10]
            np.testing.assert_array_equal(scs[b].cpu().numpy(), sc[b][order[:10]])
            walk = [j for j in order if j not in gs]
            for e, gid in zip(range(int(g_off[b]), int(g_off[b + 1])), gold[b]):
                if gid in fs:
                    assert ranks[e] == -1
                else:
                    gk = (_key(sc[b, gid], desc), gid)
                    assert ranks[e] == sum(1 for j in walk if (_key(sc[b, j], desc), j) < gk), (rank, b, gid)
