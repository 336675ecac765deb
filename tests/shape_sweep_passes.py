"""Shape sweep of the whole-pass routes (run by hand on a GPU box: `python tests/shape_sweep_passes.py`): the link-prediction rank
pass (fused sweep / count form / chunked route; TransE, TransH, TransR) and the recommendation top-n pass (preference-space sweep, hard /
soft-L1 sweeps) at ragged catalogue sizes, key counts, relation counts, gold-list lengths, widths and preference counts, against ranks /
ids derived from float64 scores of the oracle's formulas.  A rank may differ by one where two fp32 scores tie to rounding: a case is
reported when more than 2 % of its entries differ or any differs by more than 2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import numpy as np
import torch

from oracle import cpu_ref as O
from jTransUP.hip import ops

DEV = 'cuda'
bad, ran = [], [0]


def csr(lists):
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    ids = (np.concatenate(lists) if sum(len(x) for x in lists) else np.zeros(0)).astype(np.int32)
    return torch.from_numpy(off).to(DEV), torch.from_numpy(ids).to(DEV)


def want_ranks(S, gold, filt):
    """utils/misc.py:125-146 on a float64 score matrix: ascending, ties to the lower id, other golds and filtered ids skipped."""
    out = []
    for k in range(S.shape[0]):
        f = set(int(x) for x in filt[k]); gs = set(int(x) for x in gold[k])
        for g in gold[k]:
            g = int(g)
            if g in f:
                out.append(-1); continue
            sg = S[k, g]; rank = 0
            for c in range(S.shape[1]):
                if c == g or c in f or c in gs:
                    continue
                if S[k, c] < sg or (S[k, c] == sg and c < g):
                    rank += 1
            out.append(rank)
    return np.asarray(out, dtype=np.int64)


def report(name, got, want):
    ran[0] += 1
    got = np.asarray(got.cpu() if isinstance(got, torch.Tensor) else got).astype(np.int64)
    if got.shape != want.shape:
        bad.append((name, 'shape %s vs %s' % (got.shape, want.shape))); return
    if got.size == 0:
        return
    diff = np.abs(got - want)
    if (diff > 0).mean() > 0.02 or diff.max() > 2:
        bad.append((name, '%d of %d differ, max %d' % (int((diff > 0).sum()), diff.size, int(diff.max()))))


def guard(name, fn):
    try:
        fn()
    except Exception as e:                                    # noqa: BLE001
        ran[0] += 1
        bad.append((name, '%s: %s' % (type(e).__name__, str(e)[:160])))


rng = np.random.RandomState(5)
# ---------------------------------------------------------------- link prediction
for d in (20, 36, 50, 64, 100, 128, 200, 256, 300):
    for ne, nq, nr in ((1, 3, 1), (63, 1, 2), (64, 70, 7), (65, 513, 3), (1000, 130, 200), (150, 40, 1)):
        gen = torch.Generator().manual_seed(d * 7 + ne)
        E, R, N = O.make_table(ne, d, gen), O.make_table(nr, d, gen), O.make_table(nr, d, gen)
        M = torch.randn(nr, d * d, generator=gen) * 0.1 if d <= 64 else None
        q = torch.randint(0, ne, (nq,), generator=gen); r = torch.randint(0, nr, (nq,), generator=gen)
        gold = [np.unique(rng.randint(0, ne, size=rng.randint(1, 5))) for _ in range(nq)]
        filt = [np.unique(rng.randint(0, ne, size=rng.randint(0, min(20, ne) + 1))) for _ in range(nq)]
        g_off, g_ids = csr(gold); f_off, f_ids = csr(filt)
        Ed, Rd, Nd = E.to(DEV), R.to(DEV), N.to(DEV)
        for l1 in (False, True):
            for head in (False, True):
                for model in ('transe', 'transh', 'transr'):
                    if model == 'transr' and M is None:
                        continue
                    if model == 'transe':
                        S = O.eval_transe(E.double(), R.double(), q, r, l1, head)
                    elif model == 'transh':
                        S = O.eval_transh(E.double(), R.double(), N.double(), q, r, l1, head)
                    else:
                        S = O.eval_transr(E.double(), R.double(), M.double(), q, r, l1, head)
                    want = want_ranks(S.numpy(), gold, filt)
                    tag = '%s d=%d ne=%d nq=%d nr=%d %s %s' % (model, d, ne, nq, nr, 'l1' if l1 else 'l2', 'head' if head else 'tail')
                    if model == 'transr':
                        guard('kg_ranks_transr ' + tag, lambda: report('kg_ranks_transr ' + tag, ops.eval_kg_ranks_transr(
                            Ed, Rd, M.to(DEV), q.to(DEV), r.to(DEV), l1, head, False, g_off, g_ids, f_off, f_ids)[:len(want)], want))
                        continue
                    Nn = None if model == 'transe' else Nd
                    for fused in (None, False):
                        guard('kg_ranks ' + tag, lambda: report('kg_ranks%s %s' % ('' if fused is None else '[chunked]', tag), ops.eval_kg_ranks(
                            Ed, Rd, Nn, q.to(DEV), r.to(DEV), l1, head, False, g_off, g_ids, f_off, f_ids, fused=fused, chunk=64)[:len(want)], want))
    print('kg d=%d done: %d cases, %d problems' % (d, ran[0], len(bad)), flush=True)

# ---------------------------------------------------------------- recommendation top-n passes
for d in (20, 36, 64, 100, 128, 168, 172, 200, 212, 216, 256, 260):
    for P in (1, 4, 20, 32):
        for ni, nq, topn in ((1, 3, 1), (7, 5, 10), (63, 37, 10), (64, 1, 16), (65, 300, 10), (500, 64, 5)):
            gen = torch.Generator().manual_seed(d + 31 * P + ni)
            nu, ne = 90, 150
            mk = lambda rws: O.make_table(rws, d, gen)
            W = dict(U=mk(nu), I=mk(ni), E=torch.cat([mk(ne), torch.zeros(1, d)]), P=mk(P), Pn=mk(P), R=mk(P), Rn=mk(P))
            i2e = torch.randint(0, ne + 1, (ni,), generator=gen)
            u = torch.randint(0, nu, (nq,), generator=gen)
            filt = [np.unique(rng.randint(0, ni, size=rng.randint(0, min(20, ni) + 1))) for _ in range(nq)]
            f_off, f_ids = csr(filt)
            D = {k: v.to(DEV) for k, v in W.items()}
            i2e_d = i2e.to(DEV, torch.int32)
            for ktup in (False, True):
                for l1 in (False, True):
                    tag = '%s d=%d P=%d ni=%d nq=%d topn=%d %s' % ('ktup' if ktup else 'tup', d, P, ni, nq, topn, 'l1' if l1 else 'l2')

                    def run():
                        items = ops.eval_pref_items(D['I'], D['E'] if ktup else None, D['P'], D['Pn'], D['R'] if ktup else None,
                                                    D['Rn'] if ktup else None, i2e_d if ktup else None)
                        got = ops.eval_pref_topk(D['U'], u.to(DEV), items, l1, topn, f_off, f_ids)
                        if got is None:
                            ran[0] += 1
                            return
                        Wd = {k: v.double() for k, v in W.items()}
                        S = (O.eval_ktup_rec(Wd['U'], Wd['I'], Wd['E'], Wd['P'], Wd['Pn'], Wd['R'], Wd['Rn'], i2e, u, l1) if ktup
                             else O.eval_tup(Wd['U'], Wd['I'], Wd['P'], Wd['Pn'], u, l1)).numpy()
                        want = np.full((nq, topn), -1, dtype=np.int64)
                        for k in range(nq):
                            keep = np.setdiff1d(np.arange(ni), filt[k])
                            order = keep[np.lexsort((keep, S[k, keep]))][:topn]
                            want[k, :len(order)] = order
                        ran[0] += 1
                        g = got.cpu().numpy().astype(np.int64)
                        rows_off = int((g != want).any(1).sum())
                        if rows_off > max(1, nq // 50):                       # a swap of two near-tied neighbours is not a bug
                            bad.append(('rec_topk ' + tag, '%d of %d users differ' % (rows_off, nq)))
                    guard('rec_topk ' + tag, run)
    print('rec d=%d done: %d cases, %d problems' % (d, ran[0], len(bad)), flush=True)
for b in bad:
    print('PROBLEM %s: %s' % b)
sys.exit(1 if bad else 0)
