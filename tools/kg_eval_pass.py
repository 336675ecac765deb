"""One link-prediction evaluation pass at ml1m-kg shape (14,709 entities, 20 relations, d=100, TransH as in KTUP's KG half):
20,480 (entity, relation) keys in batches of 512, 1-3 gold entities and ~20 filtered entities per key, through
_driver.kg_eval_pass (what the drivers' periodic evaluation runs: ranks stay on the device, one copy back per pass) --
the walk over the batches (K13 + K18 per batch from python) vs the whole pass behind one call (model.rank_entities ->
ktup_eval_kg_ranks).  Prints ms per pass and the mean rank of both routes (equal)."""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch


def main():
    from jTransUP.models import _driver as D
    from jTransUP.models import transE, transH
    dev = torch.device('cuda')
    rng = np.random.RandomState(1)
    torch.manual_seed(1)
    ne, nr, d, nq = 14709, 20, 100, 20480
    model = sys.argv[1] if len(sys.argv) > 1 else 'transh'
    l1 = len(sys.argv) > 2 and sys.argv[2] == 'l1'
    m = (transE.TransEModel if model == 'transe' else transH.TransHModel)(l1, d, ne, nr).to(dev)
    m.eval(); m.disable_grad()
    FL = types.SimpleNamespace(topn=10)
    keys = list(dict.fromkeys((int(rng.randint(ne)), int(rng.randint(nr))) for _ in range(nq + 2000)))[:nq]
    gold = {k: set(rng.randint(0, ne, size=rng.randint(1, 4)).tolist()) for k in keys}
    filt = {k: set(rng.randint(0, ne, size=20).tolist()) for k in keys}
    batches = [keys[s:s + 512] for s in range(0, len(keys), 512)]
    score_fn = lambda q, r: m.evaluateTail(q, r)
    rank_fn = lambda q, r, desc, go, gi, fo, fi: m.rank_entities(q, r, False, desc, go, gi, fo, fi)

    def timed(**kw):
        D.kg_eval_pass(FL, score_fn, batches, gold, [filt], False, want_rows=False, **kw)       # builds the index, warms up
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            out = D.kg_eval_pass(FL, score_fn, batches, gold, [filt], False, want_rows=False, **kw)
        return 1e3 * (time.perf_counter() - t0) / reps, out

    from jTransUP.hip import lib as L
    walk_ms, a = timed()
    pass_ms, b = timed(rank_fn=rank_fn)
    # squared L2: the pass decides comparisons near a gold in fp64 (option kg_exact); the walk ranks by its fp32 score matrix alone
    differ = int((a[:, 1] != b[:, 1]).sum())
    assert differ <= a.shape[0] // 100 and np.abs(a[:, 1] - b[:, 1]).max() <= 2
    old = L.set_option('kg_exact', 0)
    raw_ms, c = timed(rank_fn=rank_fn)
    L.set_option('kg_exact', old)
    assert np.array_equal(a, c)
    print('KG evaluation pass, %d keys x %d entities (%s%s, d=%d), %d gold entries:' % (len(keys), ne, model, ' L1' if l1 else '', d, a.shape[0]))
    print('  walk over %d batches (K13 + K18 per batch from python): %8.2f ms per pass   mean rank %.2f' % (len(batches), walk_ms, a[:, 1].mean()))
    print('  whole pass behind one call (ktup_eval_kg_ranks):        %8.2f ms per pass   mean rank %.2f   (%d ranks moved by the fp64 referee near a gold)' % (pass_ms, b[:, 1].mean(), differ))
    print('  the same with option kg_exact = 0 (fp32 scores alone):   %8.2f ms per pass   mean rank %.2f' % (raw_ms, c[:, 1].mean()))


if __name__ == '__main__':
    import contextlib
    with contextlib.redirect_stderr(open(os.devnull, 'w')):      # tqdm bars of the walk
        main()
