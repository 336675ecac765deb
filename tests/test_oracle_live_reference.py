"""The oracle against the reference ITSELF on random worlds (tests/golden/live_check.py): runs only where /root/reference exists
-- the build container -- in a subprocess of its own, because the reference's package and this build's mirror of its interface
are both called jTransUP.  Skipped on the GPU box, where the reference cannot be (the committed goldens are what travels)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'jTransUP')), reason='the reference is present in the build container only')
@pytest.mark.parametrize('seed', [0, 1])
def test_oracle_matches_the_live_reference_on_random_worlds(seed):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'live_check.py'), '--worlds', '5', '--seed', str(seed), '--ref', REF],
                         capture_output=True, text=True, timeout=600, env=env, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    worst = rep['worst_excess_over_bar']
    assert worst.pop('ranking.mismatches') == 0
    assert len(worst) == 24                                      # every family ran: scores / evaluation matrices of eight model kinds, step loss + gradients of four
    # 1.0 = the goldens' bars (|diff| <= 1e-6 + 1e-5 |reference|; TransR 1e-5 + 1e-4; gradients 1e-6 + 1e-4)
    assert all(v <= 1.0 for v in worst.values()), worst


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'jTransUP', 'utils', 'data.py')), reason='the reference is present in the build container only')
def test_host_samplers_draw_what_the_reference_draws():
    """jTransUP/utils/data.py keeps the per-user union of the rating dicts between batches and slices the epoch's order as an array
    (the reference rebuilds the union per rating: 5.6 -> 0.7 ms per batch of 512): under the same `random` seed the negatives, the
    corrupted triples and the batches are the reference's, draw for draw.  The reference's module is loaded by path (it imports
    random, copy and numpy only)."""
    import importlib.util
    import random
    import sys as _sys
    import numpy as np
    spec = importlib.util.spec_from_file_location('ktup_reference_utils_data', os.path.join(REF, 'jTransUP', 'utils', 'data.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    _sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'joint-kg-recommender_amd'))
    from jTransUP.utils import data as mine
    rng = np.random.RandomState(3)
    NU, NI, NE, NR = 120, 90, 200, 7
    splits = [{u: set(rng.randint(0, NI, rng.randint(1, 25)).tolist()) for u in range(NU) if rng.rand() < p} for p in (1.0, 0.6, 0.5)]
    ratings = [(int(u), int(i)) for u in range(NU) for i in sorted(splits[0][u])]
    triples = [(int(a), int(b), int(c)) for a, b, c in zip(rng.randint(0, NE, 900), rng.randint(0, NE, 900), rng.randint(0, NR, 900))]
    hd, td = {}, {}
    for h, t, r in triples:
        hd.setdefault((t, r), set()).add(h); td.setdefault((h, r), set()).add(t)

    def run(mod):
        random.seed(11)
        out = []
        it = mod.MakeTrainIterator(ratings, 64)
        it2 = mod.MakeTrainIterator(triples, 48, negtive_samples=2)
        for _ in range(3 * (len(ratings) // 64) + 2):                       # several epochs: the order is reshuffled, users recur
            out.append(mod.getNegRatings(next(it), NI, all_dicts=splits))
        for _ in range(2 * (2 * len(triples) // 48) + 2):
            batch = [tuple(x) for x in next(it2)]
            out.append(mod.getTrainTripleBatch(batch, NE, all_head_dicts=[hd], all_tail_dicts=[td]))
        out.append(mod.getNegRatings(ratings[:40], NI, all_dicts=splits[:2]))     # another set of dicts: another cache entry
        out.append(mod.getNegRatings(ratings[:40], NI, all_dicts=splits))
        return out

    assert run(mine) == run(ref)
    assert run(mine) == run(ref)                                            # and again, with the unions already cached
