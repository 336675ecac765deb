"""The drivers' one-sweep evaluation (ktup_eval_pref_topk: scores + filtered top-n without the users x items matrix, then K18b's
per-user metrics on the device) against a whole pass of the REFERENCE: model.evaluate / evaluateRec for every user, its own
evalRecProcess (worker processes, filter sets, stable argsort) and the metric means (tests/golden/make_goldens.py
eval_pass_cases: item_recommendation.py:27-53, knowledgable_recommendation.py:50-104, utils/misc.py:148-248).  Ranked ids
bit-exact, metrics to 1e-12."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KTUP_NAMES = ['user_embeddings.weight', 'item_embeddings.weight', 'ent_embeddings.weight', 'pref_embeddings.weight',
              'pref_norm_embeddings.weight', 'rel_embeddings.weight', 'norm_embeddings.weight']
TUP_NAMES = ['user_embeddings.weight', 'item_embeddings.weight', 'pref_embeddings.weight', 'pref_norm_embeddings.weight']


def _csr(dicts, users, dtype=torch.int32):
    off, ids = [0], []
    for u in users:
        s = set()
        for dct in dicts:
            s |= set(dct.get(str(u), []))
        ids += sorted(s)
        off.append(len(ids))
    return torch.tensor(off, dtype=torch.int64, device=DEV), torch.tensor(ids, dtype=dtype, device=DEV)


@pytest.mark.parametrize('name,d,l1', [('tup', 64, False), ('ktup', 64, False), ('tup', 100, False), ('ktup', 100, False),
                                       ('tup', 100, True), ('ktup', 100, True)])
@pytest.mark.parametrize('route', ['one_sweep', 'score_matrix'])
def test_fused_eval_pass_reproduces_the_reference_pass(name, d, l1, route):
    """l1: the soft gate with the L1 distance -- the reference's own run scripts (ktup.sh / transup.sh); its one-sweep pass is
    sweep_soft_kernel (the pair kernel's arithmetic), the squared-L2 one the preference-space sweep."""
    from jTransUP.hip import ops
    g = np.load(os.path.join(GOLDEN, 'eval_pass.npz'))
    tag = ('%s.L1.d%d.' if l1 else '%s.d%d.') % (name, d)
    J = json.load(open(os.path.join(GOLDEN, 'eval_pass.json')))[tag.rstrip('.')]
    t = lambda n: torch.from_numpy(g[tag + n]).to(DEV)
    users = J['users']                                                   # the users that have test items, ascending
    u = torch.tensor(users, dtype=torch.int64, device=DEV)
    f_off, f_ids = _csr([J['train'], J['valid']], users)
    g_off, g_ids = _csr([J['eval']], users)
    if name == 'tup':
        U, I, P, Pn = (t(n) for n in TUP_NAMES)
        items = ops.eval_pref_items(I, None, P, Pn, None, None, None)
        full = lambda: ops.eval_tup(U, I, P, Pn, u, l1, items=items)
    else:
        U, I, E, P, Pn, R, Rn = (t(n) for n in KTUP_NAMES)
        i2e = t('item2ent').to(torch.int32)
        items = ops.eval_pref_items(I, E, P, Pn, R, Rn, i2e)
        full = lambda: ops.eval_ktup(U, I, E, P, Pn, R, Rn, i2e, u, l1, items=items)
    if route == 'one_sweep':
        top = ops.eval_pref_topk(U, u, items, l1, 10, f_off, f_ids)
        assert top is not None                                           # a one-sweep pass covers d = 64 / 100, soft gate, both distances
    else:
        top = ops.topk_filtered(full(), False, 10, f_off, f_ids)
    assert top.cpu().tolist() == J['top_ids']
    perf = ops.rec_metrics(top, g_off, g_ids).cpu().numpy()
    np.testing.assert_allclose(perf, g[tag + 'perf'], rtol=1e-12, atol=0)
    np.testing.assert_allclose(perf.mean(axis=0), J['mean'], rtol=1e-12)


@pytest.mark.parametrize('name', ['tup', 'ktup'])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('route', ['one_sweep', 'score_matrix'])
def test_hard_gate_eval_pass_reproduces_the_reference_pass(name, l1, route):
    """-use_st_gumbel: the reference's whole pass with its noise recovered batch by batch (make_goldens.py shim 4) against the hard
    gate's one-sweep pass (ktup_eval_pref_topk_hard, given uniforms) and the per-batch route: ranked ids bit-exact, metrics to 1e-12."""
    from jTransUP.hip import ops
    g = np.load(os.path.join(GOLDEN, 'eval_pass.npz'))
    key = '%s.hard.%s.d100' % (name, 'L1' if l1 else 'L2')
    J = json.load(open(os.path.join(GOLDEN, 'eval_pass.json')))[key]
    tag = key + '.'
    t = lambda n: torch.from_numpy(g[tag + n]).to(DEV)
    users = J['users']
    u = torch.tensor(users, dtype=torch.int64, device=DEV)
    f_off, f_ids = _csr([J['train'], J['valid']], users)
    g_off, g_ids = _csr([J['eval']], users)
    n_items = g[tag + 'item_embeddings.weight'].shape[0]
    parts = []
    for first, n, seed in J['gumbel_seeds']:                             # the reference re-seeded before every batch of 16 users
        torch.manual_seed(seed)
        parts.append(torch.empty(n, n_items, J['n_pref']).uniform_())
    uni = torch.cat(parts)[users].contiguous().to(DEV)                   # rows of the users that have test items
    if name == 'tup':
        U, I, P, Pn = (t(n) for n in TUP_NAMES)
        items = ops.eval_pref_items(I, None, P, Pn, None, None, None)
        full = lambda: ops.eval_tup(U, I, P, Pn, u, l1, ops.GUMBEL_INPUT, uni, items=items)
    else:
        U, I, E, P, Pn, R, Rn = (t(n) for n in KTUP_NAMES)
        i2e = t('item2ent').to(torch.int32)
        items = ops.eval_pref_items(I, E, P, Pn, R, Rn, i2e)
        full = lambda: ops.eval_ktup(U, I, E, P, Pn, R, Rn, i2e, u, l1, ops.GUMBEL_INPUT, uni, items=items)
    if route == 'one_sweep':
        top = ops.eval_pref_topk_hard(U, u, items, l1, 10, ops.GUMBEL_INPUT, uni, 0, 0, f_off, f_ids)
        assert top is not None
    else:
        top = ops.topk_filtered(full(), False, 10, f_off, f_ids)
    assert top.cpu().tolist() == J['top_ids']
    perf = ops.rec_metrics(top, g_off, g_ids).cpu().numpy()
    np.testing.assert_allclose(perf, g[tag + 'perf'], rtol=1e-12, atol=0)


@pytest.mark.parametrize('name', ['transe', 'transh'])
@pytest.mark.parametrize('l1', [True, False])
@pytest.mark.parametrize('route', ['one_call', 'chunked'])
def test_kg_pass_reproduces_the_reference_pass(name, l1, route):
    """The link-prediction pass behind one call (ktup_eval_kg_ranks_fused: no score matrix; the pair kernels' COUNT form for L1, the
    matrix-core sweep for squared L2) and the chunked route (K12 / K13 + K18 per chunk) against a whole pass of the REFERENCE:
    evaluateHead / evaluateTail for every key and its own evalKGProcess (tests/golden/make_goldens.py kg_pass_cases;
    knowledge_representation.py:28-75, utils/misc.py:61-146), head and tail prediction, L1 (the distance of transe.sh / transh.sh /
    ktup.sh) and squared L2: every filtered rank and the pass's (hit ratio, mean rank) equal the reference's.  (Squared L2 runs as
    |c|^2 - 2 c.e + |e|^2 on the matrix cores -- DESIGN.md section 4's declared deviation for fp32 near-ties; none occurs in these 1208
    (key, gold) entries.)"""
    from jTransUP.hip import ops
    g = np.load(os.path.join(GOLDEN, 'kg_pass.npz'))
    tag = '%s.%s' % (name, 'L1' if l1 else 'L2')
    J = json.load(open(os.path.join(GOLDEN, 'kg_pass.json')))[tag]
    t = lambda n: torch.from_numpy(g[tag + '.' + n]).to(DEV)
    E, R = t('ent_embeddings.weight'), t('rel_embeddings.weight')
    N = t('norm_embeddings.weight') if name == 'transh' else None
    for side in ('head', 'tail'):
        S = J[side]
        keys = [(int(e), int(r)) for e, r in S['keys']]
        gold = {(int(e), int(r)): sorted(v) for e, r, v in S['eval']}
        filt = {}
        for k in ('train', 'valid'):
            for e, r, v in S[k]:
                filt.setdefault((int(e), int(r)), set()).update(v)
        want = {(int(e), int(r), int(gid)): (int(rank), int(hit)) for e, r, gid, rank, hit in S['rows']}
        g_off, g_ids, f_off, f_ids, expect = [0], [], [0], [], []
        for k in keys:
            g_ids += gold[k]; g_off.append(len(g_ids))
            f_ids += sorted(filt.get(k, ())); f_off.append(len(f_ids))
            expect += [want[k + (gid,)][0] for gid in gold[k]]
        dv = lambda a, dt: torch.tensor(a, dtype=dt, device=DEV)
        q, r = dv([k[0] for k in keys], torch.int64), dv([k[1] for k in keys], torch.int64)
        ranks = ops.eval_kg_ranks(E, R, N, q, r, l1, side == 'head', False, dv(g_off, torch.int64), dv(g_ids, torch.int32),
                                  dv(f_off, torch.int64), dv(f_ids, torch.int32), chunk=16, fused=None if route == 'one_call' else False)
        got = ranks.cpu().numpy()[:len(expect)].astype(np.int64)
        expect = np.asarray(expect, dtype=np.int64)
        assert got.tolist() == expect.tolist()
        np.testing.assert_allclose([float((got < 10).mean()), float(got.mean())], S['mean'], rtol=1e-12)
