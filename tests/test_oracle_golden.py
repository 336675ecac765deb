"""Pins oracle/cpu_ref.py against the golden vectors the REFERENCE produced (tests/golden/make_goldens.py).

CPU only.  Tolerances: the oracle replays the reference's torch op sequence, so forward values are
expected bit-identical or within a few ulp (1e-6 rel); integer outputs (ranked ids, ranks, hits) exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
T = lambda a: torch.from_numpy(np.asarray(a))
RTOL, ATOL = 1e-5, 1e-6


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def leaf(g, key):
    return T(g[key]).clone().requires_grad_(True)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
def test_bprmf_scores_loss_grads(golden, d):
    g = golden('score_d%d' % d)
    U, I = leaf(g, 'bprmf.user_embeddings.weight'), leaf(g, 'bprmf.item_embeddings.weight')
    u, pi, ni = T(g['u']), T(g['pi']), T(g['ni'])
    pos, neg = O.score_bprmf(U, I, u, pi), O.score_bprmf(U, I, u, ni)
    close(pos, g['bprmf.pos']); close(neg, g['bprmf.neg'])
    loss = O.bpr_loss(pos, neg, 1.0)
    close(loss, g['bprmf.loss'])
    loss.backward()
    close(U.grad, g['bprmf.grad.user_embeddings.weight']); close(I.grad, g['bprmf.grad.item_embeddings.weight'])


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('name', ['transe', 'transh', 'transr'])
def test_kg_scorers(golden, d, l1, name):
    if name == 'transr' and d == 256:
        pytest.skip('no TransR golden at d=256 (the projection table would make the fixture 6 MB)')
    g = golden('score_d%d' % d)
    tag = '%s.%s.' % (name, 'L1' if l1 else 'L2')
    E, R = leaf(g, name + '.ent_embeddings.weight'), leaf(g, name + '.rel_embeddings.weight')
    ph, pt, pr, nh, nt = (T(g[k]) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
    extra = None
    if name == 'transe':
        f = lambda h, t, r: O.score_transe(E, R, h, t, r, l1)
    elif name == 'transh':
        extra = leaf(g, 'transh.norm_embeddings.weight')
        f = lambda h, t, r: O.score_transh(E, R, extra, h, t, r, l1)
    else:
        extra = leaf(g, 'transr.proj_embeddings.weight')
        f = lambda h, t, r: O.score_transr(E, R, extra, h, t, r, l1)
    pos, neg = f(ph, pt, pr), f(nh, nt, pr)
    close(pos, g[tag + 'pos']); close(neg, g[tag + 'neg'])
    loss = O.margin_loss(pos, neg, 1.0)
    rel = R[torch.cat([pr, pr])]
    if name == 'transh':
        loss = loss + O.orthogonal_loss(rel, extra[torch.cat([pr, pr])])
    loss = loss + O.norm_loss(E[torch.cat([ph, pt, nh, nt])]) + O.norm_loss(rel)
    close(loss, g[tag + 'loss'], rtol=1e-5)
    loss.backward()
    close(E.grad, g[tag + 'grad.ent_embeddings.weight'], atol=1e-5)
    close(R.grad, g[tag + 'grad.rel_embeddings.weight'], atol=1e-5)
    if name == 'transh':
        close(extra.grad, g[tag + 'grad.norm_embeddings.weight'], atol=1e-5)
    if name == 'transr':
        close(extra.grad, g[tag + 'grad.proj_embeddings.weight'], atol=1e-5)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('gum', [False, True])
def test_tup(golden, d, l1, gum):
    g = golden('score_d%d' % d)
    tag = 'tup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
    U, I = leaf(g, 'tup.user_embeddings.weight'), leaf(g, 'tup.item_embeddings.weight')
    P, Pn = leaf(g, 'tup.pref_embeddings.weight'), leaf(g, 'tup.pref_norm_embeddings.weight')
    u, pi, ni = T(g['u']), T(g['pi']), T(g['ni'])
    up = T(g[tag + 'uni_pos']) if gum else None
    un = T(g[tag + 'uni_neg']) if gum else None
    pos, neg = O.score_tup(U, I, P, Pn, u, pi, l1, up), O.score_tup(U, I, P, Pn, u, ni, l1, un)
    close(pos, g[tag + 'pos']); close(neg, g[tag + 'neg'])
    loss = O.bpr_loss(pos, neg, -1.0) + O.orthogonal_loss(P, Pn) + O.norm_loss(U[u]) \
        + O.norm_loss(I[torch.cat([pi, ni])]) + O.norm_loss(P)
    close(loss, g[tag + 'loss'])
    loss.backward()
    for w, k in ((U, 'user_embeddings'), (I, 'item_embeddings'), (P, 'pref_embeddings'), (Pn, 'pref_norm_embeddings')):
        close(w.grad, g[tag + 'grad.%s.weight' % k], atol=1e-5)
    if not gum:
        pr_, re_, no_ = O.tup_preferences(U[u], I[pi], P, Pn)
        close(pr_, g[tag + 'pref.probs']); close(re_, g[tag + 'pref.r_e']); close(no_, g[tag + 'pref.norm'])


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('gum', [False, True])
def test_ktup(golden, d, l1, gum):
    g = golden('score_d%d' % d)
    tag = 'ktup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
    names = ['user_embeddings', 'item_embeddings', 'ent_embeddings', 'pref_embeddings', 'pref_norm_embeddings',
             'rel_embeddings', 'norm_embeddings']
    W = {k: leaf(g, 'ktup.%s.weight' % k) for k in names}
    i2e = T(g['ktup.item2ent'])
    u, pi, ni = T(g['u']), T(g['pi']), T(g['ni'])
    up = T(g[tag + 'uni_pos']) if gum else None
    un = T(g[tag + 'uni_neg']) if gum else None
    rec = lambda i, uni: O.score_ktup_rec(W['user_embeddings'], W['item_embeddings'], W['ent_embeddings'],
                                          W['pref_embeddings'], W['pref_norm_embeddings'], W['rel_embeddings'],
                                          W['norm_embeddings'], i2e, u, i, l1, uni)
    pos, neg = rec(pi, up), rec(ni, un)
    close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
    loss = O.bpr_loss(pos, neg, -1.0) + O.orthogonal_loss(W['pref_embeddings'], W['pref_norm_embeddings'])
    close(loss, g[tag + 'rec.loss'])
    loss.backward()
    for k in names:
        key = tag + 'rec.grad.%s.weight' % k
        if key in g:
            got = W[k].grad.clone()
            if k == 'ent_embeddings':
                got[-1].zero_()          # nn.Embedding(padding_idx=...) never accumulates a grad for the pad row
            close(got, g[key], atol=1e-5)
    if not gum:
        for w in W.values():
            w.grad = None
        ph, pt, pr, nh, nt = (T(g[k]) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
        kg = lambda h, t: O.score_ktup_kg(W['ent_embeddings'], W['rel_embeddings'], W['norm_embeddings'], h, t, pr, l1)
        pos, neg = kg(ph, pt), kg(nh, nt)
        close(pos, g[tag + 'kg.pos']); close(neg, g[tag + 'kg.neg'])
        rel = W['rel_embeddings'][torch.cat([pr, pr])]
        loss = O.margin_loss(pos, neg, 1.0) + O.orthogonal_loss(rel, W['norm_embeddings'][torch.cat([pr, pr])]) \
            + O.norm_loss(W['ent_embeddings'][torch.cat([ph, pt, nh, nt])]) + O.norm_loss(rel)
        close(loss, g[tag + 'kg.loss'])
        loss.backward()
        for k in ('ent_embeddings', 'rel_embeddings', 'norm_embeddings'):
            close(W[k].grad, g[tag + 'kg.grad.%s.weight' % k], atol=1e-5)


def test_eval_matrices(golden):
    g = golden('eval_small')
    uq, eq, rq = T(g['uq']), T(g['eq']), T(g['rq'])
    close(O.eval_bprmf(T(g['bprmf.user_embeddings.weight']), T(g['bprmf.item_embeddings.weight']), uq), g['bprmf.eval'])
    for l1 in (False, True):
        L = 'L1' if l1 else 'L2'
        E, R = T(g['transe.ent_embeddings.weight']), T(g['transe.rel_embeddings.weight'])
        close(O.eval_transe(E, R, eq, rq, l1, True), g['transe.%s.head' % L])
        close(O.eval_transe(E, R, eq, rq, l1, False), g['transe.%s.tail' % L])
        E, R, N = (T(g['transh.%s.weight' % k]) for k in ('ent_embeddings', 'rel_embeddings', 'norm_embeddings'))
        close(O.eval_transh(E, R, N, eq, rq, l1, True), g['transh.%s.head' % L])
        close(O.eval_transh(E, R, N, eq, rq, l1, False), g['transh.%s.tail' % L])
        E, R, M = (T(g['transr.%s.weight' % k]) for k in ('ent_embeddings', 'rel_embeddings', 'proj_embeddings'))
        close(O.eval_transr(E, R, M, eq, rq, l1, True), g['transr.%s.head' % L], rtol=1e-4, atol=1e-5)
        close(O.eval_transr(E, R, M, eq, rq, l1, False), g['transr.%s.tail' % L], rtol=1e-4, atol=1e-5)
        for gum in (False, True):
            H = 'hard' if gum else 'soft'
            U, I, P, Pn = (T(g['tup.%s.weight' % k]) for k in ('user_embeddings', 'item_embeddings', 'pref_embeddings', 'pref_norm_embeddings'))
            uni = T(g['tup.%s.%s.uni' % (L, H)]) if gum else None
            close(O.eval_tup(U, I, P, Pn, uq, l1, uni), g['tup.%s.%s.eval' % (L, H)])
            K = {k: T(g['ktup.%s.weight' % k]) for k in ('user_embeddings', 'item_embeddings', 'ent_embeddings', 'pref_embeddings',
                                                         'pref_norm_embeddings', 'rel_embeddings', 'norm_embeddings')}
            uni = T(g['ktup.%s.%s.uni' % (L, H)]) if gum else None
            got = O.eval_ktup_rec(K['user_embeddings'], K['item_embeddings'], K['ent_embeddings'], K['pref_embeddings'],
                                  K['pref_norm_embeddings'], K['rel_embeddings'], K['norm_embeddings'], T(g['ktup.item2ent']), uq, l1, uni)
            close(got, g['ktup.%s.%s.evalRec' % (L, H)])
            if not gum:
                close(O.eval_transh(K['ent_embeddings'], K['rel_embeddings'], K['norm_embeddings'], eq, rq, l1, True), g['ktup.%s.soft.evalHead' % L])
                close(O.eval_transh(K['ent_embeddings'], K['rel_embeddings'], K['norm_embeddings'], eq, rq, l1, False), g['ktup.%s.soft.evalTail' % L])


def test_ranking_exact(golden):
    g = golden('ranking')
    J = json.load(open(os.path.join(GOLDEN, 'ranking.json')))
    nrec = g['rec.rows'].shape[0]
    for b, c in enumerate(J['rec']):
        desc = c.get('descending', False)
        row = -g['rec.bprmf_rows'][b - nrec] if desc else g['rec.rows'][b]
        filt = set(c['filter']) if c['filter'] is not None else None
        f1, p, r, hit, ndcg, top = O.rec_performance(row, set(c['gold']), filt, 10)
        assert top == c['top_ids']
        assert hit == c['hit']
        np.testing.assert_allclose([f1, p, r, ndcg], [c['f1'], c['p'], c['r'], c['ndcg']], rtol=1e-12, atol=0)
    for b, c in enumerate(J['kg']):
        hits, ranks, ids = O.kg_performance(g['kg.rows'][b], set(c['gold']), set(c['filter']), 10)
        assert (hits, ranks, ids) == (c['hits'], c['ranks'], c['ids'])
    for r, k, method, val in J['known']['ndcg']:
        assert O.ndcg_at_k(r, k, method) == pytest.approx(val, rel=1e-14, abs=0)
    # the reference's still-valid doc examples (jTransUP/utils/evaluation.py:88-96)
    assert O.ndcg_at_k([2, 1, 2, 0], 4) == pytest.approx(0.9203032077642922, rel=1e-15)
    assert O.ndcg_at_k([2, 1, 2, 0], 4, method=1) == pytest.approx(0.96519546960144276, rel=1e-15)
    assert O.ndcg_at_k([0], 1) == 0.0 and O.ndcg_at_k([1], 2) == 1.0


def test_alignment_and_schedule():
    J = json.load(open(os.path.join(GOLDEN, 'alignment.json')))
    new_map, e_remap, i_remap, n = O.rebuild_entity_item_vocab(dict(map(tuple, J['e_vocab'])), dict(map(tuple, J['i_vocab'])), J['kg2i'])
    assert n == J['n_aligned']
    assert {str(k): list(v) for k, v in new_map.items()} == J['new_map']
    assert {str(k): v for k, v in e_remap.items()} == J['e_remap']
    assert {str(k): v for k, v in i_remap.items()} == J['i_remap']
    # joint schedule (knowledgable_recommendation.py:209,320): 0.7 -> 7 rec : 3 kg per 10 steps
    for ratio, nrec in ((0.5, 5), (0.7, 7), (0.9, 9)):
        assert sum(O.is_rec_step(s, ratio) for s in range(10)) == nrec
        assert [O.is_rec_step(s, ratio) for s in range(10)] == [s < nrec for s in range(10)]


@pytest.mark.parametrize('d', [36, 64])
def test_baselines_cke_cfkg_golden(golden, d):
    """The oracle's CKE / CFKG restatements (CKE.py:122-203, CFKG.py:66-158) against what the imported reference produced:
    scores, losses and the all-candidate evaluation matrices."""
    g = golden('baselines')
    p = 'd%d.' % d
    t = lambda k: torch.from_numpy(g[p + k])
    u, pi, ni, ph, pt, pr, nh, nt, uq, eq, rq = (t(k).long() for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt', 'uq', 'eq', 'rq'))
    for l1 in (False, True):
        tag = p + 'cke.%s.' % ('L1' if l1 else 'L2')
        U, I, E, R, M = (t('cke.' + k) for k in ('user_embeddings.weight', 'item_embeddings.weight', 'ent_embeddings.weight',
                                                 'rel_embeddings.weight', 'proj_embeddings.weight'))
        i2e = t('cke.item2ent').long()
        pos, neg = O.score_cke_rec(U, I, E, i2e, u, pi), O.score_cke_rec(U, I, E, i2e, u, ni)
        close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
        close(O.bpr_loss(pos, neg, -1.0), g[tag + 'rec.loss'])
        kp, kn = O.score_transr(E, R, M, ph, pt, pr, l1), O.score_transr(E, R, M, nh, nt, pr, l1)
        close(kp, g[tag + 'kg.pos'], rtol=2e-5, atol=1e-5); close(kn, g[tag + 'kg.neg'], rtol=2e-5, atol=1e-5)
        close(O.eval_cke_rec(U, I, E, i2e, uq), g[tag + 'evalRec'])
        close(O.eval_transr(E, R, M, eq, rq, l1, True), g[tag + 'evalHead'], rtol=2e-5, atol=1e-5)
        close(O.eval_transr(E, R, M, eq, rq, l1, False), g[tag + 'evalTail'], rtol=2e-5, atol=1e-5)
        tag = p + 'cfkg.%s.' % ('L1' if l1 else 'L2')
        U, E, R = (t('cfkg.' + k) for k in ('user_embeddings.weight', 'ent_embeddings.weight', 'rel_embeddings.weight'))
        pos, neg = O.score_cfkg_rec(U, E, R, u, pi, l1), O.score_cfkg_rec(U, E, R, u, ni, l1)
        close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
        close(O.bpr_loss(pos, neg, -1.0), g[tag + 'rec.loss'])
        close(O.score_transe(E, R, ph, pt, pr, l1), g[tag + 'kg.pos']); close(O.score_transe(E, R, nh, nt, pr, l1), g[tag + 'kg.neg'])
        close(O.eval_cfkg_rec(U, E, R, uq, l1), g[tag + 'evalRec'])
        close(O.eval_transe(E, R, eq, rq, l1, True), g[tag + 'evalHead']); close(O.eval_transe(E, R, eq, rq, l1, False), g[tag + 'evalTail'])


def test_transr_d256_seeded_golden(golden):
    """TransR at d = 256: the projection table is re-created from the fixture's seed (same torch CPU generator calls as the
    generator script), everything else is stored; the oracle must reproduce the reference's scores."""
    g = golden('transr_d256')
    NE, NR, d = 53, 7, 256
    gen = torch.Generator().manual_seed(int(g['seed'][0]))
    E = torch.randn(NE, d, generator=gen) * 0.3
    R = torch.randn(NR, d, generator=gen) * 0.3
    M = torch.randn(NR, d * d, generator=gen) * 0.06
    ph, pt, pr, nh, nt = (torch.from_numpy(g[k]).long() for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
    for l1 in (False, True):
        tag = 'L1.' if l1 else 'L2.'
        close(O.score_transr(E, R, M, ph, pt, pr, l1), g[tag + 'pos'], rtol=2e-5, atol=2e-5)
        close(O.score_transr(E, R, M, nh, nt, pr, l1), g[tag + 'neg'], rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------------ G4 / G8: whole training steps
def _steps_fixture():
    return np.load(os.path.join(GOLDEN, 'train_steps.npz')), json.load(open(os.path.join(GOLDEN, 'train_steps.json')))


def _params(g, prefix, names):
    return [torch.nn.Parameter(T(g[prefix + n]).clone()) for n in names]


KTUP_NAMES = ['user_embeddings.weight', 'item_embeddings.weight', 'ent_embeddings.weight', 'pref_embeddings.weight',
              'pref_norm_embeddings.weight', 'rel_embeddings.weight', 'norm_embeddings.weight']
TUP_NAMES = ['user_embeddings.weight', 'item_embeddings.weight', 'pref_embeddings.weight', 'pref_norm_embeddings.weight']
STEP_TOL = dict(rtol=2e-5, atol=2e-6)        # several optimizer steps of fp32 sums in a different association order


@pytest.mark.parametrize('fname,opt,lr,l2', [('train_steps_d100.npz', 'Adagrad', 0.05, 0.0), ('train_steps_d100.npz', 'Adam', 0.01, 1e-5),
                                             ('train_steps_d256.npz', 'Adagrad', 0.05, 0.0), ('train_steps_d256.npz', 'Adam', 0.01, 1e-5)])
def test_ktup_training_steps_golden_at_baseline_widths(fname, opt, lr, l2):
    """The same six KTUP steps at d = 100 (BASELINE configs[1]-[3]) and d = 256 (config 5): the oracle is pinned at the widths the
    fused step kernels are instantiated for, not only at the toy world's d = 64."""
    g = np.load(os.path.join(GOLDEN, fname))
    W = _params(g, 'ktup.init.', KTUP_NAMES)
    i2e = T(g['ktup.item2ent'])
    optim = O.make_optimizer(W, opt, lr, l2)
    tag = 'ktup.%s.l2_%g.' % (opt, l2)
    kg_lambda, margin = float(g['ktup.kg_lambda'][0]), float(g['ktup.margin'][0])
    for s, is_rec in enumerate(g['ktup.kinds']):
        b = {k: T(g['ktup.batch%d.%s' % (s, k)]) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt')}
        if is_rec:
            fn = lambda: O.ktup_rec_step_loss(*W, i2e, b['u'], b['pi'], b['ni'])
        else:
            fn = lambda: O.kg_step_loss(W[2], W[5], W[6], b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr'], margin=margin, kg_lambda=kg_lambda)
        loss, norm = O.train_step(W, optim, fn, 5.0, pad_row_of=W[2])
        np.testing.assert_allclose(loss, g[tag + 'losses'][s], rtol=2e-5)
        np.testing.assert_allclose(norm, g[tag + 'gradnorms'][s], rtol=2e-5)
    c = np.load(os.path.join(GOLDEN, fname.replace('.npz', '_cond.npz')))
    for w, n in zip(W, KTUP_NAMES):
        got, want = w.data, T(g[tag + 'final.' + n])
        # the rule of tests/test_hip_train_golden.py: the band, widened by twice the fixture's conditioning where that is not 0
        bad = (got - want).abs() > 2e-5 + 1e-4 * want.abs() + 2 * T(c[tag + 'cond.' + n])
        assert int(bad.sum()) == 0, (n, int(bad.sum()))


@pytest.mark.parametrize('fname,opt,lr,l2', [('train_steps.npz', 'Adam', 0.01, 0.0), ('train_steps_d100.npz', 'Adam', 0.01, 0.0),
                                             ('train_steps_d100.npz', 'Adagrad', 0.05, 1e-5), ('train_steps_d256.npz', 'Adam', 0.01, 1e-5)])
def test_conditioning_fixture_against_the_oracle_in_fp64(fname, opt, lr, l2):
    """<fixture>_cond.npz (make_goldens.py conditioning(): the reference's modules replayed in fp64, in other batch orders, and with
    gradient noise of its own rounding size) is what tests/test_hip_train_golden.py takes its per-element tolerance from.  Its fp64
    piece is checked here against an independent replay: the oracle's steps on double tables must land on final32 + d64 -- i.e. the
    elements the fixture calls ill-conditioned are those where the reference's fp32 result leaves its own fp64 result, and by that
    much.  Plus the file's own invariants: |d64| <= cond, fp64 gradient norms = fp32 ones to 1e-5, ill-conditioned elements <= 1.25 %
    of a case."""
    g = np.load(os.path.join(GOLDEN, fname))
    c = np.load(os.path.join(GOLDEN, fname.replace('.npz', '_cond.npz')))
    tag = 'ktup.%s.l2_%g.' % (opt, l2)
    W = [torch.nn.Parameter(torch.from_numpy(g['ktup.init.' + n]).double()) for n in KTUP_NAMES]
    i2e = T(g['ktup.item2ent'])
    optim = O.make_optimizer(W, opt, lr, l2)
    kg_lambda, margin = float(g['ktup.kg_lambda'][0]), float(g['ktup.margin'][0])
    for s, is_rec in enumerate(g['ktup.kinds']):
        b = {k: T(g['ktup.batch%d.%s' % (s, k)]) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt')}
        if is_rec:
            fn = lambda: O.ktup_rec_step_loss(*W, i2e, b['u'], b['pi'], b['ni'])
        else:
            fn = lambda: O.kg_step_loss(W[2], W[5], W[6], b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr'], margin=margin, kg_lambda=kg_lambda)
        loss, norm = O.train_step(W, optim, fn, 5.0, pad_row_of=W[2])
        np.testing.assert_allclose(norm, c[tag + 'gradnorms64'][s], rtol=1e-9)
        np.testing.assert_allclose(c[tag + 'gradnorms64'][s], g[tag + 'gradnorms'][s], rtol=1e-5)
    n_ill = n_all = 0
    for w, n in zip(W, KTUP_NAMES):
        want32 = g[tag + 'final.' + n].astype(np.float64)
        band = 2e-5 + 1e-4 * np.abs(want32)
        d64, cond = c[tag + 'd64.' + n].astype(np.float64), c[tag + 'cond.' + n].astype(np.float64)
        # entries below band / 64 are stored as 0; the stored ones are float32
        assert np.all(np.abs(w.data.numpy() - (want32 + d64)) <= band / 64 + 1e-6 * np.abs(d64) + 1e-9), n
        assert np.all(np.abs(d64) <= cond * (1 + 1e-6))
        n_ill += int((cond > band / 4).sum()); n_all += cond.size
    assert 0 < n_ill <= n_all // 80, (n_ill, n_all)


def test_joint_schedule_golden():
    """G8: which branch runs at step s (knowledgable_recommendation.py:209,320), as the reference's own expression decided it."""
    _, J = _steps_fixture()
    for ratio, flags in J['joint_schedule'].items():
        assert [bool(O.is_rec_step(s, float(ratio))) for s in range(len(flags))] == flags


@pytest.mark.parametrize('opt,lr,l2', [('Adagrad', 0.05, 0.0), ('Adagrad', 0.05, 1e-5), ('Adam', 0.01, 0.0), ('Adam', 0.01, 1e-5),
                                       ('SGD', 0.05, 1e-5)])
def test_ktup_training_steps_golden(opt, lr, l2):
    """G4: six consecutive KTUP steps (rec, rec, kg, rec, kg, kg) through the reference's ModelTrainer -- losses, pre-clip
    gradient norms and the tables after the last step."""
    g, _ = _steps_fixture()
    W = _params(g, 'ktup.init.', KTUP_NAMES)
    i2e = T(g['ktup.item2ent'])
    optim = O.make_optimizer(W, opt, lr, l2)
    tag = 'ktup.%s.l2_%g.' % (opt, l2)
    kg_lambda, margin = float(g['ktup.kg_lambda'][0]), float(g['ktup.margin'][0])
    for s, is_rec in enumerate(g['ktup.kinds']):
        b = {k: T(g['ktup.batch%d.%s' % (s, k)]) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt')}
        if is_rec:
            fn = lambda: O.ktup_rec_step_loss(*W, i2e, b['u'], b['pi'], b['ni'])
        else:
            fn = lambda: O.kg_step_loss(W[2], W[5], W[6], b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr'], margin=margin, kg_lambda=kg_lambda)
        loss, norm = O.train_step(W, optim, fn, 5.0, pad_row_of=W[2])
        np.testing.assert_allclose(loss, g[tag + 'losses'][s], rtol=2e-5)
        np.testing.assert_allclose(norm, g[tag + 'gradnorms'][s], rtol=2e-5)
    for w, n in zip(W, KTUP_NAMES):
        close(w.data, g[tag + 'final.' + n], **STEP_TOL)
    assert float(W[2].data[-1].abs().sum()) == 0.0 or l2 > 0 or opt != 'Adagrad'      # the pad row only ever moves by weight decay


@pytest.mark.parametrize('gum', [False, True])
@pytest.mark.parametrize('opt,lr', [('Adagrad', 0.05), ('Adam', 0.01)])
def test_tup_training_steps_golden(gum, opt, lr):
    g, _ = _steps_fixture()
    W = _params(g, 'tup.init.', TUP_NAMES)
    optim = O.make_optimizer(W, opt, lr, 1e-5)
    tag = 'tup.%s.%s.' % ('hard' if gum else 'soft', opt)
    clip = float(g[tag + 'clip'][0])
    for s in range(3):
        b = {k: T(g['tup.batch%d.%s' % (s, k)]) for k in ('u', 'pi', 'ni')}
        up = T(g[tag + 'uni%d.pos' % s]) if gum else None
        un = T(g[tag + 'uni%d.neg' % s]) if gum else None
        loss, norm = O.train_step(W, optim, lambda: O.tup_rec_step_loss(*W, b['u'], b['pi'], b['ni'], uni_pos=up, uni_neg=un), clip)
        np.testing.assert_allclose(loss, g[tag + 'losses'][s], rtol=2e-5)
        np.testing.assert_allclose(norm, g[tag + 'gradnorms'][s], rtol=2e-5)
    for w, n in zip(W, TUP_NAMES):
        close(w.data, g[tag + 'final.' + n], **STEP_TOL)


@pytest.mark.parametrize('name', ['transe', 'transh'])
@pytest.mark.parametrize('opt,lr', [('Adagrad', 0.05), ('Adam', 0.01)])
def test_kg_training_steps_golden(name, opt, lr):
    g, _ = _steps_fixture()
    names = ['ent_embeddings.weight', 'rel_embeddings.weight'] + (['norm_embeddings.weight'] if name == 'transh' else [])
    W = _params(g, name + '.init.', names)
    optim = O.make_optimizer(W, opt, lr, 1e-5)
    tag = '%s.%s.' % (name, opt)
    for s in range(3):
        b = {k: T(g['kg.batch%d.%s' % (s, k)]) for k in ('ph', 'pt', 'pr', 'nh', 'nt')}
        N = W[2] if name == 'transh' else None
        loss, norm = O.train_step(W, optim, lambda: O.kg_step_loss(W[0], W[1], N, b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr']), 5.0)
        np.testing.assert_allclose(loss, g[tag + 'losses'][s], rtol=2e-5)
        np.testing.assert_allclose(norm, g[tag + 'gradnorms'][s], rtol=2e-5)
    for w, n in zip(W, names):
        close(w.data, g[tag + 'final.' + n], **STEP_TOL)


# ------------------------------------------------------------------------------------------------ whole evaluation pass
@pytest.mark.parametrize('name,d,l1', [('tup', 64, False), ('ktup', 64, False), ('tup', 100, False), ('ktup', 100, False),
                                       ('tup', 100, True), ('ktup', 100, True)])
def test_eval_pass_golden(name, d, l1):
    """item_recommendation.py:27-53 / knowledgable_recommendation.py:50-104 through the reference's own evalRecProcess: the
    oracle's all-item scores + ranking walk reproduce the per-user metric rows, the ranked ids and the pass means.  L1 = the distance
    of the reference's own run scripts (ktup.sh / transup.sh: -L1_flag with the soft gate)."""
    g = np.load(os.path.join(GOLDEN, 'eval_pass.npz'))
    tag = ('%s.L1.d%d.' if l1 else '%s.d%d.') % (name, d)
    J = json.load(open(os.path.join(GOLDEN, 'eval_pass.json')))[tag.rstrip('.')]
    eval_dict = {int(u): set(v) for u, v in J['eval'].items()}
    all_dicts = [{int(u): set(v) for u, v in J[k].items()} for k in ('train', 'valid')]
    users = torch.arange(37)
    if name == 'tup':
        W = [T(g[tag + n]) for n in TUP_NAMES]
        scores = O.eval_tup(*W, users, l1)
    else:
        W = [T(g[tag + n]) for n in KTUP_NAMES]
        scores = O.eval_ktup_rec(*W, T(g[tag + 'item2ent']), users, l1)
    rows = O.eval_rec_rows(list(zip(users.tolist(), scores.numpy())), eval_dict, all_dicts, descending=False, topn=10)
    rows.sort(key=lambda r: r[-1][0])
    assert [r[-1][0] for r in rows] == J['users']
    assert [[int(x) for x in r[-1][1]] for r in rows] == J['top_ids']
    np.testing.assert_allclose(np.array([r[:5] for r in rows]), g[tag + 'perf'], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.array([r[:5] for r in rows]).mean(axis=0), J['mean'], rtol=1e-12)


def _pass_uniforms(J, n_items):
    """The noise of a hard-gate pass: the reference re-seeded torch's global generator before every batch of users (make_goldens.py
    shim 4), so the uniforms of batch (first, n, seed) are torch.manual_seed(seed); torch.empty(n, n_items, P).uniform_()."""
    parts = []
    for first, n, seed in J['gumbel_seeds']:
        torch.manual_seed(seed)
        parts.append(torch.empty(n, n_items, J['n_pref']).uniform_())
    return torch.cat(parts)


@pytest.mark.parametrize('name', ['tup', 'ktup'])
@pytest.mark.parametrize('l1', [False, True])
def test_eval_pass_hard_gate_golden(name, l1):
    """The same pass with -use_st_gumbel (transUP.py:92,143-170: noise per (user, item, preference)), both distances: the oracle's
    scores for the recorded uniforms + the ranking walk reproduce the reference's ranked ids and metric rows."""
    g = np.load(os.path.join(GOLDEN, 'eval_pass.npz'))
    key = '%s.hard.%s.d100' % (name, 'L1' if l1 else 'L2')
    J = json.load(open(os.path.join(GOLDEN, 'eval_pass.json')))[key]
    tag = key + '.'
    eval_dict = {int(u): set(v) for u, v in J['eval'].items()}
    all_dicts = [{int(u): set(v) for u, v in J[k].items()} for k in ('train', 'valid')]
    users = torch.arange(37)
    n_items = g[tag + 'item_embeddings.weight'].shape[0]
    uni = _pass_uniforms(J, n_items)
    if name == 'tup':
        W = [T(g[tag + n]) for n in TUP_NAMES]
        scores = O.eval_tup(*W, users, l1, uni)
    else:
        W = [T(g[tag + n]) for n in KTUP_NAMES]
        scores = O.eval_ktup_rec(*W, T(g[tag + 'item2ent']), users, l1, uni)
    rows = O.eval_rec_rows(list(zip(users.tolist(), scores.numpy())), eval_dict, all_dicts, descending=False, topn=10)
    rows.sort(key=lambda r: r[-1][0])
    assert [r[-1][0] for r in rows] == J['users']
    assert [[int(x) for x in r[-1][1]] for r in rows] == J['top_ids']
    np.testing.assert_allclose(np.array([r[:5] for r in rows]), g[tag + 'perf'], rtol=1e-12, atol=0)


# ------------------------------------------------------------------------------------------------ FM / coFM
@pytest.mark.parametrize('d', [36, 64])
def test_fm_cofm_golden(golden, d):
    g = golden('fm_cofm')
    p = 'd%d.' % d
    ids = {k: T(g[p + k]).long() for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt', 'uq', 'eq', 'rq')}
    W = {k: leaf(g, p + 'fm.' + k) for k in ('user_embeddings.weight', 'item_embeddings.weight', 'user_bias.weight', 'item_bias.weight', 'bias')}
    args = (W['user_embeddings.weight'], W['item_embeddings.weight'], W['user_bias.weight'], W['item_bias.weight'], W['bias'])
    pos, neg = O.score_fm(*args, ids['u'], ids['pi']), O.score_fm(*args, ids['u'], ids['ni'])
    close(pos, g[p + 'fm.pos']); close(neg, g[p + 'fm.neg'])
    lo = O.bpr_loss(pos, neg, 1.0)
    close(lo, g[p + 'fm.loss'])
    lo.backward()
    for k, w in W.items():
        close(w.grad, g[p + 'fm.grad.' + k], rtol=1e-4, atol=1e-6)
    close(O.eval_fm(*[w.detach() for w in args], ids['uq']), g[p + 'fm.eval'])
    for kind in ('own', 'share'):
        for l1 in (False, True):
            tag = p + 'cofm.%s.%s.' % (kind, 'L1' if l1 else 'L2')
            q = p + 'cofm.%s.' % kind
            E, R = T(g[q + 'ent_embeddings.weight']), T(g[q + 'rel_embeddings.weight'])
            I = E if kind == 'share' else T(g[q + 'item_embeddings.weight'])
            a = (T(g[q + 'user_embeddings.weight']), I, T(g[q + 'user_bias.weight']), T(g[q + 'item_bias.weight']), T(g[q + 'bias']))
            close(O.score_fm(*a, ids['u'], ids['pi']), g[tag + 'rec.pos'])
            close(O.score_transe(E, R, ids['ph'], ids['pt'], ids['pr'], l1), g[tag + 'kg.pos'])
            close(O.eval_fm(*a, ids['uq']), g[tag + 'evalRec'])
            close(O.eval_transe(E, R, ids['eq'], ids['rq'], l1, True), g[tag + 'evalHead'])
            close(O.eval_transe(E, R, ids['eq'], ids['rq'], l1, False), g[tag + 'evalTail'])


# ------------------------------------------------------------------------------------------------ whole link-prediction pass
def _kg_side(S):
    key = lambda e, r: (int(e), int(r))
    eval_dict = {key(e, r): set(v) for e, r, v in S['eval']}
    all_dicts = [{key(e, r): set(v) for e, r, v in S[k]} for k in ('train', 'valid')]
    return [key(e, r) for e, r in S['keys']], eval_dict, all_dicts


@pytest.mark.parametrize('name', ['transe', 'transh'])
@pytest.mark.parametrize('l1', [True, False])
def test_kg_pass_golden(name, l1):
    """knowledge_representation.py:28-75 through the reference's own evalKGProcess (tests/golden/make_goldens.py kg_pass_cases): the
    oracle's all-entity scores + ranking walk reproduce hit and filtered rank of every (key, gold entity) and the pass means, head and
    tail prediction, both distances (L1 = transe.sh / transh.sh / ktup.sh)."""
    g = np.load(os.path.join(GOLDEN, 'kg_pass.npz'))
    tag = '%s.%s' % (name, 'L1' if l1 else 'L2')
    J = json.load(open(os.path.join(GOLDEN, 'kg_pass.json')))[tag]
    E, R = T(g[tag + '.ent_embeddings.weight']), T(g[tag + '.rel_embeddings.weight'])
    N = T(g[tag + '.norm_embeddings.weight']) if name == 'transh' else None
    for side in ('head', 'tail'):
        keys, eval_dict, all_dicts = _kg_side(J[side])
        q, r = torch.tensor([k[0] for k in keys]), torch.tensor([k[1] for k in keys])
        scores = O.eval_transe(E, R, q, r, l1, side == 'head') if N is None else O.eval_transh(E, R, N, q, r, l1, side == 'head')
        rows = O.eval_kg_rows(list(zip(keys, scores.numpy())), eval_dict, all_dicts, descending=False, topn=10)
        got = sorted([k[0], k[1], int(gid), int(rank), int(hit)] for hit, rank, k, gid in rows)
        assert got == J[side]['rows']
        np.testing.assert_allclose(np.array([[r_[4], r_[3]] for r_ in got], dtype=np.float64).mean(axis=0), J[side]['mean'], rtol=1e-12)
