"""GPU parity of the all-candidate evaluation kernels (K11-K16) and the ranking kernels (K17/K18).

Scores: golden matrices the reference produced + the CPU oracle at sizes it finishes in seconds (incl. ml1m-sized
catalogues).  Ranked id lists / ranks are integer results and must be bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from jTransUP.hip import lib as L

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def dv(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


def close(got, want, rtol=1e-4, atol=1e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


def ops():
    from jTransUP.hip import ops as _ops
    return _ops


def test_eval_matrices_golden(golden):
    g = golden('eval_small')
    uq, eq, rq = dv(g['uq']), dv(g['eq']), dv(g['rq'])
    close(ops().eval_bprmf(dv(g['bprmf.user_embeddings.weight']), dv(g['bprmf.item_embeddings.weight']), uq), g['bprmf.eval'])
    for l1 in (False, True):
        L = 'L1' if l1 else 'L2'
        E, R = dv(g['transe.ent_embeddings.weight']), dv(g['transe.rel_embeddings.weight'])
        close(ops().eval_transe(E, R, eq, rq, l1, True), g['transe.%s.head' % L])
        close(ops().eval_transe(E, R, eq, rq, l1, False), g['transe.%s.tail' % L])
        E, R, N = (dv(g['transh.%s.weight' % k]) for k in ('ent_embeddings', 'rel_embeddings', 'norm_embeddings'))
        close(ops().eval_transh(E, R, N, eq, rq, l1, True), g['transh.%s.head' % L])
        close(ops().eval_transh(E, R, N, eq, rq, l1, False), g['transh.%s.tail' % L])
        E, R, M = (dv(g['transr.%s.weight' % k]) for k in ('ent_embeddings', 'rel_embeddings', 'proj_embeddings'))
        close(ops().eval_transr(E, R, M, eq, rq, l1, True), g['transr.%s.head' % L])
        close(ops().eval_transr(E, R, M, eq, rq, l1, False), g['transr.%s.tail' % L])
        for gum in (False, True):
            H = 'hard' if gum else 'soft'
            mode = ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF
            U, I, P, Pn = (dv(g['tup.%s.weight' % k]) for k in ('user_embeddings', 'item_embeddings', 'pref_embeddings', 'pref_norm_embeddings'))
            uni = dv(g['tup.%s.%s.uni' % (L, H)]) if gum else None
            close(ops().eval_tup(U, I, P, Pn, uq, l1, mode, uni), g['tup.%s.%s.eval' % (L, H)])
            K = {k: dv(g['ktup.%s.weight' % k]) for k in ('user_embeddings', 'item_embeddings', 'ent_embeddings', 'pref_embeddings',
                                                          'pref_norm_embeddings', 'rel_embeddings', 'norm_embeddings')}
            uni = dv(g['ktup.%s.%s.uni' % (L, H)]) if gum else None
            got = ops().eval_ktup(K['user_embeddings'], K['item_embeddings'], K['ent_embeddings'], K['pref_embeddings'],
                                  K['pref_norm_embeddings'], K['rel_embeddings'], K['norm_embeddings'],
                                  dv(g['ktup.item2ent'], torch.int32), uq, l1, mode, uni)
            close(got, g['ktup.%s.%s.evalRec' % (L, H)])
            if not gum:   # pad entity row is a candidate too: (BQ, NE + 1)
                close(ops().eval_transh(K['ent_embeddings'], K['rel_embeddings'], K['norm_embeddings'], eq, rq, l1, True), g['ktup.%s.soft.evalHead' % L])
                close(ops().eval_transh(K['ent_embeddings'], K['rel_embeddings'], K['norm_embeddings'], eq, rq, l1, False), g['ktup.%s.soft.evalTail' % L])


def world(seed, nu, ni, ne, nr, d):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda r: O.make_table(r, d, gen)
    W = dict(U=mk(nu), I=mk(ni), E=torch.cat([mk(ne), torch.zeros(1, d)]), P=mk(nr), Pn=mk(nr), R=mk(nr), Rn=mk(nr))
    i2e = torch.randint(0, ne, (ni,), generator=gen)
    i2e[torch.rand(ni, generator=gen) < 0.1] = ne
    return W, i2e, gen


@pytest.mark.parametrize('d,ni,nq,npref', [(100, 3240, 70, 20), (64, 130, 33, 4), (128, 1000, 5, 13), (100, 63, 1, 20), (36, 200, 9, 40), (50, 301, 17, 6),
                                           (7, 40, 5, 3),
                                           (320, 211, 9, 20), (514, 70, 5, 6), (256, 300, 9, 20), (216, 130, 5, 6), (212, 130, 5, 6),
                                           (64, 130, 9, 100)])     # 100 preferences: past the hard gate's lane-per-preference tile kernels     # 50, 7, 514: not a multiple of 4 (ops stages them with a zero tail); > 256: the row kernels
def test_pref_eval_vs_oracle(d, ni, nq, npref):
    """TUP / KTUP all-item scores at the ml1m catalogue size (3240 items) and ragged small shapes, soft and hard gate (40 preferences:
    the hard gate's squared-L2 score takes its two-pass form beyond 32).  d = 256 (config 5's width) and 216: the pair kernels' staged
    item vectors pass the LDS there and the one-wave-per-pair forward scores the batch; 212 is the last width they hold."""
    W, i2e, gen = world(d + ni, 300, ni, 500, npref, d)
    u = torch.randint(0, 300, (nq,), generator=gen)
    D = {k: v.to(DEV) for k, v in W.items()}
    for l1 in (False, True):
        for gum in (False, True):
            uni = torch.rand(nq, ni, npref, generator=gen) if gum else None
            mode = ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF
            ud = uni.to(DEV) if gum else None
            close(ops().eval_tup(D['U'], D['I'], D['P'], D['Pn'], u.to(DEV), l1, mode, ud), O.eval_tup(W['U'], W['I'], W['P'], W['Pn'], u, l1, uni))
            close(ops().eval_ktup(D['U'], D['I'], D['E'], D['P'], D['Pn'], D['R'], D['Rn'], i2e.to(DEV, torch.int32), u.to(DEV), l1, mode, ud),
                  O.eval_ktup_rec(W['U'], W['I'], W['E'], W['P'], W['Pn'], W['R'], W['Rn'], i2e, u, l1, uni))


@pytest.mark.parametrize('d,ne,nq', [(100, 14709, 40), (50, 777, 19), (64, 64, 4), (7, 65, 3)])
def test_kg_eval_vs_oracle(d, ne, nq):
    """TransE / TransH / TransR all-entity scores incl. the ml1m entity count and d % 4 != 0."""
    gen = torch.Generator().manual_seed(d * 7 + ne)
    E, R, N = O.make_table(ne, d, gen), O.make_table(11, d, gen), O.make_table(11, d, gen)
    M = torch.randn(11, d * d, generator=gen) * 0.1
    q = torch.randint(0, ne, (nq,), generator=gen); r = torch.randint(0, 11, (nq,), generator=gen)
    Ed, Rd, Nd, Md = E.to(DEV), R.to(DEV), N.to(DEV), M.to(DEV)
    for l1 in (False, True):
        for head in (True, False):
            close(ops().eval_transe(Ed, Rd, q.to(DEV), r.to(DEV), l1, head), O.eval_transe(E, R, q, r, l1, head))
            close(ops().eval_transh(Ed, Rd, Nd, q.to(DEV), r.to(DEV), l1, head), O.eval_transh(E, R, N, q, r, l1, head))
            if ne <= 1000:
                close(ops().eval_transr(Ed, Rd, Md, q.to(DEV), r.to(DEV), l1, head), O.eval_transr(E, R, M, q, r, l1, head))


@pytest.mark.parametrize('d,nu,ni,nq', [(64, 6040, 3240, 512), (100, 50, 70, 33), (30, 40, 129, 65)])
def test_bprmf_eval_mfma_vs_oracle(d, nu, ni, nq):
    gen = torch.Generator().manual_seed(nq)
    U, I = O.make_table(nu, d, gen), O.make_table(ni, d, gen)
    I[3] = I[3] * 4.0      # asymmetric operands: a transposed C/D mapping cannot pass
    u = torch.randint(0, nu, (nq,), generator=gen)
    close(ops().eval_bprmf(U.to(DEV), I.to(DEV), u.to(DEV)), O.eval_bprmf(U, I, u), rtol=1e-5, atol=1e-6)


def test_module_evaluate_surface():
    from jTransUP.models import bprmf, jTransUP as jt, transE, transH, transR, transUP
    torch.manual_seed(1)
    u = torch.tensor([0, 5, 7], device=DEV)
    assert bprmf.BPRMF(64, 20, 33).evaluate(u).shape == (3, 33)
    assert transUP.TransUPModel(True, 100, 20, 33, 5, False).evaluate(u).shape == (3, 33)
    assert transUP.TransUPModel(True, 100, 20, 33, 5, True).evaluate(u).shape == (3, 33)      # Philox noise
    r = torch.tensor([0, 1, 2], device=DEV)
    for cls in (transE.TransEModel, transH.TransHModel, transR.TransRModel):
        m = cls(False, 20, 41, 3)
        assert m.evaluateHead(u, r).shape == (3, 41) and m.evaluateTail(u, r).shape == (3, 41)
    im = {i: i for i in range(33)}
    nm = {i: ((i if i % 3 else -1), i) for i in range(33)}
    k = jt.jTransUPModel(False, 100, 20, 33, 41, 3, im, nm, False, False)
    assert k.evaluateRec(u).shape == (3, 33)
    assert k.evaluateHead(u, r).shape == (3, 42)          # pad entity is ranked too (jTransUP.py:195-196)
    # evaluateRec row == forward on the same (u, i) pairs (soft gate)
    items = torch.arange(33, device=DEV)
    fw = k((u[1].expand(33).contiguous(), items), None, is_rec=True)
    close(k.evaluateRec(u)[1], fw.detach())


def test_ranking_golden_exact(golden):
    from jTransUP.utils import ranking as RK
    g = golden('ranking')
    J = json.load(open(os.path.join(GOLDEN, 'ranking.json')))
    nrec = g['rec.rows'].shape[0]
    for b, c in enumerate(J['rec']):
        desc = c.get('descending', False)
        row = g['rec.bprmf_rows'][b - nrec] if desc else g['rec.rows'][b]
        all_dicts = None if c['filter'] is None else [{7: set(c['filter'])}]
        out = RK.evalRecProcess([(7, row)], {7: set(c['gold'])}, all_dicts=all_dicts, descending=desc, topn=10)
        f1, p, r, hit, ndcg, (key, top_ids, gold) = out[0]
        assert top_ids == c['top_ids']
        assert hit == c['hit']
        np.testing.assert_allclose([f1, p, r, ndcg], [c['f1'], c['p'], c['r'], c['ndcg']], rtol=1e-12, atol=0)
    for b, c in enumerate(J['kg']):
        out = RK.evalKGProcess([((1, 2), g['kg.rows'][b])], {(1, 2): set(c['gold'])}, all_dicts=[{(1, 2): set(c['filter'])}],
                               descending=False, topn=10)
        want = sorted(zip(c['ranks'], c['ids'], c['hits']))
        assert sorted((rk, gid, h) for h, rk, _, gid in out) == want


def test_ranking_batched_vs_oracle_with_ties_and_filters():
    """A whole batch through RankIndex slices: quantised scores (many exact ties), per-row filters, gold overlapping filter."""
    from jTransUP.utils import ranking as RK
    rng = np.random.RandomState(5)
    nq, nc = 37, 3240
    scores = (rng.randint(0, 50, size=(nq, nc)) / 7.0).astype(np.float32)
    scores[3] = 0.0; scores[4, ::2] = -0.0
    keys = list(range(100, 100 + nq))
    eval_dict = {k: set(rng.permutation(nc)[:rng.randint(1, 30)].tolist()) for k in keys if k % 5}
    train = {k: set(rng.permutation(nc)[:rng.randint(0, 400)].tolist()) for k in keys}
    other = {k: set(rng.permutation(nc)[:5].tolist()) for k in keys[::2]}
    mat = torch.from_numpy(scores).to(DEV)
    for desc in (False, True):
        idx = RK.RankIndex(keys, eval_dict, [train, other], mat.device)
        got = RK.evalRecProcess((keys[8:30], mat[8:30].contiguous()), eval_dict, [train, other], descending=desc, topn=10, index=idx)
        want = O.eval_rec_rows(list(zip(keys[8:30], scores[8:30])), eval_dict, [train, other], descending=desc, topn=10)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a[5][1] == b[5][1] and a[5][0] == b[5][0]
            np.testing.assert_allclose(a[:5], b[:5], rtol=1e-12)
        gotk = RK.evalKGProcess((keys, mat), eval_dict, [train, other], descending=desc, topn=10, index=idx)
        wantk = O.eval_kg_rows(list(zip(keys, scores)), eval_dict, [train, other], descending=desc, topn=10)
        assert sorted(gotk) == sorted((int(h), int(r), k, int(g)) for h, r, k, g in wantk)


def test_topk_properties_full_catalogue():
    """ml1m entity count: output is sorted by (score, id), unfiltered, and is exactly the head of the stable argsort."""
    rng = np.random.RandomState(1)
    nq, nc = 64, 14709
    scores = rng.rand(nq, nc).astype(np.float32)
    scores[:, 100:200] = scores[:, :100]                       # duplicated scores -> ties across ids
    f_off = np.arange(0, (nq + 1) * 300, 300, dtype=np.int64)
    f_ids = np.concatenate([np.sort(rng.permutation(nc)[:300]) for _ in range(nq)]).astype(np.int32)
    top, ts = ops().topk_filtered(dv(scores), False, 10, dv(f_off), dv(f_ids), with_scores=True)
    top, ts = top.cpu().numpy(), ts.cpu().numpy()
    for b in range(nq):
        filt = set(f_ids[f_off[b]:f_off[b + 1]].tolist())
        order = [j for j in np.argsort(scores[b], kind='stable') if j not in filt][:10]
        assert top[b].tolist() == order
        np.testing.assert_array_equal(ts[b], scores[b][order])


@pytest.mark.gpu
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('noise', ['philox', 'philox_odd_offset', 'input'])
def test_hard_gate_pass_in_one_sweep(l1, noise):
    """ktup_eval_pref_topk_hard (the ST-Gumbel gate's evaluation pass in one sweep, L1 and squared L2, TUP and KTUP): the ids AND
    the scores of eval_tup / eval_ktup + topk_filtered for the same noise -- Philox stream positions of one call over all users,
    with an offset that is and is not a multiple of a Philox block, or a given uniform tensor -- bit for bit; users repeated and in
    any order, a user whose items are all filtered, one with none, several catalogue splits with a ragged last stage."""
    d, nu, ni, nq, topn, P, ne = 100, 90, 700, 150, 10, 20, 200
    if noise == 'input':
        ni, nq = 200, 70                                                      # (nq x ni x P) uniforms
    gen = torch.Generator().manual_seed(7 + ni)
    mk = lambda r: O.make_table(r, d, gen).to(DEV)
    U, I, E, Pm, Pn, R, Rn = mk(nu), mk(ni), torch.cat([O.make_table(ne, d, gen), torch.zeros(1, d)]).to(DEV), mk(P), mk(P), mk(P), mk(P)
    i2e = torch.randint(0, ne + 1, (ni,), generator=gen).to(DEV, torch.int32)
    u = torch.randint(0, nu, (nq,), generator=gen).to(DEV)
    rng = np.random.RandomState(ni)
    filt = [np.sort(rng.choice(ni, size=min(ni, int(rng.randint(0, 170))), replace=False)).astype(np.int32) for _ in range(nq)]
    filt[1] = np.arange(ni, dtype=np.int32)
    filt[2] = np.zeros(0, np.int32)
    f_off = dv(np.concatenate([[0], np.cumsum([len(f) for f in filt])]).astype(np.int64))
    f_ids = dv(np.concatenate(filt).astype(np.int32))
    G = ops()
    mode = G.GUMBEL_INPUT if noise == 'input' else G.GUMBEL_PHILOX
    uni = torch.rand(nq, ni, P, generator=gen).to(DEV) if noise == 'input' else None
    seed, off = 0x1234567, (4 * 977 if noise == 'philox' else 4 * 977 + 3)
    for ktup in (True, False):
        items = G.eval_pref_items(I, E if ktup else None, Pm, Pn, R if ktup else None, Rn if ktup else None, i2e if ktup else None)
        got = G.eval_pref_topk_hard(U, u, items, l1, topn, mode, uni, seed, off, f_off, f_ids, with_scores=True)
        mat = G.eval_ktup(U, I, E, Pm, Pn, R, Rn, i2e, u, l1, mode, uni, seed, off, items=items) if ktup else \
            G.eval_tup(U, I, Pm, Pn, u, l1, mode, uni, seed, off, items=items)
        want = G.topk_filtered(mat, False, topn, f_off, f_ids, with_scores=True)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert got[0][1].tolist() == [-1] * topn
        nf = G.eval_pref_topk_hard(U, u, items, l1, topn, mode, uni, seed, off)
        assert torch.equal(nf, G.topk_filtered(mat, False, topn))


def _host_philox_uniforms(seed, first, count):
    """u01 of the library's Philox4x32-10 stream (csrc/ktup_common.h: counter = (block, 0x4b545550), key = seed; draw i is word i & 3 of
    block i >> 2; 24-bit lattice) at positions first .. first + count - 1, in numpy."""
    idx = np.arange(first, first + count, dtype=np.uint64)
    blk = idx >> np.uint64(2)
    c = [(blk & np.uint64(0xffffffff)).astype(np.uint64), (blk >> np.uint64(32)).astype(np.uint64),
         np.full(count, 0x4b545550, np.uint64), np.zeros(count, np.uint64)]
    a, b = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    M = np.uint64(0xffffffff)
    for _ in range(10):
        m0, m1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = m0 >> np.uint64(32), m0 & M, m1 >> np.uint64(32), m1 & M
        c = [(hi1 ^ c[1] ^ a) & M, lo1, (hi0 ^ c[3] ^ b) & M, lo0]
        a, b = (a + np.uint64(0x9E3779B9)) & M, (b + np.uint64(0xBB67AE85)) & M
    words = np.stack(c, axis=1)[np.arange(count), (idx & np.uint64(3)).astype(np.int64)]
    return ((words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('P,offset', [(20, 0), (20, 4 * 555 + 3), (13, 7), (4, 1)])
def test_eval_philox_stream_is_the_documented_one(P, offset):
    """The hard gate's device noise: pair (b, j), preference p draws at stream position ((b n_items + j) P + p) + offset of
    Philox4x32-10(seed) -- the scores of a Philox call equal those of the same call fed the uniforms a host Philox produces for those
    positions (P % 4 == 0 walks the stream block by block, other P index by index: both against the same host stream)."""
    d, nu, ni, nq = 64, 40, 90, 11
    gen = torch.Generator().manual_seed(P)
    mk = lambda r: O.make_table(r, d, gen).to(DEV)
    U, I, Pm, Pn = mk(nu), mk(ni), mk(P), mk(P)
    u = torch.randint(0, nu, (nq,), generator=gen).to(DEV)
    seed = 0x0123456789abcdef
    uni = torch.from_numpy(_host_philox_uniforms(seed, offset, nq * ni * P).reshape(nq, ni, P)).to(DEV)
    G = ops()
    for l1 in (False, True):
        a = G.eval_tup(U, I, Pm, Pn, u, l1, G.GUMBEL_PHILOX, None, seed, offset)
        b = G.eval_tup(U, I, Pm, Pn, u, l1, G.GUMBEL_INPUT, uni)
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_fast_gate_choice_equals_the_logf_choice_at_scale():
    """207,360 (user, item) pairs at ml1m's item count: the Philox mode (hardware logarithm + a logf redo inside the proven margin) against
    the given-uniforms mode (logf throughout) on the same stream -- every score identical, i.e. every gate choice."""
    d, nu, ni, nq, P = 100, 64, 3240, 64, 20
    gen = torch.Generator().manual_seed(99)
    mk = lambda r: O.make_table(r, d, gen).to(DEV)
    U, I, Pm, Pn = mk(nu), mk(ni), mk(P), mk(P)
    u = torch.arange(nq).to(DEV)
    seed, offset = 0xfeedface12345678, 4 * 123456 + 2
    uni = torch.from_numpy(_host_philox_uniforms(seed, offset, nq * ni * P).reshape(nq, ni, P)).to(DEV)
    G = ops()
    assert torch.equal(G.eval_tup(U, I, Pm, Pn, u, False, G.GUMBEL_PHILOX, None, seed, offset), G.eval_tup(U, I, Pm, Pn, u, False, G.GUMBEL_INPUT, uni))


class _opt(object):
    """with _opt('kg_exact', 0): ... -- a library option for the duration of a block."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = L.set_option(self.name, self.value)

    def __exit__(self, *exc):
        L.set_option(self.name, self.old)


def _rank_case(rng, nq, nc, nf, max_gold, quant):
    scores = (rng.randint(0, quant, size=(nq, nc)) / 7.0).astype(np.float32) if quant else rng.randn(nq, nc).astype(np.float32)
    scores[0, ::3] = -0.0
    filt = [np.sort(rng.choice(nc, size=rng.randint(0, nf + 1), replace=False)).astype(np.int32) for _ in range(nq)]
    gold = [rng.choice(nc, size=rng.randint(0, max_gold + 1), replace=False).astype(np.int32) for _ in range(nq)]
    gold[1] = np.concatenate([gold[1], filt[1][:3]]).astype(np.int32)          # golds that are themselves filtered
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    cat = lambda parts: np.concatenate(parts).astype(np.int32) if sum(len(p) for p in parts) else np.zeros(0, np.int32)
    return scores, filt, gold, off(filt), cat(filt), off(gold), cat(gold)


def _rank_oracle(scores, filt, gold, desc, topn):
    """Plain walk of the stable (score, id) order: misc.py:125-146 / :213-248 with the declared tie rule."""
    tops, ranks = [], []
    for b in range(scores.shape[0]):
        s = -scores[b] if desc else scores[b]
        order = np.argsort(s + 0.0, kind='stable')
        fs, gs = set(filt[b].tolist()), set(gold[b].tolist())
        top = [int(j) for j in order if j not in fs][:topn]
        tops.append(top + [-1] * (topn - len(top)))
        pos, r = {}, 0
        for j in order:
            j = int(j)
            if j in fs:
                continue
            if j in gs:
                pos[j] = r
            else:
                r += 1
        ranks += [pos.get(int(g), -1) for g in gold[b]]
    return np.array(tops, np.int32), np.array(ranks, np.int32)


@pytest.mark.parametrize('nc,quant', [(19001, 0), (50000, 40), (131072 + 17, 0)])
@pytest.mark.parametrize('desc', [False, True])
def test_chunked_ranking_large_catalogue_vs_oracle(nc, quant, desc):
    """Catalogues beyond the single-workgroup LDS path (amazon-book / last-fm entities): chunked K17/K18, bit-exact."""
    rng = np.random.RandomState(nc % 1000 + int(desc))
    nq = 9
    scores, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, nc, 2000, 40, quant)
    want_top, want_ranks = _rank_oracle(scores, filt, gold, desc, 10)
    mat = dv(scores)
    top, ts = ops().topk_filtered(mat, desc, 10, dv(f_off), dv(f_ids), with_scores=True)
    np.testing.assert_array_equal(top.cpu().numpy(), want_top)
    np.testing.assert_array_equal(ts.cpu().numpy(), np.take_along_axis(scores, np.maximum(want_top, 0), 1) * (want_top >= 0))
    ranks = ops().gold_ranks(mat, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))
    np.testing.assert_array_equal(ranks.cpu().numpy(), want_ranks)
    top_nf = ops().topk_filtered(mat, desc, 10)                                  # no filter at all
    want_nf, _ = _rank_oracle(scores, [np.zeros(0, np.int32)] * nq, gold, desc, 10)
    np.testing.assert_array_equal(top_nf.cpu().numpy(), want_nf)


@pytest.mark.parametrize('chunk', [64, 1000, 4096])
def test_chunked_ranking_equals_lds_path(chunk):
    """Option rank_chunk (ktup_set_option) forces the chunked kernels at small N: same integers as the single-workgroup kernels, incl. many
    golds (more than one gold batch), topn larger than the unfiltered set, and everything filtered."""
    rng = np.random.RandomState(chunk)
    nq, nc = 12, 5000
    scores, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, nc, 600, 30, 25)
    gold[2] = rng.choice(nc, size=2500, replace=False).astype(np.int32)          # > GOLD_BATCH golds
    filt[3] = np.arange(nc, dtype=np.int32)                                      # nothing left to rank
    filt[4] = np.sort(rng.choice(nc, size=nc - 4, replace=False)).astype(np.int32)   # 4 candidates left, topn = 50
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    f_off, f_ids, g_off, g_ids = off(filt), np.concatenate(filt), off(gold), np.concatenate(gold)
    mat = dv(scores)
    args = (dv(f_off), dv(f_ids))
    for desc in (False, True):
        for topn in (1, 10, 50):
            a = ops().topk_filtered(mat, desc, topn, *args, with_scores=True)
            ra = ops().gold_ranks(mat, desc, dv(g_off), dv(g_ids), *args)
            old = L.set_option('rank_chunk', chunk)
            try:
                b = ops().topk_filtered(mat, desc, topn, *args, with_scores=True)
                rb = ops().gold_ranks(mat, desc, dv(g_off), dv(g_ids), *args)
            finally:
                L.set_option('rank_chunk', old)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(ra, rb)
    want_top, want_ranks = _rank_oracle(scores, filt, gold, False, 10)
    np.testing.assert_array_equal(ops().topk_filtered(mat, False, 10, *args).cpu().numpy(), want_top)
    np.testing.assert_array_equal(ops().gold_ranks(mat, False, dv(g_off), dv(g_ids), *args).cpu().numpy(), want_ranks)


def test_rec_metrics_on_device_match_host_and_golden(golden):
    """K18b: (f1, p, r, hit, ndcg) per ranked list on the device == the host arithmetic (ranking.rec_metrics, itself pinned by
    the golden ranking cases), incl. short lists (-1 padded), no hits, all hits, a single gold id."""
    from jTransUP.utils import ranking as RK
    rng = np.random.RandomState(2)
    nq, nc, topn = 300, 500, 10
    top = np.stack([rng.permutation(nc)[:topn] for _ in range(nq)]).astype(np.int32)
    gold = [np.sort(rng.choice(nc, size=rng.randint(1, 40), replace=False)).astype(np.int32) for _ in range(nq)]
    top[0, 4:] = -1; top[1, :] = -1; top[2, 1:] = -1
    gold[3] = np.sort(top[3].copy()); gold[4] = top[4, :1].copy(); gold[5] = np.array([nc + 5], np.int32)
    gold[0] = np.unique(np.concatenate([top[0, :2], [0]])).astype(np.int32)
    g_off = np.concatenate([[0], np.cumsum([len(g) for g in gold])]).astype(np.int64)
    got = ops().rec_metrics(dv(top), dv(g_off), dv(np.concatenate(gold))).cpu().numpy()
    want = np.array([RK.rec_metrics([int(i) for i in row if i >= 0], set(g.tolist())) if (row >= 0).any() else (0, 0, 0, 0, 0)
                     for row, g in zip(top, gold)], dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    assert got[3, 3] == 1 and got[3, 1] == 1.0 and got[3, 4] == 1.0 and got[5].sum() == 0
    # the golden rec cases through the whole device pipeline
    g = golden('ranking')
    J = json.load(open(os.path.join(GOLDEN, 'ranking.json')))
    nrec = g['rec.rows'].shape[0]
    for b, c in enumerate(J['rec']):
        desc = c.get('descending', False)
        row = g['rec.bprmf_rows'][b - nrec] if desc else g['rec.rows'][b]
        all_dicts = None if c['filter'] is None else [{7: set(c['filter'])}]
        cols = RK.evalRecProcess([(7, row)], {7: set(c['gold'])}, all_dicts=all_dicts, descending=desc, topn=10, as_array=True)
        np.testing.assert_allclose(cols[0], [c['f1'], c['p'], c['r'], c['hit'], c['ndcg']], rtol=1e-12, atol=0)


def test_eval_passes_fast_paths_equal_the_row_paths():
    """_driver.{rec,kg}_eval_pass with want_rows=False (metric columns / ranks stay on the device, one copy back per pass)
    report the same numbers as the per-row paths the reference's report mode uses."""
    import types
    from jTransUP.models import _driver as D
    rng = np.random.RandomState(4)
    FL = types.SimpleNamespace(topn=10)
    nq, nc = 150, 700
    scores = torch.from_numpy(rng.rand(nq, nc).astype(np.float32)).to(DEV)
    # rec: integer user keys in batches of 64
    users = list(range(nq))
    gold = {u: set(rng.choice(nc, size=rng.randint(1, 20), replace=False).tolist()) for u in users if u % 7}
    train = {u: set(rng.choice(nc, size=60, replace=False).tolist()) for u in users}
    batches = [np.array(users[s:s + 64]) for s in range(0, nq, 64)]
    fn = lambda u: scores[u]
    rows = D.rec_eval_pass(FL, fn, batches, gold, [train], False, want_rows=True)
    cols = D.rec_eval_pass(FL, fn, batches, gold, [train], False, want_rows=False)
    assert cols.shape == (len(rows), 5)
    np.testing.assert_allclose(cols, np.array([r[:5] for r in rows], dtype=np.float64), rtol=1e-12)
    # kg: (t, r) keys, several golds per key, filters that contain some golds
    keys = [(int(t), int(t) % 5) for t in range(nq)]
    kgold = {k: set(rng.choice(nc, size=rng.randint(1, 6), replace=False).tolist()) for k in keys}
    kfilt = {k: set(rng.choice(nc, size=40, replace=False).tolist()) | (set(list(kgold[k])[:1]) if k[0] % 3 == 0 else set()) for k in keys}
    kb = [keys[s:s + 64] for s in range(0, nq, 64)]
    kfn = lambda q, r: scores[q]
    krows = D.kg_eval_pass(FL, kfn, kb, kgold, [kfilt], False, want_rows=True)
    kcols = D.kg_eval_pass(FL, kfn, kb, kgold, [kfilt], False, want_rows=False)
    assert kcols.shape == (len(krows), 2) and len(krows) > 0
    want = np.array(sorted((float(h), float(rk)) for h, rk, _, _ in krows))
    np.testing.assert_array_equal(np.array(sorted(map(tuple, kcols.tolist()))), want)


@pytest.mark.parametrize('d,ne,nq,nrel', [(100, 3000, 70, 20), (128, 1500, 33, 7), (64, 2049, 65, 20)])
def test_transr_l2_matrix_core_route_vs_oracle(d, ne, nq, nrel):
    """TransR squared-L2 evaluation on the matrix cores (queries folded through M_r^T, |M_r e|^2 from the K4 forward) against
    the oracle, and against the VALU route (option eval_mc = 0) at the ml1m entity count."""
    gen = torch.Generator().manual_seed(d + ne)
    E, R = O.make_table(ne, d, gen), O.make_table(nrel, d, gen)
    M = torch.eye(d).reshape(1, d * d).repeat(nrel, 1) + torch.randn(nrel, d * d, generator=gen) * 0.05   # ~ the ctor's identity init
    q = torch.randint(0, ne, (nq,), generator=gen); r = torch.randint(0, nrel, (nq,), generator=gen)
    r[:3] = r[0]                                                   # several queries of one relation
    Ed, Rd, Md = E.to(DEV), R.to(DEV), M.to(DEV)
    for head in (True, False):
        got = ops().eval_transr(Ed, Rd, Md, q.to(DEV), r.to(DEV), False, head)
        close(got, O.eval_transr(E, R, M, q, r, False, head))
    big = O.make_table(14709, d, gen).to(DEV)
    qb = torch.randint(0, 14709, (200,), generator=gen).to(DEV); rb = torch.randint(0, nrel, (200,), generator=gen).to(DEV)
    a = ops().eval_transr(big, Rd, Md, qb, rb, False, True)
    old = L.set_option('eval_mc', 0)
    try:
        b = ops().eval_transr(big, Rd, Md, qb, rb, False, True)
    finally:
        L.set_option('eval_mc', old)
    close(a, b)
    assert torch.equal(a.argsort(1)[:, :5], b.argsort(1)[:, :5])   # same best candidates either way


@pytest.mark.parametrize('gum', [False, True])
@pytest.mark.parametrize('l1', [False, True])
def test_prepared_item_side_gives_the_same_scores(l1, gum):
    """K15 / K16 with the item-side projections prepared once per pass (ktup_eval_pref_items_prepare +
    ktup_eval_pref_scores_prepared) == the one-call form, bit for bit, for batches of different sizes."""
    from jTransUP.models import jTransUP as jt, transUP
    torch.manual_seed(3)
    NU, NI, NE, NR, D = 90, 130, 150, 7, 100
    i_map = {i: i for i in range(NI)}
    new_map = {i: ((i * 3) % NE if i % 5 else -1, i) for i in range(NI)}
    mk = jt.jTransUPModel(l1, D, NU, NI, NE, NR, i_map, new_map, False, gum)
    mt = transUP.TransUPModel(l1, D, NU, NI, NR, gum)
    for m, fn in ((mk, lambda m, u, **kw: m.evaluateRec(u, **kw)), (mt, lambda m, u, **kw: m.evaluate(u, **kw))):
        m.eval(); m.disable_grad()
        items = m.prepare_items()
        for nq in (64, 17, 1):
            u = torch.randint(0, NU, (nq,), device=DEV)
            uni = torch.rand(nq, NI, NR, device=DEV) if gum else None
            a = fn(m, u, uniform=uni)
            b = fn(m, u, uniform=uni, items=items)
            assert torch.equal(a, b)


@pytest.mark.parametrize('d,nu,ni,nq,topn', [(100, 6040, 3240, 6040, 10), (64, 300, 177, 65, 16), (128, 90, 16, 1, 3), (100, 70, 5, 63, 10),
                                             (100, 500, 1000, 129, 1)])
def test_fused_pass_topk_equals_matrix_route(d, nu, ni, nq, topn):
    """ktup_eval_pref_topk (scores + filtered top-n of a whole pass in one sweep, no score matrix) against the matrix
    route (ktup_eval_pref_scores_prepared + ktup_eval_topk_filtered).  The sweep contracts every cross term but u.v in preference
    space -- the same sums in another association -- so scores agree to fp32 rounding and the lists are the matrix route's up to
    swaps between items whose scores differ by less than that.  KTUP and TUP; per-user filters incl. empty and everything."""
    _fused_pass_case(d, nu, ni, nq, topn, True)


def _same_lists_up_to_rounding(ids, sc, want_ids, want_sc, filt, ni, rtol=2e-5, atol=2e-5):
    """Ranked lists of two routes whose scores differ by fp32 rounding: same length, valid unfiltered distinct ids, position-wise
    scores equal within the tolerance (so the lists can only differ by swaps / substitutions among near-ties)."""
    ids, sc, want_ids, want_sc = ids.cpu().numpy(), sc.cpu().numpy(), want_ids.cpu().numpy(), want_sc.cpu().numpy()
    assert ids.shape == want_ids.shape
    assert ((ids >= 0) == (want_ids >= 0)).all()
    live = want_ids >= 0
    np.testing.assert_allclose(sc[live], want_sc[live], rtol=rtol, atol=atol)
    assert (ids[live] < ni).all()
    for b in range(ids.shape[0]):
        row = ids[b][ids[b] >= 0]
        assert len(set(row.tolist())) == len(row)
        if filt is not None:
            assert not set(row.tolist()) & set(filt[b].tolist())
    assert float((ids == want_ids).mean()) > 0.98                          # and almost always they are simply identical


def _fused_pass_case(d, nu, ni, nq, topn, pspace):
    gen = torch.Generator().manual_seed(d + ni)
    P, ne = 20, 200
    mk = lambda r: O.make_table(r, d, gen).to(DEV)
    U, I, E, Pm, Pn, R, Rn = mk(nu), mk(ni), torch.cat([O.make_table(ne, d, gen), torch.zeros(1, d)]).to(DEV), mk(P), mk(P), mk(P), mk(P)
    i2e = torch.randint(0, ne + 1, (ni,), generator=gen).to(DEV, torch.int32)
    u = torch.randperm(nu, generator=gen)[:nq].to(DEV) if nq <= nu else torch.arange(nu).to(DEV)
    rng = np.random.RandomState(ni)
    filt = [np.sort(rng.choice(ni, size=min(ni, int(rng.randint(0, 170))), replace=False)).astype(np.int32) for _ in range(len(u))]
    if len(filt) > 2:
        filt[1] = np.arange(ni, dtype=np.int32)                              # everything filtered -> all -1
        filt[2] = np.zeros(0, np.int32)
    f_off = dv(np.concatenate([[0], np.cumsum([len(f) for f in filt])]).astype(np.int64))
    f_ids = dv(np.concatenate(filt).astype(np.int32)) if sum(len(f) for f in filt) else torch.zeros(0, dtype=torch.int32, device=DEV)
    for ktup in (True, False):
        items = ops().eval_pref_items(I, E if ktup else None, Pm, Pn, R if ktup else None, Rn if ktup else None, i2e if ktup else None)
        got = ops().eval_pref_topk(U, u, items, False, topn, f_off, f_ids, with_scores=True)
        assert got is not None
        want_ids, want_sc = [], []
        for s in range(0, len(u), 512):
            ub = u[s:s + 512]
            if ktup:
                mat = ops().eval_ktup(U, I, E, Pm, Pn, R, Rn, i2e, ub, False, items=items)
            else:
                mat = ops().eval_tup(U, I, Pm, Pn, ub, False, items=items)
            lo = int(f_off[s])
            fo = f_off[s:s + len(ub) + 1] - lo
            a, b = ops().topk_filtered(mat, False, topn, fo, f_ids[lo:], with_scores=True)
            want_ids.append(a); want_sc.append(b)
        want_ids, want_sc = torch.cat(want_ids), torch.cat(want_sc)
        if not pspace:
            assert torch.equal(got[0], want_ids)
            assert torch.equal(got[1], want_sc)
        else:
            _same_lists_up_to_rounding(got[0], got[1], want_ids, want_sc, filt, ni)
        if len(filt) > 2:
            assert got[0][1].tolist() == [-1] * topn
        nf = ops().eval_pref_topk(U, u, items, False, topn)                   # no filter at all
        mat = ops().eval_ktup(U, I, E, Pm, Pn, R, Rn, i2e, u[:64], False, items=items) if ktup else ops().eval_tup(U, I, Pm, Pn, u[:64], False, items=items)
        if not pspace:
            assert torch.equal(nf[:64], ops().topk_filtered(mat, False, topn))
        else:
            a, b = ops().topk_filtered(mat, False, topn, with_scores=True)
            sc = torch.gather(mat, 1, nf[:64].clamp(min=0).long())
            _same_lists_up_to_rounding(nf[:64], sc, a, b, None, ni)
    # L1 does not decompose into preference space: the pair kernel's arithmetic swept with the top-n in its epilogue -- that route's bits
    for ktup in (True, False):
        items = ops().eval_pref_items(I, E if ktup else None, Pm, Pn, R if ktup else None, Rn if ktup else None, i2e if ktup else None)
        got = ops().eval_pref_topk(U, u, items, True, topn, f_off, f_ids, with_scores=True)
        mat = ops().eval_ktup(U, I, E, Pm, Pn, R, Rn, i2e, u, True, items=items) if ktup else ops().eval_tup(U, I, Pm, Pn, u, True, items=items)
        want = ops().topk_filtered(mat, False, topn, f_off, f_ids, with_scores=True)
        assert got is not None and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.parametrize('d,ne,nrel', [(100, 3000, 20), (36, 500, 5), (64, 2049, 7)])
def test_transr_entity_side_prepared_once_per_pass(d, ne, nrel):
    """ktup_eval_transr_prepare + ents_ws: the entity side (|M_r e|^2 for the squared-L2 matrix-core route, M_r e otherwise) computed
    once and shared by the batches of a pass gives bit-identical scores to recomputing it inside every call -- L1 and L2, head
    and tail, matrix-core widths and d = 36 (VALU route); a workspace prepared for the other distance kind is refused."""
    gen = torch.Generator().manual_seed(d * 7 + ne)
    E, R = O.make_table(ne, d, gen).to(DEV), O.make_table(nrel, d, gen).to(DEV)
    M = (torch.eye(d).reshape(1, d * d).repeat(nrel, 1) + torch.randn(nrel, d * d, generator=gen) * 0.05).to(DEV)
    for l1 in (False, True):
        ents = ops().eval_transr_entities(E, M, nrel, l1)
        for nq in (65, 512):
            q = torch.randint(0, ne, (nq,), generator=gen).to(DEV); r = torch.randint(0, nrel, (nq,), generator=gen).to(DEV)
            for head in (True, False):
                a = ops().eval_transr(E, R, M, q, r, l1, head)
                b = ops().eval_transr(E, R, M, q, r, l1, head, ents=ents)
                assert torch.equal(a, b), (d, l1, nq, head)
        with pytest.raises(L.KtupError):
            ops().eval_transr(E, R, M, q, r, not l1, True, ents=ents)


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['transe', 'transh'])
def test_kg_ranks_whole_pass_equals_the_batch_walk(model):
    """ktup_eval_kg_ranks (the loop over 512-key batches under the C ABI) yields the integers of the per-batch route -- K12 / K13
    score matrix + K18 on the batch's CSR slice -- and of the oracle's walk over the oracle's scores: 1,100 keys = two full
    chunks + a ragged one, head and tail, L1 and L2, ascending and descending, with and without filters, golds that are filtered,
    keys without golds."""
    rng = np.random.RandomState(17)
    ne, nr, d, nq = 700, 9, 100, 1100
    gen = torch.Generator().manual_seed(4)
    E, R, N = O.make_table(ne, d, gen), O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    _, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, ne, 60, 5, 0)      # <= 8 golds per key: the fused pass applies
    Ed, Rd, Nd = dv(E.numpy()), dv(R.numpy()), dv(N.numpy())
    qd, rd = q.to(DEV), r.to(DEV)
    for head in (True, False):
        for l1 in (False, True):
            for desc in (False, True):
                for with_filter in (True, False):
                    fo, fi = (dv(f_off), dv(f_ids)) if with_filter else (None, None)
                    with _opt('kg_exact', 0):          # (the matrix route ranks by its fp32 scores alone: compare like with like)
                        got = ops().eval_kg_ranks(Ed, Rd, Nd if model == 'transh' else None, qd, rd, l1, head, desc, dv(g_off), dv(g_ids), fo, fi)
                    # squared L2 takes the pass without a score matrix (counts in the score kernel's epilogue); the chunked matrix
                    # route stays reachable and must give the same integers
                    got_chunked = ops().eval_kg_ranks(Ed, Rd, Nd if model == 'transh' else None, qd, rd, l1, head, desc, dv(g_off), dv(g_ids), fo, fi,
                                                      fused=False)
                    assert torch.equal(got, got_chunked)
                    want = []
                    for s in range(0, nq, 512):
                        e = min(nq, s + 512)
                        sc = ops().eval_transe(Ed, Rd, qd[s:e], rd[s:e], l1, head) if model == 'transe' else \
                            ops().eval_transh(Ed, Rd, Nd, qd[s:e], rd[s:e], l1, head)
                        lo = int(g_off[s])
                        args = (dv(f_off[s:e + 1] - f_off[s]), dv(f_ids[int(f_off[s]):])) if with_filter else ()
                        want.append(ops().gold_ranks(sc, desc, dv(g_off[s:e + 1] - lo), dv(g_ids[lo:]), *args)[:int(g_off[e]) - lo])
                    assert torch.equal(got[:len(g_ids)], torch.cat(want))
    # against the oracle end to end (scores of the oracle, stable walk), a few keys of each kind
    sel = list(range(0, 40)) + list(range(500, 530)) + list(range(1080, 1100))
    for head in (True, False):
        sc = (O.eval_transe(E, R, q[sel], r[sel], False, head) if model == 'transe' else O.eval_transh(E, R, N, q[sel], r[sel], False, head)).numpy()
        _, want_ranks = _rank_oracle(sc, [filt[i] for i in sel], [gold[i] for i in sel], False, 10)
        got = ops().eval_kg_ranks(Ed, Rd, Nd if model == 'transh' else None, qd, rd, False, head, False, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids)).cpu().numpy()
        got_sel = np.concatenate([got[int(g_off[i]):int(g_off[i + 1])] for i in sel])
        # the oracle's scores differ from the device's in the last bits: ranks may move by the number of near-ties only
        assert np.array_equal(got_sel < 0, want_ranks < 0) and np.abs(got_sel - want_ranks).max() <= 1 and (got_sel == want_ranks).mean() > 0.98


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['transe', 'transh'])
@pytest.mark.parametrize('d', [20, 36, 64, 100, 128])
def test_kg_ranks_without_score_matrix(model, d):
    """ktup_eval_kg_ranks_fused at every instantiated width, at ml1m-kg's entity count (14,709: 230 candidate stages in 8 bands, the
    last one ragged), keys whose gold is filtered, keys without golds, duplicate keys: the integers of the matrix route."""
    lib = __import__('jTransUP.hip.lib', fromlist=['x'])
    rng = np.random.RandomState(5 + d)
    ne, nr, nq = 14709, 11, 333
    gen = torch.Generator().manual_seed(d)
    E, R, N = O.make_table(ne, d, gen), O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    q[7], r[7] = q[3], r[3]
    _, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, ne, 40, 5, 0)      # + 3 filtered golds on key 1: 8 at most
    assert lib.load().ktup_eval_kg_ranks_fused_supported(0 if model == 'transe' else 1, d, 0, int(np.diff(g_off).max())) == 1
    Ed, Rd, Nd = dv(E.numpy()), dv(R.numpy()), dv(N.numpy())
    Nn = Nd if model == 'transh' else None
    for head in (True, False):
        for desc in (False, True):
            with _opt('kg_exact', 0):                  # the sweep's own fp32 scores against the matrix route's (same bits); the fp64
                a = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))    # referee has its own test below
            b = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids), fused=False)
            assert torch.equal(a, b) and int((a[:len(g_ids)] >= 0).sum()) > nq // 2
            x = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))
            assert int((x != a).sum()) <= 8 and int((x - a).abs().max()) <= 2        # the referee moves a rank only past a near tie
        with _opt('kg_exact', 0):
            a = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, False, dv(g_off), dv(g_ids))      # no filter at all
        b = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, False, dv(g_off), dv(g_ids), fused=False)
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('model,d,l1', [('transe', 100, True), ('transh', 100, True), ('transe', 50, False), ('transh', 36, True),
                                         ('transe', 256, True), ('transh', 64, False), ('transh', 300, True), ('transh', 300, False)])
def test_kg_ranks_count_route(model, d, l1):
    """The pass without the score matrix on the VALU route (ktup_eval_kg_ranks_fused for L1, for widths without a matrix-core sweep
    -- d = 50 is not even a multiple of 4 -- and for keys with more than 8 golds): the pair kernels count where they score, the
    list scores come from the same function; repeated entity rows (ties everywhere), a NaN row, keys whose gold is filtered, keys
    without golds -- the integers of the matrix route."""
    rng = np.random.RandomState(17 + d)
    ne, nr, nq = 3100, 9, 150
    gen = torch.Generator().manual_seed(d)
    E = O.make_table(ne, d, gen)
    E[2000:] = E[:1100]                                               # a third of the table repeats another third: exact ties
    E[1500] = float('nan')
    R, N = O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    _, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, ne, 40, 12 if d == 64 else 5, 0)    # d = 64: up to 15 golds per key
    Ed, Rd, Nd = dv(E.numpy()), dv(R.numpy()), dv(N.numpy())
    Nn = Nd if model == 'transh' else None
    for head in (True, False):
        for desc in (False, True):
            with _opt('kg_exact', 0):           # (d = 64, squared L2 takes the matrix-core sweep: compare its own scores with the matrix route's)
                a = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), l1, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))
            b = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), l1, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids), fused=False)
            assert torch.equal(a, b)
        with _opt('kg_exact', 0):
            a = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), l1, head, False, dv(g_off), dv(g_ids))
        b = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), l1, head, False, dv(g_off), dv(g_ids), fused=False)
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('model,wtab', [('transe', 1), ('transh', 1), ('transh', 0)])
def test_kg_ranks_without_score_matrix_ties_and_modes(model, wtab):
    """The fused pass compares scores as floats and falls back to the 64-bit keys where a lane sees equality or a NaN: an entity table
    of 40 distinct rows repeated (every candidate ties with hundreds of others, golds included), one NaN row and one inf row, keys
    with up to 8 golds (the second sweep launch) -- the integers of the matrix route, for TransH with w.e from the (relation x
    candidate) table (option kg_wtab = 1) and computed in the sweep (0)."""
    L = __import__('jTransUP.hip.lib', fromlist=['x'])
    rng = np.random.RandomState(11)
    ne, nr, nq, d = 3000, 37, 200, 100
    gen = torch.Generator().manual_seed(3)
    E = O.make_table(40, d, gen).repeat(75, 1).contiguous()
    E[1234] = float('nan'); E[77] = float('inf')
    R, N = O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    q[5] = 1234                                                       # a key whose own row is NaN: every score of it is
    _, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, ne, 40, 5, 0)
    Ed, Rd, Nd = dv(E.numpy()), dv(R.numpy()), dv(N.numpy())
    Nn = Nd if model == 'transh' else None
    old = L.set_option('kg_wtab', wtab)
    try:
        for head in (True, False):
            for desc in (False, True):
                a = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))
                b = ops().eval_kg_ranks(Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, desc, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids), fused=False)
                assert torch.equal(a, b)
    finally:
        L.set_option('kg_wtab', old)


def _referee_ranks(E, R, N, q, r, head, g_off, g_ids, f_off, f_ids, dtype, chunk=256):
    """Filtered gold ranks (utils/misc.py:125-146: filtered ids and the other golds are skipped) from the reference's OWN score formula
    -- sum_k (c_k - e_k)^2 with c = t - r / h + r (transE.py:65-105), both sides projected for TransH (transH.py:73-121) -- evaluated
    on the device in `dtype`, ascending, ids break exact ties.  float64: the referee; float32: what the reference's arithmetic gives."""
    Ed, Rd = E.to(DEV).to(dtype), R.to(DEV).to(dtype)
    Nd = None if N is None else N.to(DEV).to(dtype)
    ne, nq = Ed.shape[0], len(q)
    ids = torch.arange(ne, device=DEV)
    qd, rd = q.to(DEV), r.to(DEV)
    g_off_t, f_off_t = torch.from_numpy(g_off), torch.from_numpy(f_off)
    g_ids_d, f_ids_d = torch.from_numpy(g_ids).to(DEV).long(), torch.from_numpy(f_ids).to(DEV).long()
    out = torch.empty(len(g_ids), dtype=torch.int32, device=DEV)
    for s in range(0, nq, chunk):
        e = min(nq, s + chunk)
        qe, rr = Ed[qd[s:e]], Rd[rd[s:e]]
        if Nd is None:
            c = qe - rr if head else qe + rr
            S = ((c[:, None, :] - Ed[None]) ** 2).sum(-1)
        else:
            w = Nd[rd[s:e]]
            pq = qe - (qe * w).sum(-1, keepdim=True) * w
            c = pq - rr if head else pq + rr
            pe = Ed[None] - (Ed[None] * w[:, None, :]).sum(-1, keepdim=True) * w[:, None, :]
            S = ((c[:, None, :] - pe) ** 2).sum(-1)
        g0, g1, f0, f1 = int(g_off[s]), int(g_off[e]), int(f_off[s]), int(f_off[e])
        grow = torch.repeat_interleave(torch.arange(e - s), g_off_t[s + 1:e + 1] - g_off_t[s:e]).to(DEV)
        frow = torch.repeat_interleave(torch.arange(e - s), f_off_t[s + 1:e + 1] - f_off_t[s:e]).to(DEV)
        gid, fid = g_ids_d[g0:g1], f_ids_d[f0:f1]
        filt = torch.zeros(e - s, ne, dtype=torch.bool, device=DEV)
        filt[frow, fid] = True
        excl = filt.clone()
        excl[grow, gid] = True
        gs = S[grow, gid]
        below = (S[grow] < gs[:, None]) | ((S[grow] == gs[:, None]) & (ids[None] < gid[:, None]))
        rank = (below & ~excl[grow]).sum(1).to(torch.int32)
        rank[filt[grow, gid]] = -1
        out[g0:g1] = rank
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['transe', 'transh'])
@pytest.mark.parametrize('head', [True, False])
def test_kg_ranks_are_those_of_the_exact_scores(model, head):
    """BASELINE configs[1] at full size: 20,480 keys x 14,709 entities, d = 100, squared L2, two golds and ~20 filtered ids per key.
    The pass without the score matrix scores through dot products on the matrix cores, the reference sums squared differences; near a
    gold the two round differently.  EVERY rank of the pass must be the rank under the exact scores of the fp32 tables (fp64 referee,
    ids on exact ties) -- frac_equal == 1.0; the reference's formula in fp32 is compared with the same referee to show what its own
    rounding loses, and without the referee (option kg_exact = 0) the pass is allowed its round-3 error: a few ranks off by one."""
    rng = np.random.RandomState(41 + int(head))
    ne, nr, d, nq = 14709, 20, 100, 20480
    gen = torch.Generator().manual_seed(9)
    E, R, N = O.make_table(ne, d, gen), O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    if model == 'transe':
        N = None
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    g1 = rng.randint(0, ne, size=nq); g2 = (g1 + 1 + rng.randint(0, ne - 1, size=nq)) % ne
    gold = np.sort(np.stack([g1, g2], 1), 1).astype(np.int32)
    g_off, g_ids = np.arange(nq + 1, dtype=np.int64) * 2, gold.reshape(-1)
    filt = [np.unique(rng.randint(0, ne, size=22)).astype(np.int32) for _ in range(nq)]
    filt[5] = np.unique(np.concatenate([filt[5], gold[5, :1]])).astype(np.int32)          # a gold that is itself filtered: rank -1
    f_off = np.concatenate([[0], np.cumsum([len(x) for x in filt])]).astype(np.int64); f_ids = np.concatenate(filt).astype(np.int32)
    Ed, Rd = dv(E.numpy()), dv(R.numpy())
    Nn = None if N is None else dv(N.numpy())
    args = (Ed, Rd, Nn, q.to(DEV), r.to(DEV), False, head, False, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids))
    got = ops().eval_kg_ranks(*args)[:len(g_ids)]
    want = _referee_ranks(E, R, N, q, r, head, g_off, g_ids, f_off, f_ids, torch.float64)
    assert int(want[2 * 5 + int(np.searchsorted(gold[5], gold[5, 0]))]) == -1 and int((want >= 0).sum()) >= len(g_ids) - 400
    n_bad = int((got != want).sum())
    assert n_bad == 0, 'ranks differ from the exact order at %d of %d gold entries (max %d)' % (n_bad, len(g_ids), int((got - want).abs().max()))
    ref32 = _referee_ranks(E, R, N, q, r, head, g_off, g_ids, f_off, f_ids, torch.float32)
    with _opt('kg_exact', 0):
        raw = ops().eval_kg_ranks(*args)[:len(g_ids)]
    lost32, lost_raw = int((ref32 != want).sum()), int((raw != want).sum())
    print('exact-order check (%s, head=%s): device == fp64 referee at all %d entries; the reference formula in fp32 loses %d, the pass without '
          'the referee %d' % (model, head, len(g_ids), lost32, lost_raw))
    assert lost_raw <= 200 and int((raw - want).abs().max()) <= 2 and lost32 <= 200


@pytest.mark.gpu
@pytest.mark.parametrize('d,l1', [(64, False), (100, False), (36, False), (64, True)])
def test_kg_ranks_whole_pass_transr(d, l1):
    """ktup_eval_kg_ranks_transr (TransR's pass under the C ABI: K14 per chunk of 512 keys against the once-prepared entity side +
    K18, the rank kernel of a chunk on a second stream beside the next chunk's scores) returns the integers of the per-batch
    route: 1,300 keys = two full chunks + a ragged one, head and tail, with and without a prepared entity side."""
    rng = np.random.RandomState(29)
    ne, nr, nq = 500, 5, 1300
    gen = torch.Generator().manual_seed(6)
    E, R = O.make_table(ne, d, gen), O.make_table(nr, d, gen)
    M = torch.randn(nr, d * d, generator=gen) * 0.1
    q = torch.from_numpy(rng.randint(0, ne, size=nq)); r = torch.from_numpy(rng.randint(0, nr, size=nq))
    _, filt, gold, f_off, f_ids, g_off, g_ids = _rank_case(rng, nq, ne, 40, 5, 0)
    Ed, Rd, Md = dv(E.numpy()), dv(R.numpy()), dv(M.numpy())
    qd, rd = q.to(DEV), r.to(DEV)
    ents = ops().eval_transr_entities(Ed, Md, nr, l1)
    for head in (True, False):
        for use_ents in (True, False):
            got = ops().eval_kg_ranks_transr(Ed, Rd, Md, qd, rd, l1, head, False, dv(g_off), dv(g_ids), dv(f_off), dv(f_ids),
                                             ents=ents if use_ents else None)
            want = []
            for s in range(0, nq, 512):
                e = min(nq, s + 512)
                sc = ops().eval_transr(Ed, Rd, Md, qd[s:e], rd[s:e], l1, head, ents=ents if use_ents else None)
                lo = int(g_off[s])
                want.append(ops().gold_ranks(sc, False, dv(g_off[s:e + 1] - lo), dv(g_ids[lo:]), dv(f_off[s:e + 1] - f_off[s]),
                                             dv(f_ids[int(f_off[s]):]))[:int(g_off[e]) - lo])
            assert torch.equal(got[:len(g_ids)], torch.cat(want))


@pytest.mark.gpu
@pytest.mark.parametrize('want_rows', [False, True])
def test_kg_eval_pass_whole_pass_route_equals_the_batch_walk(want_rows, monkeypatch):
    """_driver.kg_eval_pass with a model's rank_entities (scores + filtered gold ranks of the WHOLE pass behind one call) returns
    exactly what the walk over the batches returns -- the (hit, rank) array of the periodic evaluations and the report rows
    (hit, rank, key, gold id) -- for TransH through an entity remap, keys without golds, golds that are filtered, and with
    KTUP_EVAL_PASS=0 it steps aside."""
    import types
    from jTransUP.models import _driver as D
    from jTransUP.models import transH
    rng = np.random.RandomState(8)
    torch.manual_seed(2)
    ne, nr, nq = 600, 6, 700
    m = transH.TransHModel(False, 64, ne, nr).to(DEV)
    m.eval(); m.disable_grad()
    FL = types.SimpleNamespace(topn=10)
    remap = {e: (e * 7) % ne for e in range(ne)}                  # a permutation of the entity ids (e_map of the joint drivers)
    keys = [(int(rng.randint(ne)), int(rng.randint(nr))) for _ in range(nq)]
    keys = list(dict.fromkeys(keys))
    gold = {k: set(rng.choice(ne, size=rng.randint(1, 5), replace=False).tolist()) for k in keys if k[0] % 11}
    filt = {k: set(rng.choice(ne, size=30, replace=False).tolist()) | (set(list(gold[k])[:1]) if k in gold and k[0] % 3 == 0 else set()) for k in keys}
    batches = [keys[s:s + 128] for s in range(0, len(keys), 128)]
    score_fn = lambda q, r: m.evaluateTail(q, r)
    rank_fn = lambda q, r, desc, go, gi, fo, fi: m.rank_entities(q, r, False, desc, go, gi, fo, fi)
    for desc in (False, True):
        walk = D.kg_eval_pass(FL, score_fn, batches, gold, [filt], desc, remap=remap, want_rows=want_rows)
        fused = D.kg_eval_pass(FL, score_fn, batches, gold, [filt], desc, remap=remap, want_rows=want_rows, rank_fn=rank_fn)
        if want_rows:
            assert fused == walk and len(walk) > 0
        else:
            assert fused.shape == walk.shape and walk.shape[0] > 0
            np.testing.assert_array_equal(fused, walk)
    calls = []
    monkeypatch.setenv('KTUP_EVAL_PASS', '0')
    D.kg_eval_pass(FL, score_fn, batches, gold, [filt], False, remap=remap, want_rows=want_rows,
                   rank_fn=lambda *a: calls.append(1))
    assert not calls


@pytest.mark.gpu
def test_fused_rec_pass_replayed_as_a_graph_follows_the_tables(monkeypatch):
    """_driver._rec_eval_fused with a graph key: the first pass runs eagerly, the second is captured (item side, sweep, merge,
    metrics, copy into a pinned buffer) and every later one is a replay -- which must see the tables as they are NOW (the
    optimizer updates them in place between evaluations).  Every pass equals the eager route on the same tables; a model whose
    tables moved gets its own graph; KTUP_EVAL_GRAPH=0 keeps everything eager."""
    import types
    from jTransUP.models import _driver as D
    from jTransUP.models import jTransUP as jt
    torch.manual_seed(5)
    rng = np.random.RandomState(5)
    NU, NI, NE, NR, Dm = 300, 177, 150, 7, 100
    i_map = {i: i for i in range(NI)}
    new_map = {i: ((i * 3) % NE if i % 5 else -1, i) for i in range(NI)}
    m = jt.jTransUPModel(False, Dm, NU, NI, NE, NR, i_map, new_map, False, False)
    m.eval(); m.disable_grad()
    FL = types.SimpleNamespace(topn=10)
    users = list(range(NU))
    gold = {u: set(rng.choice(NI, size=rng.randint(1, 9), replace=False).tolist()) for u in users if u % 9}
    train = {u: set(rng.choice(NI, size=25, replace=False).tolist()) for u in users}
    batches = [users[s:s + 64] for s in range(0, NU, 64)]
    index = D.rank_index(batches, gold, [train])
    pass_fn = lambda u, fo, fi, n: m.evaluate_topk(u, m.prepare_items(), n, fo, fi)
    D._EVAL_GRAPHS.clear()
    for step in range(5):
        key = D.model_graph_key(m)
        got = D._rec_eval_fused(FL, pass_fn, batches, index, key)
        want = D._rec_eval_fused(FL, pass_fn, batches, index, None)
        assert got.shape == (sum(1 for u in users if u % 9), 5)
        np.testing.assert_array_equal(got, want)
        entry = D._EVAL_GRAPHS[(id(batches), id(index), 10, key)]
        assert (entry[0] is None) == (step == 0)                  # eager once, then a graph
        with torch.no_grad():                                      # a training step's worth of change, in place
            for p in m.parameters():
                p.add_(torch.randn_like(p) * 0.05)
    moved = m.user_embeddings.weight.data.clone()
    m.user_embeddings.weight.data = moved                          # the table now lives elsewhere: another key, another graph
    assert D.model_graph_key(m) != key
    np.testing.assert_array_equal(D._rec_eval_fused(FL, pass_fn, batches, index, D.model_graph_key(m)),
                                  D._rec_eval_fused(FL, pass_fn, batches, index, None))
    monkeypatch.setenv('KTUP_EVAL_GRAPH', '0')
    n = len(D._EVAL_GRAPHS)
    D._rec_eval_fused(FL, pass_fn, batches, index, ('other',))
    assert len(D._EVAL_GRAPHS) == n


@pytest.mark.gpu
@pytest.mark.parametrize('ktup', [False, True])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('d', [50, 300, 256, 200])
def test_pref_eval_pass_any_embedding_size(ktup, l1, d):
    """-embedding_size 50 (the reference takes any integer, models/base.py:52): the item side prepared once per pass, the per-batch
    scores and the one-sweep filtered top-n all run on rows staged with a zero tail -- the oracle's scores, and the ids of the
    matrix route.  -embedding_size 300: the one-wave-per-pair forward (ktup_score_pref_row.hip) makes the per-batch scores.
    200 / 256: the soft gate's sweep and the per-batch pair kernel with two of their three item arrays staged (HybridCand)."""
    nu, ni, ne, P, nq, topn = 90, 211, 150, 6, 37, 10
    W, i2e, gen = world(5, nu, ni, ne, P, d)
    D = {k: v.to(DEV) for k, v in W.items()}
    u = torch.randint(0, nu, (nq,), generator=gen)
    rng = np.random.RandomState(3)
    filt = [np.sort(rng.choice(ni, size=rng.randint(0, 20), replace=False)).astype(np.int32) for _ in range(nq)]
    f_off = dv(np.concatenate([[0], np.cumsum([len(x) for x in filt])]).astype(np.int64)); f_ids = dv(np.concatenate(filt).astype(np.int32))
    i2e_d = i2e.to(DEV, torch.int32)
    items = ops().eval_pref_items(D['I'], D['E'] if ktup else None, D['P'], D['Pn'], D['R'] if ktup else None, D['Rn'] if ktup else None,
                                  i2e_d if ktup else None)
    if ktup:
        mat = ops().eval_ktup(D['U'], D['I'], D['E'], D['P'], D['Pn'], D['R'], D['Rn'], i2e_d, u.to(DEV), l1, items=items)
        want = O.eval_ktup_rec(W['U'], W['I'], W['E'], W['P'], W['Pn'], W['R'], W['Rn'], i2e, u, l1)
    else:
        mat = ops().eval_tup(D['U'], D['I'], D['P'], D['Pn'], u.to(DEV), l1, items=items)
        want = O.eval_tup(W['U'], W['I'], W['P'], W['Pn'], u, l1)
    close(mat, want)
    ids = ops().topk_filtered(mat, False, topn, f_off, f_ids)
    got = ops().eval_pref_topk(D['U'], u.to(DEV), items, l1, topn, f_off, f_ids)
    if d > 256:          # no one-sweep pass beyond 256 columns: the drivers keep the per-batch scores + topk_filtered
        assert got is None
    else:
        assert got is not None and torch.equal(got, ids)
