#!/usr/bin/env python3
"""Randomized LIVE comparison of oracle/cpu_ref.py with the reference itself (build container only: it imports /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/live_check.py [--worlds 6] [--seed 0] [--ref /root/reference]

The goldens pin the oracle on a fixed toy world; this draws random worlds -- table sizes, widths (incl. widths that are not a
multiple of 4), batch sizes, weights, index batches, recorded Gumbel uniforms, gold / filter sets -- runs the REFERENCE's
modules on them (through make_goldens.py's harness-side shims; no reference file is touched or copied) and the oracle on the
same inputs, and reports the worst disagreement per family: forward scores (bprmf.py, transE/H/R.py, transUP.py, jTransUP.py),
all-candidate evaluation matrices (evaluate / evaluateRec / evaluateHead / evaluateTail), the drivers' step losses with every
table's gradient (utils/loss.py + the loss assembly of the three drivers) and the ranking walks
(utils/misc.py getRecPerformance / getKGPerformance).  Prints ONE JSON line; tests/test_oracle_live_reference.py runs it where
/root/reference exists (it cannot exist on the GPU box) and holds the numbers to the goldens' bars."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
argv = sys.argv[1:]


def opt(name, default):
    return argv[argv.index(name) + 1] if name in argv else default


WORLDS, SEED, REF = int(opt('--worlds', 6)), int(opt('--seed', 0)), opt('--ref', '/root/reference')
sys.argv = [sys.argv[0], '--ref', REF, '--out', os.devnull]      # make_goldens parses its own command line on import
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))       # repo root: oracle/
import numpy as np                                               # noqa: E402
import make_goldens as MG                                        # noqa: E402  (imports the reference + installs shims 1, 2)
import torch                                                     # noqa: E402
from oracle import cpu_ref as O                                  # noqa: E402

V, npy = MG.V, MG.npy
worst = {}


def note(family, got, want, rtol=1e-5, atol=1e-6):
    """Relative excess over the bar |got - want| <= atol + rtol |want| (<= 1 passes), kept per family."""
    got = got.detach().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (family, got.shape, want.shape)
    if got.size == 0:
        return
    ex = float((np.abs(got.astype(np.float64) - want) / (atol + rtol * np.abs(want))).max())
    worst[family] = max(worst.get(family, 0.0), ex)


def params(m, names):
    sd = dict(m.named_parameters())
    return [sd[n].data.clone() for n in names]


def leafs(W):
    return [w.clone().requires_grad_(True) for w in W]


def step_family(family, m, names, loss_ref, W, loss_oracle, pad_ent=None):
    """Loss value and every table's gradient: the reference's modules + utils/loss.py against the oracle's step-loss assembly."""
    m.zero_grad()
    loss_ref.backward()
    L = loss_oracle(*W)
    L.backward()
    note(family + '.loss', L.detach(), npy(loss_ref), rtol=2e-5)
    sd = dict(m.named_parameters())
    for w, n in zip(W, names):
        want = sd[n].grad
        got = w.grad if w.grad is not None else torch.zeros_like(w)
        if pad_ent == n:
            got = got.clone(); got[-1].zero_()                   # nn.Embedding's padding row takes no gradient (jTransUP.py:96)
        note(family + '.grad', got, npy(want) if want is not None else np.zeros(tuple(w.shape), np.float32), rtol=1e-4, atol=1e-6)


TUP = ['user_embeddings.weight', 'item_embeddings.weight', 'pref_embeddings.weight', 'pref_norm_embeddings.weight']
KTUP = ['user_embeddings.weight', 'item_embeddings.weight', 'ent_embeddings.weight', 'pref_embeddings.weight',
        'pref_norm_embeddings.weight', 'rel_embeddings.weight', 'norm_embeddings.weight']


def world(rng, gen, wi):
    NU, NI, NE, NR = (int(rng.randint(lo, hi)) for lo, hi in ((3, 50), (4, 60), (5, 70), (2, 9)))
    NP = int(rng.randint(2, 8))
    d = int(rng.choice([6, 20, 50, 64, 100, 130]))
    B, BQ = int(rng.randint(1, 33)), int(rng.randint(1, 9))
    ids = lambda n, k: torch.from_numpy(rng.randint(0, n, k)).long()
    u, pi = ids(NU, B), ids(NI, B)
    h, t, r = ids(NE, B), ids(NE, B), ids(NR, B)
    nh, nt, ni = ids(NE, B), ids(NE, B), ids(NI, B)
    uq, eq, rq = ids(NU, BQ), ids(NE, BQ), ids(NR, BQ)

    m = MG.bprmf.BPRMF(d, NU, NI); MG.set_weights(m, gen)
    W = params(m, ['user_embeddings.weight', 'item_embeddings.weight'])
    note('bprmf.score', O.score_bprmf(*W, u, pi), npy(m(V(u), V(pi))))
    note('bprmf.eval', O.eval_bprmf(*W, uq), npy(m.evaluate(V(uq))))

    for name, mod, cls, extra in (('transe', MG.transE, 'TransEModel', None), ('transh', MG.transH, 'TransHModel', 'norm_embeddings.weight'),
                                  ('transr', MG.transR, 'TransRModel', 'proj_embeddings.weight')):
        if name == 'transr' and d > 64:
            continue                                             # (a d x d projection per relation: keep the live run short)
        for l1 in (False, True):
            m = getattr(mod, cls)(l1, d, NE, NR); MG.set_weights(m, gen)
            W = params(m, ['ent_embeddings.weight', 'rel_embeddings.weight'] + ([extra] if extra else []))
            sc = {'transe': O.score_transe, 'transh': O.score_transh, 'transr': O.score_transr}[name]
            ev = {'transe': O.eval_transe, 'transh': O.eval_transh, 'transr': O.eval_transr}[name]
            tol = dict(rtol=1e-4, atol=1e-5) if name == 'transr' else {}
            note(name + '.score', sc(*W, h, t, r, l1), npy(m(V(h), V(t), V(r))), **tol)
            note(name + '.eval', ev(*W, eq, rq, l1, True), npy(m.evaluateHead(V(eq), V(rq))), **tol)
            note(name + '.eval', ev(*W, eq, rq, l1, False), npy(m.evaluateTail(V(eq), V(rq))), **tol)
            if name != 'transr':                                 # the step loss of knowledge_representation.py:189-204
                pos, neg = m(V(h), V(t), V(r)), m(V(nh), V(nt), V(r))
                loss = MG.rloss.marginLoss()(pos, neg, 1.0)
                ent = m.ent_embeddings(V(torch.cat([h, t, nh, nt]))); rel = m.rel_embeddings(V(torch.cat([r, r])))
                if name == 'transh':
                    loss = loss + MG.rloss.orthogonalLoss(rel, m.norm_embeddings(V(torch.cat([r, r]))))
                loss = loss + MG.rloss.normLoss(ent) + MG.rloss.normLoss(rel)
                names = ['ent_embeddings.weight', 'rel_embeddings.weight'] + ([extra] if extra else [])
                step_family(name + '.step', m, names, loss, leafs(W),
                            lambda E_, R_, N_=None: O.kg_step_loss(E_, R_, N_, h, t, r, nh, nt, r, l1))

    for l1 in (False, True):
        for gum in (False, True):
            m = MG.transUP.TransUPModel(l1, d, NU, NI, NP, gum); MG.set_weights(m, gen)
            W = params(m, TUP)
            s1, s2 = 1000 + 10 * wi, 1001 + 10 * wi
            uni = MG.uniforms(s1, (B, NP)) if gum else None      # shim 4: the draw the reference is about to make
            torch.manual_seed(s1)
            note('tup.score', O.score_tup(*W, u, pi, l1, uni), npy(m(V(u), V(pi))))
            uni = MG.uniforms(s2, (BQ, NI, NP)) if gum else None
            torch.manual_seed(s2)
            note('tup.eval', O.eval_tup(*W, uq, l1, uni), npy(m.evaluate(V(uq))))
            s3, s4 = 1002 + 10 * wi, 1003 + 10 * wi                # the step loss of item_recommendation.py:171-180
            up, un = (MG.uniforms(s3, (B, NP)), MG.uniforms(s4, (B, NP))) if gum else (None, None)
            torch.manual_seed(s3); pos = m(V(u), V(pi))
            torch.manual_seed(s4); neg = m(V(u), V(ni))
            loss = MG.rloss.bprLoss(pos, neg, target=-1) + MG.rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight) \
                + MG.rloss.normLoss(m.user_embeddings(V(u))) + MG.rloss.normLoss(m.item_embeddings(V(torch.cat([pi, ni])))) \
                + MG.rloss.normLoss(m.pref_embeddings.weight)
            step_family('tup.step', m, TUP, loss, leafs(W), lambda *w: O.tup_rec_step_loss(*w, u, pi, ni, l1, -1.0, up, un))

    i_map = MG.IntKeyDict({i: i for i in range(NI)})
    new_map = {i: ((int(rng.randint(0, NE)) if rng.rand() < 0.7 else -1), i) for i in range(NI)}
    for l1 in (False, True):
        for gum in (False, True):
            m = MG.jtup.jTransUPModel(l1, d, NU, NI, NE, NR, i_map, new_map, False, gum); MG.set_weights(m, gen)
            W = params(m, KTUP)
            i2e = torch.from_numpy(np.asarray(m.paddingItems(torch.arange(NI), m.ent_total - 1), dtype=np.int64))
            s1, s2 = 2000 + 10 * wi, 2001 + 10 * wi
            uni = MG.uniforms(s1, (B, NR)) if gum else None
            torch.manual_seed(s1)
            note('ktup.score', O.score_ktup_rec(*W, i2e, u, pi, l1, uni), npy(m((V(u), V(pi)), None, is_rec=True)))
            uni = MG.uniforms(s2, (BQ, NI, NR)) if gum else None
            torch.manual_seed(s2)
            note('ktup.eval', O.eval_ktup_rec(*W, i2e, uq, l1, uni), npy(m.evaluateRec(V(uq))))
            s3, s4 = 2002 + 10 * wi, 2003 + 10 * wi                # the rec step loss of knowledgable_recommendation.py:335-344
            up, un = (MG.uniforms(s3, (B, NR)), MG.uniforms(s4, (B, NR))) if gum else (None, None)
            torch.manual_seed(s3); pos = m((V(u), V(pi)), None, is_rec=True)
            torch.manual_seed(s4); neg = m((V(u), V(ni)), None, is_rec=True)
            loss = MG.rloss.bprLoss(pos, neg, target=-1) + MG.rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight)
            step_family('ktup.step', m, KTUP, loss, leafs(W), lambda *w: O.ktup_rec_step_loss(*w, i2e, u, pi, ni, l1, -1.0, up, un),
                        pad_ent='ent_embeddings.weight')
            if not gum:
                E, R, Rn = W[2], W[5], W[6]
                note('ktup.score', O.score_ktup_kg(E, R, Rn, h, t, r, l1), npy(m(None, (V(h), V(t), V(r)), is_rec=False)))
                note('ktup.eval', O.eval_transh(E, R, Rn, eq, rq, l1, True), npy(m.evaluateHead(V(eq), V(rq))))
                note('ktup.eval', O.eval_transh(E, R, Rn, eq, rq, l1, False), npy(m.evaluateTail(V(eq), V(rq))))

    # the two baselines that share this path's kernels (CKE.py, CFKG.py): rec scores and all-item evaluation
    from jTransUP.models import CKE as rcke, CFKG as rcfkg           # (the REFERENCE's, like every jTransUP import here)
    for l1 in (False, True):
        if d <= 64:                                              # CKE carries TransR's d x d projections
            m = rcke.CKE(l1, d, NU, NI, NE, NR, i_map, new_map); MG.set_weights(m, gen)
            U_, I_, E_ = params(m, ['user_embeddings.weight', 'item_embeddings.weight', 'ent_embeddings.weight'])
            i2e = torch.from_numpy(np.asarray(m.paddingItems(torch.arange(NI), m.ent_total - 1), dtype=np.int64))
            note('cke.score', O.score_cke_rec(U_, I_, E_, i2e, u, pi), npy(m((V(u), V(pi)), None, is_rec=True)))
            note('cke.eval', O.eval_cke_rec(U_, I_, E_, i2e, uq), npy(m.evaluateRec(V(uq))))
        n_ent = max(NE, NI)                                      # CFKG: item ids index the shared entity table
        m = rcfkg.CFKG(l1, d, NU, NI, n_ent, NR); MG.set_weights(m, gen)
        U_, E_, R_ = params(m, ['user_embeddings.weight', 'ent_embeddings.weight', 'rel_embeddings.weight'])
        note('cfkg.score', O.score_cfkg_rec(U_, E_, R_, u, pi, l1), npy(m((V(u), V(pi)), None, is_rec=True)))
        note('cfkg.eval', O.eval_cfkg_rec(U_, E_, R_, uq, l1), npy(m.evaluateRec(V(uq))))

    # the ranking walks on random rows with repeated scores (stable argsort: shim 3, the tie rule the build declares)
    real_argsort = np.argsort
    np.argsort = lambda a, *aa, **kw: real_argsort(a, *aa, **dict(kw, kind='stable'))
    mism = 0
    try:
        for _ in range(8):
            n = int(rng.randint(3, 80))
            row = np.round(rng.randn(n), int(rng.randint(0, 3))).astype(np.float32)      # coarse values: ties happen
            perm = rng.permutation(n)
            ng, nf = int(rng.randint(1, min(6, n))), int(rng.randint(0, n // 2 + 1))
            gold, filt = set(int(x) for x in perm[:ng]), set(int(x) for x in perm[ng:ng + nf])
            topn = int(rng.randint(1, 12))
            a = MG.rmisc.getRecPerformance(row, gold, fliter_samples=filt, topn=topn)
            b = O.rec_performance(row, gold, filt, topn)
            mism += int([float(x) for x in a[:5]] != [float(x) for x in b[:5]] or [int(x) for x in a[5]] != [int(x) for x in b[5]])
            a = MG.rmisc.getKGPerformance(row, gold, fliter_samples=filt, topn=topn)
            b = O.kg_performance(row, gold, filt, topn)
            mism += int([[int(x) for x in part] for part in a] != [[int(x) for x in part] for part in b])
    finally:
        np.argsort = real_argsort
    worst['ranking.mismatches'] = worst.get('ranking.mismatches', 0) + mism


def main():
    rng = np.random.RandomState(SEED)
    gen = torch.Generator().manual_seed(SEED + 1)
    for wi in range(WORLDS):
        world(rng, gen, wi)
    print(json.dumps({'worlds': WORLDS, 'seed': SEED, 'worst_excess_over_bar': worst}))


if __name__ == '__main__':
    main()
