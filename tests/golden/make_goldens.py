#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [--ref /root/reference]

The reference's Python cannot travel to the GPU box, so what is committed is data only: inputs
(weights, index batches, recorded Gumbel uniforms, filter/gold sets) and the outputs the reference
produced for them (scores, losses, gradients, eval matrices, ranked id lists, metrics).

Harness-side shims (none touches reference files):
  1. i_map is a dict subclass whose __getitem__ does int(k): jTransUPModel.paddingItems iterates a
     LongTensor and uses its elements as dict keys (jTransUP/models/jTransUP.py:116-117).
  2. np.asfarray was removed in NumPy 2 (jTransUP/utils/evaluation.py:69).
  3. np.argsort is forced to kind='stable' while the ranking goldens are produced, which pins the tie
     rule (ascending score, then ascending id) the build declares; on tie-free rows it is a no-op.
  4. The Gumbel uniforms the reference draws from torch's global generator are recovered by re-seeding
     and re-drawing a tensor of the same shape (torch.manual_seed(s); torch.empty(shape).uniform_()).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True

ap = argparse.ArgumentParser()
ap.add_argument('--ref', default='/root/reference')
ap.add_argument('--out', default=os.path.dirname(os.path.abspath(__file__)))
args = ap.parse_args()
sys.path.insert(0, args.ref)

if not hasattr(np, 'asfarray'):
    np.asfarray = lambda a: np.asarray(a, dtype=np.float64)   # shim 2

import warnings
warnings.filterwarnings('ignore')
import torch
from torch.autograd import Variable as V

from jTransUP.models import bprmf, transE, transH, transR, transUP, jTransUP as jtup   # the REFERENCE
from jTransUP.utils import loss as rloss
from jTransUP.utils import misc as rmisc
from jTransUP.utils import evaluation as reval
from jTransUP.data import load_kg_rating_data as rkgload

torch.set_num_threads(1)


class IntKeyDict(dict):                                       # shim 1
    def __getitem__(self, k):
        return dict.__getitem__(self, int(k))


def npy(t):
    return t.detach().cpu().numpy().copy()


def uniforms(seed, shape):                                    # shim 4
    torch.manual_seed(seed)
    return torch.empty(*shape).uniform_()


def save(name, **arrs):
    path = os.path.join(args.out, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-28s %7.1f KB  %d arrays' % (name + '.npz', os.path.getsize(path) / 1024, len(arrs)))


# ------------------------------------------------------------------ toy world shared by all cases
NU, NI, NE, NR = 37, 45, 53, 7           # users, items, entities, relations (== preferences for KTUP)
NP_TUP = 5                               # TUP preference count (differs from NR on purpose)
B = 48
ALIGNED = 33                             # items 0..44, of which 33 are aligned with an entity


def make_alignment(rng):
    """i_map: item id -> new index, new_map: new index -> (ent id | -1, item id | -1) built by the
    reference's own rebuildEntityItemVocab from toy vocab files' contents."""
    e_vocab = {'uri%d' % e: e for e in range(NE)}                         # org -> mapped id
    i_vocab = {'item%d' % i: i for i in range(NI)}
    aligned_items = rng.permutation(NI)[:ALIGNED]
    aligned_ents = rng.permutation(NE)[:ALIGNED]
    kg2i = {'uri%d' % int(e): 'item%d' % int(i) for e, i in zip(aligned_ents, aligned_items)}
    new_map, e_remap, i_remap, n_aligned = rkgload.rebuildEntityItemVocab(e_vocab, i_vocab, kg2i)
    return e_vocab, i_vocab, kg2i, new_map, e_remap, i_remap, n_aligned


def set_weights(model, gen, scale_rows=True):
    """Overwrite every table with seeded values (row norms deliberately spread around 1 so that
    normLoss and the projections are exercised away from the unit sphere)."""
    sd = {}
    for name, p in model.named_parameters():
        w = torch.randn(p.shape, generator=gen) * 0.3
        if name.startswith('proj_embeddings'):
            w = torch.randn(p.shape, generator=gen) * 0.15
        p.data.copy_(w)
        sd[name] = npy(p.data)
    if hasattr(model, 'ent_embeddings') and getattr(model.ent_embeddings, 'padding_idx', None) is not None:
        model.ent_embeddings.weight.data[model.ent_embeddings.padding_idx].zero_()
        sd['ent_embeddings.weight'] = npy(model.ent_embeddings.weight.data)
    return sd


def grads_of(model):
    return {'grad.' + n: npy(p.grad) for n, p in model.named_parameters() if p.grad is not None}


def zero_grads(model):
    for p in model.parameters():
        p.grad = None


def score_cases():
    rng = np.random.RandomState(11)
    gen = torch.Generator().manual_seed(5)
    e_vocab, i_vocab, kg2i, new_map, e_remap, i_remap, n_aligned = make_alignment(rng)
    i_map = IntKeyDict(i_remap)
    pad = NE

    for d in (64, 100, 36, 256):          # 256 (config 5's size) last: the earlier files keep their random draws
        out = {}
        u = torch.from_numpy(rng.randint(0, NU, B)).long()
        pi = torch.from_numpy(rng.randint(0, NI, B)).long()
        ni = torch.from_numpy(rng.randint(0, NI, B)).long()
        ph = torch.from_numpy(rng.randint(0, NE, B)).long()
        pt = torch.from_numpy(rng.randint(0, NE, B)).long()
        pr = torch.from_numpy(rng.randint(0, NR, B)).long()
        nh = torch.from_numpy(rng.randint(0, NE, B)).long()
        nt = torch.from_numpy(rng.randint(0, NE, B)).long()
        out.update(u=npy(u), pi=npy(pi), ni=npy(ni), ph=npy(ph), pt=npy(pt), pr=npy(pr), nh=npy(nh), nt=npy(nt))

        # ---- BPRMF (bprmf.py) : bprLoss target = +1 (trainer.py:15-17)
        m = bprmf.BPRMF(d, NU, NI)
        sd = set_weights(m, gen)
        out.update({'bprmf.' + k: v for k, v in sd.items()})
        pos, neg = m(V(u), V(pi)), m(V(u), V(ni))
        loss = rloss.bprLoss(pos, neg, target=1)
        zero_grads(m); loss.backward()
        out.update({'bprmf.pos': npy(pos), 'bprmf.neg': npy(neg), 'bprmf.loss': npy(loss)})
        out.update({'bprmf.' + k: v for k, v in grads_of(m).items()})

        # ---- TransE / TransH / TransR : marginLoss + normLoss (+ orthogonalLoss for transh)
        #      exactly as knowledge_representation.py:189-204 assembles the loss
        for name, mod, cls in (('transe', transE, 'TransEModel'), ('transh', transH, 'TransHModel'), ('transr', transR, 'TransRModel')):
            if name == 'transr' and d == 256:
                continue                   # a 7 x 65536 projection table plus its gradients would make the fixture 6 MB
            for l1 in (False, True):
                tag = '%s.%s.' % (name, 'L1' if l1 else 'L2')
                m = getattr(mod, cls)(l1, d, NE, NR)
                if not l1:
                    sd = set_weights(m, gen)
                    out.update({name + '.' + k: v for k, v in sd.items()})
                    keep = {k: p.data.clone() for k, p in m.named_parameters()}
                else:
                    for k, p in m.named_parameters():
                        p.data.copy_(keep[k])
                pos, neg = m(V(ph), V(pt), V(pr)), m(V(nh), V(nt), V(pr))
                loss = rloss.marginLoss()(pos, neg, 1.0)
                ent = m.ent_embeddings(V(torch.cat([ph, pt, nh, nt])))
                rel = m.rel_embeddings(V(torch.cat([pr, pr])))
                if name == 'transh':
                    nrm = m.norm_embeddings(V(torch.cat([pr, pr])))
                    loss = loss + rloss.orthogonalLoss(rel, nrm)
                loss = loss + rloss.normLoss(ent) + rloss.normLoss(rel)
                zero_grads(m); loss.backward()
                out.update({tag + 'pos': npy(pos), tag + 'neg': npy(neg), tag + 'loss': npy(loss)})
                out.update({tag + k: v for k, v in grads_of(m).items()})

        # ---- TUP (transUP.py), soft and ST-Gumbel; loss as item_recommendation.py:171-180
        for l1 in (False, True):
            for gum in (False, True):
                tag = 'tup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
                m = transUP.TransUPModel(l1, d, NU, NI, NP_TUP, gum)
                if not l1 and not gum:
                    sd = set_weights(m, gen)
                    out.update({'tup.' + k: v for k, v in sd.items()})
                    keep = {k: p.data.clone() for k, p in m.named_parameters()}
                else:
                    for k, p in m.named_parameters():
                        p.data.copy_(keep[k])
                if gum:
                    seed_p, seed_n = 100 + d, 200 + d
                    out[tag + 'uni_pos'] = npy(uniforms(seed_p, (B, NP_TUP)))
                    out[tag + 'uni_neg'] = npy(uniforms(seed_n, (B, NP_TUP)))
                    torch.manual_seed(seed_p); pos = m(V(u), V(pi))
                    torch.manual_seed(seed_n); neg = m(V(u), V(ni))
                else:
                    pos, neg = m(V(u), V(pi)), m(V(u), V(ni))
                loss = rloss.bprLoss(pos, neg, target=-1)
                ue = m.user_embeddings(V(u)); ie = m.item_embeddings(V(torch.cat([pi, ni])))
                loss = loss + rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight) \
                    + rloss.normLoss(ue) + rloss.normLoss(ie) + rloss.normLoss(m.pref_embeddings.weight)
                zero_grads(m); loss.backward()
                out.update({tag + 'pos': npy(pos), tag + 'neg': npy(neg), tag + 'loss': npy(loss)})
                out.update({tag + k: v for k, v in grads_of(m).items()})
                if not gum:   # getPreferences outputs (probs, r_e, norm) for the reporting path
                    pr_, re_, no_ = m.getPreferences(m.user_embeddings(V(u)), m.item_embeddings(V(pi)), use_st_gumbel=False)
                    out.update({tag + 'pref.probs': npy(pr_), tag + 'pref.r_e': npy(re_), tag + 'pref.norm': npy(no_)})

        # ---- KTUP (jTransUP.py): rec branch (bpr + orthogonal, knowledgable_recommendation.py:337-344)
        #      and kg branch (margin + orth + norm, :368-383)
        for l1 in (False, True):
            for gum in (False, True):
                tag = 'ktup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
                m = jtup.jTransUPModel(l1, d, NU, NI, NE, NR, i_map, new_map, False, gum)
                if not l1 and not gum:
                    sd = set_weights(m, gen)
                    out.update({'ktup.' + k: v for k, v in sd.items()})
                    keep = {k: p.data.clone() for k, p in m.named_parameters()}
                    e_ids = m.paddingItems(torch.arange(NI), m.ent_total - 1)
                    out['ktup.item2ent'] = np.asarray(e_ids, dtype=np.int64)
                else:
                    for k, p in m.named_parameters():
                        p.data.copy_(keep[k])
                if gum:
                    seed_p, seed_n = 300 + d, 400 + d
                    out[tag + 'uni_pos'] = npy(uniforms(seed_p, (B, NR)))
                    out[tag + 'uni_neg'] = npy(uniforms(seed_n, (B, NR)))
                    torch.manual_seed(seed_p); pos = m((V(u), V(pi)), None, is_rec=True)
                    torch.manual_seed(seed_n); neg = m((V(u), V(ni)), None, is_rec=True)
                else:
                    pos, neg = m((V(u), V(pi)), None, is_rec=True), m((V(u), V(ni)), None, is_rec=True)
                loss = rloss.bprLoss(pos, neg, target=-1)
                loss = loss + rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight)
                zero_grads(m); loss.backward()
                out.update({tag + 'rec.pos': npy(pos), tag + 'rec.neg': npy(neg), tag + 'rec.loss': npy(loss)})
                out.update({tag + 'rec.' + k: v for k, v in grads_of(m).items()})
                if not gum:
                    pos, neg = m(None, (V(ph), V(pt), V(pr)), is_rec=False), m(None, (V(nh), V(nt), V(pr)), is_rec=False)
                    loss = rloss.marginLoss()(pos, neg, 1.0)
                    ent = m.ent_embeddings(V(torch.cat([ph, pt, nh, nt])))
                    rel = m.rel_embeddings(V(torch.cat([pr, pr])))
                    nrm = m.norm_embeddings(V(torch.cat([pr, pr])))
                    loss = loss + rloss.orthogonalLoss(rel, nrm)
                    loss = loss + rloss.normLoss(ent) + rloss.normLoss(rel)
                    zero_grads(m); loss.backward()
                    out.update({tag + 'kg.pos': npy(pos), tag + 'kg.neg': npy(neg), tag + 'kg.loss': npy(loss)})
                    out.update({tag + 'kg.' + k: v for k, v in grads_of(m).items()})
        save('score_d%d' % d, **out)

    # alignment maps (G9) are plain JSON
    with open(os.path.join(args.out, 'alignment.json'), 'w') as f:
        json.dump({'e_vocab': [[k, v] for k, v in e_vocab.items()],      # ordered pairs: iteration order matters
                   'i_vocab': [[k, v] for k, v in i_vocab.items()], 'kg2i': kg2i,
                   'new_map': {str(k): list(v) for k, v in new_map.items()},
                   'e_remap': {str(k): v for k, v in e_remap.items()},
                   'i_remap': {str(k): v for k, v in i_remap.items()},
                   'n_aligned': n_aligned}, f, indent=0, sort_keys=True)
    return i_map, new_map


def eval_cases(i_map, new_map):
    """G5 (full score matrices) + G6 (ranking) on a small world, d=20 so B x N x d stays tiny."""
    rng = np.random.RandomState(23)
    gen = torch.Generator().manual_seed(9)
    d, BQ = 20, 9
    out = {}
    uq = torch.from_numpy(rng.randint(0, NU, BQ)).long()
    eq = torch.from_numpy(rng.randint(0, NE, BQ)).long()
    rq = torch.from_numpy(rng.randint(0, NR, BQ)).long()
    out.update(uq=npy(uq), eq=npy(eq), rq=npy(rq))

    m = bprmf.BPRMF(d, NU, NI); sd = set_weights(m, gen)
    out.update({'bprmf.' + k: v for k, v in sd.items()})
    out['bprmf.eval'] = npy(m.evaluate(V(uq)))

    for name, mod, cls in (('transe', transE, 'TransEModel'), ('transh', transH, 'TransHModel'), ('transr', transR, 'TransRModel')):
        keep = None
        for l1 in (False, True):
            m = getattr(mod, cls)(l1, d, NE, NR)
            if keep is None:
                sd = set_weights(m, gen); out.update({name + '.' + k: v for k, v in sd.items()})
                keep = {k: p.data.clone() for k, p in m.named_parameters()}
            else:
                for k, p in m.named_parameters():
                    p.data.copy_(keep[k])
            tag = '%s.%s.' % (name, 'L1' if l1 else 'L2')
            out[tag + 'head'] = npy(m.evaluateHead(V(eq), V(rq)))
            out[tag + 'tail'] = npy(m.evaluateTail(V(eq), V(rq)))

    keep = None
    for l1 in (False, True):
        for gum in (False, True):
            m = transUP.TransUPModel(l1, d, NU, NI, NP_TUP, gum)
            if keep is None:
                sd = set_weights(m, gen); out.update({'tup.' + k: v for k, v in sd.items()})
                keep = {k: p.data.clone() for k, p in m.named_parameters()}
            else:
                for k, p in m.named_parameters():
                    p.data.copy_(keep[k])
            tag = 'tup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
            if gum:
                out[tag + 'uni'] = npy(uniforms(77, (BQ, NI, NP_TUP)))
                torch.manual_seed(77)
            out[tag + 'eval'] = npy(m.evaluate(V(uq)))

    keep = None
    for l1 in (False, True):
        for gum in (False, True):
            m = jtup.jTransUPModel(l1, d, NU, NI, NE, NR, i_map, new_map, False, gum)
            if keep is None:
                sd = set_weights(m, gen); out.update({'ktup.' + k: v for k, v in sd.items()})
                keep = {k: p.data.clone() for k, p in m.named_parameters()}
                out['ktup.item2ent'] = np.asarray(m.paddingItems(torch.arange(NI), m.ent_total - 1), dtype=np.int64)
            else:
                for k, p in m.named_parameters():
                    p.data.copy_(keep[k])
            tag = 'ktup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
            if gum:
                out[tag + 'uni'] = npy(uniforms(78, (BQ, NI, NR)))
                torch.manual_seed(78)
            out[tag + 'evalRec'] = npy(m.evaluateRec(V(uq)))
            if not gum:
                out[tag + 'evalHead'] = npy(m.evaluateHead(V(eq), V(rq)))     # (BQ, NE+1): pad row is a candidate
                out[tag + 'evalTail'] = npy(m.evaluateTail(V(eq), V(rq)))
    save('eval_small', **out)

    # ---------------- ranking goldens (G6) with np.argsort forced stable (shim 3)
    real_argsort = np.argsort
    np.argsort = lambda a, *aa, **kw: real_argsort(a, *aa, **dict(kw, kind='stable'))
    try:
        rk = {}
        rows = out['tup.L1.soft.eval'].copy()           # (BQ, NI), lower = better
        rows[2, 5] = rows[2, 9] = rows[2, 1]            # exact ties inside the top region
        rows[3, :] = 1.0                                # a fully tied row
        rk['rec.rows'] = rows
        rec_json = []
        for b in range(BQ):
            gold = set(int(x) for x in rng.permutation(NI)[:rng.randint(1, 6)])
            filt = set(int(x) for x in rng.permutation(NI)[:rng.randint(0, 15)]) - gold
            f1, p, r, hit, ndcg, top = rmisc.getRecPerformance(rows[b], gold, fliter_samples=filt, topn=10)
            rec_json.append({'gold': sorted(gold), 'filter': sorted(filt), 'f1': f1, 'p': p, 'r': r, 'hit': hit,
                             'ndcg': ndcg, 'top_ids': [int(x) for x in top]})
        # descending convention (BPRMF): scores negated before the ascending sort (misc.py:180)
        brow = out['bprmf.eval']
        rk['rec.bprmf_rows'] = brow
        for b in range(BQ):
            gold = set(int(x) for x in rng.permutation(NI)[:3])
            f1, p, r, hit, ndcg, top = rmisc.getRecPerformance(-brow[b], gold, fliter_samples=None, topn=10)
            rec_json.append({'gold': sorted(gold), 'filter': None, 'f1': f1, 'p': p, 'r': r, 'hit': hit,
                             'ndcg': ndcg, 'top_ids': [int(x) for x in top], 'descending': True})
        krows = out['ktup.L2.soft.evalTail'].copy()     # (BQ, NE+1) incl. the pad-entity column
        krows[1, 4] = krows[1, 8]
        krows[4, :] = 0.5
        rk['kg.rows'] = krows
        kg_json = []
        for b in range(BQ):
            gold = set(int(x) for x in rng.permutation(NE)[:rng.randint(1, 5)])
            filt = set(int(x) for x in rng.permutation(NE)[:rng.randint(0, 20)]) - gold
            hits, ranks, ids = rmisc.getKGPerformance(krows[b], gold, fliter_samples=filt, topn=10)
            kg_json.append({'gold': sorted(gold), 'filter': sorted(filt), 'hits': [int(h) for h in hits],
                            'ranks': [int(x) for x in ranks], 'ids': [int(x) for x in ids]})
        save('ranking', **rk)
        known = {'ndcg': [[[2, 1, 2, 0], 4, 0, reval.ndcg_at_k([2, 1, 2, 0], 4)],
                          [[2, 1, 2, 0], 4, 1, reval.ndcg_at_k([2, 1, 2, 0], 4, method=1)],
                          [[0], 1, 0, reval.ndcg_at_k([0], 1)], [[1], 2, 0, reval.ndcg_at_k([1], 2)],
                          [[1, 0, 1, 1, 0, 0, 1, 0, 0, 0], 10, 0, reval.ndcg_at_k([1, 0, 1, 1, 0, 0, 1, 0, 0, 0], 10)]]}
        with open(os.path.join(args.out, 'ranking.json'), 'w') as f:
            json.dump({'rec': rec_json, 'kg': kg_json, 'known': known}, f, indent=0, sort_keys=True)
    finally:
        np.argsort = real_argsort


def baseline_cases():
    """CKE (CKE.py: BPRMF on item + aligned entity rows, TransR on the KG) and CFKG (CFKG.py: TransE with an extra "buy"
    relation between users and item-entities) -- the reference baselines that reuse the accelerated kernels.  Own seeds and an
    own file, so the earlier fixtures keep their random draws."""
    from jTransUP.models import CKE as rcke, CFKG as rcfkg
    rng = np.random.RandomState(23)
    gen = torch.Generator().manual_seed(29)
    e_vocab, i_vocab, kg2i, new_map, e_remap, i_remap, n_aligned = make_alignment(rng)
    i_map = IntKeyDict(i_remap)
    out = {}
    BQ = 9
    for d in (36, 64):
        u = torch.from_numpy(rng.randint(0, NU, B)).long()
        pi = torch.from_numpy(rng.randint(0, NI, B)).long(); ni = torch.from_numpy(rng.randint(0, NI, B)).long()
        ph = torch.from_numpy(rng.randint(0, NE, B)).long(); pt = torch.from_numpy(rng.randint(0, NE, B)).long()
        pr = torch.from_numpy(rng.randint(0, NR, B)).long()
        nh = torch.from_numpy(rng.randint(0, NE, B)).long(); nt = torch.from_numpy(rng.randint(0, NE, B)).long()
        uq = torch.from_numpy(rng.randint(0, NU, BQ)).long(); eq = torch.from_numpy(rng.randint(0, NE, BQ)).long()
        rq = torch.from_numpy(rng.randint(0, NR, BQ)).long()
        pre = 'd%d.' % d
        out.update({pre + k: npy(v) for k, v in dict(u=u, pi=pi, ni=ni, ph=ph, pt=pt, pr=pr, nh=nh, nt=nt, uq=uq, eq=eq, rq=rq).items()})
        for l1 in (False, True):
            tag = pre + 'cke.%s.' % ('L1' if l1 else 'L2')
            m = rcke.CKE(l1, d, NU, NI, NE, NR, i_map, new_map)
            if not l1:
                sd = set_weights(m, gen)
                out.update({pre + 'cke.' + k: v for k, v in sd.items()})
                keep = {k: p.data.clone() for k, p in m.named_parameters()}
                out[pre + 'cke.item2ent'] = np.asarray(m.paddingItems(torch.arange(NI), m.ent_total - 1), dtype=np.int64)
            else:
                for k, p in m.named_parameters():
                    p.data.copy_(keep[k])
            # rec step (knowledgable_recommendation.py:335-342; target -1 for every model but bprmf / fm / cofm, trainer.py:15-17)
            pos, neg = m((V(u), V(pi)), None, is_rec=True), m((V(u), V(ni)), None, is_rec=True)
            loss = rloss.bprLoss(pos, neg, target=-1)
            zero_grads(m); loss.backward()
            out.update({tag + 'rec.pos': npy(pos), tag + 'rec.neg': npy(neg), tag + 'rec.loss': npy(loss)})
            out.update({tag + 'rec.' + k: v for k, v in grads_of(m).items() if 'proj' not in k})
            # kg step (:345-382): margin + normLoss(ent rows) + normLoss(rel rows)
            pos, neg = m(None, (V(ph), V(pt), V(pr)), is_rec=False), m(None, (V(nh), V(nt), V(pr)), is_rec=False)
            loss = rloss.marginLoss()(pos, neg, 1.0)
            loss = loss + rloss.normLoss(m.ent_embeddings(V(torch.cat([ph, pt, nh, nt])))) + rloss.normLoss(m.rel_embeddings(V(torch.cat([pr, pr]))))
            zero_grads(m); loss.backward()
            out.update({tag + 'kg.pos': npy(pos), tag + 'kg.neg': npy(neg), tag + 'kg.loss': npy(loss)})
            out.update({tag + 'kg.' + k: v for k, v in grads_of(m).items()})
            out[tag + 'evalRec'] = npy(m.evaluateRec(V(uq)))
            out[tag + 'evalHead'] = npy(m.evaluateHead(V(eq), V(rq)))
            out[tag + 'evalTail'] = npy(m.evaluateTail(V(eq), V(rq)))
        for l1 in (False, True):
            tag = pre + 'cfkg.%s.' % ('L1' if l1 else 'L2')
            m = rcfkg.CFKG(l1, d, NU, NI, NE, NR)
            if not l1:
                sd = set_weights(m, gen)
                out.update({pre + 'cfkg.' + k: v for k, v in sd.items()})
                keep = {k: p.data.clone() for k, p in m.named_parameters()}
            else:
                for k, p in m.named_parameters():
                    p.data.copy_(keep[k])
            pos, neg = m((V(u), V(pi)), None, is_rec=True), m((V(u), V(ni)), None, is_rec=True)   # item ids index the entity table
            loss = rloss.bprLoss(pos, neg, target=-1)
            zero_grads(m); loss.backward()
            out.update({tag + 'rec.pos': npy(pos), tag + 'rec.neg': npy(neg), tag + 'rec.loss': npy(loss)})
            out.update({tag + 'rec.' + k: v for k, v in grads_of(m).items()})
            pos, neg = m(None, (V(ph), V(pt), V(pr)), is_rec=False), m(None, (V(nh), V(nt), V(pr)), is_rec=False)
            loss = rloss.marginLoss()(pos, neg, 1.0)
            loss = loss + rloss.normLoss(m.ent_embeddings(V(torch.cat([ph, pt, nh, nt])))) + rloss.normLoss(m.rel_embeddings(V(torch.cat([pr, pr]))))
            zero_grads(m); loss.backward()
            out.update({tag + 'kg.pos': npy(pos), tag + 'kg.neg': npy(neg), tag + 'kg.loss': npy(loss)})
            out.update({tag + 'kg.' + k: v for k, v in grads_of(m).items()})
            out[tag + 'evalRec'] = npy(m.evaluateRec(V(uq)))
            out[tag + 'evalHead'] = npy(m.evaluateHead(V(eq), V(rq)))
            out[tag + 'evalTail'] = npy(m.evaluateTail(V(eq), V(rq)))
    save('baselines', **out)


def transr_d256_case():
    """TransR at config 5's width.  The (relations x d*d) projection table would make the fixture 2 MB per copy, so it is NOT
    stored: the test re-creates it from the seed below with the same torch CPU generator calls; stored are the small tables, the
    ids, the scores and the entity / relation gradients, plus per-relation sums and a strided sample of the projection gradient."""
    d, seed = 256, 4242
    rng = np.random.RandomState(31)
    out = {'seed': np.asarray([seed], dtype=np.int64)}
    ph = torch.from_numpy(rng.randint(0, NE, B)).long(); pt = torch.from_numpy(rng.randint(0, NE, B)).long()
    pr = torch.from_numpy(rng.randint(0, NR, B)).long()
    nh = torch.from_numpy(rng.randint(0, NE, B)).long(); nt = torch.from_numpy(rng.randint(0, NE, B)).long()
    out.update(ph=npy(ph), pt=npy(pt), pr=npy(pr), nh=npy(nh), nt=npy(nt))
    for l1 in (False, True):
        tag = 'L1.' if l1 else 'L2.'
        m = transR.TransRModel(l1, d, NE, NR)
        g = torch.Generator().manual_seed(seed)                  # the test repeats exactly these three draws
        m.ent_embeddings.weight.data.copy_(torch.randn(NE, d, generator=g) * 0.3)
        m.rel_embeddings.weight.data.copy_(torch.randn(NR, d, generator=g) * 0.3)
        m.proj_embeddings.weight.data.copy_(torch.randn(NR, d * d, generator=g) * 0.06)
        pos, neg = m(V(ph), V(pt), V(pr)), m(V(nh), V(nt), V(pr))
        loss = rloss.marginLoss()(pos, neg, 1.0)
        zero_grads(m); loss.backward()
        gp = m.proj_embeddings.weight.grad
        out.update({tag + 'pos': npy(pos), tag + 'neg': npy(neg), tag + 'loss': npy(loss),
                    tag + 'grad.ent': npy(m.ent_embeddings.weight.grad), tag + 'grad.rel': npy(m.rel_embeddings.weight.grad),
                    tag + 'grad.proj.rowsum': npy(gp.sum(1)), tag + 'grad.proj.sample': npy(gp[:, ::997])})
    save('transr_d256', **out)


if __name__ == '__main__':
    i_map, new_map = score_cases()
    eval_cases(i_map, new_map)
    baseline_cases()
    transr_d256_case()
