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
  5. (train_step_cases only) Optimizer.zero_grad() zero-FILLS, as it did in the torch 0.3 the reference is written for
     (README.md:5-9): modern torch's default set_to_none=True would make torch.optim skip every table a step does not touch
     (no weight decay, no Adam moment decay on the user / item tables during KTUP's kg steps), which is not what
     utils/trainer.py:79-84 did when it was written.  A table whose gradient is still None (never touched) is skipped in
     both.
  6. (fm_cases only) fm.py / cofm.py do not construct on a modern torch: `nn.Parameter(bias, 1)` passes an int where a bool is
     required now, and their bias tables are nn.Embedding modules whose weight is 1-D, which torch.embedding no longer accepts.
     While those two models are built and run, nn.Parameter coerces requires_grad with bool() and F.embedding serves a 1-D weight
     by weight.index_select(0, ids) (what the torch-0.3 backend did).  The models' own lines are untouched.
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
ap.add_argument('--only', default='', help="'train_steps': write only the three train_steps*.npz fixtures and their _cond files")
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

    for d in (64, 100, 36, 256, 50):      # 256 (config 5's size), then 50 (not a multiple of 4) last: the earlier files keep their random draws
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


def train_step_cases(d=64, out_name='train_steps', seeds=(47, 53), ktup_cases=None, tup_cases=None, kg_cases=None):
    """(d, out_name, seeds and the case filters: the defaults write train_steps.npz exactly as before; train_steps_d100 / _d256 are the
    same step bodies at BASELINE's widths -- the fused step kernels are per-width templates -- with their own draws.)
    G4 (post-optimizer-step tables) + G8 (joint schedule), SURVEY.md 8(c): the step bodies of the three drivers replayed line by
    line on the reference's own modules -- knowledgable_recommendation.py:335-403 (jtransup: rec and kg steps under the joint
    schedule of :209,320), item_recommendation.py:160-192 (transup, soft and ST-Gumbel gate), knowledge_representation.py:176-216
    (transe / transh) -- with the reference's own ModelTrainer (utils/trainer.py:20-81: optimizer construction with
    weight_decay = l2_lambda, optimizer_zero_grad, optimizer_step) and nn.utils.clip_grad_norm.  The drivers themselves cannot be
    imported (models/base.py:2 needs gflags; losses.data[0] raises on 0-dim tensors), so their lines are restated here; FLAGS is
    a plain namespace with the fields ModelTrainer reads.  Own seeds, own file: the earlier fixtures keep their draws.
    Stored per case: initial tables (once per model), the batches, recorded Gumbel uniforms, per-step losses and pre-clip global
    gradient norms, and the tables after the last step."""
    import logging
    import types
    from jTransUP.utils import trainer as rtrainer
    real_zero_grad = torch.optim.Optimizer.zero_grad
    torch.optim.Optimizer.zero_grad = lambda self, set_to_none=False: real_zero_grad(self, set_to_none=False)      # shim 5
    clip = getattr(torch.nn.utils, 'clip_grad_norm', None) or torch.nn.utils.clip_grad_norm_
    rng = np.random.RandomState(seeds[0])
    gen = torch.Generator().manual_seed(seeds[1])
    e_vocab, i_vocab, kg2i, new_map, e_remap, i_remap, n_aligned = make_alignment(rng)
    i_map = IntKeyDict(i_remap)
    NSTEP = 3
    log = logging.getLogger('goldens'); log.setLevel(logging.ERROR)
    out = {}
    cond_out = {}                        # <out_name>_cond.npz: conditioning of every final table (see conditioning() below)

    def flags(model_type, opt, l2, lr):
        return types.SimpleNamespace(model_type=model_type, optimizer_type=opt, l2_lambda=l2, learning_rate=lr,
                                     learning_rate_decay_when_no_progress=1.0, momentum=0.9, eval_interval_steps=10,
                                     ckpt_path='/tmp', experiment_name='goldens', eval_only_mode=False, load_experiment_name=None)

    def ids(hi, n=B):
        return torch.from_numpy(rng.randint(0, hi, n)).long()

    def replay(model, FL, steps, clip_max, after_backward=None, grads=None):
        """The same loop as run() below on another copy of the model (conditioning replays): -> (final tables, gradient norms)."""
        tr = rtrainer.ModelTrainer(model, log, 10, FL)
        norms = []
        for t, body in enumerate(steps):
            tr.optimizer_zero_grad()
            l = body(model, tr)
            l.backward()
            if grads is not None:
                grads.append({k: p.grad.data.double().numpy().copy() for k, p in model.named_parameters() if p.grad is not None})
            if after_backward is not None:
                after_backward(model, t)
            norms.append(float(clip([p for _, p in model.named_parameters()], clip_max)))
            tr.optimizer_step()
        return {k: p.data.double().numpy().copy() for k, p in model.named_parameters()}, norms

    def conditioning(tag, model0, FL, spec, clip_max, grads32):
        """How far the tables after the last step move under changes that are mathematically (nearly) nothing -- the measure of which
        elements a 1e-4 band can pin at all.  (An Adam element whose clipped gradient is of the size of eps = 1e-8 turns the last bits
        of that gradient into a step of 1e-4; a gradient that is the small difference of large per-pair terms carries a rounding
        error of several per cent in ANY fp32 evaluation, and each evaluation order rounds it differently.)  Samples, all through
        the reference's own modules and ModelTrainer:
          (1) the steps in fp64 (`.double()`; the ST-Gumbel uniforms are the fp32 draws);
          (2) N_PERM fp32 replays with every batch in another order (the reference shuffles its batches every epoch,
              utils/data.py:87-110: the sums over the batch and the embedding backward then round differently);
          (3) N_NOISE fp64 replays in which, between backward() and clip_grad_norm, every gradient row receives Gaussian noise of
              TWICE the size of the rounding error the reference's own fp32 step committed on that row at that step -- measured as
              1.4826 x the median over the row of |fp32 gradient - fp64 gradient| (a robust scale: a few elements whose
              trajectories already differ do not inflate it).  This is what another fp32 evaluation of the same step amounts to.
        cond.<table> = max over the samples of |sample - final32|; d64.<table> = final64 - final32.  Entries below 1/64 of the band
        2e-5 + 1e-4 |final32| are stored as 0 (they decide nothing; the file stays small)."""
        import copy
        import zlib
        N_PERM, N_NOISE = 8, 16
        prs = np.random.RandomState(zlib.crc32((out_name + ':' + tag).encode()) & 0x7fffffff)
        final32 = {k: out[tag + 'final.' + k].astype(np.float64) for k, _ in model0.named_parameters()}
        real_uniform = torch.Tensor.uniform_
        ctx = {'perm': None}

        def uniform_(self, *a, **kw):                         # the fp32 draws, whatever the table's dtype; rows follow the batch order
            if self.dtype != torch.float64 and ctx['perm'] is None:
                return real_uniform(self, *a, **kw)
            tmp = real_uniform(torch.empty(self.shape, dtype=torch.float32), *a, **kw)
            self.copy_(tmp if ctx['perm'] is None else tmp[ctx['perm']])
            return self

        def bodies(perm=None):
            return [f(b if perm is None else {k: v[perm] for k, v in b.items()}, **kw) for f, b, kw in spec]

        torch.Tensor.uniform_ = uniform_
        try:
            grads64 = []
            f64, n64 = replay(copy.deepcopy(model0).double(), FL, bodies(), clip_max, grads=grads64)
            sigma = [{k: 1.4826 * np.median(np.abs(g32[k] - g64[k]), axis=1, keepdims=True) for k in g64 if k in g32}
                     for g32, g64 in zip(grads32, grads64)]

            def noise(model, t):
                for k, p in model.named_parameters():
                    if p.grad is not None and k in sigma[t]:
                        p.grad.data.add_(torch.from_numpy(2.0 * sigma[t][k] * prs.standard_normal(tuple(p.shape))))

            samples = [f64]
            for _ in range(N_PERM):
                ctx['perm'] = torch.from_numpy(prs.permutation(B))
                samples.append(replay(copy.deepcopy(model0), FL, bodies(ctx['perm']), clip_max)[0])
                ctx['perm'] = None
            for _ in range(N_NOISE):
                samples.append(replay(copy.deepcopy(model0).double(), FL, bodies(), clip_max, after_backward=noise)[0])
        finally:
            torch.Tensor.uniform_ = real_uniform
            ctx['perm'] = None
        cond_out[tag + 'gradnorms64'] = np.asarray(n64, dtype=np.float64)
        for k, w in final32.items():
            floor = (2e-5 + 1e-4 * np.abs(w)) / 64.0
            c = np.max(np.stack([np.abs(s[k] - w) for s in samples]), axis=0)
            d64 = f64[k] - w
            cond_out[tag + 'cond.' + k] = np.where(c < floor, 0.0, c).astype(np.float32)
            cond_out[tag + 'd64.' + k] = np.where(np.abs(d64) < floor, 0.0, d64).astype(np.float32)

    def run(tag, model, FL, steps, clip_max, spec=None):
        """steps: list of callables(model, trainer) -> loss Variable (the driver's loss lines).  Returns nothing; fills `out`.
        spec: the same steps as (factory, batch, keyword arguments) triples, for the conditioning replays."""
        import copy
        model0 = copy.deepcopy(model)
        if spec is not None:
            steps = [f(b, **kw) for f, b, kw in spec]
        tr = rtrainer.ModelTrainer(model, log, 10, FL)
        losses, norms, grads32 = [], [], []
        for body in steps:
            tr.optimizer_zero_grad()
            l = body(model, tr)
            l.backward()
            grads32.append({k: p.grad.data.double().numpy().copy() for k, p in model.named_parameters() if p.grad is not None})
            gn = clip([p for _, p in model.named_parameters()], clip_max)
            tr.optimizer_step()
            losses.append(float(l)); norms.append(float(gn))
        out[tag + 'losses'] = np.asarray(losses, dtype=np.float64)
        out[tag + 'gradnorms'] = np.asarray(norms, dtype=np.float64)
        for k, p in model.named_parameters():
            out[tag + 'final.' + k] = npy(p.data)
        assert tr.step == len(steps)
        if spec is not None:
            conditioning(tag, model0, FL, spec, clip_max, grads32)

    # ---------------- KTUP (jtransup), the joint schedule: step s is a rec step iff s % 10 < 10 * joint_ratio
    sched = {}
    for jr in (0.5, 0.7, 0.9):
        step_to_switch = 10 * jr                                            # knowledgable_recommendation.py:209
        sched[str(jr)] = [bool(s % 10 < step_to_switch) for s in range(30)]   # :320
    kt_batches = []
    for s in range(6):
        kt_batches.append(dict(u=ids(NU), pi=ids(NI), ni=ids(NI), ph=ids(NE), pt=ids(NE), pr=ids(NR), nh=ids(NE), nt=ids(NE)))
    for s, b in enumerate(kt_batches):
        out.update({'ktup.batch%d.%s' % (s, k): npy(v) for k, v in b.items()})
    kg_lambda, margin = 0.5, 1.0
    base = jtup.jTransUPModel(False, d, NU, NI, NE, NR, i_map, new_map, False, False)
    sd = set_weights(base, gen)
    out.update({'ktup.init.' + k: v for k, v in sd.items()})
    out['ktup.item2ent'] = np.asarray(base.paddingItems(torch.arange(NI), base.ent_total - 1), dtype=np.int64)
    keep = {k: p.data.clone() for k, p in base.named_parameters()}
    kinds6 = [True, True, False, True, False, False]                       # steps 5..10 of joint_ratio 0.7 would read R R K K K R; any mix serves

    def ktup_rec(b):
        def body(m, tr):
            pos = m((V(b['u']), V(b['pi'])), None, is_rec=True)
            neg = m((V(b['u']), V(b['ni'])), None, is_rec=True)
            l = rloss.bprLoss(pos, neg, target=tr.model_target)
            return l + rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight)
        return body

    def ktup_kg(b):
        def body(m, tr):
            pos = m(None, (V(b['ph']), V(b['pt']), V(b['pr'])), is_rec=False)
            neg = m(None, (V(b['nh']), V(b['nt']), V(b['pr'])), is_rec=False)
            l = rloss.marginLoss()(pos, neg, margin)
            ent = m.ent_embeddings(V(torch.cat([b['ph'], b['pt'], b['nh'], b['nt']])))
            rel = m.rel_embeddings(V(torch.cat([b['pr'], b['pr']])))
            nrm = m.norm_embeddings(V(torch.cat([b['pr'], b['pr']])))
            l = l + rloss.orthogonalLoss(rel, nrm)
            l = l + rloss.normLoss(ent) + rloss.normLoss(rel)
            return kg_lambda * l
        return body
    out['ktup.kinds'] = np.asarray(kinds6, dtype=np.int64)
    out['ktup.kg_lambda'] = np.asarray([kg_lambda]); out['ktup.margin'] = np.asarray([margin])
    for opt, lr in (('Adagrad', 0.05), ('Adam', 0.01), ('SGD', 0.05)):
        for l2 in (0.0, 1e-5):
            if opt == 'SGD' and l2 == 0.0:
                continue
            if ktup_cases is not None and (opt, l2) not in ktup_cases:
                continue
            m = jtup.jTransUPModel(False, d, NU, NI, NE, NR, i_map, new_map, False, False)
            for k, p in m.named_parameters():
                p.data.copy_(keep[k])
            spec = [(ktup_rec if r else ktup_kg, b, {}) for r, b in zip(kinds6, kt_batches)]
            run('ktup.%s.l2_%g.' % (opt, l2), m, flags('jtransup', opt, l2, lr), None, 5.0, spec=spec)

    # ---------------- TUP (transup), soft and ST-Gumbel gate, item_recommendation.py:160-192
    tup_batches = [dict(u=ids(NU), pi=ids(NI), ni=ids(NI)) for _ in range(NSTEP)]
    for s, b in enumerate(tup_batches):
        out.update({'tup.batch%d.%s' % (s, k): npy(v) for k, v in b.items()})
    base = transUP.TransUPModel(False, d, NU, NI, NP_TUP, False)
    sd = set_weights(base, gen)
    out.update({'tup.init.' + k: v for k, v in sd.items()})
    keep = {k: p.data.clone() for k, p in base.named_parameters()}

    def tup_rec(b, seeds=None):
        def body(m, tr):
            if seeds is not None:
                torch.manual_seed(seeds[0])
            pos = m(V(b['u']), V(b['pi']))
            if seeds is not None:
                torch.manual_seed(seeds[1])
            neg = m(V(b['u']), V(b['ni']))
            l = rloss.bprLoss(pos, neg, target=tr.model_target)
            ue = m.user_embeddings(V(b['u'])); ie = m.item_embeddings(V(torch.cat([b['pi'], b['ni']])))
            return l + rloss.orthogonalLoss(m.pref_embeddings.weight, m.pref_norm_embeddings.weight) + rloss.normLoss(ue) + \
                rloss.normLoss(ie) + rloss.normLoss(m.pref_embeddings.weight)
        return body
    for gum in (False, True):
        for opt, lr, cmax in (('Adagrad', 0.05, 5.0), ('Adam', 0.01, 0.05)):          # 0.05: a clip that certainly bites
            if tup_cases is not None and (gum, opt) not in tup_cases:
                continue
            m = transUP.TransUPModel(False, d, NU, NI, NP_TUP, gum)
            for k, p in m.named_parameters():
                p.data.copy_(keep[k])
            tag = 'tup.%s.%s.' % ('hard' if gum else 'soft', opt)
            spec = []
            for s, b in enumerate(tup_batches):
                seeds = None
                if gum:
                    seeds = (500 + 2 * s, 501 + 2 * s)
                    out[tag + 'uni%d.pos' % s] = npy(uniforms(seeds[0], (B, NP_TUP)))
                    out[tag + 'uni%d.neg' % s] = npy(uniforms(seeds[1], (B, NP_TUP)))
                spec.append((tup_rec, b, dict(seeds=seeds)))
            out[tag + 'clip'] = np.asarray([cmax])
            run(tag, m, flags('transup', opt, 1e-5, lr), None, cmax, spec=spec)

    # ---------------- TransE / TransH, knowledge_representation.py:176-216
    kg_batches = [dict(ph=ids(NE), pt=ids(NE), pr=ids(NR), nh=ids(NE), nt=ids(NE)) for _ in range(NSTEP)]
    for s, b in enumerate(kg_batches):
        out.update({'kg.batch%d.%s' % (s, k): npy(v) for k, v in b.items()})
    for name, mod, cls in (('transe', transE, 'TransEModel'), ('transh', transH, 'TransHModel')):
        base = getattr(mod, cls)(False, d, NE, NR)
        sd = set_weights(base, gen)
        out.update({name + '.init.' + k: v for k, v in sd.items()})
        keep = {k: p.data.clone() for k, p in base.named_parameters()}

        def kg_body(b, name=name):
            def body(m, tr):
                pos, neg = m(V(b['ph']), V(b['pt']), V(b['pr'])), m(V(b['nh']), V(b['nt']), V(b['pr']))
                l = rloss.marginLoss()(pos, neg, margin)
                ent = m.ent_embeddings(V(torch.cat([b['ph'], b['pt'], b['nh'], b['nt']])))
                rel = m.rel_embeddings(V(torch.cat([b['pr'], b['pr']])))
                if name == 'transh':
                    l = l + rloss.orthogonalLoss(rel, m.norm_embeddings(V(torch.cat([b['pr'], b['pr']]))))
                return l + rloss.normLoss(ent) + rloss.normLoss(rel)
            return body
        for opt, lr in (('Adagrad', 0.05), ('Adam', 0.01)):
            if kg_cases is not None and (name, opt) not in kg_cases:
                continue
            m = getattr(mod, cls)(False, d, NE, NR)
            for k, p in m.named_parameters():
                p.data.copy_(keep[k])
            run('%s.%s.' % (name, opt), m, flags(name, opt, 1e-5, lr), None, 5.0, spec=[(kg_body, b, {}) for b in kg_batches])
    torch.optim.Optimizer.zero_grad = real_zero_grad
    save(out_name, **out)
    save(out_name + '_cond', **cond_out)
    if out_name == 'train_steps':
        with open(os.path.join(args.out, 'train_steps.json'), 'w') as f:
            json.dump({'joint_schedule': sched, 'B': B, 'd': d, 'NP_TUP': NP_TUP}, f, indent=0, sort_keys=True)



def eval_pass_cases():
    """The whole evaluation pass of the rec drivers (item_recommendation.py:27-53 / knowledgable_recommendation.py:50-104):
    model.evaluate / evaluateRec for EVERY user in batches, the reference's own evalRecProcess (worker processes, filter = union
    of all_dicts, sign flip for `descending`), and the mean of the metric columns -- at d = 64 and d = 100, where the build's
    one-sweep evaluation (scores + filtered top-n without the score matrix) applies.  Stored: tables, the eval / filter
    dictionaries, per-user (f1, p, r, hit, ndcg) and top ids, and the pass means.  np.argsort forced stable (shim 3).  Own seeds."""
    rng = np.random.RandomState(61)
    gen = torch.Generator().manual_seed(67)
    e_vocab, i_vocab, kg2i, new_map, e_remap, i_remap, n_aligned = make_alignment(rng)
    i_map = IntKeyDict(i_remap)
    NIe = 230                                # enough items for a filtered top-10 to be a real selection (a few 16-item tiles per user)
    i_map_big = IntKeyDict({i: i for i in range(NIe)})
    new_map_big = {i: ((int(rng.randint(0, NE)) if i % 4 else -1), i) for i in range(NIe)}
    out, meta = {}, {}
    real_argsort = np.argsort
    np.argsort = lambda a, *aa, **kw: real_argsort(a, *aa, **dict(kw, kind='stable'))
    try:
        # (d, model, L1_flag, use_st_gumbel): the soft gate at both widths; the hard gate (noise per (user, item, preference), drawn by
        # the reference from torch's global generator: re-seeded per batch and recovered by shim 4) with both distances at d = 100
        # ... and, appended last so that the earlier cases keep their random draws, the soft gate with the L1 distance: what the
        # reference's own run scripts evaluate (ktup.sh / transup.sh: -L1_flag -nouse_st_gumbel)
        cases = [(d, name, False, False) for d in (64, 100) for name in ('tup', 'ktup')] + \
                [(100, name, l1, True) for name in ('tup', 'ktup') for l1 in (False, True)] + \
                [(100, name, True, False) for name in ('tup', 'ktup')]
        for ci, (d, name, l1, gum) in enumerate(cases):
                if name == 'tup':
                    m = transUP.TransUPModel(l1, d, NU, NIe, NP_TUP, gum)
                else:
                    m = jtup.jTransUPModel(l1, d, NU, NIe, NE, NR, i_map_big, new_map_big, False, gum)
                sd = set_weights(m, gen)
                tag = '%s.hard.%s.d%d.' % (name, 'L1' if l1 else 'L2', d) if gum else ('%s.L1.d%d.' if l1 else '%s.d%d.') % (name, d)
                out.update({tag + k: v for k, v in sd.items()})
                if name == 'ktup':
                    out[tag + 'item2ent'] = np.asarray(m.paddingItems(torch.arange(NIe), m.ent_total - 1), dtype=np.int64)
                eval_dict, train_dict, valid_dict = {}, {}, {}
                for u in range(NU):
                    if u % 9 == 8:
                        continue                                         # a user without test items is skipped (misc.py:169)
                    perm = rng.permutation(NIe)
                    ng, nt, nv = rng.randint(1, 6), rng.randint(0, 40), rng.randint(0, 8)
                    eval_dict[u] = set(int(x) for x in perm[:ng])
                    train_dict[u] = set(int(x) for x in perm[ng:ng + nt])
                    if nv:
                        valid_dict[u] = set(int(x) for x in perm[ng + nt:ng + nt + nv])
                all_dicts = [train_dict, valid_dict]
                users = list(range(NU))
                results, seeds = [], []
                for b0 in range(0, NU, 16):                              # the eval iterator's batches
                    u_ids = users[b0:b0 + 16]
                    if gum:
                        seeds.append([b0, len(u_ids), 5000 + 100 * ci + b0])
                        torch.manual_seed(seeds[-1][2])                  # the batch's noise = uniforms(seed, (len(u_ids), NIe, P))
                    scores = m.evaluate(V(torch.LongTensor(u_ids))) if name == 'tup' else m.evaluateRec(V(torch.LongTensor(u_ids)))
                    preds = zip(u_ids, scores.data.cpu().numpy())
                    results.extend(rmisc.evalRecProcess(list(preds), eval_dict, all_dicts=all_dicts, descending=False, num_processes=2,
                                                        topn=10, queue_limit=10))
                results.sort(key=lambda r: r[-1][0])                     # worker order is arbitrary; key by user id
                perf = np.array([r[:5] for r in results], dtype=np.float64)
                meta[tag.rstrip('.')] = {
                    'eval': {str(u): sorted(v) for u, v in eval_dict.items()},
                    'train': {str(u): sorted(v) for u, v in train_dict.items()},
                    'valid': {str(u): sorted(v) for u, v in valid_dict.items()},
                    'users': [int(r[-1][0]) for r in results],
                    'top_ids': [[int(x) for x in r[-1][1]] for r in results],
                    'mean': [float(x) for x in perf.mean(axis=0)]}
                if gum:
                    meta[tag.rstrip('.')]['gumbel_seeds'] = seeds        # [first user, users, seed] per batch
                    meta[tag.rstrip('.')]['n_pref'] = NP_TUP if name == 'tup' else NR
                out[tag + 'perf'] = perf
    finally:
        np.argsort = real_argsort
    save('eval_pass', **out)
    with open(os.path.join(args.out, 'eval_pass.json'), 'w') as f:
        json.dump(meta, f, indent=0, sort_keys=True)



def kg_pass_cases():
    """A whole link-prediction evaluation pass of the REFERENCE (knowledge_representation.py:28-75): evaluateHead / evaluateTail for every
    (t, r) / (h, r) key in the eval iterator's batches and its own evalKGProcess (worker processes, filter sets; stable argsort, shim 3):
    hit and filtered rank per (key, gold entity) and the pass means -- TransE and TransH with both distances (L1 is what transe.sh /
    transh.sh / ktup.sh run) at d = 100 on 230 entities.  Own seeds, own files (kg_pass.npz / kg_pass.json)."""
    rng = np.random.RandomState(81)
    gen = torch.Generator().manual_seed(83)
    NEk, d, NKEY = 230, 100, 60
    out, meta = {}, {}
    real_argsort = np.argsort
    np.argsort = lambda a, *aa, **kw: real_argsort(a, *aa, **dict(kw, kind='stable'))
    try:
        for name, mod, cls in (('transe', transE, 'TransEModel'), ('transh', transH, 'TransHModel')):
            for l1 in (True, False):
                m = getattr(mod, cls)(l1, d, NEk, NR)
                sd = set_weights(m, gen)
                tag = '%s.%s.' % (name, 'L1' if l1 else 'L2')
                out.update({tag + k: v for k, v in sd.items()})
                meta[tag.rstrip('.')] = {}
                for side in ('head', 'tail'):                           # head prediction: keys (t, r), golds are heads
                    keys = []
                    while len(keys) < NKEY:
                        k = (int(rng.randint(NEk)), int(rng.randint(NR)))
                        if k not in keys:
                            keys.append(k)
                    eval_dict, train_dict, valid_dict = {}, {}, {}
                    for k in keys:
                        perm = rng.permutation(NEk)
                        ng, nt, nv = rng.randint(1, 5), rng.randint(0, 26), rng.randint(0, 6)
                        eval_dict[k] = set(int(x) for x in perm[:ng])
                        if nt:
                            train_dict[k] = set(int(x) for x in perm[ng:ng + nt])
                        if nv:
                            valid_dict[k] = set(int(x) for x in perm[ng + nt:ng + nt + nv])
                    results = []
                    for b0 in range(0, NKEY, 16):                       # the eval iterator's batches
                        batch = keys[b0:b0 + 16]
                        e = V(torch.LongTensor([k[0] for k in batch]))
                        r = V(torch.LongTensor([k[1] for k in batch]))
                        scores = m.evaluateHead(e, r) if side == 'head' else m.evaluateTail(e, r)
                        preds = zip(batch, scores.data.cpu().numpy())
                        results.extend(rmisc.evalKGProcess(list(preds), eval_dict, all_dicts=[train_dict, valid_dict], descending=False,
                                                           num_processes=2, topn=10, queue_limit=10))
                    results = sorted((tuple(int(x) for x in r[2]), int(r[3]), int(r[1]), int(r[0])) for r in results)   # worker order is arbitrary
                    perf = np.array([[r[3], r[2]] for r in results], dtype=np.float64)
                    ser = lambda dct: [[k[0], k[1], sorted(v)] for k, v in sorted(dct.items())]
                    meta[tag.rstrip('.')][side] = {
                        'keys': [list(k) for k in keys], 'eval': ser(eval_dict), 'train': ser(train_dict), 'valid': ser(valid_dict),
                        'rows': [[r[0][0], r[0][1], r[1], r[2], r[3]] for r in results],   # entity, relation, gold id, filtered rank, hit
                        'mean': [float(x) for x in perf.mean(axis=0)]}                     # (hit ratio, mean rank): knowledge_representation.py:73-75
    finally:
        np.argsort = real_argsort
    save('kg_pass', **out)
    with open(os.path.join(args.out, 'kg_pass.json'), 'w') as f:
        json.dump(meta, f, indent=0, sort_keys=True)


def fm_cases():
    """FM (fm.py) and coFM (cofm.py, shared and separate item tables): scores, BPR (target +1, trainer.py:15-17) / margin losses with
    the drivers' regularisers, gradients, and the all-candidate evaluation matrices.  Shim 6.  Own seeds, own file."""
    import torch.nn as tnn
    import torch.nn.functional as TF
    real_new, real_emb = tnn.Parameter.__new__, TF.embedding

    def param_new(cls, data=None, requires_grad=True):
        return real_new(cls, data, bool(requires_grad))

    def emb(input, weight, *a, **kw):
        if weight.dim() == 1:
            return weight.index_select(0, input.reshape(-1)).reshape(input.shape)
        return real_emb(input, weight, *a, **kw)
    tnn.Parameter.__new__ = staticmethod(param_new)
    TF.embedding = emb
    try:
        from jTransUP.models import fm as rfm, cofm as rcofm
        rng = np.random.RandomState(71)
        gen = torch.Generator().manual_seed(73)
        out = {}
        BQ = 9
        for d in (36, 64):
            pre = 'd%d.' % d
            u = torch.from_numpy(rng.randint(0, NU, B)).long()
            pi = torch.from_numpy(rng.randint(0, NI, B)).long(); ni = torch.from_numpy(rng.randint(0, NI, B)).long()
            ph = torch.from_numpy(rng.randint(0, NE, B)).long(); pt = torch.from_numpy(rng.randint(0, NE, B)).long()
            pr = torch.from_numpy(rng.randint(0, NR, B)).long()
            nh = torch.from_numpy(rng.randint(0, NE, B)).long(); nt = torch.from_numpy(rng.randint(0, NE, B)).long()
            uq = torch.from_numpy(rng.randint(0, NU, BQ)).long(); eq = torch.from_numpy(rng.randint(0, NE, BQ)).long()
            rq = torch.from_numpy(rng.randint(0, NR, BQ)).long()
            out.update({pre + k: npy(v) for k, v in dict(u=u, pi=pi, ni=ni, ph=ph, pt=pt, pr=pr, nh=nh, nt=nt, uq=uq, eq=eq, rq=rq).items()})
            m = rfm.FM(d, NU, NI)
            sd = set_weights(m, gen)
            out.update({pre + 'fm.' + k: v for k, v in sd.items()})
            pos, neg = m(V(u), V(pi)), m(V(u), V(ni))
            loss = rloss.bprLoss(pos, neg, target=1)
            zero_grads(m); loss.backward()
            out.update({pre + 'fm.pos': npy(pos), pre + 'fm.neg': npy(neg), pre + 'fm.loss': npy(loss), pre + 'fm.eval': npy(m.evaluate(V(uq)))})
            out.update({pre + 'fm.' + k: v for k, v in grads_of(m).items()})
            for share in (False, True):
                for l1 in (False, True):
                    n_items = NE if share else NI                      # shared: the item table IS the entity table (cofm.py:86-88)
                    tag = pre + 'cofm.%s.%s.' % ('share' if share else 'own', 'L1' if l1 else 'L2')
                    m = rcofm.coFM(l1, d, NU, n_items, NE, NR, share)
                    if not l1:
                        sd = set_weights(m, gen)
                        out.update({pre + 'cofm.%s.' % ('share' if share else 'own') + k: v for k, v in sd.items()})
                        keep = {k: p.data.clone() for k, p in m.named_parameters()}
                    else:
                        for k, p in m.named_parameters():
                            p.data.copy_(keep[k])
                    pos, neg = m((V(u), V(pi)), None, is_rec=True), m((V(u), V(ni)), None, is_rec=True)
                    loss = rloss.bprLoss(pos, neg, target=1)
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
        save('fm_cofm', **out)
    finally:
        tnn.Parameter.__new__ = real_new
        TF.embedding = real_emb


def train_step_fixtures():
    train_step_cases()
    # the same step bodies at BASELINE's widths (configs[1]-[3]: d = 100; config 5: d = 256)
    train_step_cases(d=100, out_name='train_steps_d100', seeds=(59, 61),
                     ktup_cases=[('Adagrad', 0.0), ('Adagrad', 1e-5), ('Adam', 0.0), ('Adam', 1e-5)],
                     tup_cases=[(True, 'Adagrad'), (True, 'Adam'), (False, 'Adagrad')], kg_cases=[('transh', 'Adagrad'), ('transh', 'Adam')])
    train_step_cases(d=256, out_name='train_steps_d256', seeds=(67, 71), ktup_cases=[('Adagrad', 0.0), ('Adam', 1e-5)],
                     tup_cases=[(False, 'Adagrad')], kg_cases=[('transh', 'Adagrad')])


if __name__ == '__main__':
    if args.only == 'train_steps':
        train_step_fixtures()
        sys.exit(0)
    i_map, new_map = score_cases()
    eval_cases(i_map, new_map)
    baseline_cases()
    transr_d256_case()
    train_step_fixtures()
    eval_pass_cases()
    fm_cases()
    kg_pass_cases()
