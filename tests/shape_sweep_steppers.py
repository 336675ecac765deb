"""Shape sweep of RecStepper (TUP, BPRMF) and KGStepper (TransE / TransH / TransR) against the autograd route that mirrors
item_recommendation.py:160-195 / knowledge_representation.py:176-211 (run by hand on a GPU box: `python tests/shape_sweep_steppers.py`):
widths, preference / relation counts, batch sizes, both distances, SGD (an update is then linear in the gradient: no conditioning
to argue about).  Reports exceptions, losses off by more than 2e-5 relative, tables with more than 0.2 % of their elements off.
What round 6's run reported and was not a bug: with L1, a coordinate z below its rounding flips sign(z) between the two routes and
moves one row pair by a clipped learning-rate step (~3e-4)."""
import copy
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

from test_fast_train import _trainer_for

DEV = torch.device('cuda', 0)
bad, ran = [], [0]


def tables_off(m1, m2):
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        err = (b - a).abs()
        off = err > 2e-6 + 2e-5 * a.abs()
        if int(off.sum()) > max(2, int(2e-3 * off.numel())):
            return '%s: %d of %d elements off, max %.3g' % (k, int(off.sum()), off.numel(), float(err.max()))
    return None


def rec_case(model_type, D, P, B, l1):
    from jTransUP.models import bprmf, transUP
    from jTransUP.utils import loss
    from jTransUP.utils.fast_train import RecStepper
    NU, NI = 50, 40
    tmp = tempfile.mkdtemp()
    torch.manual_seed(4)
    mk = (lambda: transUP.TransUPModel(l1, D, NU, NI, P, False)) if model_type == 'transup' else (lambda: bprmf.BPRMF(D, NU, NI))
    m1, m2 = mk(), mk()
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    FLAGS, tr1 = _trainer_for(tmp, model_type, m1, 'SGD')
    _, tr2 = _trainer_for(tmp, model_type, m2, 'SGD')
    fast = RecStepper(m2, tr2, FLAGS, B)
    gen = torch.Generator().manual_seed(9)
    rnd = lambda hi: torch.randint(0, hi, (B,), generator=gen).to(DEV)
    for step in range(4):
        u, pi, ni = rnd(NU), rnd(NI), rnd(NI)
        tr1.optimizer_zero_grad()
        losses = loss.bprLoss(m1(u, pi), m1(u, ni), target=tr1.model_target)
        if model_type == 'transup':
            losses = losses + loss.orthogonalLoss(m1.pref_embeddings.weight, m1.pref_norm_embeddings.weight) \
                + loss.normLoss(m1.user_embeddings.weight, ids=u) + loss.normLoss(m1.item_embeddings.weight, ids=torch.cat([pi, ni])) \
                + loss.normLoss(m1.pref_embeddings.weight)
        losses.backward(); tr1.clip_and_step(FLAGS.clipping_max_value)
        fl = fast.rec_step(u, pi, ni)
        if not torch.allclose(fl, losses.detach(), rtol=2e-5, atol=2e-6):
            return 'step %d loss %.7g vs %.7g' % (step, float(fl), float(losses))
        msg = tables_off(m1, m2)
        if msg:
            return 'step %d %s' % (step, msg)
    return None


def kg_case(model_type, D, NR, B, l1):
    from jTransUP.models import transE, transH, transR
    from jTransUP.utils import loss
    from jTransUP.utils.fast_train import KGStepper
    NE = 70
    tmp = tempfile.mkdtemp()
    torch.manual_seed(4)
    mk = {'transh': lambda: transH.TransHModel(l1, D, NE, NR), 'transe': lambda: transE.TransEModel(l1, D, NE, NR),
          'transr': lambda: transR.TransRModel(l1, D, NE, NR)}[model_type]
    m1, m2 = mk(), mk()
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    FLAGS, tr1 = _trainer_for(tmp, model_type, m1, 'SGD')
    _, tr2 = _trainer_for(tmp, model_type, m2, 'SGD')
    fast = KGStepper(m2, tr2, FLAGS, B)
    gen = torch.Generator().manual_seed(9)
    rnd = lambda hi: torch.randint(0, hi, (B,), generator=gen).to(DEV)
    for step in range(4):
        ph, pt, pr, nh, nt = rnd(NE), rnd(NE), rnd(NR), rnd(NE), rnd(NE)
        tr1.optimizer_zero_grad()
        losses = loss.marginLoss()(m1(ph, pt, pr), m1(nh, nt, pr), FLAGS.margin)
        rel_ids = torch.cat([pr, pr])
        if model_type == 'transh':
            losses = losses + loss.orthogonalLoss(m1.rel_embeddings.weight, m1.norm_embeddings.weight, ids=rel_ids)
        losses = losses + loss.normLoss(m1.ent_embeddings.weight, ids=torch.cat([ph, pt, nh, nt])) + loss.normLoss(m1.rel_embeddings.weight, ids=rel_ids)
        losses.backward(); tr1.clip_and_step(FLAGS.clipping_max_value)
        fl = fast.kg_step(ph, pt, pr, nh, nt, pr)
        if not torch.allclose(fl, losses.detach(), rtol=2e-5, atol=2e-6):
            return 'step %d loss %.7g vs %.7g' % (step, float(fl), float(losses))
        msg = tables_off(m1, m2)
        if msg:
            return 'step %d %s' % (step, msg)
    return None


def run(tag, fn):
    ran[0] += 1
    try:
        msg = fn()
    except Exception as e:                                    # noqa: BLE001
        msg = '%s: %s' % (type(e).__name__, str(e)[:200])
    if msg:
        bad.append((tag, msg))


for D in [int(x) for x in sys.argv[1:]] or [7, 20, 36, 50, 64, 100, 128, 132, 200, 256, 300]:
    for B in (1, 63, 513):
        for l1 in (False, True):
            t = 'B=%d %s' % (B, 'l1' if l1 else 'l2')
            for P in (1, 5, 13, 33, 40) if D % 4 == 0 else ():       # (other widths: the drivers take the autograd route, item_recommendation.py:70)
                run('transup D=%d P=%d %s' % (D, P, t), lambda: rec_case('transup', D, P, B, l1))
            for NR in (1, 6, 50):
                for mt in ('transe', 'transh') + (('transr',) if D <= 64 else ()):
                    run('%s D=%d NR=%d %s' % (mt, D, NR, t), lambda: kg_case(mt, D, NR, B, l1))
        run('bprmf D=%d B=%d' % (D, B), lambda: rec_case('bprmf', D, 1, B, False))
    print('D=%d done: %d cases, %d problems' % (D, ran[0], len(bad)), flush=True)
for b in bad:
    print('PROBLEM %s: %s' % b)
sys.exit(1 if bad else 0)
