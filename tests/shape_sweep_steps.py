"""Shape sweep of the GPU-resident training steps (run by hand on a GPU box: `python tests/shape_sweep_steps.py`): JointStepper's rec /
kg steps (fused kernels where they exist, the multi-launch sequence elsewhere) against the autograd route that mirrors the reference's
step body (knowledgable_recommendation.py:330-401), over widths, preference / relation counts, batch sizes and both distances.  The
comparison is test_fast_train.test_fast_steps_match_the_autograd_route's: the loss to 1e-5, tables element-wise with a capped count
of strays."""
import copy
import logging
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch

DEV = torch.device('cuda', 0)
bad, ran = [], [0]


def build(tmp, D, NR, l1):
    from jTransUP.models import jTransUP as jt
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    from jTransUP.utils.trainer import ModelTrainer
    get_flags(); FLAGS.reset()
    FLAGS(['prog', '-model_type', 'jtransup', '-noshare_embeddings', '-log_path', tmp, '-experiment_name', 'ft',
           '-optimizer_type', os.environ.get('SWEEP_OPT', 'SGD'), '-learning_rate', '0.05', '-kg_lambda', '0.5'])
    FLAGS.ckpt_path = tmp
    NU, NI, NE = 50, 40, 70
    i_map = {i: i for i in range(NI)}
    new_map = {i: ((i * 3) % NE if i % 5 else -1, i) for i in range(NI)}
    torch.manual_seed(4)
    m = jt.jTransUPModel(l1, D, NU, NI, NE, NR, i_map, new_map, False, False)
    tr = ModelTrainer(m, logging.getLogger('ft'), 10, FLAGS)
    return FLAGS, m, tr, (NU, NI, NE, NR)


def case(D, NR, B, l1):
    from jTransUP.utils import loss
    from jTransUP.utils.fast_train import JointStepper
    tmp = tempfile.mkdtemp()
    FLAGS, m1, tr1, (NU, NI, NE, NR) = build(tmp, D, NR, l1)
    _, m2, tr2, _ = build(tmp, D, NR, l1)
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    fast = JointStepper(m2, tr2, FLAGS, B)
    gen = torch.Generator().manual_seed(9)
    rnd = lambda hi: torch.randint(0, hi, (B,), generator=gen).to(DEV)
    for step, is_rec in enumerate([True, False, True, False]):
        tr1.optimizer_zero_grad()
        if is_rec:
            u, pi, ni = rnd(NU), rnd(NI), rnd(NI)
            pos, neg = m1((u, pi), None, is_rec=True), m1((u, ni), None, is_rec=True)
            losses = loss.bprLoss(pos, neg, target=tr1.model_target) + loss.orthogonalLoss(m1.pref_embeddings.weight, m1.pref_norm_embeddings.weight)
            losses.backward(); tr1.clip_and_step(FLAGS.clipping_max_value)
            fast_loss = fast.rec_step(u, pi, ni)
        else:
            ph, pt, pr, nh, nt = rnd(NE), rnd(NE), rnd(NR), rnd(NE), rnd(NE)
            pos, neg = m1(None, (ph, pt, pr), is_rec=False), m1(None, (nh, nt, pr), is_rec=False)
            rel_ids = torch.cat([pr, pr])
            losses = loss.marginLoss()(pos, neg, FLAGS.margin) + loss.orthogonalLoss(m1.rel_embeddings.weight, m1.norm_embeddings.weight, ids=rel_ids) \
                + loss.normLoss(m1.ent_embeddings.weight, ids=torch.cat([ph, pt, nh, nt])) + loss.normLoss(m1.rel_embeddings.weight, ids=rel_ids)
            losses = FLAGS.kg_lambda * losses
            losses.backward(); tr1.clip_and_step(FLAGS.clipping_max_value)
            fast_loss = fast.kg_step(ph, pt, pr, nh, nt, pr)
        if not torch.allclose(fast_loss, losses.detach(), rtol=2e-5, atol=2e-6):
            return 'step %d loss %.7g vs %.7g' % (step, float(fast_loss), float(losses))
        for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
            err = (b - a).abs()
            off = err > 2e-6 + 2e-5 * a.abs()
            if int(off.sum()) > max(6, int(2e-3 * off.numel())) or float(err.max()) > 2.1 * 0.05:
                return 'step %d table %s: %d of %d elements off, max %.3g' % (step, k, int(off.sum()), off.numel(), float(err.max()))
    if float(m2.ent_embeddings.weight[m2.ent_total - 1].abs().sum()) != 0.0:
        return 'pad row moved'
    return None


widths = [int(x) for x in sys.argv[1:]] or [20, 36, 64, 100, 128, 132, 200, 256, 300]
for D in widths:
    for NR in (1, 6, 20, 32, 33, 40):
        for B in (1, 63, 64, 513):
            for l1 in (False, True):
                if (NR, B) not in ((6, 64), (20, 513), (33, 63), (40, 64), (1, 1), (32, 64)) and D not in (64, 100, 256):
                    continue                                  # the full grid at three widths, a diagonal elsewhere
                ran[0] += 1
                tag = 'D=%d NR=%d B=%d %s' % (D, NR, B, 'l1' if l1 else 'l2')
                try:
                    msg = case(D, NR, B, l1)
                except Exception as e:                        # noqa: BLE001
                    msg = '%s: %s' % (type(e).__name__, str(e)[:200])
                if msg:
                    bad.append((tag, msg))
    print('D=%d done: %d cases, %d problems' % (D, ran[0], len(bad)), flush=True)
for b in bad:
    print('PROBLEM %s: %s' % b)
sys.exit(1 if bad else 0)
