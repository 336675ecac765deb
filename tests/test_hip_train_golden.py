"""The DEFAULT training path of the three drivers -- the GPU-resident steppers of utils/fast_train.py (fused step kernels, one-launch
clip + optimizer, HIP-graph replay) -- against goldens produced by the REFERENCE's own step lines and its own ModelTrainer
(tests/golden/make_goldens.py train_step_cases: knowledgable_recommendation.py:335-403, item_recommendation.py:160-192,
knowledge_representation.py:176-216, utils/trainer.py:63-81): per-step losses, and every table after the last step.  Not a
comparison with this repo's own autograd route (tests/test_fast_train.py does that): the other end of this one is the reference.
Tolerance: 1e-4 on the tables (north_star), with the same stray-element allowance as test_fast_train for Adagrad / Adam elements
whose accumulated gradient is itself rounding noise."""
import json
import logging
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NU, NI, NE, NR, NP_TUP, B, D = 37, 45, 53, 7, 5, 48, 64


@pytest.fixture(scope='module')
def G():
    return np.load(os.path.join(GOLDEN, 'train_steps.npz')), json.load(open(os.path.join(GOLDEN, 'train_steps.json')))


def _flags(tmp_path, model_type, opt, lr, l2, clip, extra=()):
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    get_flags(); FLAGS.reset()
    FLAGS(['prog', '-model_type', model_type, '-log_path', str(tmp_path), '-experiment_name', 'g4', '-optimizer_type', opt,
           '-learning_rate', str(lr), '-l2_lambda', str(l2), '-clipping_max_value', str(clip), '-margin', '1.0'] + list(extra))
    FLAGS.ckpt_path = str(tmp_path)
    return FLAGS


def _load(model, g, prefix):
    sd = {k: torch.from_numpy(g[prefix + k]) for k in model.state_dict().keys()}
    model.load_state_dict(sd)


def _compare(model, g, tag, opt):
    for k, v in model.state_dict().items():
        want = torch.from_numpy(g[tag + 'final.' + k])
        got = v.detach().cpu()
        err = (got - want).abs()
        bad = err > 1e-5 + 1e-4 * want.abs()
        # a last-bit difference of an accumulated gradient becomes visible where Adagrad's sum / Adam's v is ~eps^2
        assert float(bad.float().mean()) <= (2e-2 if opt == 'Adam' else 2e-3) and float(err.max()) <= 5e-4, \
            '%s: %d of %d elements off, max %.3g' % (k, int(bad.sum()), bad.numel(), float(err.max()))


def _t(g, key):
    return torch.from_numpy(g[key]).to(DEV)


@pytest.mark.parametrize('opt,lr,l2', [('Adagrad', 0.05, 0.0), ('Adagrad', 0.05, 1e-5), ('Adam', 0.01, 0.0), ('Adam', 0.01, 1e-5),
                                       ('SGD', 0.05, 1e-5)])
def test_joint_stepper_reproduces_the_reference_steps(tmp_path, G, opt, lr, l2):
    """KTUP: rec, rec, kg, rec, kg, kg -- the fused rec and kg step kernels + ktup_optim_clip_step, replayed from graphs from the
    third step of each kind on."""
    from jTransUP.models import jTransUP as jt
    from jTransUP.utils.fast_train import JointStepper
    from jTransUP.utils.trainer import ModelTrainer
    g, _ = G
    FLAGS = _flags(tmp_path, 'jtransup', opt, lr, l2, 5.0, ['-noshare_embeddings', '-kg_lambda', str(float(g['ktup.kg_lambda'][0]))])
    i2e = g['ktup.item2ent']
    i_map = {i: i for i in range(NI)}
    new_map = {i: (int(i2e[i]) if int(i2e[i]) != NE else -1, i) for i in range(NI)}
    m = jt.jTransUPModel(False, D, NU, NI, NE, NR, i_map, new_map, False, False)
    _load(m, g, 'ktup.init.')
    assert torch.equal(m._item2ent.cpu().long(), torch.from_numpy(i2e))
    tr = ModelTrainer(m, logging.getLogger('g4'), 10, FLAGS)
    st = JointStepper(m, tr, FLAGS, B)
    tag = 'ktup.%s.l2_%g.' % (opt, l2)
    for s, is_rec in enumerate(g['ktup.kinds']):
        b = {k: _t(g, 'ktup.batch%d.%s' % (s, k)) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt')}
        loss = st.rec_step(b['u'], b['pi'], b['ni']) if is_rec else st.kg_step(b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr'])
        np.testing.assert_allclose(float(loss), g[tag + 'losses'][s], rtol=1e-4)
    assert st.fused_step and tr.step == 6
    _compare(m, g, tag, opt)


@pytest.mark.parametrize('gum', [False, True])
@pytest.mark.parametrize('opt,lr', [('Adagrad', 0.05), ('Adam', 0.01)])
def test_rec_stepper_reproduces_the_reference_steps(tmp_path, G, gum, opt, lr):
    """TUP, soft gate and ST-Gumbel gate (the reference's recorded uniforms fed through the parity hook), three steps."""
    from jTransUP.models import transUP as tu
    from jTransUP.utils.fast_train import RecStepper
    from jTransUP.utils.trainer import ModelTrainer
    g, _ = G
    tag = 'tup.%s.%s.' % ('hard' if gum else 'soft', opt)
    FLAGS = _flags(tmp_path, 'transup', opt, lr, 1e-5, float(g[tag + 'clip'][0]), ['-num_preferences', str(NP_TUP)] + (['-use_st_gumbel'] if gum else []))
    m = tu.TransUPModel(False, D, NU, NI, NP_TUP, gum)
    _load(m, g, 'tup.init.')
    tr = ModelTrainer(m, logging.getLogger('g4'), 10, FLAGS)
    st = RecStepper(m, tr, FLAGS, B)
    for s in range(3):
        b = {k: _t(g, 'tup.batch%d.%s' % (s, k)) for k in ('u', 'pi', 'ni')}
        if gum:
            st.set_gumbel_uniforms(torch.cat([_t(g, tag + 'uni%d.pos' % s), _t(g, tag + 'uni%d.neg' % s)]))
        loss = st.rec_step(b['u'], b['pi'], b['ni'])
        np.testing.assert_allclose(float(loss), g[tag + 'losses'][s], rtol=1e-4)
    assert st.fused_step
    _compare(m, g, tag, opt)


@pytest.mark.parametrize('name', ['transe', 'transh'])
@pytest.mark.parametrize('opt,lr', [('Adagrad', 0.05), ('Adam', 0.01)])
def test_kg_stepper_reproduces_the_reference_steps(tmp_path, G, name, opt, lr):
    from jTransUP.models import transE, transH
    from jTransUP.utils.fast_train import KGStepper
    from jTransUP.utils.trainer import ModelTrainer
    g, _ = G
    FLAGS = _flags(tmp_path, name, opt, lr, 1e-5, 5.0)
    m = (transE.TransEModel if name == 'transe' else transH.TransHModel)(False, D, NE, NR)
    _load(m, g, name + '.init.')
    tr = ModelTrainer(m, logging.getLogger('g4'), 10, FLAGS)
    st = KGStepper(m, tr, FLAGS, B)
    tag = '%s.%s.' % (name, opt)
    for s in range(3):
        b = {k: _t(g, 'kg.batch%d.%s' % (s, k)) for k in ('ph', 'pt', 'pr', 'nh', 'nt')}
        loss = st.kg_step(b['ph'], b['pt'], b['pr'], b['nh'], b['nt'], b['pr'])
        np.testing.assert_allclose(float(loss), g[tag + 'losses'][s], rtol=1e-4)
    assert st.fused_step
    _compare(m, g, tag, opt)
