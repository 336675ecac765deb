"""GPU parity of the reference baselines that reuse the accelerated kernels -- CKE (BPRMF over item + entity rows, TransR) and
CFKG (TransE with a "buy" relation) -- against the vectors the imported reference produced (tests/golden/baselines.npz: scores,
losses, gradients, all-candidate matrices), plus the seeded TransR d = 256 golden (projection table re-created from the seed)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NU, NI, NE, NR = 37, 45, 53, 7


def close(got, want, rtol=1e-4, atol=1e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


def _load(model, g, prefix):
    sd = {k: torch.from_numpy(g[prefix + k]).to(DEV) for k in model.state_dict()}
    model.load_state_dict(sd)


def _margin_and_norms(model, pos, neg, ph, pt, nh, nt, pr):
    from jTransUP.utils import loss
    out = loss.marginLoss()(pos, neg, 1.0)
    return out + loss.normLoss(model.ent_embeddings.weight, ids=torch.cat([ph, pt, nh, nt])) \
        + loss.normLoss(model.rel_embeddings.weight, ids=torch.cat([pr, pr]))


@pytest.mark.parametrize('d', [36, 64])
@pytest.mark.parametrize('l1', [False, True])
def test_cke_golden(golden, d, l1):
    from jTransUP.models import CKE
    from jTransUP.utils import loss
    g = golden('baselines')
    p = 'd%d.' % d
    tag = p + 'cke.%s.' % ('L1' if l1 else 'L2')
    i2e = g[p + 'cke.item2ent']
    i_map = {i: i for i in range(NI)}
    new_map = {i: ((int(i2e[i]) if i2e[i] != NE else -1), i) for i in range(NI)}
    m = CKE.CKE(l1, d, NU, NI, NE, NR, i_map, new_map)
    _load(m, g, p + 'cke.')
    assert m._item2ent.cpu().tolist() == i2e.tolist()
    ids = {k: torch.from_numpy(g[p + k]).long().to(DEV) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt', 'uq', 'eq', 'rq')}
    pos, neg = m((ids['u'], ids['pi']), None, is_rec=True), m((ids['u'], ids['ni']), None, is_rec=True)
    close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
    lo = loss.bprLoss(pos, neg, target=-1)
    close(lo, g[tag + 'rec.loss'])
    m.zero_grad(); lo.backward()
    for k, prm in m.named_parameters():
        key = tag + 'rec.grad.' + k
        if key in g:
            close(prm.grad, g[key], rtol=2e-4, atol=3e-5)
    assert float(m.ent_embeddings.weight.grad[NE].abs().sum()) == 0.0          # padding_idx row: no gradient
    pos, neg = m(None, (ids['ph'], ids['pt'], ids['pr']), is_rec=False), m(None, (ids['nh'], ids['nt'], ids['pr']), is_rec=False)
    close(pos, g[tag + 'kg.pos']); close(neg, g[tag + 'kg.neg'])
    lo = _margin_and_norms(m, pos, neg, ids['ph'], ids['pt'], ids['nh'], ids['nt'], ids['pr'])
    close(lo, g[tag + 'kg.loss'], rtol=2e-4)
    m.zero_grad(); lo.backward()
    for k, prm in m.named_parameters():
        key = tag + 'kg.grad.' + k
        if key in g and prm.grad is not None:
            close(prm.grad, g[key], rtol=3e-4, atol=1e-4)
    close(m.evaluateRec(ids['uq']), g[tag + 'evalRec'])
    close(m.evaluateHead(ids['eq'], ids['rq']), g[tag + 'evalHead'])
    close(m.evaluateTail(ids['eq'], ids['rq']), g[tag + 'evalTail'])


@pytest.mark.parametrize('d', [36, 64])
@pytest.mark.parametrize('l1', [False, True])
def test_cfkg_golden(golden, d, l1):
    from jTransUP.models import CFKG
    from jTransUP.utils import loss
    g = golden('baselines')
    p = 'd%d.' % d
    tag = p + 'cfkg.%s.' % ('L1' if l1 else 'L2')
    m = CFKG.CFKG(l1, d, NU, NI, NE, NR)
    sd = {k: torch.from_numpy(g[p + 'cfkg.' + k]).to(DEV) for k in ('user_embeddings.weight', 'ent_embeddings.weight', 'rel_embeddings.weight')}
    m.load_state_dict(sd, strict=False)
    assert m.item_embeddings is m.ent_embeddings and m.rel_embeddings.weight.shape[0] == NR + 1
    ids = {k: torch.from_numpy(g[p + k]).long().to(DEV) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt', 'uq', 'eq', 'rq')}
    pos, neg = m((ids['u'], ids['pi']), None, is_rec=True), m((ids['u'], ids['ni']), None, is_rec=True)
    close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
    lo = loss.bprLoss(pos, neg, target=-1)
    close(lo, g[tag + 'rec.loss'])
    m.zero_grad(); lo.backward()
    for k in ('user_embeddings.weight', 'ent_embeddings.weight', 'rel_embeddings.weight'):
        close(dict(m.named_parameters())[k].grad, g[tag + 'rec.grad.' + k], rtol=2e-4, atol=3e-5)
    pos, neg = m(None, (ids['ph'], ids['pt'], ids['pr']), is_rec=False), m(None, (ids['nh'], ids['nt'], ids['pr']), is_rec=False)
    close(pos, g[tag + 'kg.pos']); close(neg, g[tag + 'kg.neg'])
    lo = _margin_and_norms(m, pos, neg, ids['ph'], ids['pt'], ids['nh'], ids['nt'], ids['pr'])
    close(lo, g[tag + 'kg.loss'])
    m.zero_grad(); lo.backward()
    for k in ('ent_embeddings.weight', 'rel_embeddings.weight'):
        close(dict(m.named_parameters())[k].grad, g[tag + 'kg.grad.' + k], rtol=2e-4, atol=3e-5)
    close(m.evaluateRec(ids['uq']), g[tag + 'evalRec'])
    close(m.evaluateHead(ids['eq'], ids['rq']), g[tag + 'evalHead'])
    close(m.evaluateTail(ids['eq'], ids['rq']), g[tag + 'evalTail'])


@pytest.mark.parametrize('l1', [False, True])
def test_transr_d256_seeded_golden(golden, l1):
    """TransR at config 5's width against the reference: the (7 x 65536) projection table is drawn from the fixture's seed."""
    from jTransUP.hip import ops
    g = golden('transr_d256')
    d = 256
    gen = torch.Generator().manual_seed(int(g['seed'][0]))
    E = (torch.randn(NE, d, generator=gen) * 0.3).to(DEV).requires_grad_(True)
    R = (torch.randn(NR, d, generator=gen) * 0.3).to(DEV).requires_grad_(True)
    M = (torch.randn(NR, d * d, generator=gen) * 0.06).to(DEV).requires_grad_(True)
    ph, pt, pr, nh, nt = (torch.from_numpy(g[k]).long().to(DEV) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
    tag = 'L1.' if l1 else 'L2.'
    pos, neg = ops.score_transr(E, R, M, ph, pt, pr, l1), ops.score_transr(E, R, M, nh, nt, pr, l1)
    close(pos, g[tag + 'pos']); close(neg, g[tag + 'neg'])
    torch.sum(torch.clamp(pos - neg + 1.0, min=0.0)).backward()
    close(E.grad, g[tag + 'grad.ent'], rtol=3e-4, atol=2e-4); close(R.grad, g[tag + 'grad.rel'], rtol=3e-4, atol=2e-4)
    close(M.grad.sum(1), g[tag + 'grad.proj.rowsum'], rtol=1e-3, atol=5e-3)
    close(M.grad[:, ::997], g[tag + 'grad.proj.sample'], rtol=3e-4, atol=2e-4)


@pytest.mark.parametrize('d', [36, 64])
def test_fm_golden(golden, d):
    """fm.py: global + user + item bias + u . i, bprLoss with target +1 (trainer.py:15-17), gradients of all five parameters, and the
    all-item evaluation matrix."""
    from jTransUP.models import fm
    from jTransUP.utils import loss
    g = golden('fm_cofm')
    p = 'd%d.' % d
    m = fm.FM(d, NU, NI)
    assert all(tuple(v.shape) == g[p + 'fm.' + k].shape for k, v in m.state_dict().items())      # biases are 1-D like the reference's
    assert set(m.state_dict()) == {'user_embeddings.weight', 'item_embeddings.weight', 'user_bias.weight', 'item_bias.weight', 'bias'}
    _load(m, g, p + 'fm.')
    ids = {k: torch.from_numpy(g[p + k]).long().to(DEV) for k in ('u', 'pi', 'ni', 'uq')}
    pos, neg = m(ids['u'], ids['pi']), m(ids['u'], ids['ni'])
    close(pos, g[p + 'fm.pos']); close(neg, g[p + 'fm.neg'])
    lo = loss.bprLoss(pos, neg, target=1)
    close(lo, g[p + 'fm.loss'])
    m.zero_grad(); lo.backward()
    for k, prm in m.named_parameters():
        close(prm.grad, g[p + 'fm.grad.' + k], rtol=2e-4, atol=3e-5)
    close(m.evaluate(ids['uq']), g[p + 'fm.eval'])


@pytest.mark.parametrize('d', [36, 64])
@pytest.mark.parametrize('share', [False, True])
@pytest.mark.parametrize('l1', [False, True])
def test_cofm_golden(golden, d, share, l1):
    """cofm.py: FM on the ratings, TransE on the triples, item table = entity table with -share_embeddings."""
    from jTransUP.models import cofm
    from jTransUP.utils import loss
    g = golden('fm_cofm')
    p = 'd%d.' % d
    kind = 'share' if share else 'own'
    tag = p + 'cofm.%s.%s.' % (kind, 'L1' if l1 else 'L2')
    m = cofm.coFM(l1, d, NU, NE if share else NI, NE, NR, share)
    pre = p + 'cofm.%s.' % kind                    # a shared item / entity table is stored once (named_parameters de-duplicates)
    m.load_state_dict({k: torch.from_numpy(g[pre + k]).to(DEV) for k in m.state_dict() if pre + k in g}, strict=False)
    assert (m.item_embeddings is m.ent_embeddings) == share
    ids = {k: torch.from_numpy(g[p + k]).long().to(DEV) for k in ('u', 'pi', 'ni', 'ph', 'pt', 'pr', 'nh', 'nt', 'uq', 'eq', 'rq')}
    pos, neg = m((ids['u'], ids['pi']), None, is_rec=True), m((ids['u'], ids['ni']), None, is_rec=True)
    close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
    lo = loss.bprLoss(pos, neg, target=1)
    close(lo, g[tag + 'rec.loss'])
    m.zero_grad(); lo.backward()
    for k, prm in m.named_parameters():
        key = tag + 'rec.grad.' + k
        if key in g:
            close(prm.grad, g[key], rtol=2e-4, atol=3e-5)
    pos, neg = m(None, (ids['ph'], ids['pt'], ids['pr']), is_rec=False), m(None, (ids['nh'], ids['nt'], ids['pr']), is_rec=False)
    close(pos, g[tag + 'kg.pos']); close(neg, g[tag + 'kg.neg'])
    lo = _margin_and_norms(m, pos, neg, ids['ph'], ids['pt'], ids['nh'], ids['nt'], ids['pr'])
    close(lo, g[tag + 'kg.loss'], rtol=2e-4)
    m.zero_grad(); lo.backward()
    for k, prm in m.named_parameters():
        key = tag + 'kg.grad.' + k
        if key in g and prm.grad is not None:
            close(prm.grad, g[key], rtol=3e-4, atol=1e-4)
    close(m.evaluateRec(ids['uq']), g[tag + 'evalRec'])
    close(m.evaluateHead(ids['eq'], ids['rq']), g[tag + 'evalHead'])
    close(m.evaluateTail(ids['eq'], ids['rq']), g[tag + 'evalTail'])
