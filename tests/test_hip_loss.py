"""GPU parity of the loss / regulariser kernels (K8-K10) against the golden losses + gradients and the oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, rtol=1e-4, atol=1e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
def test_full_transh_loss_and_grads_golden(golden, d, l1):
    """marginLoss + orthogonalLoss + normLoss x2 exactly as knowledge_representation.py:189-204, all through HIP."""
    from jTransUP.hip import ops
    from jTransUP.utils import loss as Lf
    g = golden('score_d%d' % d)
    tag = 'transh.%s.' % ('L1' if l1 else 'L2')
    E, R, N = (dv(g['transh.%s.weight' % k]).requires_grad_(True) for k in ('ent_embeddings', 'rel_embeddings', 'norm_embeddings'))
    ph, pt, pr, nh, nt = (dv(g[k]) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
    pos, neg = ops.score_transh(E, R, N, ph, pt, pr, l1), ops.score_transh(E, R, N, nh, nt, pr, l1)
    rel_ids = torch.cat([pr, pr])
    loss = Lf.marginLoss()(pos, neg, 1.0) + Lf.orthogonalLoss(R, N, ids=rel_ids) \
        + Lf.normLoss(E, ids=torch.cat([ph, pt, nh, nt])) + Lf.normLoss(R, ids=rel_ids)
    close(loss, g[tag + 'loss'])
    loss.backward()
    close(E.grad, g[tag + 'grad.ent_embeddings.weight'], atol=3e-5)
    close(R.grad, g[tag + 'grad.rel_embeddings.weight'], atol=3e-5)
    close(N.grad, g[tag + 'grad.norm_embeddings.weight'], atol=3e-5)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('gum', [False, True])
def test_full_tup_loss_and_grads_golden(golden, d, gum):
    """bprLoss(target=-1) + orthogonalLoss(P, Pn) + normLoss(users) + normLoss(items) + normLoss(P), item_recommendation.py:175-180."""
    from jTransUP.hip import ops
    from jTransUP.utils import loss as Lf
    g = golden('score_d%d' % d)
    tag = 'tup.L1.%s.' % ('hard' if gum else 'soft')
    U, I, P, Pn = (dv(g['tup.%s.weight' % k]).requires_grad_(True) for k in ('user_embeddings', 'item_embeddings', 'pref_embeddings', 'pref_norm_embeddings'))
    u, pi, ni = dv(g['u']), dv(g['pi']), dv(g['ni'])
    mode = ops.GUMBEL_INPUT if gum else ops.GUMBEL_OFF
    pos = ops.score_tup(U, I, P, Pn, u, pi, True, mode, dv(g[tag + 'uni_pos']) if gum else None)
    neg = ops.score_tup(U, I, P, Pn, u, ni, True, mode, dv(g[tag + 'uni_neg']) if gum else None)
    loss = Lf.bprLoss(pos, neg, target=-1) + Lf.orthogonalLoss(P, Pn) + Lf.normLoss(U, ids=u) \
        + Lf.normLoss(I, ids=torch.cat([pi, ni])) + Lf.normLoss(P)
    close(loss, g[tag + 'loss'])
    loss.backward()
    for w, k in ((U, 'user_embeddings'), (I, 'item_embeddings'), (P, 'pref_embeddings'), (Pn, 'pref_norm_embeddings')):
        close(w.grad, g[tag + 'grad.%s.weight' % k], atol=3e-5)


def test_bpr_target_plus_one_golden(golden):
    from jTransUP.utils import loss as Lf
    g = golden('score_d64')
    pos, neg = dv(g['bprmf.pos']).requires_grad_(True), dv(g['bprmf.neg']).requires_grad_(True)
    loss = Lf.bprLoss(pos, neg, target=1)
    close(loss, g['bprmf.loss'])
    loss.backward()
    p2, n2 = torch.from_numpy(g['bprmf.pos']).requires_grad_(True), torch.from_numpy(g['bprmf.neg']).requires_grad_(True)
    O.bpr_loss(p2, n2, 1.0).backward()
    close(pos.grad, p2.grad, atol=1e-7); close(neg.grad, n2.grad, atol=1e-7)


@pytest.mark.parametrize('n', [1, 5, 512, 100000])
def test_losses_vs_oracle_sizes(n):
    from jTransUP.utils import loss as Lf
    gen = torch.Generator().manual_seed(n)
    pos, neg = torch.randn(n, generator=gen) * 3, torch.randn(n, generator=gen) * 3
    for tgt in (1.0, -1.0):
        pc, nc = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
        pd, nd = pos.to(DEV).requires_grad_(True), neg.to(DEV).requires_grad_(True)
        want = O.bpr_loss(pc, nc, tgt); got = Lf.bprLoss(pd, nd, tgt)
        close(got, want, rtol=2e-5, atol=1e-6)
        (want * 3).backward(); (got * 3).backward()
        close(pd.grad, pc.grad, atol=1e-7); close(nd.grad, nc.grad, atol=1e-7)
    pc, nc = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
    pd, nd = pos.to(DEV).requires_grad_(True), neg.to(DEV).requires_grad_(True)
    want = O.margin_loss(pc, nc, 1.0); got = Lf.marginLoss()(pd, nd, 1.0)
    close(got, want, rtol=2e-5, atol=1e-4)
    want.backward(); got.backward()
    close(pd.grad, pc.grad, atol=0); close(nd.grad, nc.grad, atol=0)


def test_gathered_call_shape_still_works():
    """The reference's call shape: normLoss(model.ent_embeddings(ids)) on an already-gathered (n x d) tensor."""
    from jTransUP.utils import loss as Lf
    gen = torch.Generator().manual_seed(0)
    T = (torch.randn(50, 100, generator=gen) * 0.2)
    ids = torch.randint(0, 50, (333,), generator=gen)
    Tc = T.clone().requires_grad_(True); Td = T.to(DEV).requires_grad_(True)
    want = O.norm_loss(Tc[ids]); got = Lf.normLoss(Td[ids.to(DEV)])
    close(got, want)
    want.backward(); got.backward()
    close(Td.grad, Tc.grad, atol=1e-5)


@pytest.mark.parametrize('n,d', [(1, 64), (333, 100), (2048, 36)])
def test_fused_value_and_gradient_entries_vs_oracle(n, d):
    """The one-launch value+gradient entry points used by the GPU-resident steppers: the loss is ADDED to the accumulator,
    the gradient (scaled by the device scalar `gloss`) is ADDED to the table gradient / written to gpos, gneg."""
    from jTransUP.hip import lib as L
    gen = torch.Generator().manual_seed(n + d)
    p = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    gl = torch.full((), 0.75, device=DEV)
    # pair losses
    pos, neg = torch.randn(n, generator=gen) * 2, torch.randn(n, generator=gen) * 2
    for name, param, ref in (('ktup_loss_bpr_fused', -1.0, lambda a, b: O.bpr_loss(a, b, -1.0)),
                             ('ktup_loss_margin_fused', 1.0, lambda a, b: O.margin_loss(a, b, 1.0))):
        pc, nc = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
        want = ref(pc, nc); (want * 0.75).backward()
        pd, nd = pos.to(DEV), neg.to(DEV)
        acc = torch.full((1,), 2.0, device=DEV); gp, gn = torch.empty_like(pd), torch.empty_like(nd)
        L.call(name, p(pd), p(nd), n, param, p(gl), p(acc), p(gp), p(gn), st)
        close(acc[0] - 2.0, want, rtol=1e-4, atol=1e-4)
        close(gp, pc.grad, atol=1e-7); close(gn, nc.grad, atol=1e-7)
    # regularisers over gathered rows (duplicates on purpose) and over whole tables (ids = NULL)
    rows = 57
    T, N = torch.randn(rows, d, generator=gen) * 0.3, torch.randn(rows, d, generator=gen) * 0.3
    ids = torch.randint(0, rows, (n,), generator=gen)
    for use_ids in (True, False):
        Tc, Nc = T.clone().requires_grad_(True), N.clone().requires_grad_(True)
        sel = (lambda w: w[ids]) if use_ids else (lambda w: w)
        want_n = O.norm_loss(sel(Tc)); want_o = O.orthogonal_loss(sel(Tc), sel(Nc))
        ((want_n + want_o) * 0.75).backward()
        Td, Nd = T.to(DEV), N.to(DEV)
        gT, gN = torch.ones_like(Td), torch.ones_like(Nd)          # accumulate on top of existing gradients
        acc = torch.zeros(2, device=DEV)
        idp, cnt = (p(ids.to(DEV)), n) if use_ids else (None, rows)
        idd = ids.to(DEV)
        idp = p(idd) if use_ids else None
        L.call('ktup_reg_norm_fused', p(Td), Td.stride(0), d, idp, cnt, p(gl), p(acc[0:]), p(gT), st)
        L.call('ktup_reg_orth_fused', p(Td), Td.stride(0), p(Nd), Nd.stride(0), d, idp, cnt, p(gl), p(acc[1:]), p(gT), p(gN), st)
        close(acc[0], want_n, rtol=1e-4, atol=1e-5); close(acc[1], want_o, rtol=1e-4, atol=1e-5)
        close(gT - 1.0, Tc.grad, rtol=1e-4, atol=2e-5); close(gN - 1.0, Nc.grad, rtol=1e-4, atol=2e-5)
