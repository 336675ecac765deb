"""GPU parity of the fused gather+score kernels (K1-K7) and their backward, through the C ABI.

Checked against (1) the golden vectors the reference produced, (2) the CPU oracle on seeded inputs at sizes it
finishes in seconds, including ragged tails, unaligned/odd-d tables and the ml1m shape.
Tolerance: north_star asks for 1e-4 on fp32 results; the kernels sum in a different order than torch, so scores
use rtol 1e-4 / atol 1e-5 and gradients (atomic accumulation order) rtol 1e-4 / atol 3e-5.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
RT, AT, GAT = 1e-4, 1e-5, 3e-5


def dv(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


def close(got, want, rtol=RT, atol=AT):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


def leaf(a):
    return dv(a).clone().requires_grad_(True)


def ops():
    from jTransUP.hip import ops as _ops
    return _ops


# torch-op regularisers on the device, used only to assemble the golden losses' gradients here
def t_norm_loss(e):
    return torch.sum(torch.clamp(torch.sum(e ** 2, dim=1, keepdim=True) - 1.0, min=0.0))


def t_orth_loss(rel, nrm):
    return torch.sum(torch.sum(nrm * rel, dim=1, keepdim=True) ** 2 / torch.sum(rel ** 2, dim=1, keepdim=True))


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
def test_bprmf_golden(golden, d):
    g = golden('score_d%d' % d)
    U, I = leaf(g['bprmf.user_embeddings.weight']), leaf(g['bprmf.item_embeddings.weight'])
    u, pi, ni = dv(g['u']), dv(g['pi']), dv(g['ni'])
    pos, neg = ops().score_bprmf(U, I, u, pi), ops().score_bprmf(U, I, u, ni)
    close(pos, g['bprmf.pos']); close(neg, g['bprmf.neg'])
    (-F.logsigmoid(pos - neg)).mean().backward()
    close(U.grad, g['bprmf.grad.user_embeddings.weight'], atol=GAT)
    close(I.grad, g['bprmf.grad.item_embeddings.weight'], atol=GAT)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('name', ['transe', 'transh', 'transr'])
def test_kg_golden(golden, d, l1, name):
    if name == 'transr' and d == 256:
        pytest.skip('no TransR golden at d=256 (the projection table would make the fixture 6 MB)')
    g = golden('score_d%d' % d)
    tag = '%s.%s.' % (name, 'L1' if l1 else 'L2')
    E, R = leaf(g[name + '.ent_embeddings.weight']), leaf(g[name + '.rel_embeddings.weight'])
    ph, pt, pr, nh, nt = (dv(g[k]) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
    X = None
    if name == 'transe':
        f = lambda h, t: ops().score_transe(E, R, h, t, pr, l1)
    elif name == 'transh':
        X = leaf(g['transh.norm_embeddings.weight'])
        f = lambda h, t: ops().score_transh(E, R, X, h, t, pr, l1)
    else:
        X = leaf(g['transr.proj_embeddings.weight'])
        f = lambda h, t: ops().score_transr(E, R, X, h, t, pr, l1)
    pos, neg = f(ph, pt), f(nh, nt)
    close(pos, g[tag + 'pos']); close(neg, g[tag + 'neg'])              # TransR too: measured 3e-7 relative against the goldens
    loss = torch.sum(torch.clamp(pos - neg + 1.0, min=0.0))
    rel = R[torch.cat([pr, pr])]
    if name == 'transh':
        loss = loss + t_orth_loss(rel, X[torch.cat([pr, pr])])
    loss = loss + t_norm_loss(E[torch.cat([ph, pt, nh, nt])]) + t_norm_loss(rel)
    close(loss, g[tag + 'loss'], rtol=1e-4)
    loss.backward()
    gat = 2e-4 if name == 'transr' else GAT
    close(E.grad, g[tag + 'grad.ent_embeddings.weight'], atol=gat)
    close(R.grad, g[tag + 'grad.rel_embeddings.weight'], atol=gat)
    if name == 'transh':
        close(X.grad, g[tag + 'grad.norm_embeddings.weight'], atol=gat)
    if name == 'transr':
        close(X.grad, g[tag + 'grad.proj_embeddings.weight'], atol=gat)


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('gum', [False, True])
def test_tup_golden(golden, d, l1, gum):
    g = golden('score_d%d' % d)
    tag = 'tup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
    U, I = leaf(g['tup.user_embeddings.weight']), leaf(g['tup.item_embeddings.weight'])
    P, Pn = leaf(g['tup.pref_embeddings.weight']), leaf(g['tup.pref_norm_embeddings.weight'])
    u, pi, ni = dv(g['u']), dv(g['pi']), dv(g['ni'])
    mode = ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF
    up = dv(g[tag + 'uni_pos']) if gum else None
    un = dv(g[tag + 'uni_neg']) if gum else None
    pos = ops().score_tup(U, I, P, Pn, u, pi, l1, mode, up)
    neg = ops().score_tup(U, I, P, Pn, u, ni, l1, mode, un)
    close(pos, g[tag + 'pos']); close(neg, g[tag + 'neg'])
    loss = (-F.logsigmoid(-(pos - neg))).mean() + t_orth_loss(P, Pn) + t_norm_loss(U[u]) \
        + t_norm_loss(I[torch.cat([pi, ni])]) + t_norm_loss(P)
    close(loss, g[tag + 'loss'])
    loss.backward()
    for w, k in ((U, 'user_embeddings'), (I, 'item_embeddings'), (P, 'pref_embeddings'), (Pn, 'pref_norm_embeddings')):
        want = g[tag + 'grad.%s.weight' % k]
        close(w.grad, want, atol=max(GAT, 2e-6 * float(np.abs(want).max())))   # fp32 sums of a batch: the floor scales with the gradient


KT = ['user_embeddings', 'item_embeddings', 'ent_embeddings', 'pref_embeddings', 'pref_norm_embeddings', 'rel_embeddings',
      'norm_embeddings']


@pytest.mark.parametrize('d', [36, 50, 64, 100, 256])
@pytest.mark.parametrize('l1', [False, True])
@pytest.mark.parametrize('gum', [False, True])
def test_ktup_golden(golden, d, l1, gum):
    g = golden('score_d%d' % d)
    tag = 'ktup.%s.%s.' % ('L1' if l1 else 'L2', 'hard' if gum else 'soft')
    W = {k: leaf(g['ktup.%s.weight' % k]) for k in KT}
    i2e = dv(g['ktup.item2ent'], torch.int32)
    pad = W['ent_embeddings'].shape[0] - 1
    u, pi, ni = dv(g['u']), dv(g['pi']), dv(g['ni'])
    mode = ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF
    up = dv(g[tag + 'uni_pos']) if gum else None
    un = dv(g[tag + 'uni_neg']) if gum else None
    rec = lambda i, uni: ops().score_ktup(W['user_embeddings'], W['item_embeddings'], W['ent_embeddings'], W['pref_embeddings'],
                                          W['pref_norm_embeddings'], W['rel_embeddings'], W['norm_embeddings'], i2e, u, i, l1,
                                          mode, uni, ent_pad=pad)
    pos, neg = rec(pi, up), rec(ni, un)
    close(pos, g[tag + 'rec.pos']); close(neg, g[tag + 'rec.neg'])
    loss = (-F.logsigmoid(-(pos - neg))).mean() + t_orth_loss(W['pref_embeddings'], W['pref_norm_embeddings'])
    close(loss, g[tag + 'rec.loss'])
    loss.backward()
    for k in KT:
        key = tag + 'rec.grad.%s.weight' % k
        if key in g:
            close(W[k].grad, g[key], atol=max(GAT, 2e-6 * float(np.abs(g[key]).max())))      # incl. the pad row staying exactly zero
    if not gum:
        for w in W.values():
            w.grad = None
        ph, pt, pr, nh, nt = (dv(g[k]) for k in ('ph', 'pt', 'pr', 'nh', 'nt'))
        kg = lambda h, t: ops().score_transh(W['ent_embeddings'], W['rel_embeddings'], W['norm_embeddings'], h, t, pr, l1)
        pos, neg = kg(ph, pt), kg(nh, nt)
        close(pos, g[tag + 'kg.pos']); close(neg, g[tag + 'kg.neg'])


def rand_world(seed, nu, ni, ne, nr, d, device=DEV):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda r: O.make_table(r, d, gen)
    W = dict(U=mk(nu), I=mk(ni), E=torch.cat([mk(ne), torch.zeros(1, d)]), P=mk(nr), Pn=mk(nr), R=mk(nr), Rn=mk(nr))
    i2e = torch.randint(0, ne, (ni,), generator=gen)
    i2e[torch.rand(ni, generator=gen) < 0.1] = ne          # ~10 % of items map to the pad row
    return W, i2e, gen


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 130, 1000])
def test_ragged_tails_vs_oracle(n):
    W, i2e, gen = rand_world(1, 50, 60, 70, 20, 100)
    u = torch.randint(0, 50, (n,), generator=gen); i = torch.randint(0, 60, (n,), generator=gen)
    h = torch.randint(0, 70, (n,), generator=gen); t = torch.randint(0, 70, (n,), generator=gen)
    r = torch.randint(0, 20, (n,), generator=gen)
    D = {k: v.to(DEV) for k, v in W.items()}
    for l1 in (False, True):
        close(ops().score_transe(D['E'], D['R'], h.to(DEV), t.to(DEV), r.to(DEV), l1), O.score_transe(W['E'], W['R'], h, t, r, l1))
        close(ops().score_transh(D['E'], D['R'], D['Rn'], h.to(DEV), t.to(DEV), r.to(DEV), l1),
              O.score_transh(W['E'], W['R'], W['Rn'], h, t, r, l1))
        close(ops().score_tup(D['U'], D['I'], D['P'], D['Pn'], u.to(DEV), i.to(DEV), l1), O.score_tup(W['U'], W['I'], W['P'], W['Pn'], u, i, l1))
        close(ops().score_ktup(D['U'], D['I'], D['E'], D['P'], D['Pn'], D['R'], D['Rn'], i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), l1),
              O.score_ktup_rec(W['U'], W['I'], W['E'], W['P'], W['Pn'], W['R'], W['Rn'], i2e, u, i, l1))
    close(ops().score_bprmf(D['U'], D['I'], u.to(DEV), i.to(DEV)), O.score_bprmf(W['U'], W['I'], u, i))


@pytest.mark.parametrize('d', [50, 7, 130, 256])
def test_odd_and_unaligned_tables_vs_oracle(d):
    """d % 4 != 0 and tables that are views at a 4-byte offset take the 4-byte-lane path of K1-K3."""
    gen = torch.Generator().manual_seed(d)
    n = 300
    E = O.make_table(40, d, gen); R = O.make_table(9, d, gen); N = O.make_table(9, d, gen)
    h = torch.randint(0, 40, (n,), generator=gen); t = torch.randint(0, 40, (n,), generator=gen); r = torch.randint(0, 9, (n,), generator=gen)
    for shift in (0, 1):
        def place(x):
            buf = torch.zeros(x.numel() + 4, device=DEV)
            v = buf[shift:shift + x.numel()].view(x.shape)
            v.copy_(x)
            return v
        Ed, Rd, Nd = place(E), place(R), place(N)
        for l1 in (False, True):
            close(ops().score_transe(Ed, Rd, h.to(DEV), t.to(DEV), r.to(DEV), l1), O.score_transe(E, R, h, t, r, l1))
            close(ops().score_transh(Ed, Rd, Nd, h.to(DEV), t.to(DEV), r.to(DEV), l1), O.score_transh(E, R, N, h, t, r, l1))
        close(ops().score_bprmf(Ed, Ed, h.to(DEV), t.to(DEV)), O.score_bprmf(E, E, h, t))


def _l1_knife_edge(W, i2e, u, i, uni, ktup, eps=1e-7):
    """Pairs with a coordinate |z| < eps, z = proj(u) + r - proj(v) from the oracle's helpers in fp64 (transUP.py:69-82)."""
    D = {k: v.double() for k, v in W.items()}
    un = None if uni is None else uni.double()
    u_e = D['U'][u]
    if ktup:
        v_e = D['I'][i] + D['E'][i2e[i]]
        _, r_e, nrm = O.ktup_preferences(u_e, v_e, D['P'], D['Pn'], D['R'], D['Rn'], un)
    else:
        v_e = D['I'][i]
        _, r_e, nrm = O.tup_preferences(u_e, v_e, D['P'], D['Pn'], un)
    z = O.projection_transH(u_e, nrm) + r_e - O.projection_transH(v_e, nrm)
    return z.abs().min(dim=1).values < eps


@pytest.mark.parametrize('d,npref', [(100, 20), (64, 4), (128, 13), (256, 20), (200, 20), (8, 33), (320, 20), (516, 7), (1028, 3), (322, 5),
                                     (100, 40), (64, 100)])     # more preferences than the tile backward holds (32 at d = 100): the row kernels
def test_ml1m_shape_vs_oracle_fwd_bwd(d, npref):
    """ml1m-shape tables (scaled down in rows for d=256), B=512*3 pairs, forward and full backward.  Widths beyond 256 (any multiple
    of 4; others staged with a zero tail) run the one-wave-per-pair kernels of ktup_score_pref_row.hip."""
    nu, ni, ne = (6040, 3240, 14708) if d <= 128 else (600, 300, 1500)
    W, i2e, gen = rand_world(7, nu, ni, ne, npref, d)
    n = 1536
    u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
    gs = torch.randn(n, generator=gen)
    for l1 in (False, True):
        for gum in (False, True):
            uni = torch.rand(n, npref, generator=gen) if gum else None
            if l1:
                # |z| has no derivative at 0: a pair with a coordinate z below its fp32 rounding gets sign(z) = +1 from one summation
                # order and -1 from another, and its three rows then differ by 2 g in that coordinate.  Such pairs (found with the
                # oracle's own formulas in fp64, |z| < 1e-7: a handful of 1536 at most) leave the batch.
                keep = ~(_l1_knife_edge(W, i2e, u, i, uni, True) | _l1_knife_edge(W, i2e, u, i, uni, False))
                assert int((~keep).sum()) <= 8
                u, i, gs, n = u[keep], i[keep], gs[keep], int(keep.sum())
                uni = uni[keep] if gum else None
            Wc = {k: v.clone().requires_grad_(True) for k, v in W.items()}
            Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
            want = O.score_ktup_rec(Wc['U'], Wc['I'], Wc['E'], Wc['P'], Wc['Pn'], Wc['R'], Wc['Rn'], i2e, u, i, l1, uni)
            got = ops().score_ktup(Wd['U'], Wd['I'], Wd['E'], Wd['P'], Wd['Pn'], Wd['R'], Wd['Rn'], i2e.to(DEV, torch.int32),
                                   u.to(DEV), i.to(DEV), l1, ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF,
                                   uni.to(DEV) if gum else None, ent_pad=ne)
            close(got, want)
            want.backward(gs); got.backward(gs.to(DEV))
            for k in W:
                wg = Wc[k].grad.clone()
                if k == 'E':
                    wg[-1].zero_()
                # north_star's 1e-4, with the floor of test_tup_golden: a table gradient is an fp32 sum over the batch, whose rounding
                # scales with the gradient's size (2e-6 of its largest element ~ sqrt(n) ulp of the summands), not with each element
                close(Wd[k].grad, wg, rtol=RT, atol=max(GAT, 2e-6 * float(wg.abs().max())))
            # TUP on the same tables
            Wc = {k: v.clone().requires_grad_(True) for k, v in W.items()}
            Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
            want = O.score_tup(Wc['U'], Wc['I'], Wc['P'], Wc['Pn'], u, i, l1, uni)
            got = ops().score_tup(Wd['U'], Wd['I'], Wd['P'], Wd['Pn'], u.to(DEV), i.to(DEV), l1,
                                  ops().GUMBEL_INPUT if gum else ops().GUMBEL_OFF, uni.to(DEV) if gum else None)
            close(got, want)
            want.backward(gs); got.backward(gs.to(DEV))
            for k in ('U', 'I', 'P', 'Pn'):
                close(Wd[k].grad, Wc[k].grad, rtol=RT, atol=max(GAT, 2e-6 * float(Wc[k].grad.abs().max())))


def test_philox_gate_is_deterministic_and_one_hot():
    """Production ST-Gumbel: same (seed, offset) -> same scores; every score equals the score of SOME one-hot preference."""
    W, i2e, gen = rand_world(3, 100, 80, 90, 6, 100)
    n = 777
    u = torch.randint(0, 100, (n,), generator=gen); i = torch.randint(0, 80, (n,), generator=gen)
    D = {k: v.to(DEV) for k, v in W.items()}
    f = lambda seed, off: ops().score_tup(D['U'], D['I'], D['P'], D['Pn'], u.to(DEV), i.to(DEV), True, ops().GUMBEL_PHILOX, None, seed, off)
    a, b, c = f(11, 0), f(11, 0), f(11, 12345)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    u_e, i_e = W['U'][u], W['I'][i]
    cands = []
    for p in range(6):
        oh = torch.zeros(n, 6); oh[:, p] = 1
        r_e, nrm = oh @ W['P'], oh @ W['Pn']
        cands.append(O._tup_tail(u_e, i_e, r_e, nrm, True))
    cands = torch.stack(cands, 1)
    err = (cands - a.cpu().unsqueeze(1)).abs().min(1)[0]
    assert float(err.max()) < 1e-4
    picked = (cands - a.cpu().unsqueeze(1)).abs().argmin(1)
    assert len(set(picked.tolist())) == 6          # every preference gets sampled over 777 draws


def test_module_surface_matches_reference_names():
    from jTransUP.models import jTransUP as jt, transUP
    m = transUP.TransUPModel(True, 100, 30, 40, 5, True)
    assert sorted(m.state_dict()) == ['item_embeddings.weight', 'pref_embeddings.weight', 'pref_norm_embeddings.weight',
                                      'user_embeddings.weight']
    s = m(torch.tensor([1, 2, 3], device=DEV), torch.tensor([4, 5, 6], device=DEV))
    assert s.shape == (3,) and s.requires_grad
    s.sum().backward()
    assert m.user_embeddings.weight.grad is not None
    im = {i: i for i in range(40)}
    nm = {i: ((i if i % 3 else -1), i) for i in range(40)}
    k = jt.jTransUPModel(False, 100, 30, 40, 50, 7, im, nm, False, False)
    assert k.ent_embeddings.weight.shape == (51, 100)
    with pytest.raises(NotImplementedError):
        k(None, None, is_rec=True)
    s = k((torch.tensor([1, 2], device=DEV), torch.tensor([0, 3], device=DEV)), None, is_rec=True)
    s.sum().backward()
    assert float(k.ent_embeddings.weight.grad[-1].abs().sum()) == 0.0
    k.disable_grad(); assert not k.user_embeddings.weight.requires_grad
    k.enable_grad(); assert k.user_embeddings.weight.requires_grad


@pytest.mark.parametrize('n', [1, 15, 16, 17, 49, 300])
def test_matrix_core_kernels_ragged_fwd_bwd(n):
    """Wave tiles of the matrix-core kernels hold 16 rows: sizes around the tile edge, forward AND backward, soft and hard gate
    (uniforms supplied, so the oracle sees the same Gumbel draw), plus the relation-bucketed TransR forward when most relations
    have no triple at all."""
    W, i2e, gen = rand_world(2, 50, 60, 70, 20, 100)
    u = torch.randint(0, 50, (n,), generator=gen); i = torch.randint(0, 60, (n,), generator=gen)
    uni = torch.rand(n, 20, generator=gen)
    wgt = torch.randn(n, generator=gen)
    names = ('U', 'I', 'E', 'P', 'Pn', 'R', 'Rn')
    for l1 in (False, True):
        for hard in (False, True):
            Wc = {k: W[k].clone().requires_grad_(True) for k in names}
            Wd = {k: W[k].to(DEV).requires_grad_(True) for k in names}
            ref = O.score_ktup_rec(*(Wc[k] for k in names), i2e, u, i, l1, uniform=uni if hard else None)
            got = ops().score_ktup(*(Wd[k] for k in names), i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), l1,
                                   ops().GUMBEL_INPUT if hard else ops().GUMBEL_OFF, uni.to(DEV) if hard else None, ent_pad=70)
            close(got, ref)
            (ref * wgt).sum().backward(); (got * wgt.to(DEV)).sum().backward()
            Wc['E'].grad[70] = 0.0                                        # nn.Embedding(padding_idx=...) of the reference: no gradient
            for k in names:
                close(Wd[k].grad, Wc[k].grad, atol=GAT)
            assert float(Wd['E'].grad[70].abs().sum()) == 0.0            # the pad entity row never receives a gradient
    # TransR: d = 100 takes the bucketed matrix-core forward; only relations 3 and 17 occur
    M = torch.nn.functional.normalize(torch.randn(20, 100 * 100, generator=gen), dim=1)
    h = torch.randint(0, 70, (n,), generator=gen); t = torch.randint(0, 70, (n,), generator=gen)
    r = torch.where(torch.rand(n, generator=gen) < 0.5, torch.tensor(3), torch.tensor(17))
    for l1 in (False, True):
        close(ops().score_transr(W['E'].to(DEV), W['R'].to(DEV), M.to(DEV), h.to(DEV), t.to(DEV), r.to(DEV), l1),
              O.score_transr(W['E'], W['R'], M, h, t, r, l1))


@pytest.mark.parametrize('d,P', [(100, 20), (64, 4), (36, 7)])
def test_philox_stream_position_in_device_memory(d, P):
    """KTUP_GUMBEL_PHILOX_DEV (graph-capturable hard gate): {seed, offset} read from device memory give the same scores and
    gradients, bit for bit, as the same values passed as launch arguments -- for TUP and KTUP, mc and generic kernels."""
    W, i2e, gen = rand_world(d + P, 90, 70, 80, P, d)
    n = 1500
    u = torch.randint(0, 90, (n,), generator=gen).to(DEV); i = torch.randint(0, 70, (n,), generator=gen).to(DEV)
    seed, off = 987654321987, 4242
    state = torch.tensor([seed, off], dtype=torch.int64, device=DEV)
    o = ops()
    for ktup in (False, True):
        outs = []
        for mode, uni, sd, of in ((o.GUMBEL_PHILOX, None, seed, off), (o.GUMBEL_PHILOX_DEV, state, 0, 0)):
            D = {k: v.to(DEV).clone().requires_grad_(True) for k, v in W.items()}
            if ktup:
                s = o.score_ktup(D['U'], D['I'], D['E'], D['P'], D['Pn'], D["R"], D["Rn"], i2e.to(DEV, torch.int32), u, i, False, mode, uni, sd, of,
                                 ent_pad=D['E'].shape[0] - 1)
            else:
                s = o.score_tup(D['U'], D['I'], D['P'], D['Pn'], u, i, False, mode, uni, sd, of)
            (s * torch.linspace(0.5, 1.5, n, device=DEV)).sum().backward()
            outs.append((s.detach(), D))
        assert torch.equal(outs[0][0], outs[1][0])
        for k in ('U', 'I', 'P', 'Pn') + (('E', 'R', 'Rn') if ktup else ()):
            torch.testing.assert_close(outs[0][1][k].grad, outs[1][1][k].grad, rtol=1e-4, atol=1e-5)     # atomics order only
    state[1] += n * P                                                  # the next step's draws
    D = {k: v.to(DEV) for k, v in W.items()}
    a = o.score_tup(D['U'], D['I'], D['P'], D['Pn'], u, i, False, o.GUMBEL_PHILOX_DEV, state)
    b = o.score_tup(D['U'], D['I'], D['P'], D['Pn'], u, i, False, o.GUMBEL_PHILOX, None, seed, off + n * P)
    assert torch.equal(a, b) and not torch.equal(a, outs[0][0])


@pytest.mark.parametrize('npref', [20, 13, 7])
@pytest.mark.parametrize('n', [4096, 16 * 256 * 3 + 5, 20011])
def test_wide_forward_d256_vs_oracle(n, npref):
    """d = 256 with 4096 pairs or more: the coordinate-split forward (pref_fwd_wide_kernel: four waves share a 16-pair tile, the three
    contractions over d summed across them through LDS) against the oracle -- KTUP and TUP, both distances, one / two / three tile
    groups per workgroup, a ragged last tile, P on and off a four-preference group -- and against the one-wave-per-tile kernel
    (option fwd_wide = 0) on the same inputs."""
    from jTransUP.hip import lib as L
    nu, ni, ne, d = 700, 400, 900, 256
    W, i2e, gen = rand_world(11 + npref, nu, ni, ne, npref, d)
    u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
    D = {k: v.to(DEV) for k, v in W.items()}
    for l1 in (False, True):
        want_k = O.score_ktup_rec(W['U'], W['I'], W['E'], W['P'], W['Pn'], W['R'], W['Rn'], i2e, u, i, l1)
        want_t = O.score_tup(W['U'], W['I'], W['P'], W['Pn'], u, i, l1)
        got = {}
        for wide in (1, 0):
            old = L.set_option('fwd_wide', wide)
            try:
                got[wide] = (ops().score_ktup(D['U'], D['I'], D['E'], D['P'], D['Pn'], D['R'], D['Rn'], i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), l1).cpu(),
                             ops().score_tup(D['U'], D['I'], D['P'], D['Pn'], u.to(DEV), i.to(DEV), l1).cpu())
            finally:
                L.set_option('fwd_wide', old)
            close(got[wide][0], want_k); close(got[wide][1], want_t)
        close(got[1][0], got[0][0], rtol=2e-5, atol=2e-6); close(got[1][1], got[0][1], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('d', [100, 50])
def test_gradients_added_straight_into_grad_equal_the_autograd_handover(d):
    """ops._grad_targets: leaves whose `.grad` exists receive the kernels' adds directly (the Functions report None for them).  Two
    steps of the reference's loop shape -- positive and negative call, BPR / margin loss, gradients accumulated over both steps into
    buffers that start NON-zero -- against the same with set_direct_grad(False) (zero-filled temporaries + AccumulateGrad).
    d = 50: the zero-tail staged tables are not leaves and keep the hand-over in both runs."""
    nu, ni, ne, P, n = 300, 200, 400, 7, 257
    W, i2e, gen = rand_world(11, nu, ni, ne, P, d)
    u = torch.randint(0, nu, (2, n), generator=gen); pi = torch.randint(0, ni, (2, n), generator=gen); nj = torch.randint(0, ni, (2, n), generator=gen)
    h = torch.randint(0, ne, (2, n), generator=gen); t = torch.randint(0, ne, (2, n), generator=gen); r = torch.randint(0, P, (2, n), generator=gen)
    start = {k: torch.randn(v.shape, generator=gen) for k, v in W.items()}
    i2e_d = i2e.to(DEV, torch.int32)

    def run(direct):
        was = ops().set_direct_grad(direct)
        try:
            Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
            for k in Wd:
                Wd[k].grad = start[k].to(DEV).clone()
            for s in range(2):
                ud, pd, nd = u[s].to(DEV), pi[s].to(DEV), nj[s].to(DEV)
                kt = lambda ii: ops().score_ktup(Wd['U'], Wd['I'], Wd['E'], Wd['P'], Wd['Pn'], Wd['R'], Wd['Rn'], i2e_d, ud, ii, False, ent_pad=ne)
                tu = lambda ii: ops().score_tup(Wd['U'], Wd['I'], Wd['P'], Wd['Pn'], ud, ii, True)
                th = lambda tt: ops().score_transh(Wd['E'], Wd['R'], Wd['Rn'], h[s].to(DEV), tt, r[s].to(DEV), False)
                te = lambda tt: ops().score_transe(Wd['E'], Wd['R'], h[s].to(DEV), tt, r[s].to(DEV), True)
                loss = (-F.logsigmoid(-(kt(pd) - kt(nd)))).mean() + (-F.logsigmoid(-(tu(pd) - tu(nd)))).mean() \
                    + torch.sum(torch.clamp(th(t[s].to(DEV)) - th(h[s].to(DEV)) + 1.0, min=0.0)) * 1e-2 \
                    + torch.sum(torch.clamp(te(t[s].to(DEV)) - te(h[s].to(DEV)) + 1.0, min=0.0)) * 1e-2 \
                    + ops().score_bprmf(Wd['U'], Wd['I'], ud, pd).mean()
                loss.backward()
            return {k: v.grad.clone() for k, v in Wd.items()}
        finally:
            ops().set_direct_grad(was)

    a, b = run(True), run(False)
    for k in W:
        moved = float((b[k].cpu() - start[k]).abs().max())
        assert moved > 0
        close(a[k], b[k], rtol=RT, atol=max(GAT, 2e-6 * moved))
