"""GPU parity of the large-batch backward route (per-pair row gradients + reduction by sorted segments, ktup_segreduce.hip),
of the d = 256 coordinate-sliced matrix-core backward (ktup_score_pref_bwd_wide.hip), and of the generic K5-K7 kernels forced
onto a matrix-core shape.  Oracle = oracle/cpu_ref.py (autograd gives the gradient oracle); tolerances as test_hip_score.py."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from jTransUP.hip import lib as L

pytestmark = pytest.mark.gpu
DEV = 'cuda'
RT, AT, GAT = 1e-4, 1e-5, 3e-5
NAMES = ('U', 'I', 'E', 'P', 'Pn', 'R', 'Rn')


def ops():
    from jTransUP.hip import ops as _ops
    return _ops


def close(got, want, rtol=RT, atol=AT):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


def world(seed, nu, ni, ne, P, d):
    gen = torch.Generator().manual_seed(seed)
    W = dict(U=O.make_table(nu, d, gen), I=O.make_table(ni, d, gen), E=torch.cat([O.make_table(ne, d, gen), torch.zeros(1, d)]),
             P=O.make_table(P, d, gen), Pn=O.make_table(P, d, gen), R=O.make_table(P, d, gen), Rn=O.make_table(P, d, gen))
    i2e = torch.randint(0, ne + 1, (ni,), generator=gen)          # some items map to the pad entity (row ne)
    return W, i2e, gen


@pytest.mark.parametrize('d', [64, 100, 128, 256, 36])
def test_segment_reduce_rows_vs_index_add(d):
    """ktup_segment_reduce_rows on hot rows, empty rows, two roles with opposite signs and the mapped second table."""
    gen = torch.Generator().manual_seed(d)
    if d % 4:
        with pytest.raises(L.KtupError):
            ops().segment_reduce_rows(torch.zeros(8, d, device=DEV), torch.zeros(8, dtype=torch.long, device=DEV), torch.zeros(4, d, device=DEV))
        return
    for n, rows in ((1, 5), (37, 3), (5000, 700), (70001, 41)):
        G = torch.randn(n, d, generator=gen)
        ids = torch.randint(0, rows, (n,), generator=gen)
        ids[: n // 3] = rows - 1                                   # a hot row; several rows stay empty when rows > n
        want = torch.zeros(rows, d).index_add_(0, ids, G)
        got = ops().segment_reduce_rows(G.to(DEV), ids.to(DEV), torch.zeros(rows, d, device=DEV))
        close(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
        # accumulates into what is there
        base = torch.randn(rows, d, generator=gen)
        got = ops().segment_reduce_rows(G.to(DEV), ids.to(DEV), base.to(DEV).clone())
        close(got, base + want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
        # two roles (TransE: + for heads, - for tails)
        ids2 = torch.randint(0, rows, (n,), generator=gen)
        want2 = torch.zeros(rows, d).index_add_(0, ids, G).index_add_(0, ids2, -G)
        got2 = ops().segment_reduce_rows(G.to(DEV), torch.cat([ids, ids2]).to(DEV), torch.zeros(rows, d, device=DEV), sign_split=n)
        close(got2, want2, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
        # mapped second table with a pad target that receives nothing
        rows2 = 9
        m2 = torch.randint(0, rows2, (rows,), generator=gen).to(torch.int32)
        want_t2 = torch.zeros(rows2, d).index_add_(0, m2.long(), want)
        want_t2[rows2 - 1] = 0
        t2 = torch.zeros(rows2, d, device=DEV)
        ops().segment_reduce_rows(G.to(DEV), ids.to(DEV), torch.zeros(rows, d, device=DEV), map2=m2.to(DEV), pad2=rows2 - 1, gtable2=t2)
        close(t2, want_t2, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
        assert float(t2[rows2 - 1].abs().sum()) == 0.0


@pytest.mark.parametrize('d,P', [(100, 20), (256, 20), (64, 4)])
def test_backward_by_segments_matches_oracle_and_atomics(d, P):
    """n above the seg_bwd_min threshold: the backward writes per-pair row gradients and reduces them by sorted segments.  Small
    tables on purpose (hundreds of pairs per row: the contention case).  Against the oracle's autograd and against the atomics
    route (option seg_bwd_min = 0) of the same kernels; soft and hard gate; the pad entity row stays zero."""
    nu, ni, ne, n = 90, 70, 110, 9000
    W, i2e, gen = world(11 + d, nu, ni, ne, P, d)
    u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
    u[:2000] = 7                                                   # one very hot user
    wgt = torch.randn(n, generator=gen)
    assert L.load().ktup_score_pref_bwd_workspace_bytes(n, d, nu, ni) > 0
    for l1, hard in ((False, False), (True, True)):
        uni = torch.rand(n, P, generator=gen) if hard else None
        Wc = {k: W[k].clone().requires_grad_(True) for k in NAMES}
        ref = O.score_ktup_rec(*(Wc[k] for k in NAMES), i2e, u, i, l1, uniform=uni)
        (ref * wgt).sum().backward()
        Wc['E'].grad[ne] = 0.0
        grads = {}
        # 'one stream': option side_sort = 0 keeps the two counting sorts on the caller's stream instead of the library's side stream
        for route, thresh, side in (('segments', 8192, 1), ('segments, one stream', 8192, 0), ('atomics', 0, 1)):
            old = L.set_option('seg_bwd_min', thresh)
            old_side = L.set_option('side_sort', side)
            try:
                Wd = {k: W[k].to(DEV).requires_grad_(True) for k in NAMES}
                got = ops().score_ktup(*(Wd[k] for k in NAMES), i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), l1,
                                       ops().GUMBEL_INPUT if hard else ops().GUMBEL_OFF, uni.to(DEV) if hard else None, ent_pad=ne)
                (got * wgt.to(DEV)).sum().backward()
            finally:
                L.set_option('seg_bwd_min', old)
                L.set_option('side_sort', old_side)
            close(got, ref)
            grads[route] = {k: Wd[k].grad.cpu() for k in NAMES}
            for k in NAMES:
                scale = float(Wc[k].grad.abs().max())
                close(grads[route][k], Wc[k].grad, rtol=2e-4, atol=2e-5 * max(scale, 1.0))
            assert float(grads[route]['E'][ne].abs().sum()) == 0.0
        # TUP through the same route
        Wc = {k: W[k].clone().requires_grad_(True) for k in ('U', 'I', 'P', 'Pn')}
        ref = O.score_tup(Wc['U'], Wc['I'], Wc['P'], Wc['Pn'], u, i, l1, uni)
        (ref * wgt).sum().backward()
        Wd = {k: W[k].to(DEV).requires_grad_(True) for k in ('U', 'I', 'P', 'Pn')}
        got = ops().score_tup(Wd['U'], Wd['I'], Wd['P'], Wd['Pn'], u.to(DEV), i.to(DEV), l1,
                              ops().GUMBEL_INPUT if hard else ops().GUMBEL_OFF, uni.to(DEV) if hard else None)
        (got * wgt.to(DEV)).sum().backward()
        for k in ('U', 'I', 'P', 'Pn'):
            scale = float(Wc[k].grad.abs().max())
            close(Wd[k].grad, Wc[k].grad, rtol=2e-4, atol=2e-5 * max(scale, 1.0))


@pytest.mark.parametrize('n', [1, 15, 16, 17, 49, 300, 4100])
def test_wide_backward_d256_ragged_vs_oracle(n):
    """d = 256 takes the coordinate-sliced matrix-core backward (four waves per 16-pair tile): tile edges, more tiles than
    workgroups, soft and hard gate, L1 and squared L2, KTUP and TUP."""
    P, d = 20, 256
    W, i2e, gen = world(5, 50, 60, 70, P, d)
    u = torch.randint(0, 50, (n,), generator=gen); i = torch.randint(0, 60, (n,), generator=gen)
    uni = torch.rand(n, P, generator=gen)
    wgt = torch.randn(n, generator=gen)
    for l1 in (False, True):
        for hard in (False, True):
            Wc = {k: W[k].clone().requires_grad_(True) for k in NAMES}
            Wd = {k: W[k].to(DEV).requires_grad_(True) for k in NAMES}
            ref = O.score_ktup_rec(*(Wc[k] for k in NAMES), i2e, u, i, l1, uniform=uni if hard else None)
            got = ops().score_ktup(*(Wd[k] for k in NAMES), i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), l1,
                                   ops().GUMBEL_INPUT if hard else ops().GUMBEL_OFF, uni.to(DEV) if hard else None, ent_pad=70)
            close(got, ref)
            (ref * wgt).sum().backward(); (got * wgt.to(DEV)).sum().backward()
            Wc['E'].grad[70] = 0.0
            for k in NAMES:
                scale = float(Wc[k].grad.abs().max())
                close(Wd[k].grad, Wc[k].grad, rtol=2e-4, atol=GAT * max(scale, 1.0))
            assert float(Wd['E'].grad[70].abs().sum()) == 0.0
    Wc = {k: W[k].clone().requires_grad_(True) for k in ('U', 'I', 'P', 'Pn')}
    Wd = {k: W[k].to(DEV).requires_grad_(True) for k in ('U', 'I', 'P', 'Pn')}
    ref = O.score_tup(Wc['U'], Wc['I'], Wc['P'], Wc['Pn'], u, i, False)
    got = ops().score_tup(Wd['U'], Wd['I'], Wd['P'], Wd['Pn'], u.to(DEV), i.to(DEV), False)
    (ref * wgt).sum().backward(); (got * wgt.to(DEV)).sum().backward()
    for k in ('U', 'I', 'P', 'Pn'):
        scale = float(Wc[k].grad.abs().max())
        close(Wd[k].grad, Wc[k].grad, rtol=2e-4, atol=GAT * max(scale, 1.0))


def test_wide_backward_preference_counts():
    """P = 13 (NP = 4 instantiation) and P = 20 share the kernel family; P = 24 exceeds its LDS budget and must fall back to the
    generic kernel with the same results."""
    d, n = 256, 130
    for P in (13, 16, 20, 24):
        W, i2e, gen = world(P, 40, 30, 50, P, d)
        u = torch.randint(0, 40, (n,), generator=gen); i = torch.randint(0, 30, (n,), generator=gen)
        Wc = {k: W[k].clone().requires_grad_(True) for k in NAMES}
        Wd = {k: W[k].to(DEV).requires_grad_(True) for k in NAMES}
        ref = O.score_ktup_rec(*(Wc[k] for k in NAMES), i2e, u, i, False)
        got = ops().score_ktup(*(Wd[k] for k in NAMES), i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), False, ent_pad=50)
        close(got, ref)
        ref.sum().backward(); got.sum().backward()
        Wc['E'].grad[50] = 0.0
        for k in NAMES:
            scale = float(Wc[k].grad.abs().max())
            close(Wd[k].grad, Wc[k].grad, rtol=2e-4, atol=GAT * max(scale, 1.0))


def test_generic_kernels_on_a_matrix_core_shape():
    """Option pref_mc = 0 routes d = 100 through the generic K5-K7 kernels (the fallback every other d takes): same scores and
    gradients as the matrix-core kernels."""
    W, i2e, gen = world(9, 80, 60, 90, 20, 100)
    n = 333
    u = torch.randint(0, 80, (n,), generator=gen); i = torch.randint(0, 60, (n,), generator=gen)
    out = {}
    for mc in (1, 0):
        old = L.set_option('pref_mc', mc)
        try:
            Wd = {k: W[k].to(DEV).requires_grad_(True) for k in NAMES}
            s = ops().score_ktup(*(Wd[k] for k in NAMES), i2e.to(DEV, torch.int32), u.to(DEV), i.to(DEV), False, ent_pad=90)
            s.sum().backward()
        finally:
            L.set_option('pref_mc', old)
        out[mc] = (s.detach().cpu(), {k: Wd[k].grad.cpu() for k in NAMES})
    close(out[0][0], out[1][0])
    for k in NAMES:
        close(out[0][1][k], out[1][1][k], rtol=2e-4, atol=GAT * max(1.0, float(out[1][1][k].abs().max())))
    with pytest.raises(L.KtupError):
        L.set_option('no_such_option', 1)


@pytest.mark.parametrize('d', [100, 64, 256])
def test_kg_and_bprmf_backward_by_segments(d):
    """K1-K3 at n above the seg_bwd_min threshold: per-row gradient vectors + reduction by sorted segments (heads and tails with
    opposite signs), relation-side tables through LDS accumulators -- against the oracle's autograd and against the atomics route
    (option seg_bwd_min = 0).  Small tables: hundreds of rows of the batch per table row."""
    ne, nr, nu, ni, n = 130, 7, 90, 70, 9000
    gen = torch.Generator().manual_seed(d)
    E, R, N = O.make_table(ne, d, gen), O.make_table(nr, d, gen), O.make_table(nr, d, gen)
    U, I = O.make_table(nu, d, gen), O.make_table(ni, d, gen)
    h = torch.randint(0, ne, (n,), generator=gen); t = torch.randint(0, ne, (n,), generator=gen); r = torch.randint(0, nr, (n,), generator=gen)
    h[:1500] = 3
    u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
    wgt = torch.randn(n, generator=gen)
    assert L.load().ktup_score_kg_bwd_workspace_bytes(n, d, ne) > 0 and L.load().ktup_score_bprmf_bwd_workspace_bytes(n, d, nu, ni) > 0
    cases = [('transe', [E, R], lambda W, l1: O.score_transe(W[0], W[1], h, t, r, l1),
              lambda W, l1: ops().score_transe(W[0], W[1], h.to(DEV), t.to(DEV), r.to(DEV), l1)),
             ('transh', [E, R, N], lambda W, l1: O.score_transh(W[0], W[1], W[2], h, t, r, l1),
              lambda W, l1: ops().score_transh(W[0], W[1], W[2], h.to(DEV), t.to(DEV), r.to(DEV), l1)),
             ('bprmf', [U, I], lambda W, l1: O.score_bprmf(W[0], W[1], u, i),
              lambda W, l1: ops().score_bprmf(W[0], W[1], u.to(DEV), i.to(DEV)))]
    for name, tabs, ref_fn, hip_fn in cases:
        for l1 in ((False, True) if name != 'bprmf' else (False,)):
            Wc = [x.clone().requires_grad_(True) for x in tabs]
            (ref_fn(Wc, l1) * wgt).sum().backward()
            for thresh, side in ((8192, 1), (8192, 0), (0, 1)):     # side_sort = 0: the id sort stays on the caller's stream
                old = L.set_option('seg_bwd_min', thresh)
                old_side = L.set_option('side_sort', side)
                try:
                    Wd = [x.to(DEV).requires_grad_(True) for x in tabs]
                    got = hip_fn(Wd, l1)
                    (got * wgt.to(DEV)).sum().backward()
                finally:
                    L.set_option('seg_bwd_min', old)
                    L.set_option('side_sort', old_side)
                close(got, ref_fn([x.detach() for x in Wc], l1))
                for a, b in zip(Wd, Wc):
                    scale = float(b.grad.abs().max())
                    close(a.grad, b.grad, rtol=2e-4, atol=2e-5 * max(scale, 1.0))


@pytest.mark.parametrize('d', [100, 64, 128])
@pytest.mark.parametrize('n', [700, 9000])
def test_transr_backward_on_the_matrix_cores(d, n):
    """K4 backward (ktup_score_transr_bwd_ws): relation buckets, gq = M^T gy and gM += gy (x) (h - t) on the matrix cores, against
    the oracle's autograd and against the generic per-triple kernel (option pref_mc = 0).  n = 700: atomics for the entity rows,
    partial 16-triple tiles and rounds with idle waves; n = 9000: entity rows by sorted segments; one relation takes most of the
    batch (several 1024-triple passes), two relations never occur."""
    ne, nr = 150, 9
    gen = torch.Generator().manual_seed(1000 * d + n)
    E, R = O.make_table(ne, d, gen), O.make_table(nr, d, gen)
    M = torch.nn.functional.normalize(torch.randn(nr, d * d, generator=gen), dim=1)
    h = torch.randint(0, ne, (n,), generator=gen); t = torch.randint(0, ne, (n,), generator=gen)
    r = torch.randint(0, nr - 2, (n,), generator=gen)
    r[torch.rand(n, generator=gen) < 0.6] = 4
    h[: n // 6] = 3
    wgt = torch.randn(n, generator=gen)
    assert L.load().ktup_score_transr_bwd_workspace_bytes(n, d, ne, nr) > 0
    for l1 in (False, True):
        Wc = [x.clone().requires_grad_(True) for x in (E, R, M)]
        ref = O.score_transr(Wc[0], Wc[1], Wc[2], h, t, r, l1)
        (ref * wgt).sum().backward()
        for mc in (1, 0):
            old = L.set_option('pref_mc', mc)
            try:
                Wd = [x.to(DEV).requires_grad_(True) for x in (E, R, M)]
                got = ops().score_transr(Wd[0], Wd[1], Wd[2], h.to(DEV), t.to(DEV), r.to(DEV), l1)
                (got * wgt.to(DEV)).sum().backward()
            finally:
                L.set_option('pref_mc', old)
            close(got, ref)
            for a, b in zip(Wd, Wc):
                scale = float(b.grad.abs().max())
                close(a.grad, b.grad, rtol=2e-4, atol=2e-5 * max(scale, 1.0))
            assert float(Wd[2].grad[7:].abs().sum()) == 0.0                   # relations that never occur receive nothing


def test_side_stream_sorts_stay_ordered_across_back_to_back_calls():
    """The id sorts of the segment reductions run on the library's side stream (fork at entry, join before the reduction).  Calls
    back to back reuse the same scratch memory (torch's caching allocator hands the freed block out again) with DIFFERENT ids:
    every call must see its own sort.  Also on a non-default caller stream."""
    d, P, nu, ni, ne, n = 100, 20, 300, 200, 260, 20000
    W, i2e, gen = world(5, nu, ni, ne, P, d)
    Wd = {k: W[k].to(DEV) for k in NAMES}
    cases = []
    for c in range(4):
        u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
        cases.append((u.to(DEV), i.to(DEV), torch.randn(n, generator=gen).to(DEV)))

    def run(side):
        old = L.set_option('side_sort', side)
        out = []
        try:
            for u, i, wgt in cases:
                T = {k: Wd[k].detach().clone().requires_grad_(True) for k in NAMES}
                (ops().score_ktup(*(T[k] for k in NAMES), i2e.to(DEV, torch.int32), u, i, False, ent_pad=ne) * wgt).sum().backward()
                out.append({k: T[k].grad for k in NAMES})
            torch.cuda.synchronize()
        finally:
            L.set_option('side_sort', old)
        return out

    want = run(0)
    got = run(1)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        got2 = run(1)
    torch.cuda.current_stream().wait_stream(st)
    for a, b, c in zip(got, want, got2):
        for k in NAMES:
            scale = max(float(b[k].abs().max()), 1.0)
            close(a[k].cpu(), b[k].cpu(), rtol=2e-4, atol=2e-5 * scale)
            close(c[k].cpu(), b[k].cpu(), rtol=2e-4, atol=2e-5 * scale)
