"""Bench-size (configs[3], ml1m shape, 716,800 pairs / 307,200 triples) parity through size-independent properties: the
oracle cannot run these sizes in seconds, but independent kernels must agree with each other, and row order must not matter.

  * K6 (pair scoring, matrix-core kernel) == K16 (all-item evaluation, decomposed gate) gathered at the same (u, i);
  * K3 (triple scoring) == K13 (all-entity tail evaluation) gathered at (h, r) -> t;
  * scoring a permutation of the batch permutes the scores, bit for bit;
  * the filtered top-10 of every user is sorted by (score, id), contains no filtered item, and is a prefix of a full sort.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.fixture(scope='module')
def world():
    import bench
    W, i2e, idx = bench.build_world(3, DEV)
    return ({k: v.to(DEV) for k, v in W.items()}, i2e.to(DEV, torch.int32), {k: v.to(DEV) for k, v in idx.items()})


def ops():
    from jTransUP.hip import ops as o
    return o


@pytest.mark.parametrize('l1', [False, True])
def test_pair_scores_agree_with_all_item_evaluation(world, l1):
    D_, i2e, X = world
    with torch.no_grad():
        pair = ops().score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, X['u'], X['i'], l1)
        users = torch.arange(0, D_['U'].shape[0], 7, device=DEV)                 # 863 users x all 3240 items
        full = ops().eval_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, users, l1)
        sel = (X['u'] % 7 == 0)
        got = full[X['u'][sel] // 7, X['i'][sel]]
    assert int(sel.sum()) > 90000
    torch.testing.assert_close(got, pair[sel], rtol=1e-4, atol=1e-5)


def test_triple_scores_agree_with_all_entity_evaluation(world):
    D_, _, X = world
    n = 4096
    h, t, r = X['h'][:n], X['t'][:n], X['r'][:n]
    with torch.no_grad():
        trip = ops().score_transh(D_['E'], D_['R'], D_['Rn'], X['h'], X['t'], X['r'], False)
        full = ops().eval_transh(D_['E'], D_['R'], D_['Rn'], h, r, False, head=False)      # (n x E): distance of every tail
    torch.testing.assert_close(full[torch.arange(n, device=DEV), t], trip[:n], rtol=1e-4, atol=1e-5)


def test_row_order_does_not_matter(world):
    D_, i2e, X = world
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    perm = torch.randperm(X['u'].numel(), generator=gen, device=DEV)
    with torch.no_grad():
        a = ops().score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, X['u'], X['i'], False)
        b = ops().score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, X['u'][perm], X['i'][perm], False)
        perm_k = perm[:X['h'].numel()] % X['h'].numel()
        c = ops().score_transh(D_['E'], D_['R'], D_['Rn'], X['h'], X['t'], X['r'], False)
        d = ops().score_transh(D_['E'], D_['R'], D_['Rn'], X['h'][perm_k], X['t'][perm_k], X['r'][perm_k], False)
    assert torch.equal(a[perm], b) and torch.equal(c[perm_k], d)


def test_filtered_topk_properties_at_full_catalogue(world):
    D_, i2e, _ = world
    nu, ni = 1024, D_['I'].shape[0]
    gen = torch.Generator().manual_seed(2)
    users = torch.arange(nu, device=DEV)
    filt = [torch.randperm(ni, generator=gen)[:165].sort().values for _ in range(nu)]
    off = torch.tensor([0] + [165 * (k + 1) for k in range(nu)], dtype=torch.int64, device=DEV)
    ids = torch.cat(filt).to(DEV, torch.int32)
    with torch.no_grad():
        scores = ops().eval_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, users, False)
        top = ops().topk_filtered(scores, False, 10, off, ids)
    masked = scores.clone()
    masked[torch.arange(nu, device=DEV).repeat_interleave(165), ids.long()] = float('inf')
    # reference order: ascending score, ties to the lower id
    order = torch.argsort(masked, dim=1, stable=True)[:, :10]
    assert torch.equal(top.long(), order)


@pytest.mark.parametrize('d', [64, 100, 128])
def test_large_batch_tile_kernels_vs_oracle(d):
    """Batches >= 65,536 rows take the wave-tile forward kernels of K1-K3; 70,001 rows (ragged last tile) vs the oracle."""
    from oracle import cpu_ref as O
    gen = torch.Generator().manual_seed(d)
    n, NU, NI, NE, NR = 70001, 3000, 2000, 5000, 20
    U, I, E = O.make_table(NU, d, gen), O.make_table(NI, d, gen), O.make_table(NE, d, gen)
    R, Rn = O.make_table(NR, d, gen), O.make_table(NR, d, gen)
    u, i = torch.randint(0, NU, (n,), generator=gen), torch.randint(0, NI, (n,), generator=gen)
    h, t, r = torch.randint(0, NE, (n,), generator=gen), torch.randint(0, NE, (n,), generator=gen), torch.randint(0, NR, (n,), generator=gen)
    dv = lambda x: x.to(DEV)
    with torch.no_grad():
        torch.testing.assert_close(ops().score_bprmf(dv(U), dv(I), dv(u), dv(i)).cpu(), O.score_bprmf(U, I, u, i), rtol=1e-4, atol=1e-5)
        for l1 in (False, True):
            torch.testing.assert_close(ops().score_transe(dv(E), dv(R), dv(h), dv(t), dv(r), l1).cpu(), O.score_transe(E, R, h, t, r, l1),
                                       rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(ops().score_transh(dv(E), dv(R), dv(Rn), dv(h), dv(t), dv(r), l1).cpu(),
                                       O.score_transh(E, R, Rn, h, t, r, l1), rtol=1e-4, atol=1e-5)
