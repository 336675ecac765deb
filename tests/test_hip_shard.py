"""GPU checks of the sharded-table device halves (pack / unpack-add) and of ShardedTable on one GPU feeding the KTUP scorer."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('d', [100, 256, 50])
def test_pack_and_unpack_rows(d):
    from jTransUP.hip import ops
    gen = torch.Generator().manual_seed(d)
    T = torch.randn(97, d, generator=gen)
    ids = torch.randint(0, 97, (1000,), generator=gen)
    got = ops.pack_rows(T.to(DEV), ids.to(DEV))
    assert torch.equal(got.cpu(), T[ids])
    rows = torch.randn(1000, d, generator=gen)
    g = ops.unpack_rows_add(rows.to(DEV), ids.to(DEV), torch.zeros(97, d, device=DEV))
    want = torch.zeros(97, d).index_add_(0, ids, rows)
    assert torch.allclose(g.cpu(), want, rtol=1e-5, atol=1e-5)


def test_sharded_table_feeds_the_scorer_on_one_gpu():
    """The compact (unique-row) table + inverse ids returned by ShardedTable.lookup drive the unchanged TUP kernel and its
    backward; result equals scoring against the full table."""
    from jTransUP.hip import ops
    from jTransUP.parallel import ShardedTable
    gen = torch.Generator().manual_seed(1)
    nu, ni, d, P, n = 500, 300, 100, 20, 777
    U, I, Pt, Pn = (O.make_table(r, d, gen) for r in (nu, ni, P, P))
    u = torch.randint(0, nu, (n,), generator=gen); i = torch.randint(0, ni, (n,), generator=gen)
    st_u = ShardedTable(nu, d, rank=0, world=1, init=lambda g: U[g], device=DEV)
    st_i = ShardedTable(ni, d, rank=0, world=1, init=lambda g: I[g], device=DEV)
    Pd, Pnd = Pt.to(DEV).requires_grad_(True), Pn.to(DEV).requires_grad_(True)
    cu, uid = st_u.lookup(u.to(DEV)); ci, iid = st_i.lookup(i.to(DEV))
    got = ops.score_tup(cu, ci, Pd, Pnd, uid, iid, True)
    Uc, Ic, Pc, Pnc = (x.clone().requires_grad_(True) for x in (U, I, Pt, Pn))
    want = O.score_tup(Uc, Ic, Pc, Pnc, u, i, True)
    assert torch.allclose(got.cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    gs = torch.randn(n, generator=gen)
    got.backward(gs.to(DEV)); want.backward(gs)
    assert torch.allclose(st_u.weight.grad.cpu(), Uc.grad, rtol=2e-4, atol=1e-4)
    assert torch.allclose(st_i.weight.grad.cpu(), Ic.grad, rtol=2e-4, atol=1e-4)
    assert torch.allclose(Pd.grad.cpu(), Pc.grad, rtol=2e-4, atol=1e-4)
