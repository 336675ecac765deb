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


@pytest.mark.parametrize('d', [100, 256, 50])
@pytest.mark.parametrize('kind', ['adagrad', 'sgd'])
def test_sparse_row_step_kernel(d, kind):
    """ktup_shard_sparse_step on unique rows == the dense rule on those rows (clip coefficient from the device double);
    every other row of the table and of the Adagrad state is untouched, bit for bit."""
    from jTransUP.hip import ops
    gen = torch.Generator().manual_seed(d)
    T = torch.randn(211, d, generator=gen); S = torch.rand(211, d, generator=gen)
    ids = torch.randperm(211, generator=gen)[:97]
    g = torch.randn(97, d, generator=gen)
    lr, eps, max_norm = 0.05, 1e-10, 3.0
    sumsq = (g.double() ** 2).sum().reshape(1) * 4            # "job-wide" norm: other ranks contributed too
    coef = min(1.0, max_norm / (float(sumsq.sqrt()) + 1e-6))
    Td, Sd = T.to(DEV), S.to(DEV)
    ops.sparse_step(kind, Td, Sd, ids.to(DEV), g.to(DEV), lr, eps, sumsq.to(DEV), max_norm)
    Tw, Sw = T.clone(), S.clone()
    gc = g * coef
    if kind == 'adagrad':
        Sw[ids] += gc * gc
        Tw[ids] -= lr * gc / (Sw[ids].sqrt() + eps)
    else:
        Tw[ids] -= lr * gc
    torch.testing.assert_close(Td.cpu(), Tw, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(Sd.cpu(), Sw, rtol=1e-5, atol=1e-6)
    rest = torch.ones(211, dtype=torch.bool); rest[ids] = False
    assert torch.equal(Td.cpu()[rest], T[rest]) and torch.equal(Sd.cpu()[rest], S[rest])
    assert abs(float(ops.grad_sumsq([g.to(DEV)])) - float((g.double() ** 2).sum())) < 1e-6 * float((g.double() ** 2).sum())


def test_sharded_step_on_one_gpu_equals_dense():
    """Config 5's step with the HIP row kernels (world 1: no exchange): same tables as autograd + clip + torch.optim."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _sharded_case import check_against_dense
    from jTransUP.parallel import RowOps
    for kind, lr, max_norm in (('adagrad', 0.1, 0.05), ('sgd', 0.05, 0.02)):
        check_against_dense(kind, lr, max_norm, 3, torch.device(DEV), RowOps, 0, 1)
        check_against_dense(kind, lr, max_norm, 3, torch.device(DEV), RowOps, 0, 1, many=True)


def _two_rank_worker(rank, world, port):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share this box's GPU (RCCL refuses that)
    try:
        from _sharded_case import check_against_dense
        from jTransUP.parallel import RowOps
        check_against_dense('adagrad', 0.1, 0.05, 3, torch.device(DEV), RowOps, rank, world)
        check_against_dense('adagrad', 0.1, 0.05, 3, torch.device(DEV), RowOps, rank, world, many=True)
    finally:
        dist.destroy_process_group()


def test_sharded_step_two_ranks_share_the_gpu():
    """Two ranks, tables sharded by row % 2, HIP pack / unpack / sparse-step kernels, exchanges staged through gloo."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port), nprocs=2, join=True)


def _shard_eval_worker(rank, world, port):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import _shard_eval_case as C
        C.run(rank, world, DEV)                       # the HIP local kernels: ktup_eval_topk_filtered, ktup_eval_gold_rank_counts
        C.run(rank, world, DEV, layout='lattice')     # candidates rank + world * j: ktup_eval_gold_rank_counts_strided (-shard_tables)
    finally:
        dist.destroy_process_group()


def test_sharded_candidate_evaluation_two_ranks_share_the_gpu():
    """Catalogue split over two ranks: per-shard filtered top-n (K17) + merge and per-shard rank counts (ktup_eval_gold_rank_counts)
    + all-reduce reproduce the reference's golden ranked lists / ranks and a seeded tie-heavy case; also as one shard."""
    import os
    import socket
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _shard_eval_case as C
    C.run(0, 1, DEV)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_shard_eval_worker, args=(2, port), nprocs=2, join=True)
