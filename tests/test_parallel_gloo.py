"""World-size-2 tests of the multi-GPU path on CPU (gloo): gradient sync of replicas, the row-sharded table exchange
(forward and backward through two all-to-alls), and the sharded-candidate top-n merge.  The HIP scorers are GPU-only,
so plain torch embedding arithmetic stands in for them here: what is under test is the distributed logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))     # _sharded_case, also in the spawned ranks


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn), nprocs=world, join=True)


def _entry(rank, world, port, fn):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fn(rank, world)
    finally:
        dist.destroy_process_group()


def _toy_loss(U, I, P, u, i, sync=None):
    """mean term (BPR-like) + sum term (margin-like, on gathered rows) + replicated whole-table term."""
    s = (U[u] * I[i]).sum(1)
    mean_term = torch.nn.functional.softplus(s).mean()
    sum_term = torch.clamp((U[u] ** 2).sum(1) - 0.5, min=0).sum()
    rep_term = (P ** 2).sum()
    if sync is None:
        return mean_term + sum_term + rep_term
    return sync.scale(mean_term, 'mean') + sync.scale(sum_term, 'sum') + sync.scale(rep_term, 'replicated')


def _replica_worker(rank, world):
    from jTransUP.parallel import ReplicaGradSync
    gen = torch.Generator().manual_seed(0)
    U0, I0, P0 = torch.randn(13, 6, generator=gen), torch.randn(17, 6, generator=gen), torch.randn(4, 6, generator=gen)
    u = torch.randint(0, 13, (32,), generator=gen); i = torch.randint(0, 17, (32,), generator=gen)
    # single-process truth on the concatenated batch
    Ut, It, Pt = (x.clone().requires_grad_(True) for x in (U0, I0, P0))
    _toy_loss(Ut, It, Pt, u, i).backward()
    torch.nn.utils.clip_grad_norm_([Ut, It, Pt], 0.5)
    # replica: my half of the batch
    U, I, P = (torch.nn.Parameter((x + (rank * 0 if True else 0)).clone()) for x in (U0, I0, P0))
    sync = ReplicaGradSync([U, I, P])
    sync.broadcast_params()
    sl = slice(rank * 16, (rank + 1) * 16)
    _toy_loss(U, I, P, u[sl], i[sl], sync).backward()
    sync.all_reduce_grads()
    torch.nn.utils.clip_grad_norm_([U, I, P], 0.5)          # global norm of the REDUCED gradient, same on every rank
    for got, want in ((U, Ut), (I, It), (P, Pt)):
        assert torch.allclose(got.grad, want.grad, rtol=1e-5, atol=1e-6)
    opt = torch.optim.Adagrad([U, I, P], lr=0.1)
    opt.step()
    flat = torch.cat([p.data.reshape(-1) for p in (U, I, P)])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])                    # replicas stay bit-identical after the step


def _init_rows(g):
    return g[:, None].float() * 0.01 + torch.arange(8).float()[None, :] * (1 + (g[:, None] % 3).float())


def _sharded_worker(rank, world):
    from jTransUP.parallel import ShardedTable
    from _sharded_case import TorchRowOps            # no GPU here: torch stand-ins for the HIP row kernels
    table = ShardedTable(37, 8, init=_init_rows, pack=TorchRowOps.pack, unpack_add=TorchRowOps.unpack_add)
    assert table.weight.shape[0] == (19 if rank == 0 else 18)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    all_ids = [torch.randint(0, 37, (50,), generator=g) for g in gens]      # duplicates on purpose
    all_w = [torch.randn(50, 8, generator=g) for g in gens]
    ids, w = all_ids[rank], all_w[rank]
    compact, cids = table.lookup(ids)
    rows = compact[cids]
    assert torch.equal(rows, _init_rows(ids))                               # forward: the right rows, in my order
    (rows * w).sum().backward()
    # dense truth: d loss / d table[g] = sum over every rank's occurrences of g
    dense = torch.zeros(37, 8)
    for r in range(world):
        dense.index_add_(0, all_ids[r], all_w[r])
    mine = dense[torch.arange(rank, 37, world)]
    assert torch.allclose(table.weight.grad, mine, rtol=1e-6, atol=1e-6)
    # an empty request from one rank must not deadlock or corrupt the other
    compact, cids = table.lookup(ids[:0] if rank == 0 else ids[:5])
    assert compact.shape[0] == (0 if rank == 0 else len(torch.unique(ids[:5])))


def _sharded_step_worker(rank, world):
    """Config 5's whole step (lookups, row-gradient return, duplicate combine, global clip, row-sparse update) == dense."""
    from _sharded_case import TorchRowOps, check_against_dense
    for kind, lr, max_norm in (('adagrad', 0.1, 0.05), ('sgd', 0.05, 0.0), ('sgd', 0.05, 0.02)):
        check_against_dense(kind, lr, max_norm, 3, torch.device('cpu'), TorchRowOps, rank, world)
        check_against_dense(kind, lr, max_norm, 3, torch.device('cpu'), TorchRowOps, rank, world, many=True)   # combined route


def _merge_worker(rank, world):
    from jTransUP.parallel import merge_topk, shard_bounds
    rng = np.random.RandomState(3)
    nq, nc, topn = 7, 101, 10
    scores = (rng.randint(0, 12, size=(nq, nc)) / 4.0).astype(np.float32)    # many exact ties across shards
    lo, hi = shard_bounds(nc, rank, world)
    assert (lo, hi) == ((0, 51) if rank == 0 else (51, 101))
    local = scores[:, lo:hi]
    order = np.argsort(local, axis=1, kind='stable')[:, :topn]
    ids = torch.from_numpy(order + lo).to(torch.int32)
    sc = torch.from_numpy(np.take_along_axis(local, order, 1))
    if rank == 1:                                                              # a shard with fewer than topn survivors
        ids[0, 4:] = -1; sc[0, 4:] = 0.0
    got_ids, got_sc = merge_topk(ids, sc, topn)
    for b in range(nq):
        cand = list(range(nc))
        if b == 0:
            keep1 = set((order[0, :4] + lo).tolist()) if rank == 1 else None
            # rank 1's truncated list: only its first 4 survive on BOTH ranks' view (all_gather shares it)
        full = np.argsort(scores[b], kind='stable')
        if b != 0:
            assert got_ids[b].tolist() == full[:topn].tolist()
            np.testing.assert_array_equal(got_sc[b].numpy(), scores[b][full[:topn]])
    # row 0: merge of shard 0's top-10 with only 4 entries of shard 1
    s1 = np.argsort(scores[0, 51:], kind='stable')[:4] + 51
    s0 = np.argsort(scores[0, :51], kind='stable')[:topn]
    pool = sorted(list(s0) + list(s1), key=lambda j: (scores[0, j], j))[:topn]
    assert got_ids[0].tolist() == pool


def _shard_eval_worker(rank, world):
    """Sharded-candidate evaluation: per-shard filtered top-n + merge, additive KG rank counts + all-reduce, against
    the reference's golden ranked lists / ranks and a seeded case with ties (numpy stand-ins for the two local HIP kernels)."""
    import _shard_eval_case as C
    C.run(rank, world, 'cpu', local_topk=C.np_local_topk, local_counts=C.np_local_counts)


def _shard_eval_lattice_worker(rank, world):
    """The same with the candidates of a rank = the lattice rank + world * j (-shard_tables: evaluation straight from the row shards):
    strided local ids for the filter lists, global ids back for the merge, strided rank counts -- the golden lists / ranks again."""
    import _shard_eval_case as C
    C.run(rank, world, 'cpu', local_topk=C.np_local_topk, local_counts=C.np_local_counts, layout='lattice')


def _gather_table_worker(rank, world):
    """-shard_tables: before an evaluation or a checkpoint every rank rebuilds the whole tables from the shards
    (utils/sharded_train.gather_table: rows g % world == r live on rank r) -- row counts that do and do not divide by the ranks,
    a table with fewer rows than ranks, and the round trip back to the shards."""
    from jTransUP.utils.sharded_train import gather_table
    for total in (37, 40, 1, world, world + 1):
        want = _init_rows(torch.arange(total))
        mine = want[rank::world].clone()
        full = torch.full((total, 8), float('nan'))
        gather_table(full, mine, total, world)
        assert torch.equal(full, want), (total, rank)
        assert torch.equal(full[rank::world], mine)                         # load_from_model's slice is the shard again


@pytest.mark.parametrize('worker', [_replica_worker, _sharded_worker, _sharded_step_worker, _merge_worker, _shard_eval_worker,
                                    _shard_eval_lattice_worker, _gather_table_worker])
def test_world_size_2_gloo(worker):
    _spawn(worker)


def test_shard_gather_on_three_ranks():
    _spawn(_gather_table_worker, world=3)


def test_lattice_shard_evaluation_on_three_ranks():
    _spawn(_shard_eval_lattice_worker, world=3)


def test_single_process_paths():
    """world == 1 without a process group: ShardedTable degenerates to a local gather, merge_topk to a sort."""
    from jTransUP.parallel import ReplicaGradSync, ShardedTable, merge_topk
    from _sharded_case import TorchRowOps, check_against_dense
    t = ShardedTable(10, 8, rank=0, world=1, init=_init_rows, pack=TorchRowOps.pack, unpack_add=TorchRowOps.unpack_add)
    ids = torch.tensor([3, 3, 9, 0])
    compact, cids = t.lookup(ids)
    assert torch.equal(compact[cids], _init_rows(ids))
    compact[cids].sum().backward()
    assert t.weight.grad[3].tolist() == [2.0] * 8 and t.weight.grad[1].abs().sum() == 0
    ids_, sc_ = merge_topk(torch.tensor([[5, 2, -1]], dtype=torch.int32), torch.tensor([[1.0, 1.0, 0.0]]), 2)
    assert ids_.tolist() == [[2, 5]]
    s = ReplicaGradSync([torch.nn.Parameter(torch.ones(2))])
    assert float(s.scale(torch.tensor(4.0), 'mean')) == 4.0
    check_against_dense('adagrad', 0.1, 0.05, 3, torch.device('cpu'), TorchRowOps, 0, 1)
    import _shard_eval_case as C                     # one shard = the whole catalogue: same results without a process group
    C.run(0, 1, 'cpu', local_topk=C.np_local_topk, local_counts=C.np_local_counts)
    with pytest.raises(Exception):                   # the product's default row ops are the HIP kernels: CPU tensors are refused
        ShardedTable(10, 8, rank=0, world=1, init=_init_rows).lookup(ids)
