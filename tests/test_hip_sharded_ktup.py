"""The fixed-shape config-5 step (jTransUP/sharded_ktup.py: route -> pack -> fused KTUP step with row-gradient output -> segment
reduction -> clip + row-sparse optimizer, replayed as HIP graphs) against a single-process DENSE run of the reference's step on
CPU: the oracle's KTUP scorer (jTransUP.py:122-143 restated in oracle/cpu_ref.py), bprLoss with target -1 (utils/loss.py:29-31 =
softplus(pos - neg).mean()), clip_grad_norm_ and torch.optim (utils/trainer.py:63-77 with l2_lambda = 0), several consecutive
steps so that graph replay, the zero-filled gradient buffers and the Adagrad state are all exercised.  One rank, one rank in
exchange form (the several-ranks route talking to itself, with RCCL at world 1 when a process group exists), two ranks sharing
the GPU (gloo stages the collectives), at toy sizes and at one rank's share of config 5."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _world_tables(nu, ni, ne, P, d, seed, pad_every=0):
    gen = torch.Generator().manual_seed(seed)
    full = {k: torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1) for k, n in (('U', nu), ('I', ni), ('E', ne))}
    small = [torch.nn.functional.normalize(torch.randn(P, d, generator=gen), dim=1) for _ in range(4)]
    i2e = torch.randint(0, ne, (ni,), generator=gen)
    if pad_every:
        i2e[::pad_every] = -1                                   # items without an aligned entity (jTransUP.py:114-120 -> pad row)
    return full, small, i2e, gen


def _dense_reference(full, small0, i2e, batches, kind, lr, eps, max_norm, l1=False, orth=False, weight_decay=0.0, uniforms=None, adam_init=None):
    """One process, whole tables, the global batch: returns the tables after the steps and the per-step losses."""
    ne = full['E'].shape[0]
    E_pad = torch.cat([full['E'], torch.zeros(1, full['E'].shape[1])])           # pad row = ent_total - 1 (jTransUP.py:46,96)
    W = [torch.nn.Parameter(full['U'].clone()), torch.nn.Parameter(full['I'].clone()), torch.nn.Parameter(E_pad)] + \
        [torch.nn.Parameter(t.clone()) for t in small0]
    i2e_pad = torch.where(i2e < 0, torch.full_like(i2e, ne), i2e)
    wd = weight_decay
    opt = torch.optim.Adagrad(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adagrad' else \
        torch.optim.Adam(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adam' else torch.optim.SGD(W, lr=lr, weight_decay=wd)
    if adam_init is not None:                 # an optimizer that has already run t0 steps: (t0, first moments, second moments) per table
        t0, M, V = adam_init
        for w, m, v in zip(W, M, V):
            pad = w.shape[0] - m.shape[0]     # (the entity table's pad row)
            opt.state[w] = {'step': torch.tensor(float(t0)), 'exp_avg': torch.cat([m, torch.zeros(pad, m.shape[1])]).clone(),
                            'exp_avg_sq': torch.cat([v, torch.zeros(pad, v.shape[1])]).clone()}
    losses = []
    for k_step, step in enumerate(batches):
        opt.zero_grad(set_to_none=False)      # zero-FILL, like the torch 0.3 of the reference: Adam keeps moving every table it has ever stepped
        u = torch.cat([x[0] for x in step]); pi = torch.cat([x[1] for x in step]); ni_ = torch.cat([x[2] for x in step])
        un = (None, None) if uniforms is None else uniforms[k_step]        # (pos, neg): one row of n_pref uniforms per pair, as transUP.py:159-162 draws them
        pos = O.score_ktup_rec(*W, i2e_pad, u, pi, l1, un[0]); neg = O.score_ktup_rec(*W, i2e_pad, u, ni_, l1, un[1])
        loss = torch.nn.functional.softplus(pos - neg).mean()
        if orth:
            loss = loss + O.orthogonal_loss(W[3], W[4])
        loss.backward()
        W[2].grad[-1].zero_()                                                    # padding_idx: the pad row takes no gradient
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(W, max_norm)
        opt.step()
        losses.append(float(loss.detach()))
    return [w.data for w in W], losses


def _run_stepper(full, small0, i2e, batches, kind, lr, eps, max_norm, rank, world, dev, l1=False, orth=False, uniforms=None, adam_init=None, **kw):
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKtupStepper
    d = full['U'].shape[1]
    mk = lambda key: parallel.ShardedTable(full[key].shape[0], d, rank=rank, world=world, device=dev,
                                           init=lambda g: full[key][g].to(dev))
    Ut, It, Et = mk('U'), mk('I'), mk('E')
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    B = batches[0][rank][0].numel()
    st = ShardedKtupStepper(Ut, It, Et, *small, i2e.to(torch.int32).to(dev), batch=B, kind=kind, lr=lr, eps=eps, max_norm=max_norm,
                            l1=l1, orth=orth, **kw)
    if adam_init is not None:                 # every row's state as written at step t0
        t0, M, V = adam_init
        for t, m, v in zip((Ut, It, Et), M[:3], V[:3]):
            own = torch.arange(rank, t.total_rows, world)
            t.state[:, :d] = m[own].to(dev); t.state[:, d:2 * d] = v[own].to(dev)
            t.state[:, 2 * d] = torch.full((own.numel(),), t0, dtype=torch.int32).view(torch.float32).to(dev)
        for sst, m, v in zip(st.small_state, M[3:], V[3:]):
            sst[:, :d] = m.to(dev); sst[:, d:2 * d] = v.to(dev)
            sst[:, 2 * d] = torch.full((m.shape[0],), t0, dtype=torch.int32).view(torch.float32).to(dev)
        st.opt_step[0] = t0
    for k_step, step in enumerate(batches):
        if uniforms is not None:              # this rank's rows of the recorded draws, positives then negatives
            st.set_gumbel_uniforms(torch.cat([x[rank * B:(rank + 1) * B] for x in uniforms[k_step]]).to(dev))
        st(*(x.to(dev) for x in step[rank]))
    st.flush()                                # the lazy rules (Adam, weight decay): the rows the last steps did not touch, up to the last step
    torch.cuda.synchronize()
    return (Ut, It, Et), small, st


def _check(tables, small, Wd, rank, world, rtol=1e-4, atol=2e-5):
    for t, w in zip(tables, Wd[:3]):
        n = t.total_rows
        torch.testing.assert_close(t.weight.data.cpu(), w[torch.arange(rank, n, world)], rtol=rtol, atol=atol)
    for p, w in zip(small, Wd[3:]):
        torch.testing.assert_close(p.data.cpu(), w, rtol=rtol, atol=atol)


def _batches(gen, world, steps, nu, ni, b):
    return [[(torch.randint(0, nu, (b,), generator=gen), torch.randint(0, ni, (b,), generator=gen),
              torch.randint(0, ni, (b,), generator=gen)) for _ in range(world)] for _ in range(steps)]


# Adagrad's first steps divide by |g| + eps: with the default eps = 1e-10 an element whose gradient is ~1e-10 turns fp32 rounding
# noise into an O(lr) difference, so the comparisons use eps = 1e-4 (well-conditioned, same code path), as tests/test_hip_config5.py
@pytest.mark.parametrize('d,P', [(256, 20), (100, 20), (64, 4)])
@pytest.mark.parametrize('kind', ['adagrad', 'sgd'])
@pytest.mark.parametrize('form', ['one_graph', 'exchange_form', 'eager', 'one_graph_gradient_buffer', 'exchange_form_gradient_buffer'])
def test_stepper_equals_the_dense_reference_step(d, P, kind, form):
    nu, ni, ne, b, steps = 900, 300, 700, 512, 5                 # duplicates in every batch; every 7th item has no entity
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=11 + d, pad_every=7)
    batches = _batches(gen, 1, steps, nu, ni, b)
    lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (20.0, 0.5)
    Wd, losses = _dense_reference(full, small0, i2e, batches, kind, lr, 1e-4, max_norm)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}, 'eager': {'use_graphs': False},
          'one_graph_gradient_buffer': {'fused_apply': False}, 'exchange_form_gradient_buffer': {'force_exchange': True, 'fused_apply': False}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, kind, lr, 1e-4, max_norm, 0, 1, torch.device(DEV), **kw)
    assert st.steps == steps and (form == 'eager') == (st._graphs is None)
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)
    assert st.overflowed_steps() == 0
    st.check()


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form', 'one_graph_gradient_buffer'])
@pytest.mark.parametrize('d', [256, 64])
def test_stepper_hot_rows_span_many_workgroups(form, d):
    """12 users / 40 items / 30 entities under 512-pair batches: every row collects dozens of entries, so its sorted segment spans
    several workgroups of the reduction (32 entries each at d = 256) -- the boundary-row list of the two-walk form."""
    nu, ni, ne, b, steps, P = 12, 40, 30, 512, 4, 20
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=3)
    batches = _batches(gen, 1, steps, nu, ni, b)
    Wd, losses = _dense_reference(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}, 'one_graph_gradient_buffer': {'fused_apply': False}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, 0, 1, torch.device(DEV), **kw)
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)


@pytest.mark.parametrize('l1,orth,pad_every', [(True, False, 0), (False, True, 5), (False, False, 0)])
def test_stepper_l1_distance_and_orthogonal_regulariser(l1, orth, pad_every):
    """pad_every = 0: every item has an entity row, so the one-rank step gathers straight from the shards (no pack launch)."""
    nu, ni, ne, b, steps, d, P = 500, 200, 400, 256, 3, 100, 20
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=5, pad_every=pad_every)
    batches = _batches(gen, 1, steps, nu, ni, b)
    Wd, losses = _dense_reference(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, l1=l1, orth=orth)
    tables, small, st = _run_stepper(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, 0, 1, torch.device(DEV), l1=l1, orth=orth)
    assert st.direct == (pad_every == 0)
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0] + st.loss_sum[1]), sum(losses), rtol=1e-4)


@pytest.mark.parametrize('direct', [True, False, 'beside', 'exchange', 'exchange_adam'])
def test_stepper_device_fed_batches_walk_the_columns(direct):
    """set_feed: an epoch of pre-drawn batches in device columns, the step's own launches move the cursor (no per-step copy or
    argument); 5 steps over 3 batches wrap around.  'beside': the step kernel reads the id columns itself and the WHOLE route runs
    on the second graph branch (route_beside).  'exchange': the several-ranks form, where the route of step s + 1 runs at the END of
    step s beside the owner's apply walk (the cursor is then one batch ahead of the steps)."""
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKtupStepper
    nu, ni, ne, b, d, P = 700, 250, 500, 256, 128, 20
    dev = torch.device(DEV)
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=41)
    three = _batches(gen, 1, 3, nu, ni, b)
    order = [0, 1, 2, 0, 1]
    exch = str(direct).startswith('exchange')
    kind, lr, eps = ('adam', 0.01, 1e-5) if direct == 'exchange_adam' else ('adagrad', 0.05, 1e-4)
    Wd, losses = _dense_reference(full, small0, i2e, [three[k] for k in order], kind, lr, eps, 0.5)
    mk = lambda key: parallel.ShardedTable(full[key].shape[0], d, rank=0, world=1, device=dev, init=lambda g: full[key][g].to(dev))
    Ut, It, Et = mk('U'), mk('I'), mk('E')
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    st = ShardedKtupStepper(Ut, It, Et, *small, i2e.to(torch.int32).to(dev), batch=b, kind=kind, lr=lr, eps=eps, max_norm=0.5,
                            direct=None if exch else bool(direct), route_beside=direct == 'beside', force_exchange=exch)
    cols = [torch.stack([three[k][0][c] for k in range(3)]).to(dev) for c in range(3)]
    st.set_feed(cols)
    for _ in order:
        st.run()
    st.flush()
    torch.cuda.synchronize()
    assert int(st.cursor) == len(order) + (direct != 'beside') and st._graphs is not None       # (the next step's route has already run)
    _check((Ut, It, Et), small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)


def test_stepper_exchange_form_over_rccl_at_world_one():
    """The several-ranks route with REAL collectives: init_process_group('nccl', world_size=1) -- RCCL's all_to_all_single and
    all_reduce on the device buffers, between the captured segments."""
    import torch.distributed as dist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        nu, ni, ne, b, steps, d, P = 900, 300, 700, 512, 4, 256, 20
        full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=23, pad_every=9)
        batches = _batches(gen, 1, steps, nu, ni, b)
        Wd, _ = _dense_reference(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5)
        tables, small, st = _run_stepper(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, 0, 1, torch.device(DEV), force_exchange=True)
        assert st.multi and len(st._graphs) == 1          # RCCL: the five segments and the collectives between them are ONE graph
        _check(tables, small, Wd, 0, 1)
        st.close()                                        # the graph holds captured collectives: it goes before the group does
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, kind, overflow):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share this box's GPU (RCCL refuses that)
    try:
        dev = torch.device(DEV)
        nu, ni, ne, b, steps, d, P = 901, 301, 703, 512, 4, 256, 20     # odd row counts: the shards differ in size
        full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=31, pad_every=6)
        batches = _batches(gen, world, steps, nu, ni, b)
        lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (0.01, 0.5) if kind == 'adam' else (20.0, 0.5)
        eps = 1e-5 if kind == 'adam' else 1e-4
        if kind == 'adam':                                                # more, smaller steps: rows rest for a few steps between touches
            steps, b = 7, 128
            batches = _batches(gen, world, steps, nu, ni, b)
        if overflow:
            tables, small, st = _run_stepper(full, small0, i2e, batches[:3], kind, lr, 1e-4, max_norm, rank, world, dev, capacity_factor=0.01)
            assert st.overflowed_steps() > 0                            # cap = 64 + a few rows < the batch's distinct ids per owner
            with pytest.raises(Exception):
                st.check()
            for key, t in zip(('U', 'I', 'E'), tables):                  # every overflowed step was skipped on every rank
                assert torch.equal(t.weight.data.cpu(), full[key][torch.arange(rank, t.total_rows, world)])
            for p, w in zip(small, small0):
                assert torch.equal(p.data.cpu(), w)
            assert float(st.Gown.abs().sum()) == 0.0     # and left no gradient behind (the requester's buffer is stored into, never accumulated)
            return
        Wd, _ = _dense_reference(full, small0, i2e, batches, kind, lr, eps, max_norm)
        tables, small, st = _run_stepper(full, small0, i2e, batches, kind, lr, eps, max_norm, rank, world, dev)
        assert st.multi and len(st._graphs) == 5 and st.overflowed_steps() == 0
        _check(tables, small, Wd, rank, world)
        copies = [torch.empty_like(small[0].data.cpu()) for _ in range(world)]
        dist.all_gather(copies, small[0].data.cpu())
        assert all(torch.equal(copies[0], c) for c in copies)            # replicated tables stay bit-identical across ranks
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,overflow', [('adagrad', False), ('sgd', False), ('adagrad', True), ('adam', False)])
def test_stepper_two_ranks_share_the_gpu(kind, overflow):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, kind, overflow), nprocs=2, join=True)


def _zipf(gen, n_rows, n, a):
    uu = torch.rand(n, generator=gen, dtype=torch.float64)
    a1 = a - 1.0
    top = float(n_rows) ** (-a1)
    rank = (1.0 - uu * (1.0 - top)) ** (-1.0 / a1)
    return (rank.clamp(1, n_rows) - 1).to(torch.int64)


@pytest.mark.parametrize('zipf', [0.0, 1.05])
def test_stepper_at_one_ranks_share_of_config5(zipf):
    """1.25 M / 125 K / 625 K rows, d = 256, P = 20, B = 8192, uniform and Zipf(1.05) ids (a hot row takes hundreds of entries of a
    batch): the touched rows equal the dense reference run on the rows the batches touch, everything else is bit-identical."""
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKtupStepper
    NU, NI, NE, P, D, B, steps = 1_250_000, 125_000, 625_000, 20, 256, 8192, 3
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV); g.manual_seed(3)

    def table(n):
        t = parallel.ShardedTable(n, D, rank=0, world=1, device=dev)
        t.weight.data.copy_(torch.nn.functional.normalize(torch.randn(t.weight.shape, generator=g, device=DEV), dim=1))
        return t
    Ut, It, Et = table(NU), table(NI), table(NE)
    small = [torch.nn.Parameter(torch.nn.functional.normalize(torch.randn(P, D, generator=g, device=DEV), dim=1)) for _ in range(4)]
    item2ent = torch.randint(0, NE, (NI,), generator=g, device=DEV).to(torch.int32)
    cg = torch.Generator().manual_seed(9)
    draw = (lambda n_rows: torch.randint(0, n_rows, (B,), generator=cg)) if zipf <= 0 else (lambda n_rows: _zipf(cg, n_rows, B, zipf))
    batches = [[(draw(NU), draw(NI), draw(NI))] for _ in range(steps)]
    # the dense reference on the sub-world of touched rows: compact tables (row j = j-th distinct id), remapped batches
    touched = {}
    for key, parts in (('U', [b[0][0] for b in batches]), ('I', [x for b in batches for x in b[0][1:]])):
        touched[key] = torch.unique(torch.cat(parts))
    i2e_cpu = item2ent.cpu().long()
    touched['E'] = torch.unique(i2e_cpu[touched['I']])
    remap = {k: {int(v): j for j, v in enumerate(ids.tolist())} for k, ids in touched.items()}
    sub = {k: t.weight.data[touched[k].to(DEV)].cpu() for k, t in (('U', Ut), ('I', It), ('E', Et))}
    sub_i2e = torch.tensor([remap['E'][int(i2e_cpu[i])] for i in touched['I'].tolist()])
    sub_batches = [[tuple(torch.tensor([remap[k][int(v)] for v in x.tolist()]) for k, x in zip(('U', 'I', 'I'), b[0]))] for b in batches]
    before = {k: t.weight.data.clone() for k, t in (('U', Ut), ('I', It), ('E', Et))}
    Wd, losses = _dense_reference(sub, [p.data.cpu() for p in small], sub_i2e, sub_batches, 'adagrad', 0.05, 1e-4, 5.0)
    st = ShardedKtupStepper(Ut, It, Et, *small, item2ent, batch=B, kind='adagrad', lr=0.05, eps=1e-4, max_norm=5.0)
    for b in batches:
        st(*(x.to(DEV) for x in b[0]))
    torch.cuda.synchronize()
    assert st._graphs is not None and st.overflowed_steps() == 0
    if zipf > 0:
        assert touched['U'].numel() < steps * B // 2                       # hot ids: most entries share rows
    for k, t, w in (('U', Ut, Wd[0]), ('I', It, Wd[1]), ('E', Et, Wd[2][:-1])):
        ids = touched[k].to(DEV)
        torch.testing.assert_close(t.weight.data[ids].cpu(), w, rtol=1e-4, atol=2e-5)
        mask = torch.ones(t.weight.shape[0], dtype=torch.bool, device=DEV)
        mask[ids] = False
        assert torch.equal(t.weight.data[mask], before[k][mask])            # untouched rows bit-identical
        assert float(t.state[mask].abs().sum()) == 0.0
    for p, w in zip(small, Wd[3:]):
        torch.testing.assert_close(p.data.cpu(), w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)


# Adam's eps = 1e-5 here for the same reason as Adagrad's 1e-4 above (the default 1e-8 turns a gradient element of that size into a
# step of lr / 2 that fp32 rounding decides); the catch-up arithmetic is the same for any eps.
@pytest.mark.parametrize('d,P', [(256, 20), (100, 20), (64, 4)])
@pytest.mark.parametrize('form', ['one_graph', 'exchange_form', 'eager', 'one_graph_gradient_buffer'])
def test_stepper_adam_equals_the_dense_adam(d, P, form):
    """Row-sparse Adam with catch-up (include/ktup_hip.h ktup_adam_t) against torch.optim.Adam over WHOLE tables with zero-filled
    gradients -- what utils/trainer.py:63-66 builds for ktup.sh's `-optimizer_type Adam -l2_lambda 0`: nine steps on 900 users with
    128-row batches, so most rows are touched at some steps and not at others (gaps of 1-8 steps replayed on the next touch, the
    rest by the flush)."""
    nu, ni, ne, b, steps = 900, 300, 700, 128, 9
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=5 + d, pad_every=7)
    batches = _batches(gen, 1, steps, nu, ni, b)
    lr, max_norm = 0.01, 0.5
    Wd, losses = _dense_reference(full, small0, i2e, batches, 'adam', lr, 1e-5, max_norm, orth=True)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}, 'eager': {'use_graphs': False},
          'one_graph_gradient_buffer': {'fused_apply': False}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, 'adam', lr, 1e-5, max_norm, 0, 1, torch.device(DEV), orth=True, **kw)
    assert st.steps == steps and int(st.opt_step[0].item()) == steps
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum.sum()), sum(losses), rtol=1e-4)
    # every touched row of every shard was written at the last step or brought up to it; untouched rows never moved
    for t, key in zip(tables, 'UIE'):
        last = t.state[:, 2 * d].view(torch.int32)
        touched = last > 0
        assert bool((last[touched] == steps).all()) and torch.equal(t.weight.data[~touched].cpu(), full[key][~touched.cpu()])
    st.check()


@pytest.mark.parametrize('kind', ['adagrad', 'adam'])
def test_run_cycle_is_the_same_steps_in_one_graph(kind):
    """run_cycle(n): n device-fed steps as ONE graph replay (the next step's route joined in front of the next step instead of in front
    of this step's apply walk) -- the same launches in the same order, so the same tables as step-by-step replays and as the dense
    reference; ten steps over three batches: two by themselves (warm-up), then two cycles of four."""
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKtupStepper
    nu, ni, ne, b, d, P = 700, 250, 500, 256, 128, 20
    dev = torch.device(DEV)
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=43)
    three = _batches(gen, 1, 3, nu, ni, b)
    order = [0, 1, 2, 0, 1, 2, 0, 1, 2, 0]
    lr, eps = (0.01, 1e-5) if kind == 'adam' else (0.05, 1e-4)
    Wd, losses = _dense_reference(full, small0, i2e, [three[k] for k in order], kind, lr, eps, 0.5)
    mk = lambda key: parallel.ShardedTable(full[key].shape[0], d, rank=0, world=1, device=dev, init=lambda g: full[key][g].to(dev))
    Ut, It, Et = mk('U'), mk('I'), mk('E')
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    st = ShardedKtupStepper(Ut, It, Et, *small, i2e.to(torch.int32).to(dev), batch=b, kind=kind, lr=lr, eps=eps, max_norm=0.5)
    st.set_feed([torch.stack([three[k][0][c] for k in range(3)]).to(dev) for c in range(3)])
    st.run(); st.run()
    st.run_cycle(4); st.run_cycle(4)
    st.run_cycle(3)                                            # (odd: step by step)
    st.flush()
    torch.cuda.synchronize()
    assert st.steps == 13 and int(st.cursor) == 14 and st._cycles is not None and len(st._cycles) == 1
    Wd13, losses13 = _dense_reference(full, small0, i2e, [three[k] for k in order + [1, 2, 0]], kind, lr, eps, 0.5)
    _check((Ut, It, Et), small, Wd13, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses13), rtol=1e-4)
    st.check()
    st.close()


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form'])
def test_stepper_adam_on_old_states_takes_the_series(form):
    """The same comparison 5,000 steps into a run: every row carries first and second moments written at step 5,000, the batches are
    small (32 pairs over 900 users / 300 items / 700 entities), so a row rests for 5-30 steps between two touches and its catch-up is
    the SERIES form of the replay (ktup_shard_step.hip adam_zero_series: eight steps or more of a state whose bias corrections have
    settled) -- inside the stepper's own launches, catch-up and apply walk, against torch.optim.Adam handed the same state."""
    nu, ni, ne, b, steps, d, P, t0 = 900, 300, 700, 32, 40, 128, 20, 5000
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=77, pad_every=7)
    batches = _batches(gen, 1, steps, nu, ni, b)
    shapes = [full['U'].shape, full['I'].shape, full['E'].shape] + [t.shape for t in small0]
    M = [1e-3 * torch.randn(*sh, generator=gen) for sh in shapes]
    V = [(m ** 2) * (0.3 + 2 * torch.rand(*m.shape, generator=gen)) + 1e-14 for m in M]
    # (no elements with sqrt(v) below eps here: the first gradient such an element sees moves it by ~3 lr whatever the gradient's size,
    #  with the sign of a number that may be rounding noise -- test_adam_flush_replays_the_untouched_steps has them, without a model around)
    lr, max_norm = 0.01, 0.5
    init = (t0, M, V)
    Wd, losses = _dense_reference(full, small0, i2e, batches, 'adam', lr, 1e-8, max_norm, orth=True, adam_init=init)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, 'adam', lr, 1e-8, max_norm, 0, 1, torch.device(DEV), orth=True, adam_init=init, **kw)
    assert int(st.opt_step[0].item()) == t0 + steps
    _check(tables, small, Wd, 0, 1, atol=6e-5)                   # 40 steps of two float32 runs (the nine-step tests: 2e-5; measured here: 3.5e-5)
    np.testing.assert_allclose(float(st.loss_sum.sum()), sum(losses), rtol=1e-4)
    st.check()


@pytest.mark.parametrize('base', [0, 300, 20000])
@pytest.mark.parametrize('gap', [1, 7, 8, 60, 109, 110, 111, 400])
def test_adam_flush_replays_the_untouched_steps(gap, base):
    """ktup_shard_adam_flush against the dense recurrence written out in float64: a row whose state was written at step `last` is
    taken through steps last + 1 .. t with a zero gradient -- m <- beta1 m, v <- beta2 v, p <- p - lr / (1 - beta1^s) m / (sqrt(v /
    (1 - beta2^s)) + eps) -- one after the other; beyond 110 replayed steps (adam_replay: the increments have fallen below 1e-5 of
    the first) only m and v keep decaying.  Rows never touched (last = 0) stay as they are.
    `base`: how old the rows' states are.  Young states (bias correction 1 - beta2^s still moving by percents per step) are replayed
    step by step; from a few hundred steps on a replay of eight steps or more is the series form (ktup_shard_step.hip
    adam_zero_series) -- the same float64 recurrence is the reference for both."""
    import ctypes
    from jTransUP.hip import lib as L
    from jTransUP.sharded_ktup import AdamRule, adam_replay, adam_state_pitch
    gen = torch.Generator().manual_seed(gap)
    n, d, lr, eps, b1, b2 = 96, 100, 0.01, 1e-8, 0.9, 0.999
    p0 = torch.randn(n, d, generator=gen)
    m0 = 1e-2 * torch.randn(n, d, generator=gen)
    v0 = (m0 ** 2) * (0.2 + 3 * torch.rand(n, d, generator=gen)) + 1e-12
    m0[:, :7] *= 1e-7; v0[:, :7] = m0[:, :7] ** 2              # elements where eps carries the denominator (|g| ~ 1e-9 < eps)
    last = (base + torch.randint(1, 60, (n,), generator=gen)).to(torch.int32)
    last[::9] = 0
    m0[last == 0] = 0; v0[last == 0] = 0
    t = int(last.max()) + gap
    state = torch.zeros(n, adam_state_pitch(d))
    state[:, :d] = m0; state[:, d:2 * d] = v0
    state[:, 2 * d] = last.view(torch.float32)
    P, S = p0.to(DEV), state.to(DEV)
    step = torch.tensor([t, 0], dtype=torch.int64, device=DEV)
    rule = AdamRule(b1, b2, adam_replay((b1, b2)), 0, step.data_ptr())
    L.call('ktup_shard_adam_flush', P.data_ptr(), P.stride(0), S.data_ptr(), S.stride(0), d, n, lr, eps, ctypes.addressof(rule),
           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p, m, v = p0.double().clone(), m0.double().clone(), v0.double().clone()
    for s in range(base + 1, t + 1):
        on = (last.long() > 0) & (last.long() < s)
        m[on] *= b1; v[on] *= b2
        p[on] -= lr / (1 - b1 ** s) * m[on] / (v[on].sqrt() / (1 - b2 ** s) ** 0.5 + eps)
    got_p, got_s = P.cpu(), S.cpu()
    # beyond the replay cap the dropped tail is < 1e-4 of a row's FIRST replayed increment (here up to 0.2: tiny `last`, large bias correction)
    torch.testing.assert_close(got_p.double(), p, rtol=1e-5, atol=2e-6 if gap <= 110 else 3e-5)       # fp32 running sums of up to 110 increments against float64 (+ < 1e-4 of the first increment dropped)
    # (beta1 = 0.9 is 0.89999998 as the fp32 kernel argument: 2.6e-8 per step, 1e-5 after 400)
    torch.testing.assert_close(got_s[:, :d].double(), m, rtol=5e-5, atol=1e-30)
    torch.testing.assert_close(got_s[:, d:2 * d].double(), v, rtol=5e-5, atol=1e-30)
    want_last = torch.where(last > 0, torch.full_like(last, t), last)
    assert torch.equal(got_s[:, 2 * d].view(torch.int32), want_last)


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form'])
@pytest.mark.parametrize('wd', [1e-5, 1e-2])
@pytest.mark.parametrize('kind', ['adagrad', 'sgd', 'adam'])
def test_stepper_weight_decay_equals_the_dense_step(kind, wd, form):
    """-l2_lambda (utils/trainer.py:63-77: every optimizer is built with weight_decay = l2_lambda; base.py:51: 1e-5 by default): the dense
    step moves EVERY row at every step by the optimizer's step on g = wd * p.  Row-sparse: a row owes those steps until it is touched
    (or flushed) and then takes them one by one (ktup_adam_t rule / weight_decay).  Small batches over bigger tables, so that most rows
    rest for several steps -- and many are never touched at all before the flush; 1e-2 makes the owed steps large against the band."""
    nu, ni, ne, b, steps, d, P = 500, 260, 400, 64, 7, 100, 20
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=53, pad_every=7)
    batches = _batches(gen, 1, steps, nu, ni, b)
    lr, eps = (0.01, 1e-5) if kind == 'adam' else (0.05, 1e-4) if kind == 'adagrad' else (5.0, 1e-4)
    Wd, losses = _dense_reference(full, small0, i2e, batches, kind, lr, eps, 0.5, weight_decay=wd)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, kind, lr, eps, 0.5, 0, 1, torch.device(DEV), weight_decay=wd, **kw)
    assert st.lazy and tables[0].state.shape[1] == 2 * d + 4
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)
    moved = (Wd[0] - full['U']).abs().max(dim=1).values > 0              # the dense run moved EVERY user row (decay), touched or not
    assert bool(moved.all())


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form'])
@pytest.mark.parametrize('d', [100, 256])
def test_stepper_st_gumbel_gate_equals_the_dense_step(d, form):
    """-use_st_gumbel (transup.sh:1's gate; transUP.py:118-170, jTransUP.py:250-262) in the sharded rec step: the forward takes the
    one-hot of argmax(logits + Gumbel noise), the backward the softmax's Jacobian.  Parity mode: the uniforms the reference would draw
    (one (B, n_pref) tensor for the positives, one for the negatives, per step) are recorded and handed to both sides."""
    nu, ni, ne, b, steps, P = 300, 200, 250, 128, 4, 20
    full, small0, i2e, gen = _world_tables(nu, ni, ne, P, d, seed=59, pad_every=5)
    batches = _batches(gen, 1, steps, nu, ni, b)
    uniforms = [(torch.rand(b, P, generator=gen), torch.rand(b, P, generator=gen)) for _ in range(steps)]
    Wd, losses = _dense_reference(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, uniforms=uniforms)
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}}[form]
    tables, small, st = _run_stepper(full, small0, i2e, batches, 'adagrad', 0.05, 1e-4, 0.5, 0, 1, torch.device(DEV), uniforms=uniforms,
                                     use_st_gumbel=True, **kw)
    _check(tables, small, Wd, 0, 1)
    np.testing.assert_allclose(float(st.loss_sum[0]), sum(losses), rtol=1e-4)
    # and the production mode: noise from the device-resident Philox stream, which moves by 2 B P draws per step
    st.set_gumbel_uniforms(None)
    before = int(st.gstate[1])
    for step in batches[:3]:
        st(*(x.to(DEV) for x in step[0]))
    torch.cuda.synchronize()
    assert int(st.gstate[1]) - before == 3 * 2 * b * P and st.overflowed_steps() == 0
    assert all(bool(torch.isfinite(t.weight.data).all()) for t in tables)


def _dense_tup_reference(full, small0, batches, kind, lr, eps, max_norm, l1=False, weight_decay=0.0):
    """TUP's rec step as item_recommendation.py:160-192 runs it: bprLoss + orthogonalLoss(pref, pref_norm) + normLoss(user rows of the
    batch) + normLoss(item rows of [pos ; neg]) + normLoss(pref), clip, dense optimizer."""
    W = [torch.nn.Parameter(full['U'].clone()), torch.nn.Parameter(full['I'].clone())] + [torch.nn.Parameter(t.clone()) for t in small0[:2]]
    wd = weight_decay
    opt = torch.optim.Adagrad(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adagrad' else \
        torch.optim.Adam(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adam' else torch.optim.SGD(W, lr=lr, weight_decay=wd)
    losses = []
    for step in batches:
        opt.zero_grad(set_to_none=False)
        u = torch.cat([x[0] for x in step]); pi = torch.cat([x[1] for x in step]); ni_ = torch.cat([x[2] for x in step])
        pos = O.score_tup(*W, u, pi, l1); neg = O.score_tup(*W, u, ni_, l1)
        loss = torch.nn.functional.softplus(pos - neg).mean() + O.orthogonal_loss(W[2], W[3]) + O.norm_loss(W[0][u]) \
            + O.norm_loss(W[1][torch.cat([pi, ni_])]) + O.norm_loss(W[2])
        loss.backward()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(W, max_norm)
        opt.step()
        losses.append(float(loss.detach()))
    return [w.data for w in W], losses


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form', 'packed'])
@pytest.mark.parametrize('kind,wd', [('adagrad', 0.0), ('adam', 0.0), ('adagrad', 1e-5)])
@pytest.mark.parametrize('d,P', [(100, 10), (256, 20)])
def test_tup_stepper_equals_the_dense_reference_step(d, P, kind, wd, form):
    """ShardedKtupStepper WITHOUT an entity table = TUP's rec step on row-sharded user / item tables (config 3 at scale:
    run_item_recommendation.py -model_type transup -shard_tables), with the row regularisers of item_recommendation.py:177-180 (rows scaled
    so that many norms exceed 1 and the regularisers bite) against the dense reference step."""
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKtupStepper
    nu, ni, b, steps = 400, 250, 128, 5
    gen = torch.Generator().manual_seed(61)
    full = {k: 1.1 * torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1) * (0.8 + 0.4 * torch.rand(n, 1, generator=gen))
            for k, n in (('U', nu), ('I', ni))}
    small0 = [1.05 * torch.nn.functional.normalize(torch.randn(P, d, generator=gen), dim=1) for _ in range(2)]
    batches = _batches(gen, 1, steps, nu, ni, b)
    lr, eps = (0.01, 1e-5) if kind == 'adam' else (0.05, 1e-4)
    Wd, losses = _dense_tup_reference(full, small0, batches, kind, lr, eps, 0.5, weight_decay=wd)
    dev = torch.device(DEV)
    mk = lambda key: parallel.ShardedTable(full[key].shape[0], d, rank=0, world=1, device=dev, init=lambda g: full[key][g].to(dev))
    Ut, It = mk('U'), mk('I')
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}, 'packed': {'direct': False}}[form]
    st = ShardedKtupStepper(Ut, It, None, small[0], small[1], None, None, None, batch=b, kind=kind, lr=lr, eps=eps, max_norm=0.5, orth=True,
                            row_regs=True, weight_decay=wd, **kw)
    for step in batches:
        st(*(x.to(dev) for x in step[0]))
    st.flush()
    torch.cuda.synchronize()
    assert st.tup and st.E == 3 * b and st.overflowed_steps() == 0
    _check((Ut, It), [], Wd, 0, 1)
    for p_, w in zip(small, Wd[2:]):
        torch.testing.assert_close(p_.data.cpu(), w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(st.loss_sum.sum()), sum(losses), rtol=1e-4)
    assert float(st.loss_sum[2]) > 0 and float(st.loss_sum[3]) > 0        # the row regularisers did bite
