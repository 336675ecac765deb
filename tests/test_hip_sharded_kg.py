"""Config 5's kg step and the joint schedule on row-sharded tables (jTransUP/sharded_ktup.py: ShardedKgStepper, ShardedKtupJoint)
against a single-process DENSE run of the reference's step bodies on CPU: the oracle's step losses (oracle/cpu_ref.py
kg_step_loss = knowledgable_recommendation.py:368-383, ktup_rec_step_loss = :335-344), clip_grad_norm_ over ALL parameters and
torch.optim (utils/trainer.py:63-77, l2_lambda = 0), on the reference's 10-step cycle (:209,320).  One rank (one graph, direct
gathers and packed rows), one rank in exchange form, two ranks sharing the GPU, relation counts from 1 to more than the sort's
LDS histogram holds, TransE, L1, hot entities, overflow."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _tables(nu, ni, ne, P, d, seed, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    # rows of norm ~ scale: with scale > 1 normLoss is active on most rows, below 1 on none
    mk = lambda n: torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1) * (scale * (0.6 + 0.8 * torch.rand(n, 1, generator=gen)))
    full = {'U': mk(nu), 'I': mk(ni), 'E': mk(ne)}
    small = [mk(P) for _ in range(4)]                                   # pref, pref_norm, rel, norm
    i2e = torch.randint(0, ne, (ni,), generator=gen)
    return full, small, i2e, gen


def _kg_batches(gen, world, steps, ne, P, b, hot=0):
    def ent():
        if hot:
            return torch.randint(0, hot, (b,), generator=gen)
        return torch.randint(0, ne, (b,), generator=gen)
    out = []
    for _ in range(steps):
        per = []
        for _ in range(world):
            ph, pt, pr = ent(), ent(), torch.randint(0, P, (b,), generator=gen)
            flip = torch.rand(b, generator=gen) < 0.5                    # corrupt the head or the tail (utils/data.py:12-18)
            other = ent()
            per.append((ph, pt, pr, torch.where(flip, other, ph), torch.where(flip, pt, other), pr.clone()))
        out.append(per)
    return out


def _rec_batches(gen, world, steps, nu, ni, b):
    return [[(torch.randint(0, nu, (b,), generator=gen), torch.randint(0, ni, (b,), generator=gen),
              torch.randint(0, ni, (b,), generator=gen)) for _ in range(world)] for _ in range(steps)]


def _dense(full, small0, i2e, schedule, kind, lr, eps, max_norm, l1=False, margin=1.0, kg_lambda=1.0, transh=True, orth=True, weight_decay=0.0):
    """schedule: list of ('rec', per-rank batches) / ('kg', per-rank batches).  -> tables after the steps, per-step losses."""
    W = [torch.nn.Parameter(full[k].clone()) for k in ('U', 'I', 'E')] + [torch.nn.Parameter(t.clone()) for t in small0]
    U, I, E, Pf, Pn, R, Rn = W
    wd = weight_decay
    opt = torch.optim.Adagrad(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adagrad' else \
        torch.optim.Adam(W, lr=lr, eps=eps, weight_decay=wd) if kind == 'adam' else torch.optim.SGD(W, lr=lr, weight_decay=wd)
    losses = []
    for what, per in schedule:
        opt.zero_grad(set_to_none=False)      # zero-FILL (torch 0.3): a table keeps being stepped on the steps that do not touch it
        cat = [torch.cat([x[c] for x in per]) for c in range(len(per[0]))]
        if what == 'rec':
            pos = O.score_ktup_rec(U, I, E, Pf, Pn, R, Rn, i2e, cat[0], cat[1], l1)
            neg = O.score_ktup_rec(U, I, E, Pf, Pn, R, Rn, i2e, cat[0], cat[2], l1)
            loss = O.bpr_loss(pos, neg, -1.0)
            if orth:
                loss = loss + O.orthogonal_loss(Pf, Pn)
        else:
            loss = O.kg_step_loss(E, R, Rn if transh else None, *cat, l1=l1, margin=margin, kg_lambda=kg_lambda)
        loss.backward()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(W, max_norm)
        opt.step()
        losses.append((what, float(loss.detach())))
    return [w.data for w in W], losses


def _sharded(full, dev, rank, world):
    from jTransUP import parallel
    d = full['U'].shape[1]
    return [parallel.ShardedTable(full[k].shape[0], d, rank=rank, world=world, device=dev, init=lambda g, k=k: full[k][g].to(dev))
            for k in ('U', 'I', 'E')]


def _close_mostly(got, want, rtol, atol, frac=1e-3, cap=5e-4):
    """Adagrad divides by sqrt(sum g^2) + eps: an element whose gradients are ~eps (a sum of hundreds of cancelling terms in the small
    tables) turns summation-order rounding into a visible difference.  At most `frac` of the elements may leave the (rtol, atol)
    band, none by more than `cap`; a wrong gradient or a lost update moves whole rows by ~lr = 0.05."""
    err = (got - want).abs()
    bad = err > atol + rtol * want.abs()
    assert float(err.max()) <= cap and int(bad.sum()) <= max(1, int(frac * want.numel())), (float(err.max()), int(bad.sum()))


def _check(tables, keys, small, Wd, rank, world, rtol=1e-4, atol=2e-5):
    idx = {'U': 0, 'I': 1, 'E': 2}
    for t, k in zip(tables, keys):
        torch.testing.assert_close(t.weight.data.cpu(), Wd[idx[k]][torch.arange(rank, t.total_rows, world)], rtol=rtol, atol=atol)
    for p, w in zip(small, Wd[3:]):
        _close_mostly(p.data.cpu(), w, rtol, atol)


# eps = 1e-4 for Adagrad as in tests/test_hip_sharded_ktup.py (a ~1e-10 gradient element otherwise turns rounding into O(lr))
@pytest.mark.parametrize('d,P,b', [(256, 20, 512), (100, 20, 256), (64, 1, 200), (128, 600, 512), (32, 20000, 96)])
@pytest.mark.parametrize('kind', ['adagrad', 'sgd'])
@pytest.mark.parametrize('form', ['one_graph', 'packed', 'exchange_form', 'eager'])
def test_kg_stepper_equals_the_dense_reference_step(d, P, b, kind, form):
    if P > 1000 and (kind != 'adagrad' or form != 'one_graph'):
        pytest.skip('the identity-order fallback of the relation sort is covered once')
    from jTransUP.sharded_ktup import ShardedKgStepper
    ne, steps = 700, 5
    dev = torch.device(DEV)
    full, small0, i2e, gen = _tables(10, 10, ne, P, d, seed=7 + d, scale=1.1)
    batches = _kg_batches(gen, 1, steps, ne, P, b)
    lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (0.02, 0.5)
    Wd, losses = _dense(full, small0, i2e, [('kg', x) for x in batches], kind, lr, 1e-4, max_norm, margin=1.0, kg_lambda=0.5)
    _, _, Et = _sharded(full, dev, 0, 1)
    rel, norm = [torch.nn.Parameter(t.clone().to(dev)) for t in small0[2:]]
    kw = {'one_graph': {}, 'packed': {'direct': False}, 'exchange_form': {'force_exchange': True}, 'eager': {'use_graphs': False}}[form]
    st = ShardedKgStepper(Et, rel, norm, batch=b, kind=kind, lr=lr, eps=1e-4, max_norm=max_norm, margin=1.0, kg_lambda=0.5, **kw)
    for step in batches:
        st(*(x.to(dev) for x in step[0]))
    torch.cuda.synchronize()
    assert st.steps == steps and (form == 'eager') == (st._graphs is None) and st.direct == (form in ('one_graph', 'eager'))
    torch.testing.assert_close(Et.weight.data.cpu(), Wd[2], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(rel.data.cpu(), Wd[5], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(norm.data.cpu(), Wd[6], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(0.5 * float(st.loss_sum.sum()), sum(v for _, v in losses), rtol=1e-4)
    assert float(st.loss_sum[1]) > 0 and float(st.loss_sum[2]) > 0 and (P < 20 or float(st.loss_sum[3]) > 0)     # every regulariser was active
    assert st.overflowed_steps() == 0
    st.check()


@pytest.mark.parametrize('transh,l1', [(False, False), (True, True), (False, True)])
def test_kg_stepper_transe_and_l1(transh, l1):
    from jTransUP.sharded_ktup import ShardedKgStepper
    ne, P, d, b, steps = 500, 11, 100, 300, 4
    dev = torch.device(DEV)
    full, small0, i2e, gen = _tables(10, 10, ne, P, d, seed=19, scale=1.05)
    batches = _kg_batches(gen, 1, steps, ne, P, b)
    Wd, losses = _dense(full, small0, i2e, [('kg', x) for x in batches], 'adagrad', 0.05, 1e-4, 0.5, l1=l1, margin=2.0, transh=transh)
    _, _, Et = _sharded(full, dev, 0, 1)
    rel, norm = [torch.nn.Parameter(t.clone().to(dev)) for t in small0[2:]]
    st = ShardedKgStepper(Et, rel, norm if transh else None, batch=b, kind='adagrad', lr=0.05, eps=1e-4, max_norm=0.5, l1=l1, margin=2.0,
                          transh=transh, regs=7 if transh else 6)
    for step in batches:
        st(*(x.to(dev) for x in step[0]))
    torch.cuda.synchronize()
    torch.testing.assert_close(Et.weight.data.cpu(), Wd[2], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(rel.data.cpu(), Wd[5], rtol=1e-4, atol=2e-5)
    if transh:
        torch.testing.assert_close(norm.data.cpu(), Wd[6], rtol=1e-4, atol=2e-5)
    else:
        assert torch.equal(norm.data.cpu(), small0[3])
    np.testing.assert_allclose(float(st.loss_sum.sum()), sum(v for _, v in losses), rtol=1e-4)


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form', 'exchange_form_adam'])
def test_kg_stepper_hot_entities_and_device_fed_columns(form):
    """30 entities under 512-triple batches: every row's sorted segment spans several workgroups of the reduction; the batches come
    from device columns walked by the step's own cursor (5 steps over 3 batches wrap around).  Exchange form: the route of step s + 1
    runs at the end of step s (the cursor is one batch ahead), the requester's reduction stores its rows."""
    from jTransUP.sharded_ktup import ShardedKgStepper
    ne, P, d, b = 400, 20, 256, 512
    dev = torch.device(DEV)
    full, small0, i2e, gen = _tables(10, 10, ne, P, d, seed=29, scale=1.2)
    three = _kg_batches(gen, 1, 3, ne, P, b, hot=30)
    order = [0, 1, 2, 0, 1]
    kind, lr, eps = ('adam', 0.01, 1e-5) if form.endswith('adam') else ('adagrad', 0.05, 1e-4)
    Wd, losses = _dense(full, small0, i2e, [('kg', three[k]) for k in order], kind, lr, eps, 0.5)
    _, _, Et = _sharded(full, dev, 0, 1)
    rel, norm = [torch.nn.Parameter(t.clone().to(dev)) for t in small0[2:]]
    st = ShardedKgStepper(Et, rel, norm, batch=b, kind=kind, lr=lr, eps=eps, max_norm=0.5, force_exchange=form != 'one_graph')
    st.set_feed([torch.stack([three[k][0][c] for k in range(3)]).to(dev) for c in range(6)])
    for _ in order:
        st.run()
    st.flush()
    torch.cuda.synchronize()
    assert int(st.cursor) == len(order) + 1 and st._graphs is not None       # (the next step's route has already run)
    torch.testing.assert_close(Et.weight.data.cpu(), Wd[2], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(rel.data.cpu(), Wd[5], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(st.loss_sum.sum()), sum(v for _, v in losses), rtol=1e-4)


def _joint_schedule(gen, world, steps, nu, ni, ne, P, b, joint_ratio):
    sched = []
    for s in range(steps):
        if O.is_rec_step(s, joint_ratio):
            sched.append(('rec', _rec_batches(gen, world, 1, nu, ni, b)[0]))
        else:
            sched.append(('kg', _kg_batches(gen, world, 1, ne, P, b)[0]))
    return sched


@pytest.mark.parametrize('kind', ['adagrad', 'sgd', 'adam'])
@pytest.mark.parametrize('form', ['one_graph', 'exchange_form'])
def test_joint_schedule_equals_the_dense_reference(kind, form):
    """Twelve steps of the 7 : 3 cycle (rec x 7, kg x 3, rec x 2) over shared entity / rel / norm tables and Adagrad sums."""
    from jTransUP.sharded_ktup import ShardedKtupJoint
    nu, ni, ne, P, d, b, steps = 600, 250, 500, 20, 256, 256, 12
    dev = torch.device(DEV)
    full, small0, i2e, gen = _tables(nu, ni, ne, P, d, seed=43, scale=1.05)
    sched = _joint_schedule(gen, 1, steps, nu, ni, ne, P, b, 0.7)
    assert [w for w, _ in sched] == ['rec'] * 7 + ['kg'] * 3 + ['rec'] * 2
    lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (0.01, 0.5) if kind == 'adam' else (0.02, 0.5)
    eps = 1e-5 if kind == 'adam' else 1e-4     # Adam: the user / item tables decay through the three kg steps, pref / pref_norm too (dense semantics)
    Wd, losses = _dense(full, small0, i2e, sched, kind, lr, eps, max_norm, kg_lambda=0.5)
    tabs = _sharded(full, dev, 0, 1)
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}}[form]
    joint = ShardedKtupJoint.build(*tabs, *small, i2e.to(torch.int32).to(dev), batch=b, joint_ratio=0.7, kg_lambda=0.5, kind=kind, lr=lr,
                                   eps=eps, max_norm=max_norm, **kw)
    for what, per in sched:
        assert joint.is_rec() == (what == 'rec')
        (joint.rec if what == 'rec' else joint.kg).load_batch(*(x.to(dev) for x in per[0]))
        joint.run()
    joint.flush()
    torch.cuda.synchronize()
    _check(tabs, 'UIE', small, Wd, 0, 1)
    np.testing.assert_allclose(float(joint.rec.loss_sum.sum()), sum(v for w, v in losses if w == 'rec'), rtol=1e-4)
    np.testing.assert_allclose(0.5 * float(joint.kg.loss_sum.sum()), sum(v for w, v in losses if w == 'kg'), rtol=1e-4)
    joint.check()


def _two_rank_worker(rank, world, port, what):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share this box's GPU (RCCL refuses that)
    try:
        from jTransUP.sharded_ktup import ShardedKgStepper, ShardedKtupJoint
        dev = torch.device(DEV)
        nu, ni, ne, P, d, b = 601, 251, 503, 20, 256, 256                # odd row counts: the shards differ in size
        full, small0, i2e, gen = _tables(nu, ni, ne, P, d, seed=53, scale=1.05)
        if what == 'overflow':
            batches = _kg_batches(gen, world, 3, ne, P, b)
            tabs = _sharded(full, dev, rank, world)
            rel, norm = [torch.nn.Parameter(t.clone().to(dev)) for t in small0[2:]]
            st = ShardedKgStepper(tabs[2], rel, norm, batch=b, kind='adagrad', lr=0.05, eps=1e-4, max_norm=0.5, capacity_factor=0.01)
            for step in batches:
                st(*(x.to(dev) for x in step[rank]))
            torch.cuda.synchronize()
            assert st.overflowed_steps() == 3                           # every step overflowed: the sticky counter saw all three
            with pytest.raises(Exception):
                st.check()
            assert torch.equal(tabs[2].weight.data.cpu(), full['E'][torch.arange(rank, ne, world)])
            assert torch.equal(rel.data.cpu(), small0[2]) and torch.equal(norm.data.cpu(), small0[3])
            assert float(st.loss_sum.abs().sum()) == 0.0                # skipped steps leave no loss behind
            return
        if what == 'sticky':
            # The skipped-step counter is sticky: a run whose FIRST steps overflowed is still flagged after later steps ran, and the
            # skipped steps neither moved a table nor left their losses in loss_sum
            from jTransUP.sharded_ktup import ShardedKtupStepper
            wide = _rec_batches(gen, world, 2, nu, ni, b)             # ~115 distinct users per owner > cap = 0.25 * 256 / 2 + 64 = 96
            narrow = [[tuple(x % 40 for x in r) for r in per] for per in _rec_batches(gen, world, 3, nu, ni, b)]
            runs = []
            for batches, cf in ((wide + narrow, 0.25), (narrow, 1.25)):
                tabs = _sharded(full, dev, rank, world)
                small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
                st = ShardedKtupStepper(*tabs, *small, i2e.to(torch.int32).to(dev), batch=b, kind='sgd', lr=0.1, max_norm=0.5, capacity_factor=cf)
                for step in batches:
                    st(*(x.to(dev) for x in step[rank]))
                torch.cuda.synchronize()
                runs.append((tabs, small, st))
            (ta, sa, a), (tb, sb, bb) = runs
            assert a.overflowed_steps() == 2 and a.last_step_unplaced() == 0 and bb.overflowed_steps() == 0
            with pytest.raises(Exception):
                a.check()
            np.testing.assert_allclose(float(a.loss_sum[0]), float(bb.loss_sum[0]), rtol=1e-5)
            for x, y in zip(ta, tb):
                torch.testing.assert_close(x.weight.data, y.weight.data, rtol=1e-5, atol=1e-6)
            return
        sched = _joint_schedule(gen, world, 12, nu, ni, ne, P, b, 0.7)
        Wd, _ = _dense(full, small0, i2e, sched, 'adagrad', 0.05, 1e-4, 0.5, kg_lambda=0.5)
        tabs = _sharded(full, dev, rank, world)
        small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
        joint = ShardedKtupJoint.build(*tabs, *small, i2e.to(torch.int32).to(dev), batch=b, joint_ratio=0.7, kg_lambda=0.5, kind='adagrad',
                                       lr=0.05, eps=1e-4, max_norm=0.5)
        for what_, per in sched:
            (joint.rec if what_ == 'rec' else joint.kg).load_batch(*(x.to(dev) for x in per[rank]))
            joint.run()
        torch.cuda.synchronize()
        assert joint.kg.multi and len(joint.kg._graphs) == 5
        joint.check()
        _check(tabs, 'UIE', small, Wd, rank, world)
        copies = [torch.empty_like(small[2].data.cpu()) for _ in range(world)]
        dist.all_gather(copies, small[2].data.cpu())
        assert all(torch.equal(copies[0], c) for c in copies)            # replicated tables stay bit-identical across ranks
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('what', ['joint', 'overflow', 'sticky'])
def test_joint_schedule_two_ranks_share_the_gpu(what):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, what), nprocs=2, join=True)


@pytest.mark.parametrize('opt', ['Adagrad', 'Adam'])
def test_shard_files_restore_a_run_exactly(tmp_path, opt):
    """utils/sharded_train.ShardedJointDriver.save_shards / load_shards: a run that stops after 8 steps, writes its shard file and
    continues in a FRESH driver from that file ends where the run that never stopped ends (rows, Adagrad sums of the big
    and the small tables, the position in the 10-step cycle)."""
    import logging
    import types
    from jTransUP.models import jTransUP as jt
    from jTransUP.utils.sharded_train import ShardedJointDriver
    nu, ni, ne, P, d, b = 120, 90, 150, 6, 64, 64
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(71)
    i_map = {i: i for i in range(ni)}
    new_map = {i: ((i * 7) % ne if i % 5 else -1, i) for i in range(ni)}
    FL = types.SimpleNamespace(model_type='jtransup', share_embeddings=False, optimizer_type=opt, momentum=0.9, l2_lambda=0.0,
                               use_st_gumbel=False, joint_ratio=0.7, margin=1.0, kg_lambda=0.5, clipping_max_value=0.5, L1_flag=False)
    sched = _joint_schedule(gen, 1, 14, nu, ni, ne, P, b, 0.7)

    def fresh():
        torch.manual_seed(5)
        m = jt.jTransUPModel(False, d, nu, ni, ne, P, i_map, new_map, False, False)
        tr = types.SimpleNamespace(step=0, learning_rate=0.05, model_target=-1, save=lambda filename: None)
        return m, tr, ShardedJointDriver(m, tr, FL, b, logging.getLogger('shards'))

    def run(drv, steps):
        for what, per in steps:
            ids = [x.to(dev) for x in per[0]]
            drv.rec_step(*ids) if what == 'rec' else drv.kg_step(*ids)
    m1, tr1, d1 = fresh()
    run(d1, sched)
    d1.sync_model()
    m2, tr2, d2 = fresh()
    run(d2, sched[:8])
    path = str(tmp_path / 'run.ckpt')
    d2.save_shards(path)
    m3, tr3, d3 = fresh()
    d3.load_shards(path)
    assert tr3.step == 8 and d3.joint.steps == 8 and (opt != 'Adam' or int(d3.joint.rec.opt_step[0].item()) == 8)
    run(d3, sched[8:])
    d3.sync_model()
    # (the small tables' gradients are flushed by float atomics: two runs agree to rounding, not bit for bit)
    for (k, a), (_, c) in zip(m1.state_dict().items(), m3.state_dict().items()):
        # two RUNS are compared: the small tables' gradients (and, at this batch size, the row gradients of the fused step) are summed by
        # float atomics in an order that differs from run to run, and Adagrad's eps = 1e-10 / Adam's 1e-8 turn the rounding of an element
        # whose gradient is of that size into a visible step (one run in three, round 6) -- nearly every element to 1e-5, strays capped
        _close_mostly(c.cpu(), a.cpu(), 1e-5, 1e-6, frac=2e-3, cap=2e-4)
    nd = 2 * d if opt == 'Adam' else d                                   # Adam: [m | v | last]: compare the moments as floats, `last` as ints
    for ta, tc in zip(d1.tables, d3.tables):
        torch.testing.assert_close(ta.state[:, :nd], tc.state[:, :nd], rtol=1e-4 if opt == 'Adam' else 1e-5, atol=1e-7)
        assert opt != 'Adam' or torch.equal(ta.state[:, nd].view(torch.int32), tc.state[:, nd].view(torch.int32))
    for sa, sc in zip(d1.joint.rec.small_state, d3.joint.rec.small_state):
        torch.testing.assert_close(sa[:, :nd], sc[:, :nd], rtol=1e-4 if opt == 'Adam' else 1e-5, atol=1e-7)
    assert float(d3.tables[0].state.abs().sum()) > 0                       # the sums really travelled


def test_shard_files_restore_after_a_learning_rate_decay(tmp_path):
    """A run that has LOWERED its learning rate (utils/trainer.py:107-110: a fresh optimizer at the new rate), stepped on, written its shard
    file and is continued in a fresh driver: the restored trainer must carry the lowered rate, or the driver's first step would see a
    "change" of rate, throw the restored sums away and go on at the original rate (round 5's advisor finding).  Also restored: best_step /
    best_dev_performance, without which a resumed run's first evaluation overwrites the best checkpoint unconditionally."""
    import logging
    import types
    from jTransUP.models import jTransUP as jt
    from jTransUP.utils.sharded_train import ShardedJointDriver
    nu, ni, ne, P, d, b = 120, 90, 150, 6, 64, 64
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(73)
    i_map = {i: i for i in range(ni)}
    new_map = {i: ((i * 7) % ne if i % 5 else -1, i) for i in range(ni)}
    FL = types.SimpleNamespace(model_type='jtransup', share_embeddings=False, optimizer_type='Adagrad', momentum=0.9, l2_lambda=0.0,
                               use_st_gumbel=False, joint_ratio=0.7, margin=1.0, kg_lambda=0.5, clipping_max_value=0.5, L1_flag=False)
    sched = _joint_schedule(gen, 1, 13, nu, ni, ne, P, b, 0.7)

    def fresh():
        torch.manual_seed(5)
        m = jt.jTransUPModel(False, d, nu, ni, ne, P, i_map, new_map, False, False)
        tr = types.SimpleNamespace(step=0, learning_rate=0.05, model_target=-1, save=lambda filename: None, best_step=0,
                                   best_dev_performance=0.0, best_performances=None)
        tr.optimizer_reset = lambda lr: setattr(tr, 'learning_rate', lr)
        return m, tr, ShardedJointDriver(m, tr, FL, b, logging.getLogger('shards'))

    def run(drv, steps):
        for what, per in steps:
            ids = [x.to(dev) for x in per[0]]
            drv.rec_step(*ids) if what == 'rec' else drv.kg_step(*ids)
    m1, tr1, d1 = fresh()
    run(d1, sched[:5])
    tr1.optimizer_reset(0.025)                                             # ModelTrainer.new_performance: no improvement for an epoch
    tr1.best_step, tr1.best_dev_performance, tr1.best_performances = 3, 0.123, [(0.123, 0.1)]
    run(d1, sched[5:9])
    path = str(tmp_path / 'decayed.ckpt')
    d1.save_shards(path)
    run(d1, sched[9:])                                                     # the run that never stopped
    d1.sync_model()
    m2, tr2, d2 = fresh()                                                  # a fresh process: -learning_rate as on the command line
    d2.load_shards(path)
    assert tr2.learning_rate == 0.025 and d2._lr == 0.025 and tr2.step == 9
    assert (tr2.best_step, tr2.best_dev_performance, tr2.best_performances) == (3, 0.123, [(0.123, 0.1)])
    sums = d2.tables[2].state.clone()                                      # the entity shard: rec and kg steps both touch it
    assert float(sums.abs().sum()) > 0 and float(d2.tables[0].state.abs().sum()) > 0
    run(d2, sched[9:10])
    assert d2._lr == 0.025 and bool((d2.tables[2].state >= sums).all()) and float((d2.tables[2].state - sums).abs().sum()) > 0   # grown, not reset
    run(d2, sched[10:])
    d2.sync_model()
    for (k, a), (_, c) in zip(m1.state_dict().items(), m2.state_dict().items()):
        torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-6, msg=k)


@pytest.mark.parametrize('form', ['one_graph', 'exchange_form'])
@pytest.mark.parametrize('kind', ['adagrad', 'sgd', 'adam'])
def test_joint_schedule_under_weight_decay(kind, form):
    """The 7 : 3 cycle under -l2_lambda (utils/trainer.py:63-77: weight_decay on every table): the user / item / preference tables owe the
    decay steps of the three kg steps that do not touch them, every row of every table the steps of the batches that miss it -- replayed
    when the row is touched again or flushed (ktup_adam_t rule / weight_decay), against torch.optim's dense optimizers over whole tables
    with zero-filled gradients."""
    from jTransUP.sharded_ktup import ShardedKtupJoint
    nu, ni, ne, P, d, b, steps, wd = 500, 250, 450, 20, 100, 96, 12, 1e-3
    dev = torch.device(DEV)
    full, small0, i2e, gen = _tables(nu, ni, ne, P, d, seed=47, scale=1.05)
    sched = _joint_schedule(gen, 1, steps, nu, ni, ne, P, b, 0.7)
    lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (0.01, 0.5) if kind == 'adam' else (0.02, 0.5)
    eps = 1e-5 if kind == 'adam' else 1e-4
    Wd, losses = _dense(full, small0, i2e, sched, kind, lr, eps, max_norm, kg_lambda=0.5, weight_decay=wd)
    tabs = _sharded(full, dev, 0, 1)
    small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
    kw = {'one_graph': {}, 'exchange_form': {'force_exchange': True}}[form]
    joint = ShardedKtupJoint.build(*tabs, *small, i2e.to(torch.int32).to(dev), batch=b, joint_ratio=0.7, kg_lambda=0.5, kind=kind, lr=lr,
                                   eps=eps, max_norm=max_norm, weight_decay=wd, **kw)
    assert joint.rec.lazy and joint.kg.lazy and joint.kg.opt_step is joint.rec.opt_step
    for what, per in sched:
        (joint.rec if what == 'rec' else joint.kg).load_batch(*(x.to(dev) for x in per[0]))
        joint.run()
    joint.flush()
    torch.cuda.synchronize()
    _check(tabs, 'UIE', small, Wd, 0, 1)
    np.testing.assert_allclose(float(joint.rec.loss_sum.sum()), sum(v for w, v in losses if w == 'rec'), rtol=1e-4)
    joint.check()
