"""BASELINE config 5 at its own size on one rank: KTUP at d = 256, P = 20 with the user / item / entity tables of ONE of the
eight ranks (1.25 M / 125 K / 625 K rows), B = 8192 (u, pos, neg) triples per step, uniform and Zipf(1.05) ids, through
parallel.ShardedStep and the real scorer (ops.score_ktup: matrix-core forward, coordinate-sliced matrix-core backward, row
gradients reduced by sorted segments).  Checks against the oracle (jTransUP.py:122-143 restated in oracle/cpu_ref.py) and
against the dense Adagrad rule of utils/trainer.py:63-77 with l2_lambda = 0: scores, compact row gradients, touched-rows-only
update, untouched rows bit-identical.  The 2-rank variant shares the GPU (gloo stages the exchanges) at a smaller row count."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NU, NI, NE, P, D, B = 1_250_000, 125_000, 625_000, 20, 256, 8192


def _draw(gen, n_rows, n, zipf):
    if zipf <= 0:
        return torch.randint(0, n_rows, (n,), generator=gen, device=DEV)
    uu = torch.rand(n, generator=gen, device=DEV, dtype=torch.float64)      # inverse transform of the truncated power law
    a1 = zipf - 1.0
    top = float(n_rows) ** (-a1)
    rank = (1.0 - uu * (1.0 - top)) ** (-1.0 / a1)
    return (rank.clamp(1, n_rows) - 1).to(torch.int64)


@pytest.fixture(scope='module')
def tables():
    from jTransUP import parallel
    gen = torch.Generator(device=DEV); gen.manual_seed(3)

    def table(n):
        t = parallel.ShardedTable(n, D, rank=0, world=1, device=torch.device(DEV))
        t.weight.data.copy_(torch.nn.functional.normalize(torch.randn(t.weight.shape, generator=gen, device=DEV), dim=1))
        return t
    Ut, It, Et = table(NU), table(NI), table(NE)
    small = [torch.nn.Parameter(torch.nn.functional.normalize(torch.randn(P, D, generator=gen, device=DEV), dim=1)) for _ in range(4)]
    item2ent = torch.randint(0, NE, (NI,), generator=gen, device=DEV)
    return Ut, It, Et, small, item2ent, gen


@pytest.mark.parametrize('zipf', [0.0, 1.05])
def test_config5_step_at_size(tables, zipf):
    from jTransUP import parallel
    from jTransUP.hip import ops
    from jTransUP.hip import lib as L
    Ut, It, Et, small, item2ent, gen = tables
    Pm, Pn, R, Rn = small
    for p in small:
        p.grad = None
    Ut.state = It.state = Et.state = None
    lr, max_norm, eps = 0.05, 5.0, 1e-10
    step = parallel.ShardedStep('adagrad', lr=lr, max_norm=max_norm, eps=eps)
    u = _draw(gen, NU, B, zipf); pi = _draw(gen, NI, B, zipf); ni = _draw(gen, NI, B, zipf)
    items = torch.cat([pi, ni])
    before = {k: t.weight.data.clone() for k, t in (('U', Ut), ('I', It), ('E', Et))}
    small_before = [p.data.clone() for p in small]
    (u_rows, u_at), (i_rows, i_at), (e_rows, e_at) = step.lookup_many([(Ut, u), (It, items), (Et, item2ent[items])])
    if zipf > 0:
        assert int((step._pending[0][1][0] >= 0).sum()) < B // 2              # hot ids: the lookup dedupes them
    i2e_compact = torch.zeros(i_rows.shape[0], dtype=torch.int32, device=DEV)
    i2e_compact[i_at] = e_at.to(torch.int32)
    uu = torch.cat([u_at, u_at])
    assert L.load().ktup_score_pref_bwd_workspace_bytes(2 * B, D, u_rows.shape[0], i_rows.shape[0]) > 0   # the segment route runs
    score = ops.score_ktup(u_rows, i_rows, e_rows, Pm, Pn, R, Rn, i2e_compact, uu, i_at, False, ent_pad=-1)
    loss = torch.nn.functional.softplus(score[:B] - score[B:]).mean()      # = bprLoss(pos, neg, target=-1), utils/loss.py:29-31
    loss.backward()
    # ---- oracle on the same compact tables (CPU, autograd): scores and every gradient
    Wc = [t.detach().cpu().clone().requires_grad_(True) for t in (u_rows, i_rows, e_rows, Pm, Pn, R, Rn)]
    want = O.score_ktup_rec(*Wc, i2e_compact.cpu().long(), uu.cpu(), i_at.cpu(), False)
    np.testing.assert_allclose(score.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-5)
    torch.nn.functional.softplus(want[:B] - want[B:]).mean().backward()
    got_g = [u_rows.grad, i_rows.grad, e_rows.grad, Pm.grad, Pn.grad, R.grad, Rn.grad]
    for g, w in zip(got_g, Wc):
        scale = max(float(w.grad.abs().max()), 1e-6)
        np.testing.assert_allclose(g.cpu().numpy(), w.grad.numpy(), rtol=2e-4, atol=3e-5 * scale)
    # ---- apply: global-norm clip + row-sparse Adagrad == the dense rule on the touched rows; nothing else moves
    total = sum(float((g.double() ** 2).sum()) for g in got_g)
    coef = min(1.0, max_norm / (total ** 0.5 + 1e-6))
    uniqs = step._pending[0][1]
    row_grads = [g.clone() for g in got_g[:3]]
    small_grads = [g.clone() for g in got_g[3:]]
    step.apply(replicated=small)
    for name, t, uniq, g in (('U', Ut, uniqs[0], row_grads[0]), ('I', It, uniqs[1], row_grads[1]), ('E', Et, uniqs[2], row_grads[2])):
        ok = uniq >= 0
        ids, gg = uniq[ok], g[ok] * coef
        want_rows = before[name][ids] - lr * gg / (gg.abs() + eps)             # first step: sqrt(state) = |g'|
        torch.testing.assert_close(t.weight.data[ids], want_rows, rtol=2e-5, atol=2e-6)
        touched = torch.zeros(t.weight.shape[0], dtype=torch.bool, device=DEV)
        touched[ids] = True
        assert torch.equal(t.weight.data[~touched], before[name][~touched])     # untouched rows bit-identical
        assert torch.equal(t.state[~touched], torch.zeros_like(t.state[~touched]))
    for p, p0, g in zip(small, small_before, small_grads):
        gg = g * coef
        torch.testing.assert_close(p.data, p0 - lr * gg / (gg.abs() + eps), rtol=2e-5, atol=2e-6)


def _two_rank_worker(rank, world, port, kind):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share this box's GPU (RCCL refuses that)
    try:
        from jTransUP import parallel
        from jTransUP.hip import ops
        dev = torch.device(DEV)
        nu, ni, ne, d, b = 5000, 700, 3000, 256, 8192                    # b >= the segment-route threshold per rank
        gen = torch.Generator().manual_seed(11)
        full = {k: torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1) for k, n in (('U', nu), ('I', ni), ('E', ne))}
        small0 = [torch.nn.functional.normalize(torch.randn(P, d, generator=gen), dim=1) for _ in range(4)]
        i2e = torch.randint(0, ne, (ni,), generator=gen)
        batches = [[(torch.randint(0, nu, (b,), generator=gen), torch.randint(0, ni, (b,), generator=gen),
                     torch.randint(0, ni, (b,), generator=gen)) for _ in range(world)] for _ in range(2)]
        # Adagrad's first steps divide by |g| + eps: with the default eps = 1e-10 an element whose gradient is ~1e-10 turns fp32
        # rounding noise into an O(lr) difference, so the comparison uses eps = 1e-4 (well-conditioned, same code path); plain
        # SGD (linear in g) is checked with the default settings
        lr, max_norm = (0.05, 0.5) if kind == 'adagrad' else (40.0, 0.5)
        eps = 1e-4
        # sharded run
        mk = lambda key, n: parallel.ShardedTable(n, d, rank=rank, world=world, device=dev, init=lambda g: full[key][g].to(dev))
        Ut, It, Et = mk('U', nu), mk('I', ni), mk('E', ne)
        small = [torch.nn.Parameter(t.clone().to(dev)) for t in small0]
        st = parallel.ShardedStep(kind, lr=lr, max_norm=max_norm, eps=eps)
        i2e_d = i2e.to(dev)
        for step in batches:
            u, pi, ni_ = (x.to(dev) for x in step[rank])
            items = torch.cat([pi, ni_])
            (u_rows, u_at), (i_rows, i_at), (e_rows, e_at) = st.lookup_many([(Ut, u), (It, items), (Et, i2e_d[items])])
            i2e_c = torch.zeros(i_rows.shape[0], dtype=torch.int32, device=dev)
            i2e_c[i_at] = e_at.to(torch.int32)
            s = ops.score_ktup(u_rows, i_rows, e_rows, *small, i2e_c, torch.cat([u_at, u_at]), i_at, False, ent_pad=-1)
            (torch.nn.functional.softplus(s[:b] - s[b:]).sum() / (world * b)).backward()
            st.apply(replicated=small)
        # dense single-process reference on the concatenated batches: oracle scorer + clip_grad_norm_ + torch.optim.Adagrad (wd 0)
        Wd = [torch.nn.Parameter(full[k].clone()) for k in ('U', 'I', 'E')] + [torch.nn.Parameter(t.clone()) for t in small0]
        opt = torch.optim.Adagrad(Wd, lr=lr, eps=eps) if kind == 'adagrad' else torch.optim.SGD(Wd, lr=lr)
        for step in batches:
            opt.zero_grad()
            u = torch.cat([x[0] for x in step]); pi = torch.cat([x[1] for x in step]); ni_ = torch.cat([x[2] for x in step])
            pos = O.score_ktup_rec(*Wd, i2e, u, pi, False); neg = O.score_ktup_rec(*Wd, i2e, u, ni_, False)
            (torch.nn.functional.softplus(pos - neg).sum() / (world * b)).backward()
            torch.nn.utils.clip_grad_norm_(Wd, max_norm)
            opt.step()
        for t, w, n in ((Ut, Wd[0], nu), (It, Wd[1], ni), (Et, Wd[2], ne)):
            torch.testing.assert_close(t.weight.data.cpu(), w.data[torch.arange(rank, n, world)], rtol=1e-4, atol=2e-5)
        for p, w in zip(small, Wd[3:]):
            torch.testing.assert_close(p.data.cpu(), w.data, rtol=1e-4, atol=2e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['sgd', 'adagrad'])
def test_config5_two_ranks_share_the_gpu_with_the_ktup_scorer(kind):
    """Row-sharded tables (row % 2), the d = 256 KTUP scorer on the compact tables, combined id / row / gradient exchanges,
    global clip and row-sparse Adagrad on two ranks == one dense process on the concatenated batch."""
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, kind), nprocs=2, join=True)
