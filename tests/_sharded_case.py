"""Shared body of the row-sharded training-step tests (config 5): ShardedStep over two ShardedTables + one replicated table
against a single-process dense run (autograd + clip_grad_norm_ + torch.optim with weight_decay 0) on the concatenated batch.
Used on CPU with torch stand-ins for the row kernels (gloo, world 2) and on the GPU with the HIP kernels."""
import torch
import torch.distributed as dist

NU, NI, D, NP, B = 53, 41, 8, 4, 30


class TorchRowOps(object):
    """Test doubles of parallel.RowOps for runs without a GPU (the product's defaults are the HIP kernels only)."""

    @staticmethod
    def dedupe(ids):                                   # (uniq padded with -1 to len(ids), inverse); negative ids = padding everywhere
        uniq, inverse = torch.unique(ids, return_inverse=True)
        pad = torch.full((ids.numel() - uniq.numel(),), -1, dtype=ids.dtype, device=ids.device)
        return torch.cat([uniq, pad]), inverse

    @staticmethod
    def pack(table, local_ids):
        ok = local_ids >= 0
        out = table.index_select(0, local_ids.clamp(min=0))
        out[~ok] = 0
        return out

    @staticmethod
    def unpack_add(rows, local_ids, gtable):
        ok = local_ids >= 0
        return gtable.index_add_(0, local_ids[ok], rows[ok])

    @staticmethod
    def sumsq(tensors):
        out = torch.zeros(1, dtype=torch.float64, device=tensors[0].device if tensors else 'cpu')
        for t in tensors:
            out += (t.double() ** 2).sum()
        return out

    @staticmethod
    def sparse_step(kind, table, state, ids, grows, lr, eps, sumsq, max_norm):
        coef = 1.0
        if sumsq is not None and max_norm > 0:
            coef = min(1.0, max_norm / (float(sumsq.sqrt()) + 1e-6))
        ok = ids >= 0
        ids, g = ids[ok], (grows * coef)[ok]
        if kind == 'adagrad':
            state[ids] += g * g
            table[ids] -= lr * g / (state[ids].sqrt() + eps)
        else:
            table[ids] -= lr * g
        return table


def init_rows(g, d=D, salt=0):
    g = g.to(torch.float32)
    return torch.sin(g[:, None] * (0.37 + salt) + torch.arange(d, dtype=torch.float32)[None, :] * 0.11) * 0.5


def batches(world, steps):
    gen = torch.Generator().manual_seed(77)
    return [[(torch.randint(0, NU, (B,), generator=gen), torch.randint(0, NI, (B,), generator=gen),
              torch.randint(0, NI, (B,), generator=gen)) for _ in range(world)] for _ in range(steps)]


def toy_loss(u_rows, pi_rows, ni_rows, pref, u):
    """A TUP-shaped margin: translation picked by user id, squared distance, softplus of the pos - neg gap, batch SUM."""
    p = pref[u % NP]
    pos = ((u_rows + p - pi_rows) ** 2).sum(1)
    neg = ((u_rows + p - ni_rows) ** 2).sum(1)
    return torch.nn.functional.softplus(pos - neg).sum()


def dense_reference(kind, lr, max_norm, world, steps, device):
    U = torch.nn.Parameter(init_rows(torch.arange(NU)).to(device))
    I = torch.nn.Parameter(init_rows(torch.arange(NI), salt=1).to(device))
    P = torch.nn.Parameter(init_rows(torch.arange(NP), salt=2).to(device))
    opt = (torch.optim.Adagrad if kind == 'adagrad' else torch.optim.SGD)([U, I, P], lr=lr)
    for step in batches(world, steps):
        opt.zero_grad()
        u = torch.cat([b[0] for b in step]).to(device); pi = torch.cat([b[1] for b in step]).to(device)
        ni = torch.cat([b[2] for b in step]).to(device)
        loss = toy_loss(U[u], I[pi], I[ni], P, u) / (world * B)
        loss.backward()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_([U, I, P], max_norm)
        opt.step()
    return U.data, I.data, P.data


def sharded_run(kind, lr, max_norm, steps, device, ops, rank, world, group=None, many=False):
    """-> this rank's (U shard, I shard, P copy) after `steps` ShardedStep steps on its own batches."""
    from jTransUP.parallel import ShardedStep, ShardedTable
    mk = lambda n, salt: ShardedTable(n, D, rank=rank, world=world, group=group, device=device,
                                      init=lambda g: init_rows(g, salt=salt).to(device), pack=ops.pack, unpack_add=ops.unpack_add)
    Ut, It = mk(NU, 0), mk(NI, 1)
    P = torch.nn.Parameter(init_rows(torch.arange(NP), salt=2).to(device))
    st = ShardedStep(kind, lr, max_norm=max_norm, group=group, ops=ops)
    for step in batches(world, steps):
        u, pi, ni = (x.to(device) for x in step[rank])
        if many:                                                              # both tables through ONE id and ONE row exchange
            (u_rows, u_at), (i_rows, i_at) = st.lookup_many([(Ut, u), (It, torch.cat([pi, ni]))])
        else:
            u_rows, u_at = st.lookup(Ut, u)
            i_rows, i_at = st.lookup(It, torch.cat([pi, ni]))                 # one lookup per table and step
        loss = toy_loss(u_rows[u_at], i_rows[i_at[:B]], i_rows[i_at[B:]], P, u) / (world * B)
        loss.backward()
        st.apply(replicated=[P])
    return Ut.weight.data, It.weight.data, P.data


def check_against_dense(kind, lr, max_norm, steps, device, ops, rank, world, group=None, rtol=2e-5, atol=2e-6, many=False):
    Us, Is, Ps = sharded_run(kind, lr, max_norm, steps, device, ops, rank, world, group, many=many)
    Ud, Id, Pd = dense_reference(kind, lr, max_norm, world, steps, device)
    torch.testing.assert_close(Us, Ud[torch.arange(rank, NU, world, device=Ud.device)], rtol=rtol, atol=atol)
    torch.testing.assert_close(Is, Id[torch.arange(rank, NI, world, device=Id.device)], rtol=rtol, atol=atol)
    torch.testing.assert_close(Ps, Pd, rtol=rtol, atol=atol)
    moved = (Us - init_rows(torch.arange(rank, NU, world)).to(device)).abs().sum(1) > 0
    assert 0 < int(moved.sum()) <= Us.shape[0]                              # some rows moved; untouched ones are bit-identical
    if world > 1:
        copies = [torch.empty_like(Ps.cpu()) for _ in range(world)]
        dist.all_gather(copies, Ps.cpu(), group=group)
        assert all(torch.equal(copies[0], c) for c in copies)                # replicated table stays identical
