"""Multi-GPU execution of the scoring path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU tensors for the world-size-2 tests).  The reference is single-device: everything here is new.

  * ReplicaGradSync  -- config 4: every GPU holds all tables (9.7 MB at ml1m, d=100) and scores its slice of the batch;
                        ONE all-reduce of the flattened dense gradient per step, then the global-norm clip and the dense
                        optimizer run identically on every replica (clip_grad_norm is a function of the reduced gradient).
  * ShardedTable     -- config 5: a table too big to replicate is partitioned by `row % world`; a batch lookup is
                        ids -> all-to-all -> owner-side row pack -> all-to-all of rows, and the row gradients travel the
                        reverse route into the owner's shard gradient.  Duplicate ids are sent once.
  * ShardedStep      -- config 5's whole step without a shard-sized gradient: compact row gradients back to the owners, ONE
                        scalar all-reduce for the global-norm clip, row-sparse SGD / Adagrad on the touched rows.
  * merge_topk       -- evaluation with the candidate catalogue sharded across GPUs: local filtered top-n per shard, then
                        an all-gather of (score, id) pairs and a merge under the same (score, id) order.

xGMI is point-to-point (7 links per GPU): the all-to-all spreads each GPU's traffic over all links at once, whereas a ring
all-reduce is bound by one link -- which is why the big tables are sharded + all-to-all'ed and only the small replicated
tables' gradients are all-reduced.
"""
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the job torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op for a single process."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 or dist.is_initialized():
        return dist.get_rank() if dist.is_initialized() else 0, world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # KTUP_DIST_BACKEND=gloo is a test hook: several ranks can then share the one GPU of a single-GPU box
    backend = backend or os.environ.get('KTUP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    dist.init_process_group(backend, rank=int(os.environ['RANK']), world_size=world)
    return dist.get_rank(), world


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


# ------------------------------------------------------------------------------------------ config 4: replicas
class ReplicaGradSync(object):
    """Gradient exchange for data-parallel replicas.

    Loss terms scale differently when the global batch is split over G ranks:
      'mean'        a mean over the batch (bprLoss)            -> local term / G, gradients summed
      'sum'         a sum over the batch (marginLoss, normLoss on gathered rows, orthogonalLoss on gathered rows)
                                                               -> local term as is, gradients summed
      'replicated'  a term every rank computes identically on whole tables (orthogonalLoss(pref, pref_norm))
                                                               -> local term / G, gradients summed
    so that the reduced gradient equals the single-process gradient of the loss on the concatenated batch."""

    def __init__(self, params, group=None):
        self.params = [p for p in params]
        self.group = group
        self.world = _world(group)

    def scale(self, term, kind):
        if kind in ('mean', 'replicated'):
            return term / self.world
        if kind == 'sum':
            return term
        raise ValueError(kind)

    @torch.no_grad()
    def all_reduce_grads(self):
        """One bucket: flatten every table's dense gradient, all-reduce(sum), scatter back."""
        if self.world == 1:
            return
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad.reshape(-1))
        flat = torch.cat(grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n

    def broadcast_params(self, src=0):
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)


# ------------------------------------------------------------------------------------------ config 5: row-sharded tables
class RowOps(object):
    """The device halves of the exchange.  The defaults are the HIP kernels and nothing else (CPU tensors raise): the
    world-size-2 gloo tests that run without a GPU pass their own torch stand-ins."""

    @staticmethod
    def pack(table, local_ids):                                   # out[k] = table[ids[k]]
        from jTransUP.hip import ops
        return ops.pack_rows(table, local_ids)

    @staticmethod
    def unpack_add(rows, local_ids, gtable):                      # gtable[ids[k]] += rows[k]
        from jTransUP.hip import ops
        return ops.unpack_rows_add(rows.contiguous(), local_ids, gtable)

    @staticmethod
    def sumsq(tensors):                                           # -> one device double
        from jTransUP.hip import ops
        return ops.grad_sumsq([t.contiguous() for t in tensors])

    @staticmethod
    def sparse_step(kind, table, state, ids, grows, lr, eps, sumsq, max_norm):
        from jTransUP.hip import ops
        return ops.sparse_step(kind, table, state, ids, grows.contiguous(), lr, eps, sumsq, max_norm)


def _a2a(out, inp, out_splits, in_splits, group):
    """all_to_all_single.  Under the gloo test hook (several ranks sharing one GPU: RCCL refuses that) device tensors are
    staged through the host, gloo having no device all-to-all."""
    if inp.is_cuda and dist.get_backend(group) == 'gloo':
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


class _Plan(object):
    """Routing of one batch of (unique) global row ids: who owns what, in which order it is sent and received."""

    def __init__(self, uniq, world, group):
        owner = uniq % world
        order = torch.argsort(owner, stable=True)
        self.order = order
        self.send_local = (uniq // world)[order].contiguous()                 # owner-local row numbers, grouped by owner
        self.send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(self.send_counts)
        _a2a(recv_counts, self.send_counts, None, None, group)
        self.send_counts_l = self.send_counts.tolist()
        self.recv_counts_l = recv_counts.tolist()
        self.recv_local = torch.empty(int(sum(self.recv_counts_l)), dtype=uniq.dtype, device=uniq.device)
        _a2a(self.recv_local, self.send_local, self.recv_counts_l, self.send_counts_l, group)


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shard, uniq, table):
        world, group = table.world, table.group
        plan = _Plan(uniq, world, group)
        packed = table.pack(shard, plan.recv_local)                              # rows my peers asked me for
        rows_sorted = torch.empty(uniq.numel(), shard.shape[1], dtype=shard.dtype, device=shard.device)
        _a2a(rows_sorted, packed, plan.send_counts_l, plan.recv_counts_l, group)
        rows = torch.empty_like(rows_sorted)
        rows[plan.order] = rows_sorted                                           # back to the order of `uniq`
        ctx.plan, ctx.table, ctx.shape = plan, table, shard.shape
        return rows

    @staticmethod
    def backward(ctx, grows):
        plan, table = ctx.plan, ctx.table
        send = grows[plan.order].contiguous()
        recv = torch.empty(plan.recv_local.numel(), grows.shape[1], dtype=grows.dtype, device=grows.device)
        _a2a(recv, send, plan.recv_counts_l, plan.send_counts_l, table.group)
        gshard = torch.zeros(ctx.shape, dtype=grows.dtype, device=grows.device)
        table.unpack_add(recv, plan.recv_local, gshard)
        return gshard, None, None


class ShardedTable(torch.nn.Module):
    """Rows {g : g % world == rank} of a (rows x d) table; `lookup(ids)` returns a compact table of the batch's distinct
    rows plus the ids to address it with, so the scoring kernels run unchanged on (compact, compact_ids)."""

    def __init__(self, total_rows, d, rank=None, world=None, group=None, init=None, pack=None, unpack_add=None, device=None):
        super(ShardedTable, self).__init__()
        self.group = group
        self.world = world if world is not None else _world(group)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.total_rows, self.d = total_rows, d
        local_rows = (total_rows - self.rank + self.world - 1) // self.world
        w = torch.zeros(local_rows, d, dtype=torch.float32, device=device)
        if init is not None:                        # init(global_row_ids) -> (n x d) values, for reproducible tests / loading
            w.copy_(init(torch.arange(self.rank, total_rows, self.world)))
        self.weight = torch.nn.Parameter(w)
        self.pack = pack or RowOps.pack
        self.unpack_add = unpack_add or RowOps.unpack_add
        self.state = None                           # Adagrad accumulator of the shard (ShardedStep creates it)

    def lookup(self, ids):
        uniq, inverse = torch.unique(ids, return_inverse=True)
        if self.world == 1:
            return _LocalGather.apply(self.weight, uniq, self), inverse
        return _ShardedLookup.apply(self.weight, uniq, self), inverse


class _LocalGather(torch.autograd.Function):
    """world == 1: the same pack / unpack kernels without the exchange."""

    @staticmethod
    def forward(ctx, shard, uniq, table):
        ctx.save_for_backward(uniq); ctx.table, ctx.shape = table, shard.shape
        return table.pack(shard, uniq)

    @staticmethod
    def backward(ctx, grows):
        (uniq,) = ctx.saved_tensors
        g = torch.zeros(ctx.shape, dtype=grows.dtype, device=grows.device)
        ctx.table.unpack_add(grows, uniq, g)
        return g, None, None


class ShardedStep(object):
    """One training step over row-sharded tables that never materialises a shard-sized gradient:

        step = ShardedStep('adagrad', lr, max_norm=5.0)
        u_rows, u_at = step.lookup(user_table, u_ids)          # compact rows of the batch's distinct ids + positions
        ...score on (compact rows, positions), loss scaled for the global batch..., loss.backward()
        step.apply(replicated=[pref, pref_norm, rel, norm])

    lookup : ids -> all-to-all -> owner-side pack -> all-to-all of rows; `rows` is a leaf collecting the dense (compact)
             row gradients of this rank.
    apply  : row gradients -> all-to-all back to the owners, duplicates from different ranks combined (atomics into a
             compact buffer); gradients of the small replicated tables all-reduced; ONE scalar all-reduce gives the job-wide
             gradient norm for the clip; then the owner updates exactly the touched rows (K: ktup_shard_sparse_step) and
             every rank applies the same rule to its copy of the replicated tables.
    Exact w.r.t. the reference's dense step for plain SGD / Adagrad with l2_lambda = 0 (rows with zero gradient do not
    move); weight decay or momentum would touch every row of every shard each step and are refused."""

    def __init__(self, kind, lr, eps=1e-10, max_norm=0.0, group=None, ops=RowOps):
        if kind not in ('sgd', 'adagrad'):
            raise ValueError('row-sparse steps exist for plain SGD and Adagrad only')
        self.kind, self.lr, self.eps, self.max_norm = kind, float(lr), float(eps), float(max_norm)
        self.group, self.world, self.ops = group, _world(group), ops
        self._pending = []
        self._rep_state = {}

    def lookup(self, table, ids):
        if any(t is table for t, _, _, _ in self._pending):
            raise ValueError('one lookup per table and step: concatenate the ids (duplicates are sent once anyway)')
        uniq, inverse = torch.unique(ids, return_inverse=True)
        with torch.no_grad():
            if self.world == 1:
                plan, rows = None, table.pack(table.weight.data, uniq)
            else:
                plan = _Plan(uniq, self.world, self.group)
                packed = table.pack(table.weight.data, plan.recv_local)
                rows_sorted = torch.empty(uniq.numel(), table.d, dtype=packed.dtype, device=packed.device)
                _a2a(rows_sorted, packed, plan.send_counts_l, plan.recv_counts_l, self.group)
                rows = torch.empty_like(rows_sorted)
                rows[plan.order] = rows_sorted
        rows.requires_grad_(True)
        self._pending.append((table, uniq, plan, rows))
        return rows, inverse

    def _state_of(self, key, like):
        if self.kind != 'adagrad':
            return None
        st = self._rep_state.get(key)
        if st is None:
            st = self._rep_state[key] = torch.zeros_like(like)
        return st

    @torch.no_grad()
    def apply(self, replicated=()):
        ops = self.ops
        work = []
        for table, uniq, plan, rows in self._pending:
            g = rows.grad if rows.grad is not None else torch.zeros_like(rows)
            if plan is None:
                ids_local, gsum = uniq, g
            else:
                recv = torch.empty(plan.recv_local.numel(), table.d, dtype=g.dtype, device=g.device)
                _a2a(recv, g[plan.order].contiguous(), plan.recv_counts_l, plan.send_counts_l, self.group)
                ids_local, at = torch.unique(plan.recv_local, return_inverse=True)     # the same row asked for by several ranks
                gsum = torch.zeros(ids_local.numel(), table.d, dtype=g.dtype, device=g.device)
                if recv.shape[0]:
                    table.unpack_add(recv, at, gsum)
            work.append((table, ids_local, gsum))
        self._pending = []
        reps = [p for p in replicated if p.grad is not None]
        if self.world > 1:
            for p in reps:                                                              # small tables: plain all-reduce
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
        sumsq = None
        if self.max_norm > 0:
            sumsq = ops.sumsq([g for _, _, g in work if g.numel()])                     # every touched row once, at its owner
            if self.world > 1:
                dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=self.group)
            if reps:
                sumsq += ops.sumsq([p.grad for p in reps])                              # identical on every rank: counted once
        for table, ids_local, gsum in work:
            if self.kind == 'adagrad' and table.state is None:
                table.state = torch.zeros_like(table.weight.data)
            if ids_local.numel():
                ops.sparse_step(self.kind, table.weight.data, table.state, ids_local, gsum, self.lr, self.eps, sumsq, self.max_norm)
        for p in reps:
            every = torch.arange(p.shape[0], device=p.device)
            ops.sparse_step(self.kind, p.data, self._state_of(id(p), p.data), every, p.grad, self.lr, self.eps, sumsq, self.max_norm)
            p.grad = None
        return sumsq


# ------------------------------------------------------------------------------------------ sharded-candidate evaluation
def shard_bounds(n_candidates, rank, world):
    """Contiguous candidate block of a rank (first `rem` ranks get one extra)."""
    base, rem = divmod(n_candidates, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def merge_topk(local_ids, local_scores, topn, descending=False, group=None):
    """Merge per-shard filtered top-n lists (global candidate ids, -1 padded) into the global top-n on every rank.
    Order: ascending score (descending=True negates), ties -> lower id: the same total order the ranking kernel uses."""
    world = _world(group)
    if world > 1:
        ids_all = [torch.empty_like(local_ids) for _ in range(world)]
        sc_all = [torch.empty_like(local_scores) for _ in range(world)]
        dist.all_gather(ids_all, local_ids.contiguous(), group=group)
        dist.all_gather(sc_all, local_scores.contiguous(), group=group)
        ids, sc = torch.cat(ids_all, 1), torch.cat(sc_all, 1)
    else:
        ids, sc = local_ids, local_scores
    key = -sc if descending else sc.clone()
    key = key + 0.0                                         # -0.0 -> +0.0 like the kernel's key
    key[ids < 0] = float('inf')
    big = ids.to(torch.int64).clone()
    big[ids < 0] = torch.iinfo(torch.int64).max
    o1 = torch.argsort(big, dim=1, stable=True)             # secondary key first, then a stable sort on the primary
    key1 = torch.gather(key, 1, o1)
    o2 = torch.argsort(key1, dim=1, stable=True)
    order = torch.gather(o1, 1, o2)[:, :topn]
    return torch.gather(ids, 1, order), torch.gather(sc, 1, order)
