"""Multi-GPU execution of the scoring path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU tensors for the world-size-2 tests).  The reference is single-device: everything here is new.

  * ReplicaGradSync  -- config 4: every GPU holds all tables (9.7 MB at ml1m, d=100) and scores its slice of the batch;
                        ONE all-reduce of the flattened dense gradient per step, then the global-norm clip and the dense
                        optimizer run identically on every replica (clip_grad_norm is a function of the reduced gradient).
  * ShardedTable     -- config 5: a table too big to replicate is partitioned by `row % world`; a batch lookup is
                        ids -> all-to-all -> owner-side row pack -> all-to-all of rows, and the row gradients travel the
                        reverse route into the owner's shard gradient.  Duplicate ids are sent once.
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
def _default_pack(table, local_ids):
    if table.is_cuda:
        from jTransUP.hip import ops
        return ops.pack_rows(table, local_ids)
    return table.index_select(0, local_ids)          # CPU tensors only occur in the gloo tests


def _default_unpack_add(rows, local_ids, gtable):
    if gtable.is_cuda:
        from jTransUP.hip import ops
        return ops.unpack_rows_add(rows.contiguous(), local_ids, gtable)
    return gtable.index_add_(0, local_ids, rows)


class _Plan(object):
    """Routing of one batch of (unique) global row ids: who owns what, in which order it is sent and received."""

    def __init__(self, uniq, world, group):
        owner = uniq % world
        order = torch.argsort(owner, stable=True)
        self.order = order
        self.send_local = (uniq // world)[order].contiguous()                 # owner-local row numbers, grouped by owner
        self.send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(self.send_counts)
        dist.all_to_all_single(recv_counts, self.send_counts, group=group)
        self.send_counts_l = self.send_counts.tolist()
        self.recv_counts_l = recv_counts.tolist()
        self.recv_local = torch.empty(int(sum(self.recv_counts_l)), dtype=uniq.dtype, device=uniq.device)
        dist.all_to_all_single(self.recv_local, self.send_local, output_split_sizes=self.recv_counts_l,
                               input_split_sizes=self.send_counts_l, group=group)


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shard, uniq, table):
        world, group = table.world, table.group
        plan = _Plan(uniq, world, group)
        packed = table.pack(shard, plan.recv_local)                              # rows my peers asked me for
        rows_sorted = torch.empty(uniq.numel(), shard.shape[1], dtype=shard.dtype, device=shard.device)
        dist.all_to_all_single(rows_sorted, packed, output_split_sizes=plan.send_counts_l,
                               input_split_sizes=plan.recv_counts_l, group=group)
        rows = torch.empty_like(rows_sorted)
        rows[plan.order] = rows_sorted                                           # back to the order of `uniq`
        ctx.plan, ctx.table, ctx.shape = plan, table, shard.shape
        return rows

    @staticmethod
    def backward(ctx, grows):
        plan, table = ctx.plan, ctx.table
        send = grows[plan.order].contiguous()
        recv = torch.empty(plan.recv_local.numel(), grows.shape[1], dtype=grows.dtype, device=grows.device)
        dist.all_to_all_single(recv, send, output_split_sizes=plan.recv_counts_l, input_split_sizes=plan.send_counts_l,
                               group=table.group)
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
        self.pack = pack or _default_pack
        self.unpack_add = unpack_add or _default_unpack_add

    def lookup(self, ids):
        uniq, inverse = torch.unique(ids, return_inverse=True)
        if self.world == 1:
            return self.weight.index_select(0, uniq) if not self.weight.is_cuda else _LocalGather.apply(self.weight, uniq, self), inverse
        return _ShardedLookup.apply(self.weight, uniq, self), inverse


class _LocalGather(torch.autograd.Function):
    """world == 1 on a GPU: the same pack / unpack kernels without the exchange."""

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
