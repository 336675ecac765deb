"""Multi-GPU execution of the scoring path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU tensors for the world-size-2 tests).  The reference is single-device: everything here is new.

  * ReplicaGradSync  -- config 4: every GPU holds all tables (9.7 MB at ml1m, d=100) and scores its slice of the batch;
                        ONE all-reduce of the flattened dense gradient per step, then the global-norm clip and the dense
                        optimizer run identically on every replica (clip_grad_norm is a function of the reduced gradient).
  * ShardedTable     -- config 5: a table too big to replicate is partitioned by `row % world`; a batch lookup is
                        ids -> all-to-all -> owner-side row pack -> all-to-all of rows, and the row gradients travel the
                        reverse route into the owner's shard gradient.  Duplicate ids are sent once.
  * ShardedStep      -- config 5's whole step without a shard-sized gradient: device-side dedupe (no host sync on one rank,
                        one per step on several), all tables' lookups in ONE id and ONE row all-to-all, compact row gradients
                        back to the owners in ONE all-to-all, small tables' gradients + the global norm in ONE all-reduce,
                        row-sparse SGD / Adagrad on the touched rows.
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


_FORCE = [False]


def force_collectives(on=True):
    """Test / measurement hook: with an initialised process group of ONE rank, still issue every collective of the N > 1 paths
    (all-reduce of the gradient bucket, the all-to-alls of the sharded lookups, the all-gathers of the merged rankings) instead of
    short-cutting them -- a 1-GPU box then runs RCCL's device-tensor branches and their stream ordering, degenerately but for
    real.  Returns the previous setting."""
    old = _FORCE[0]
    _FORCE[0] = bool(on)
    return old


def _exchanging(group=None):
    """True when the collectives of the N > 1 paths must be issued: several ranks, or one rank under force_collectives()."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE[0])


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

    def __init__(self, params, group=None, extra=0):
        """All gradients become views into ONE persistent flat bucket (every view 16-byte aligned for the float4 kernels),
        followed by `extra` fp32 scalars that ride along in the same all-reduce (loss terms).  Gradients are zero-filled,
        like the reference's torch-0.3 zero_grad."""
        self.params = [p for p in params]
        self.group = group
        self.world = _world(group)
        sizes = [p.numel() for p in self.params]
        pad = [(-n) % 4 for n in sizes]
        dev = self.params[0].device if self.params else None
        self.flat = torch.zeros(sum(sizes) + sum(pad) + int(extra), dtype=torch.float32, device=dev)
        off = 0
        for p, n, q in zip(self.params, sizes, pad):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n + q
        self.extra = self.flat[off:off + int(extra)]

    def scale(self, term, kind):
        if kind in ('mean', 'replicated'):
            return term / self.world
        if kind == 'sum':
            return term
        raise ValueError(kind)

    @torch.no_grad()
    def all_reduce_grads(self):
        """ONE all-reduce(sum) of the flat bucket: no gather / scatter copies, the gradients are views into it."""
        if not _exchanging(self.group):
            return
        for p in self.params:                 # autograd may have replaced a view by its own tensor (set_to_none, first backward)
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                self._readopt()
                break
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def _readopt(self):
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
            off += n + ((-n) % 4)

    def broadcast_params(self, src=0):
        if _exchanging(self.group):
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)


# ------------------------------------------------------------------------------------------ config 5: row-sharded tables
class RowOps(object):
    """The device halves of the exchange.  The defaults are the HIP kernels and nothing else (CPU tensors raise): the
    world-size-2 gloo tests that run without a GPU pass their own torch stand-ins.  Negative ids are padding everywhere
    (pack -> zero row, unpack / sparse step -> skipped): id lists keep a fixed capacity so that the number of distinct ids
    of a batch never has to reach the host."""

    @staticmethod
    def dedupe(ids):                                              # -> (uniq padded with -1 to len(ids), inverse)
        from jTransUP.hip import ops
        return ops.dedupe(ids)

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


class _Route(object):
    """Routing of one step's lookups for ALL its tables at once: who owns which distinct id, in which order ids and rows travel.

    Per table t: uniq_t (distinct global ids, -1 padded), order_t = stable argsort by owner (padding last).  The buffers of the
    three exchanges are laid out peer-major, table-minor, so each is ONE all-to-all:
        counts  (world x T int64)  ->  ids (owner-local row numbers)  ->  rows (d floats each; the gradients travel back the same way)
    and the host reads the two count matrices ONCE per step (the only sync of the step: all_to_all_single wants host split
    sizes; round 1 did two .tolist() per table and direction)."""

    def __init__(self, uniqs, world, group):
        self.world, self.group, self.T = world, group, len(uniqs)
        dev = uniqs[0].device
        self.order, send_local, cnt = [], [], []
        for uniq in uniqs:
            key = torch.where(uniq < 0, torch.full_like(uniq, world), uniq % world)
            order = torch.argsort(key, stable=True)
            self.order.append(order)
            send_local.append(torch.div(uniq, world, rounding_mode='floor')[order])
            cnt.append(torch.bincount(key, minlength=world + 1)[:world])
        send_cnt = torch.stack(cnt, 1).contiguous()                               # (world, T): what I ask each peer for, per table
        recv_cnt = torch.empty_like(send_cnt)
        _a2a(recv_cnt, send_cnt, None, None, group)
        both = torch.stack([send_cnt, recv_cnt]).tolist()                         # the step's one host sync
        self.send_cnt, self.recv_cnt = both[0], both[1]                           # [peer][table]
        self.send_tot = [sum(r) for r in self.send_cnt]
        self.recv_tot = [sum(r) for r in self.recv_cnt]
        self.n_valid = [sum(self.send_cnt[p][t] for p in range(world)) for t in range(self.T)]
        # ids: peer-major, table-minor
        parts, offs = [], [[0] * self.T for _ in range(world)]
        run = [0] * self.T
        for p in range(world):
            for t in range(self.T):
                offs[p][t] = run[t]
                parts.append(send_local[t][run[t]:run[t] + self.send_cnt[p][t]])
                run[t] += self.send_cnt[p][t]
        self.send_off = offs
        send_ids = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=dev)
        self.recv_ids = torch.empty(sum(self.recv_tot), dtype=torch.int64, device=dev)
        _a2a(self.recv_ids, send_ids, self.recv_tot, self.send_tot, group)
        # the (peer, table) blocks of what I received, regrouped per table for the owner-side kernels
        self.recv_blocks = [[None] * self.T for _ in range(world)]
        at = 0
        for p in range(world):
            for t in range(self.T):
                self.recv_blocks[p][t] = (at, at + self.recv_cnt[p][t])
                at += self.recv_cnt[p][t]

    def owner_ids(self, t):
        """Owner-local row numbers my peers asked me for in table t, peer-major."""
        return torch.cat([self.recv_ids[a:b] for a, b in (self.recv_blocks[p][t] for p in range(self.world))])

    def owner_to_wire(self, per_table):
        """[rows of table t in owner_ids(t) order] -> one peer-major, table-minor buffer."""
        cur = [0] * self.T
        parts = []
        for p in range(self.world):
            for t in range(self.T):
                n = self.recv_cnt[p][t]
                parts.append(per_table[t][cur[t]:cur[t] + n])
                cur[t] += n
        return torch.cat(parts)

    def wire_to_owner(self, buf):
        """Inverse of owner_to_wire for a buffer that arrived over the reverse route."""
        out = [[] for _ in range(self.T)]
        for p in range(self.world):
            for t in range(self.T):
                a, b = self.recv_blocks[p][t]
                out[t].append(buf[a:b])
        return [torch.cat(o) for o in out]

    def requester_to_wire(self, per_table):
        """[tensor ordered like order_t (owner-sorted, valid entries first)] -> one peer-major, table-minor buffer."""
        parts = []
        for p in range(self.world):
            for t in range(self.T):
                a = self.send_off[p][t]
                parts.append(per_table[t][a:a + self.send_cnt[p][t]])
        return torch.cat(parts)

    def wire_to_requester(self, buf):
        out = [[] for _ in range(self.T)]
        at = 0
        for p in range(self.world):
            for t in range(self.T):
                n = self.send_cnt[p][t]
                out[t].append(buf[at:at + n])
                at += n
        return [torch.cat(o) for o in out]


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
        """Autograd form (dense shard gradient): for code that wants `.grad` on the shard.  The training step uses
        ShardedStep.lookup, which never materialises a shard-sized gradient."""
        uniq, inverse = torch.unique(ids, return_inverse=True)
        if not _exchanging(self.group):
            return _LocalGather.apply(self.weight, uniq, self), inverse
        return _ShardedLookup.apply(self.weight, uniq, self), inverse


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shard, uniq, table):
        route = _Route([uniq], table.world, table.group)
        packed = table.pack(shard, route.owner_ids(0))                           # rows my peers asked me for
        rows_sorted = torch.empty(uniq.numel(), shard.shape[1], dtype=shard.dtype, device=shard.device)
        _a2a(rows_sorted, route.owner_to_wire([packed]), route.send_tot, route.recv_tot, table.group)
        rows = torch.empty_like(rows_sorted)
        rows[route.order[0]] = rows_sorted                                       # back to the order of `uniq`
        ctx.route, ctx.table, ctx.shape = route, table, shard.shape
        return rows

    @staticmethod
    def backward(ctx, grows):
        route, table = ctx.route, ctx.table
        send = grows[route.order[0]].contiguous()
        recv = torch.empty(sum(route.recv_tot), grows.shape[1], dtype=grows.dtype, device=grows.device)
        _a2a(recv, send, route.recv_tot, route.send_tot, table.group)
        gshard = torch.zeros(ctx.shape, dtype=grows.dtype, device=grows.device)
        table.unpack_add(route.wire_to_owner(recv)[0], route.owner_ids(0), gshard)
        return gshard, None, None


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
        (u_rows, u_at), (i_rows, i_at), (e_rows, e_at) = step.lookup_many([(U, u_ids), (I, i_ids), (E, e_ids)])
        ...score on (compact rows, positions), loss scaled for the global batch..., loss.backward()
        step.apply(replicated=[pref, pref_norm, rel, norm])

    lookup_many : per table the batch's distinct ids (device-side hash dedupe, fixed capacity, -1 padded: no host sync);
             one rank: owner-side pack and done.  Several ranks: ONE count exchange for all tables + the step's single host
             read, ONE all-to-all of ids, owner-side pack, ONE all-to-all of rows.  `rows` is a leaf collecting the dense
             (compact) row gradients of this rank; padding rows are zero and receive zero gradients.
    apply  : row gradients -> ONE all-to-all back to the owners, duplicates from different ranks combined into a compact
             buffer; the gradients of the small replicated tables and the owners' sum of squared row gradients ride in ONE
             all-reduce (fp64 bucket); then the owner updates exactly the touched rows (ktup_shard_sparse_step) and every
             rank applies the same rule to its copy of the replicated tables.
    Collectives per step: 3 + 2 (round 1: >= 9 + 5, with >= 6 host syncs).  Exact w.r.t. the reference's dense step for plain
    SGD / Adagrad with l2_lambda = 0 (rows with zero gradient do not move); weight decay or momentum would touch every row of
    every shard each step and are refused."""

    def __init__(self, kind, lr, eps=1e-10, max_norm=0.0, group=None, ops=RowOps):
        if kind not in ('sgd', 'adagrad'):
            raise ValueError('row-sparse steps exist for plain SGD and Adagrad only')
        self.kind, self.lr, self.eps, self.max_norm = kind, float(lr), float(eps), float(max_norm)
        self.group, self.world, self.ops = group, _world(group), ops
        self._pending = []          # (tables, uniqs, route, rows) per lookup_many call
        self._rep_state = {}
        self._every = {}

    def lookup(self, table, ids):
        return self.lookup_many([(table, ids)])[0]

    def lookup_many(self, pairs):
        tables = [t for t, _ in pairs]
        for t in tables:
            if any(t is u for entry in self._pending for u in entry[0]) or sum(1 for u in tables if u is t) > 1:
                raise ValueError('one lookup per table and step: concatenate the ids (duplicates are sent once anyway)')
        with torch.no_grad():
            ded = [self.ops.dedupe(ids) for _, ids in pairs]                     # (uniq padded with -1, inverse)
            uniqs = [u for u, _ in ded]
            if not _exchanging(self.group):
                route = None
                rows = [t.pack(t.weight.data, u) for t, u in zip(tables, uniqs)]
            else:
                d = tables[0].d
                if any(t.d != d for t in tables):
                    raise ValueError('tables of one lookup_many share the row width (one row buffer on the wire)')
                route = _Route(uniqs, self.world, self.group)
                packed = [t.pack(t.weight.data, route.owner_ids(k)) for k, t in enumerate(tables)]
                wire = torch.empty(sum(route.send_tot), d, dtype=torch.float32, device=uniqs[0].device)
                _a2a(wire, route.owner_to_wire(packed), route.send_tot, route.recv_tot, self.group)
                rows = []
                for k, got in enumerate(route.wire_to_requester(wire)):
                    full = torch.zeros(uniqs[k].numel(), d, dtype=torch.float32, device=got.device)
                    full[route.order[k][:route.n_valid[k]]] = got                # back to the order of uniq; padding rows stay zero
                    rows.append(full)
        for r in rows:
            r.requires_grad_(True)
        self._pending.append((tables, uniqs, route, rows))
        return [(r, inv) for r, (_, inv) in zip(rows, ded)]

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
        for tables, uniqs, route, rows in self._pending:
            grads = [r.grad if r.grad is not None else torch.zeros_like(r) for r in rows]
            if route is None:
                work += [(t, u, g) for t, u, g in zip(tables, uniqs, grads)]
                continue
            d = tables[0].d
            send = route.requester_to_wire([g[route.order[k]] for k, g in enumerate(grads)])
            recv = torch.empty(sum(route.recv_tot), d, dtype=torch.float32, device=send.device)
            _a2a(recv, send.contiguous(), route.recv_tot, route.send_tot, self.group)
            for k, got in enumerate(route.wire_to_owner(recv)):
                ids_local, at = ops.dedupe(route.owner_ids(k))                   # the same row asked for by several ranks
                gsum = torch.zeros(ids_local.numel(), d, dtype=torch.float32, device=got.device)
                if got.shape[0]:
                    tables[k].unpack_add(got, at, gsum)
                work.append((tables[k], ids_local, gsum))
        self._pending = []
        reps = [p for p in replicated if p.grad is not None]
        sumsq = None
        local = ops.sumsq([g for _, _, g in work if g.numel()]) if self.max_norm > 0 else None   # every touched row once, at its owner
        if _exchanging(self.group) and (reps or local is not None):
            flat = torch.cat([p.grad.reshape(-1).double() for p in reps] + ([local.reshape(1)] if local is not None else []))
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)        # small tables' gradients + the norm, one bucket
            off = 0
            for p in reps:
                p.grad.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            if local is not None:
                local = flat[off:off + 1].clone()
        if self.max_norm > 0:
            sumsq = local
            if reps:
                sumsq = sumsq + ops.sumsq([p.grad for p in reps])                # identical on every rank: counted once
        for table, ids_local, gsum in work:
            if self.kind == 'adagrad' and table.state is None:
                table.state = torch.zeros_like(table.weight.data)
            if ids_local.numel():
                ops.sparse_step(self.kind, table.weight.data, table.state, ids_local, gsum, self.lr, self.eps, sumsq, self.max_norm)
        for p in reps:
            every = self._every.get(p.shape[0])
            if every is None or every.device != p.device:
                every = self._every[p.shape[0]] = torch.arange(p.shape[0], device=p.device)
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
    if _exchanging(group):
        stage = local_ids.is_cuda and dist.get_backend(group) == 'gloo'      # gloo test hook: no device all-gather there
        src_i, src_s = (local_ids.cpu(), local_scores.cpu()) if stage else (local_ids.contiguous(), local_scores.contiguous())
        ids_all = [torch.empty_like(src_i) for _ in range(world)]
        sc_all = [torch.empty_like(src_s) for _ in range(world)]
        dist.all_gather(ids_all, src_i, group=group)
        dist.all_gather(sc_all, src_s, group=group)
        ids, sc = torch.cat(ids_all, 1).to(local_ids.device), torch.cat(sc_all, 1).to(local_ids.device)
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


def _local_topk_hip(local_scores, descending, topn, f_off, f_ids_local):
    from jTransUP.hip import ops
    return ops.topk_filtered(local_scores, descending, topn, f_off, f_ids_local, with_scores=True)


def _local_counts_hip(local_scores, lo, descending, g_off, g_ids, gold_scores, f_off, f_ids, stride=1):
    from jTransUP.hip import ops
    return ops.gold_rank_counts(local_scores, lo, descending, g_off, g_ids, gold_scores, f_off, f_ids, cand_stride=stride)


@torch.no_grad()
def sharded_topk(local_scores, lo, topn, descending, f_off=None, f_ids=None, group=None, local_topk=None, stride=1):
    """Filtered top-n of a catalogue sharded over the ranks (utils/misc.py:213-248 on the union of the shards).
    local_scores: (nq, n_local) scores of candidates lo + stride * j, j in [0, n_local) (stride 1: a contiguous block; stride = world,
    lo = rank: the rows of a table sharded by `row % world`, whose order within the shard is the global id order, so ties fall the
    same way); f_off / f_ids: the filter sets as CSR with GLOBAL candidate ids (ids outside the shard are ignored by the ranking kernel).  Every rank returns the same (nq, topn) global ids
    (-1 padded) and scores: local filtered top-n -> all-gather of (score, id) -> merge under the same (score, id) order."""
    local_topk = local_topk or _local_topk_hip
    if local_scores.shape[1] == 0:
        ids = torch.full((local_scores.shape[0], topn), -1, dtype=torch.int32, device=local_scores.device)
        sc = torch.zeros(local_scores.shape[0], topn, dtype=torch.float32, device=local_scores.device)
    else:
        fl = None
        if f_ids is not None:
            off = f_ids.to(torch.int64) - lo
            fl = (off if stride == 1 else torch.where(off % stride == 0, torch.div(off, stride, rounding_mode='floor'), torch.full_like(off, -1))).to(torch.int32)
        ids, sc = local_topk(local_scores, descending, topn, f_off, fl)
        ids = torch.where(ids >= 0, ids * int(stride) + int(lo), ids)
    return merge_topk(ids, sc, topn, descending=descending, group=group)


@torch.no_grad()
def sharded_gold_ranks(local_scores, lo, descending, g_off, g_ids, g_rows, f_off=None, f_ids=None, group=None, local_counts=None, stride=1):
    """0-based filtered ranks of the gold ids (utils/misc.py:125-146) when every rank holds the scores of one candidate shard
    [lo, lo + n_local).  g_off / g_ids: gold CSR (global ids), g_rows: the query row of every gold entry (int64, len(g_ids));
    -> int32 ranks per gold entry on every rank (-1: the gold is itself filtered).  Two small collectives: the golds' own
    scores from their owners (all-reduce of a vector that is zero elsewhere), then the additive per-shard counts."""
    local_counts = local_counts or _local_counts_hip
    n_local = local_scores.shape[1]
    n = g_rows.numel()
    col = g_ids[:n].to(torch.int64) - lo
    if stride != 1:                                          # a lattice shard: candidate lo + stride * j is local column j
        col = torch.where(col % stride == 0, torch.div(col, stride, rounding_mode='floor'), torch.full_like(col, -1))
    mine = (col >= 0) & (col < n_local)
    gold_scores = torch.zeros(n, dtype=torch.float32, device=local_scores.device)
    if n_local and bool(n):
        picked = local_scores[g_rows[mine], col[mine]]
        gold_scores[mine] = picked
    if _exchanging(group):
        _all_reduce(gold_scores, group)
    counts = (local_counts(local_scores, lo, descending, g_off, g_ids, gold_scores, f_off, f_ids) if stride == 1 else
              local_counts(local_scores, lo, descending, g_off, g_ids, gold_scores, f_off, f_ids, stride=stride))[:n].clone()
    if _exchanging(group):
        _all_reduce(counts, group)
    return torch.where(counts < 0, torch.full_like(counts, -1), counts)


def _all_reduce(t, group):
    """all_reduce(sum); under the gloo test hook device tensors are staged through the host."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
