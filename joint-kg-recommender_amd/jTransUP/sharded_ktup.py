"""BASELINE config 5 as ONE replayable step: KTUP's rec step (knowledgable_recommendation.py:335-344,394-403 over
jTransUP.py:122-143: model(pos), model(neg), bprLoss, backward, clip_grad_norm, optimizer.step) on user / item / entity tables
that are row-sharded by `row % world` (parallel.ShardedTable), with every buffer of a FIXED shape -- no host synchronisation, no
torch op, no autograd in the step; the launches go through the C ABI with pre-bound arguments and are replayed as HIP graphs.
The reference is single-device: everything here is new.

One rank (the whole step is one graph of 9-10 launches):
    route      entries [u | pos ; neg | item2ent[pos ; neg]] of the cursor's batch (a user is ONE entry: its positive and its
               negative pair share the row, and the step kernel hands back the sum of both); distinct ids -> wire rows (owner-major,
               fixed capacity), inverse, item -> entity map on the wire rows, and the counting sort of the entries by wire row
               as a by-product                                                                ktup_shard_route_ktup (5 launches)
    pack       X[w] = table[ids[w]] for all three tables (skipped on one rank when every item has an entity row: the step
               kernel then gathers straight from the shards by global id)                     ktup_shard_pack_wire
    step       forward of [pos ; neg], BPR term, backward; the user-row gradient of example k is row k of GU (B rows), the item-row
               gradient of pair k row k of GV (2B rows), the small tables' gradients accumulate in gA / gC
                                                                                              ktup_train_rec_step_rows
    reduce     Gwire[w] += rows of the entries sorted to w (users | items | entities)         ktup_shard_reduce_rows
    norm       sum of squares of every gradient of the step                                   ktup_optim_gradnorm_acc
    apply      clip + row-sparse SGD / Adagrad on the touched rows of the three shards and on the four small tables; the
               gradient buffers are left zero-filled                                          ktup_shard_apply
Several ranks: the same launches as five graph segments around three all-to-alls with EQUAL splits (ids out, rows back, row
gradients out) and one fp64 all-reduce (small tables' gradients + the sum of squares + the overflow flag); the owner combines the
rows several peers asked for with the same route + reduce (a row asked for by k peers is k entries of one key).

The KG half of the joint schedule (knowledgable_recommendation.py:345-383 over jTransUP.py:144-157; three of every ten steps at
joint_ratio 0.7) is ShardedKgStepper: the same fixed-shape form over the entity shard alone -- route of [ph ; pt ; nh ; nt],
ktup_train_kg_step_rows (TransH forward x 2, marginLoss, orthogonalLoss / normLoss on the gathered rows, backward; entity-row
gradients stored per triple, relation-side gradients accumulated in relation-sorted order), the same reduce -> norm -> apply.
ShardedKtupJoint runs the two steppers on the reference's 10-step cycle over shared tables and optimizer state.

Fixed capacity: a rank may ask one owner for at most cap = ceil(capacity_factor * n / world) + 64 distinct rows of a table per
step (n = batch entries of that table).  Owners are `id % world`, so distinct ids spread like a binomial(n, 1 / world): at
n = 16384, world = 8 the default factor 1.25 is 10 standard deviations out.  A step that does overflow is SKIPPED on every rank
(the flag rides in the all-reduce): no table moves, its loss terms are dropped, and a device counter that no launch clears
(`skipped`) takes note -- `overflowed_steps()` reads it, `check()` raises when it is non-zero, however long ago the step was.
capacity_factor = world can never overflow.
"""
import ctypes
import math

import torch
import torch.distributed as dist

from jTransUP.hip import lib as L

KINDS = {'sgd': 0, 'adagrad': 1, 'adam': 2}
SLOTS = 16          # the sum of squares is accumulated in this many words (one atomic per workgroup, ~20 ns each on one address)


class AdamRule(ctypes.Structure):
    """ktup_adam_t of include/ktup_hip.h: the row-sparse rules that equal the reference's DENSE optimizers -- Adam (catch-up of the untouched
    steps), and any of Adam / Adagrad / plain SGD with weight decay (`rule` 0 / 1 / 2: the untouched steps on g = weight_decay * p)."""
    _fields_ = [('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('replay', ctypes.c_int32), ('rule', ctypes.c_int32),
                ('step', ctypes.c_void_p), ('weight_decay', ctypes.c_float), ('reserved', ctypes.c_float)]


RULES = {'adam': 0, 'adagrad': 1, 'sgd': 2}


def is_lazy(kind, weight_decay=0.0):
    """Does the row-sparse form of `kind` need the state rows [m | v | last] and the catch-up of untouched steps?  Adam always (a dense Adam
    step moves every row that ever had a gradient), everything under weight decay (the dense step moves every row at every step)."""
    return kind == 'adam' or float(weight_decay) != 0.0


def adam_state_pitch(d):
    return 2 * d + 4            # KTUP_SHARD_ADAM_STATE_PITCH: [m (d) | v (d) | last (int32) + padding]


def adam_replay(betas, tol=1e-5):
    """Zero-gradient steps replayed one by one when a row is touched again: the replayed increments fall like (beta1 / sqrt(beta2))^k, so
    after this many (110 for the default betas) the rest of the series is below 10 `tol` = 1e-4 of its first term -- < 2e-6 absolute at
    the learning rates in use, a tenth of the comparison band."""
    r = float(betas[0]) / math.sqrt(float(betas[1]))
    if r <= 0.0:
        return 0
    if r >= 1.0:
        return 1 << 20
    return min(1 << 20, int(math.ceil(math.log(tol) / math.log(r))))


def row_state(weight, kind, weight_decay=0.0):
    """Optimizer state of a table (or shard) for `kind`: None (sgd), the Adagrad sums, or the [m | v | last] rows of Adam and of every
    kind under weight decay (is_lazy; Adagrad's sum is then the `v` half)."""
    if is_lazy(kind, weight_decay):
        kind = 'adam'
    if kind == 'adagrad':
        return torch.zeros_like(weight)
    if kind == 'adam':
        return torch.zeros(weight.shape[0], adam_state_pitch(weight.shape[1]), dtype=torch.float32, device=weight.device)
    return None


def _check_state(state, weight, kind, weight_decay=0.0):
    if is_lazy(kind, weight_decay):
        kind = 'adam'
    want = None if kind == 'sgd' else (weight.shape[0], weight.shape[1] if kind == 'adagrad' else adam_state_pitch(weight.shape[1]))
    return (state is None and want is None) or (state is not None and want is not None and tuple(state.shape) == want)


def _p(t):
    return None if t is None else t.data_ptr()


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _i64s(vs):
    return (ctypes.c_int64 * len(vs))(*[int(v) for v in vs])


class _ShardedStepBase(object):
    """What the rec and the kg stepper share: eager warm-up, capture of the segments as HIP graphs, replay, the collectives
    between the segments, the skipped-step counter.  A subclass provides `_bind(stream, side)` -> list of segments (lists of
    pre-bound launches; ('fork', [...]) / ('join',) around launches bound to the side stream) and the buffers `_exchange` names."""

    def _exchange(self, k):
        """The collective after segment k (several ranks only)."""
        from jTransUP.parallel import _a2a, _all_reduce
        if not dist.is_initialized():                        # force_exchange without a process group: one rank talks to itself
            if k == 0:
                self.recv_ids.copy_(self.send_ids)
            elif k == 1:
                self.X[:self.W].copy_(self.Xsend)
            elif k == 2:
                self.Grecv.copy_(self.Gwire)
            return
        if k == 0:
            _a2a(self.recv_ids, self.send_ids, None, None, self.group)            # ids to their owners
        elif k == 1:
            _a2a(self.X[:self.W], self.Xsend, None, None, self.group)             # rows back
        elif k == 2:
            _a2a(self.Grecv, self.Gwire, None, None, self.group)                  # row gradients to the owners
        elif k == 3:
            _all_reduce(self.bucket, self.group)                                  # small gradients + norm + overflow flag

    def _whole_step_graph(self):
        """Several ranks: the five segments AND the collectives between them as ONE graph -- when the collectives can be captured (RCCL, or
        the device copies that stand for them when one rank talks to itself without a process group).  Under gloo (the CPU staging of the
        two-ranks-share-the-GPU tests) every segment stays a graph of its own with the collectives issued between the replays."""
        if not self.multi or not getattr(self, 'exchange_graph', True):
            return False
        return (not dist.is_initialized()) or dist.get_backend(self.group) == 'nccl'

    def close(self):
        """Drop the captured graphs and bound launches (stream-synchronised first).  A process group must not be destroyed while a graph
        that holds its captured collectives is alive: `destroy_process_group()` then waits forever (seen with RCCL's all-to-all inside the
        whole-step graph) -- call this (ShardedKtupJoint.close closes both steppers) before tearing the group down."""
        torch.cuda.synchronize(self.dev)
        self._graphs = self._graphs1 = self._cycles = None
        self._graph_keep = self._graph_keep1 = None
        self._eager = None

    def _pipelined(self):
        """Batches fed from device columns (set_feed): the route of step s + 1 -- the hash of its ids, the wire slots, the send buffer, the
        counting sort: 30-35 us of small dependent launches that depend on nothing but the id columns -- runs on the second stream at the
        END of step s, beside the (owner's) apply walk, instead of at the head of step s + 1.  (Not with load_batch: the next batch is
        not there yet.)  The first step after set_feed routes itself (run()).  One rank: the walks of step s read the route's output
        to their end, so the route's buffers exist TWICE (`_sets`) and steps alternate between them -- two graphs, `_graphs` and
        `_graphs1`; several ranks: the requester's side of the route is free once the bucket launch has read the overflow word."""
        return self._fed and getattr(self, 'pipeline_route', True) and not getattr(self, 'route_beside', False)

    def _double(self):
        return self._pipelined()

    def _use(self, par):
        """Point the attribute names of the route's buffers (entries, send_ids, sort_ws, ...) at buffer set `par`: what _bind bakes in."""
        for k, v in self._sets[par].items():
            setattr(self, k, v)
        self._bind_par = par

    def _open_branch(self):
        import os
        return os.environ.get('KTUP_EXCHANGE_LAYOUT', 'open') == 'open'

    def run(self):
        """One step on the ids in the static buffers (or the cursor's batch of the feed columns).  The first two steps issue the
        launches directly (warm-up), then the segments are captured once and replayed."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        dbl = self._double()
        par = self._par if dbl else 0
        self._use(par)
        if self._pipelined() and not self._routed:               # nobody has routed this step's batch yet
            self._route_now = self._bind_route(stream)            # (kept: the ctypes arrays behind the bound launch)
            self._route_now[0]()
            self._routed = True
        if dbl:
            self._par ^= 1
        if not self.use_graphs or self.steps < 2:
            if self._eager is None:
                self._eager = {}
            hit = self._eager.get(par)
            if hit is None or hit[0] != stream:
                side = self._side if ((self.direct or self.multi or dbl) and self.overlap_route) else None
                hit = self._eager[par] = (stream, self._bind(stream, None if side is None else side.cuda_stream), self._keep)
            for k, seg in enumerate(hit[1]):
                self._issue(seg, self._side)
                if self.multi:
                    self._exchange(k)
            self.steps += 1
            return
        graphs = self._graphs1 if par else self._graphs
        if graphs is None:
            graphs = self._capture(par)
        for k, g in enumerate(graphs):
            g.replay()
            if self.multi and len(graphs) > 1:
                self._exchange(k)
        self.steps += 1

    def run_cycle(self, n):
        """`n` steps (even) on the next n batches of the feed columns as ONE graph replay.  One rank, device-fed, pipelined route: inside a
        graph of several steps the route of step s + 1 -- forked at the first launch of step s -- is joined in front of step s + 1's
        first launch instead of in front of step s's apply walk, where the main branch WAITED for it (the route beside the step kernel ends
        ~5 us after the norm walk, and a cross-queue join that has to wait costs ~12 us on top: profiles/r06_config5_timeline.txt; a join
        whose branch ended long ago costs ~6), and n steps pay for one graph launch.  Anything else (several ranks, host-fed batches, an
        odd n, the warm-up steps) runs the steps one by one.  Same launches, same order per step, same results as n calls of run()."""
        n = int(n)
        if n <= 0:
            return
        ok = self.use_graphs and not self.multi and self._double() and self.overlap_route and self._routed and self.steps >= 2 and n % 2 == 0 \
            and getattr(self, 'fused_apply', True)
        if not ok:
            for _ in range(n):
                self.run()
            return
        par = self._par
        if self._cycles is None:
            self._cycles = {}
        hit = self._cycles.get((n, par))
        if hit is None:
            graph, keeps = torch.cuda.CUDAGraph(), []
            with L.capture(graph):
                cs = torch.cuda.current_stream(self.dev).cuda_stream
                for j in range(n):
                    self._use(par ^ (j & 1))
                    seg = self._bind(cs, self._side.cuda_stream)[0]
                    keeps.append(self._keep)
                    # the step's launches with the join moved behind its last one (= in front of the next step's first)
                    self._issue([x for x in seg if x != ('join',)] + [('join',)], self._side)
            hit = self._cycles[(n, par)] = (graph, keeps)
        self._use(par)
        hit[0].replay()
        self.steps += n

    def _issue(self, seg, side):
        for item in seg:
            if callable(item):
                item()
            elif item[0] == 'fork':                           # launches bound to the side stream, ordered after everything so far
                side.wait_stream(torch.cuda.current_stream(self.dev))
                for launch in item[1]:
                    launch()
            elif item[0] == 'beside':                         # (main launch, side launches): both ordered after everything so far.  The
                main = torch.cuda.current_stream(self.dev)    # MAIN launch is enqueued first: a captured graph keeps the branch it meets first
                ev = torch.cuda.Event()                       # on the queue of the launches around it, and the other branch pays the
                ev.record(main)                               # cross-queue latency (~12 us each way)
                for launch in item[1]:
                    launch()
                side.wait_event(ev)
                for launch in item[2]:
                    launch()
            elif item[0] == 'join':
                torch.cuda.current_stream(self.dev).wait_stream(side)

    def _capture(self, par=0):
        """Each segment becomes one HIP graph (the collectives between them are issued by torch.distributed).  A captured segment
        also RUNS nothing: the step that triggers the capture replays the fresh graphs.  `par`: the buffer set the launches are
        bound to (`_use(par)` has been called)."""
        graphs, keeps = [], []
        n_seg = 5 if self.multi else 1
        if self._whole_step_graph():
            graph = torch.cuda.CUDAGraph()
            with L.capture(graph):
                cs = torch.cuda.current_stream(self.dev).cuda_stream
                side = self._side if self.overlap_route else None
                segs = self._bind(cs, None if side is None else side.cuda_stream)
                for k, seg in enumerate(segs):
                    self._issue(seg, side)
                    self._exchange(k)
            graphs, keeps = [graph], [self._keep]
        else:
            for k in range(n_seg):
                graph = torch.cuda.CUDAGraph()
                with L.capture(graph):
                    cs = torch.cuda.current_stream(self.dev).cuda_stream
                    side = self._side if ((self.direct or self.multi or self._double()) and self.overlap_route) else None
                    segs = self._bind(cs, None if side is None else side.cuda_stream)
                    keeps.append(self._keep)
                    self._issue(segs[k], side)
                graphs.append(graph)
        if par:
            self._graphs1, self._graph_keep1 = graphs, keeps
        else:
            self._graphs, self._graph_keep = graphs, keeps
        return graphs

    def _catchup(self, ids, caps, adam, stream, arr):
        """Adam: the launch that brings the step's rows up to date BEFORE anything reads them (ktup_shard_adam_catchup): the distinct
        owner-local rows `ids` = [caps[0] of table 0 | caps[1] of table 1 | ...] the route named (negative = unused slot), and every
        row of the small tables (pref / pref_norm rest during the kg steps and are read again by the next rec step)."""
        tabs = [t.weight.data for t in self.tables] + [p.data for p in self.small]
        sts = [t.state for t in self.tables] + list(self.small_state)
        offs = [sum(caps[:k]) for k in range(len(caps))]
        idp = [ids.data_ptr() + 8 * o for o in offs] + [None] * len(self.small)
        ns = list(caps) + [p.shape[0] for p in self.small]
        return L.bind('ktup_shard_adam_catchup', len(tabs), arr(_ptrs(tabs)), arr(_i64s([w.stride(0) for w in tabs])), arr(_ptrs(sts)),
                      arr(_i64s([x.stride(0) for x in sts])), arr((ctypes.c_void_p * len(idp))(*idp)), arr(_i64s(ns)), self.d, self.lr, self.eps,
                      adam, stream)

    def _rule(self):
        return AdamRule(self.betas[0], self.betas[1], adam_replay(self.betas), RULES[self.kind], self.opt_step.data_ptr(), self.weight_decay, 0.0)

    def flush(self):
        """Adam: bring EVERY row of this stepper's shards and small tables up to the current step (the zero-gradient steps a row has not
        been touched for; ktup_shard_adam_flush), so that what an evaluation, a gather or a checkpoint reads is what the reference's
        dense optimizer would hold.  Stream-ordered, no synchronisation; a no-op for the other optimizers."""
        if not self.lazy:
            return
        st = torch.cuda.current_stream(self.dev).cuda_stream
        rule = self._rule()
        for w, s in [(t.weight.data, t.state) for t in self.tables] + [(p.data, s) for p, s in zip(self.small, self.small_state)]:
            L.call('ktup_shard_adam_flush', w.data_ptr(), w.stride(0), s.data_ptr(), s.stride(0), self.d, w.shape[0], self.lr, self.eps,
                   ctypes.addressof(rule), st)

    # ------------------------------------------------------------------------------------------------ reporting
    def overflowed_steps(self):
        """Steps skipped so far because an exchange buffer overflowed (a device counter no launch clears; one device read -- call
        it rarely)."""
        return int(self.skipped.item())

    def last_step_unplaced(self):
        """Ids of the LAST step that found no slot on this job (0 = it ran)."""
        if not self.multi:
            return int(self._sets[self._bind_par]['counters'][-1].item())
        return int(self.bucket[-1].item())

    def check(self):
        n = self.overflowed_steps()
        if n:
            raise L.KtupError('%d sharded step(s) asked one owner for more distinct rows than capacity_factor=%.2f allows and were '
                              'skipped on every rank (tables untouched, losses dropped) -- raise capacity_factor (-shard_capacity_factor; world = always safe)'
                              % (n, self.capacity_factor))


class ShardedKtupStepper(_ShardedStepBase):
    """step = ShardedKtupStepper(Ut, It, Et, pref, pref_norm, rel, norm, item2ent, batch=8192, kind='adagrad', lr=0.005, max_norm=5.0)
    step(u, pos_items, neg_items)      # int64 device tensors of `batch` ids; returns nothing, syncs nothing
    step.loss_sum[0]                   # running sum of the steps' batch-mean BPR losses of THIS rank (a device float)
    step.check()                       # raises if ANY step so far overflowed its exchange capacity (and was therefore skipped)

    Ut / It / Et: parallel.ShardedTable (rows {g : g % world == rank}); pref, pref_norm, rel, norm: replicated (P, d) parameters;
    item2ent: int32 device table, global item -> global entity (negative or `ent_pad`: no aligned entity).  Exact w.r.t. the
    reference's dense step for plain SGD / Adagrad with l2_lambda = 0 (rows with a zero gradient do not move under either)."""

    def __init__(self, Ut, It, Et, pref, pref_norm, rel, norm, item2ent, batch, kind='adagrad', lr=0.005, eps=1e-10, max_norm=0.0,
                 l1=False, target=-1.0, orth=False, ent_pad=-1, group=None, capacity_factor=1.25, use_graphs=True, force_exchange=False,
                 direct=None, overlap_route=True, fused_apply=True, route_beside=False, betas=(0.9, 0.999), opt_step=None, exchange_graph=True,
                 weight_decay=0.0, use_st_gumbel=False, gumbel_seed=0, row_regs=False):
        self.exchange_graph = bool(exchange_graph)
        # TUP (transUP.py:69-82; run_item_recommendation.py -model_type transup): Et = rel = norm = item2ent = None -- two sharded tables,
        # two small ones.  row_regs: TUP's row regularisers (item_recommendation.py:177-180: normLoss of the batch's user rows, of its
        # [pos ; neg] item rows and of the preference table), added to the stored row gradients by a launch after the step kernel
        self.tup = Et is None
        self.row_regs = bool(row_regs)
        self.weight_decay = float(weight_decay)
        self.lazy = is_lazy(kind, weight_decay)
        self.use_st_gumbel = bool(use_st_gumbel)
        if kind not in KINDS:
            raise ValueError('row-sparse steps exist for plain SGD, Adagrad and Adam')
        self.betas = (float(betas[0]), float(betas[1]))
        self.has_state = kind != 'sgd' or self.lazy
        self.route_beside = bool(route_beside)
        if self.lazy and self.route_beside:      # the catch-up needs the route's distinct rows before the step kernel reads them
            raise ValueError('route_beside (the step kernel beside the whole route) does not exist for the lazy rules (Adam, weight decay): '
                             'the catch-up of the rows the route names comes before the step kernel')
        if self.tup and (rel is not None or norm is not None or item2ent is not None):
            raise ValueError('without an entity table there is no rel / norm / item2ent (TUP)')
        self.tables = [Ut, It] if self.tup else [Ut, It, Et]
        self.small = [pref, pref_norm] if self.tup else [pref, pref_norm, rel, norm]
        self.T = T = len(self.tables)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        for t in self.tables:
            if t.world != self.world or t.rank != self.rank:
                raise ValueError('the tables must be sharded over the stepper\'s process group')
        self.kind, self.lr, self.eps, self.max_norm = kind, float(lr), float(eps), float(max_norm)
        self.l1, self.target, self.orth = bool(l1), float(target), bool(orth)
        self.B = B = int(batch)
        self.d = d = Ut.d
        self.P = P = pref.shape[0]
        dev = self.dev = Ut.weight.device
        if dev.type != 'cuda':
            raise L.KtupError('ShardedKtupStepper runs the HIP kernels: the tables must live on the GPU (no CPU fallback)')
        if any(t.d != d for t in self.tables) or any(tuple(s.shape) != (P, d) or not s.is_contiguous() for s in self.small):
            raise ValueError('one row width for all tables; the small tables are contiguous (P, d)')
        if not L.load().ktup_train_step_supported(0, d, P):
            raise L.KtupError('no fused KTUP step kernel for d=%d, n_pref=%d (ktup_train_step_supported)' % (d, P))
        if not self.tup and (item2ent.dtype != torch.int32 or item2ent.device != dev):
            raise L.KtupError('item2ent must be an int32 device table')
        self.item2ent, self.ent_pad = (None if self.tup else item2ent.contiguous()), int(ent_pad)
        self.use_graphs = bool(use_graphs)
        self.overlap_route = bool(overlap_route)
        # fused_apply: reduce -> norm -> apply as two walks over the per-pair gradients (no W x d gradient buffer in between);
        # False keeps the three-launch form through Gwire (same results to rounding: the tests run both)
        self.fused_apply = bool(fused_apply)
        self._side = torch.cuda.Stream(device=Ut.weight.device) if self.overlap_route else None
        # force_exchange: take the several-ranks route (five segments, the three all-to-alls and the all-reduce) on ONE rank too --
        # what a rank of a bigger job runs, minus the wire; with an initialised process group the collectives are real (RCCL at
        # world 1), without one they are device copies
        self.multi = self.world > 1 or bool(force_exchange)
        self.capacity_factor = float(capacity_factor)
        # direct (one rank only): the step kernel gathers straight from the shards by global id -- no pack launch, no compact copy;
        # it needs every item's entity to be a row of Et (no negative map entries; `ent_pad`, if given, is Et's own zero row)
        can_direct = not self.multi and (self.tup or not bool((self.item2ent < 0).any()))
        if direct and not can_direct:
            raise ValueError('direct gathers need a single rank and an item2ent without negative entries')
        self.direct = can_direct if direct is None else bool(direct)
        W_ = self.world
        n_dist = [B, 2 * B, 2 * B][:T]                         # entries per table ([u], [pos ; neg], their entities) = at most this many DISTINCT ids
        if W_ == 1:
            cap = list(n_dist)
        else:
            cap = [min(n, int(math.ceil(self.capacity_factor * n / W_)) + 64) for n in n_dist]
        self.cap, self.capsum = cap, sum(cap)
        self.W = W = W_ * self.capsum
        self.E = E = (3 if self.tup else 5) * B
        i64 = lambda n, fill=None: torch.empty(n, dtype=torch.int64, device=dev) if fill is None else torch.full((n,), fill, dtype=torch.int64, device=dev)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        lib = L.load()
        self.u, self.pi, self.ni = i64(B, 0), i64(B, 0), i64(B, 0)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)       # batch the next step reads (device side, moves by itself)
        self._feed = (self.u, self.pi, self.ni, 1)
        # what a route writes and the step's launches read, TWICE: with batches fed from device columns on one rank, steps alternate between
        # the sets and the route of step s + 1 fills the other set beside the walks of step s (_pipelined); otherwise set 0 is the only one
        # in use.  The attribute names (self.entries, ...) point at the set of the step being bound / run (_use).
        self._sets = [{'entries': i64(E, -1), 'inverse': i64(E, 0), 'send_ids': i64(W, -1), 'pair_map': i32(W + 1),
                       'sort_ws': i32((lib.ktup_shard_route_sort_bytes(E, W) + 3) // 4), 'counters': i32(W_ * T + 1),
                       'acc': torch.zeros(SLOTS + 1, dtype=torch.float64, device=dev)} for _ in range(2)]   # acc: [SLOTS partial sums of squares | the job-wide total]
        self._par = 0
        self._use(0)
        self.route_ws = torch.empty((lib.ktup_shard_route_workspace_bytes(E) + 7) // 8, dtype=torch.int64, device=dev)
        self.X = f32(W + 1, d)                                # row W stays zero: "no entity" (jTransUP.py:96 padding_idx)
        self.Gcat = f32(3 * B, d)                             # [GU (B) ; GV (2B)]
        self.Gwire = f32(W, d)
        self.xkeys = i32(max(2, lib.ktup_shard_reduce_list_len(E, d)))
        self.acc_step = torch.zeros(SLOTS, dtype=torch.float64, device=dev)  # the step kernel's share, when it runs beside the route's init
        self.n_loss = 4 if self.row_regs else 2               # [batch-mean BPR terms, orthogonalLoss values (, normLoss of the rows, normLoss(pref))]
        self.loss_sum = f32(self.n_loss)                      # ... summed over the steps that ran
        self.loss_step = f32(self.n_loss)                     # the current step's terms (the apply launch folds and clears them)
        self.skipped = i32(1)                                 # steps skipped for overflow: never cleared by a launch
        n_g = 4 if (self.orth and not self.tup) else 2
        self.small_g = [f32(P, d) for _ in range(n_g)]        # orth: gP, gPn, gR, gRn; else gA (pref & rel), gC (pref_norm & norm)
        self.small_state = [row_state(s.data, kind, weight_decay) for s in self.small]
        for t in self.tables:
            if not _check_state(t.state, t.weight.data, kind, weight_decay):
                t.state = row_state(t.weight.data, kind, weight_decay)
        # Adam: the number of the step being applied, in device memory (the launches are replayed from graphs); shared by the steppers
        # of a joint schedule.  ktup_shard_step_count moves it just before every apply launch.
        self.opt_step = opt_step if opt_step is not None else torch.zeros(2, dtype=torch.int64, device=dev)      # [step, two floats of bias corrections]
        self.steps = 0
        # ST-Gumbel gate (transUP.py:118-170 / jTransUP.py:250-262, -use_st_gumbel): the draws of a step come from a Philox stream whose
        # position lives in device memory (KTUP_GUMBEL_PHILOX_DEV: the step is a replayed graph) and moves by 2 B P per step -- every rank
        # its own stream; set_gumbel_uniforms() feeds recorded uniforms instead (the parity mode of the tests)
        self.gstate = self.gadv = self.guni = None
        if self.use_st_gumbel:
            seed = (int(gumbel_seed) * 6364136223846793005 + 1442695040888963407 + 7919 * self.rank) % (1 << 62)
            self.gstate = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
            self.gadv = torch.tensor([0, 2 * B * P], dtype=torch.int64, device=dev)
        if self.multi:
            self.recv_ids = i64(W, -1)
            self.Xsend = f32(W, d)
            self.Grecv = f32(W, d)
            self.cap_own = [max(1, min(W_ * c, t.weight.shape[0])) for c, t in zip(cap, self.tables)]
            self.W_own = Wo = sum(self.cap_own)
            self.own_inverse = i64(W, 0)
            self.own_ids = i64(Wo, -1)
            self.own_sort = i32((lib.ktup_shard_route_sort_bytes(W, Wo) + 3) // 4)
            self.own_counters = i32(T + 1)
            self.own_ws = torch.empty((lib.ktup_shard_route_workspace_bytes(W) + 7) // 8, dtype=torch.int64, device=dev)
            self.Gown = f32(Wo, d)
            self.own_xkeys = i32(max(2, lib.ktup_shard_reduce_list_len(W, d)))
            self.bucket = torch.zeros(n_g * P * d + 2, dtype=torch.float64, device=dev)
        self._eager = None
        self._graphs = self._graphs1 = self._cycles = None
        self._graph_steps = 0
        self._fed = self._routed = False

    # ------------------------------------------------------------------------------------------------ launch lists
    def _bind(self, stream, side=None):
        """Pre-bound launches (lib.bind) on `stream`, as the segments between the collectives.  One rank with direct gathers: the
        step kernel needs only the entry list (the route's first launch), so the rest of the route -- hashing, slots, the
        counting sort: four small latency-bound launches -- is bound to `side` (a second stream) and runs beside it; the
        segment then reads [init, ('fork', [...]), step, ('join',), ...]."""
        B, d, P, W, E, Wn = self.B, self.d, self.P, self.W, self.E, self.world
        Ut, It = self.tables[0], self.tables[1]
        Et = None if self.tup else self.tables[2]
        T = self.T
        pref, pref_norm = self.small[0].data, self.small[1].data
        rel, norm = (None, None) if self.tup else (self.small[2].data, self.small[3].data)
        keep = self._keep = []                                 # ctypes arrays must outlive the bound launches

        def arr(x):
            keep.append(x)
            return ctypes.addressof(x)
        tabs = arr(_ptrs([t.weight.data for t in self.tables]))
        lds = arr(_i64s([t.weight.data.stride(0) for t in self.tables]))
        states = arr(_ptrs([t.state for t in self.tables])) if self.has_state else None
        slds = arr(_i64s([t.state.stride(0) for t in self.tables])) if self.has_state else lds      # interleaved tables: pitch 2d
        adam = arr(self._rule()) if self.lazy else None                 # (the lazy rules: Adam, or any kind under weight decay)
        cap = arr(_i64s(self.cap))
        kind = KINDS['adam'] if self.lazy else KINDS[self.kind]      # (the C side: 'state rows [m | v | last] + the rule' or a plain form)
        gscale = 1.0 / Wn
        g = self.small_g
        if self.tup:                                         # two small tables, each with a gradient of its own
            gP, gPn, gR, gRn = g[0], g[1], None, None
            sg_list = [gP, gPn]
            sp0, ss0 = [pref, pref_norm], list(self.small_state)
            sp1, ss1 = [None] * 2, [None] * 2
            norm_list, small_weight = [gP, gPn], 1.0
        elif self.orth:
            gP, gPn, gR, gRn = g
            sg_list = [gP, gPn, gR, gRn]
            sp0, ss0 = [pref, pref_norm, rel, norm], list(self.small_state)
            sp1, ss1 = [None] * 4, [None] * 4
            norm_list, small_weight = [gP, gPn, gR, gRn], 1.0
        else:
            gP, gPn, gR, gRn = g[0], g[1], None, None
            sg_list = [g[0], g[1]]
            sp0, ss0 = [pref, pref_norm], [self.small_state[0], self.small_state[1]]
            sp1, ss1 = [rel, norm], [self.small_state[2], self.small_state[3]]
            norm_list, small_weight = [g[0], g[0], g[1], g[1]], 2.0      # the norm runs over all four tables' gradients
        n_small = len(sg_list)
        sgp, sp0p, ss0p = arr(_ptrs(sg_list)), arr(_ptrs(sp0)), (arr(_ptrs(ss0)) if self.has_state else None)
        two = not self.orth and not self.tup                 # a gradient shared by two tables (pref & rel, pref_norm & norm)
        sp1p = arr(_ptrs(sp1)) if two else None
        ss1p = arr(_ptrs(ss1)) if (two and self.has_state) else None
        X, inv = self.X, self.inverse
        close = (_p(self.loss_step), self.n_loss, _p(self.loss_sum), _p(self.skipped))
        # one rank, two-walk form: the step kernel adds the stored rows' squared norms itself and the norm walk only corrects for
        # rows that several entries share (dup_only) -- with ids spread over millions of rows it reads almost nothing
        import os as _os
        # (not with row regularisers: they change the stored rows after the step kernel has summed their squares)
        dup = (not self.multi) and self.fused_apply and not self.row_regs and _os.environ.get('KTUP_C5_DUP','1') != '0'
        ssq = (_p(self.acc), SLOTS) if dup else (None, 0)
        # one rank, direct gathers, a second stream: the step kernel reads the id columns itself and runs beside the WHOLE route (its
        # init launch included); its squared norms go to accumulators of their own that the boundary-norm launch folds in, and that
        # launch moves the cursor once both branches are done
        # (measured at config 5: 0.151 ms per step against 0.141 with only the route's last four launches beside the kernel -- the init
        # launch's 1,024 workgroups delay the step kernel's first tiles -- so it is an option, off by default)
        beside = dup and self.direct and side is not None and self.route_beside
        cols = (None, None, 0)
        fold = (None, 0, None)
        if beside:
            ssq = (_p(self.acc_step), SLOTS)
            fold = (_p(self.acc_step), SLOTS, _p(self.cursor))
        bind = L.bind
        fu, fp, fn, nb = self._feed
        def route_phase(phase, on):
            return self._route_launch(phase, on, keep)
        route = route_phase(0, stream)
        gate = (0, None)                                     # KTUP_GUMBEL_OFF / _INPUT (recorded uniforms) / _PHILOX_DEV (device-resident stream position)
        if self.use_st_gumbel:
            gate = (1, _p(self.guni)) if self.guni is not None else (3, _p(self.gstate))
        if self.direct:                                      # global ids straight into the shards (entries = [u | pos ; neg | ...])
            ent = self.entries
            uid_p, iid_p = (_p(fu), _p(fp)) if beside else (_p(ent), ent.data_ptr() + B * 8)
            cols = (_p(fn), _p(self.cursor), nb) if beside else (None, None, 0)
            step = bind('ktup_train_rec_step_rows_ws', _p(Ut.weight.data), Ut.weight.data.stride(0), _p(It.weight.data), It.weight.data.stride(0),
                        None if self.tup else _p(Et.weight.data), 0 if self.tup else Et.weight.data.stride(0), _p(self.item2ent), self.ent_pad, _p(pref), _p(pref_norm), _p(rel),
                        _p(norm), d, P, d, uid_p, iid_p, B, int(self.l1), self.target, gscale, int(self.orth),
                        _p(self.loss_step), _p(self.Gcat), self.Gcat.data_ptr() + B * d * 4, _p(gP), _p(gPn), _p(gR), _p(gRn), *ssq, *cols, *gate, *self._small_ws(), stream)
        else:
            step = bind('ktup_train_rec_step_rows_ws', _p(X), d, _p(X), d, None if self.tup else _p(X), 0 if self.tup else d, None if self.tup else _p(self.pair_map), W, _p(pref), _p(pref_norm), _p(rel),
                        _p(norm), d, P, d, _p(inv), inv.data_ptr() + B * 8, B, int(self.l1), self.target, gscale, int(self.orth),
                        _p(self.loss_step), _p(self.Gcat), self.Gcat.data_ptr() + B * d * 4, _p(gP), _p(gPn), _p(gR), _p(gRn), *ssq, None, None, 0, *gate, *self._small_ws(), stream)
        if self.row_regs:                                    # TUP: normLoss of the gathered rows and of pref, onto the stored row gradients
            if self.direct:
                ru, ri, ldr_u, ldr_i, idu, idi = _p(Ut.weight.data), _p(It.weight.data), Ut.weight.data.stride(0), It.weight.data.stride(0), \
                    _p(self.entries), self.entries.data_ptr() + B * 8
            else:
                ru, ri, ldr_u, ldr_i, idu, idi = _p(X), _p(X), d, d, _p(inv), inv.data_ptr() + B * 8
            reg = bind('ktup_train_rec_reg_rows', ru, ldr_u, ri, ldr_i, d, idu, idi, B, _p(self.Gcat), self.Gcat.data_ptr() + B * d * 4,
                       _p(pref), P, _p(gP), 1.0, gscale, self.loss_step.data_ptr() + 8, stream)   # (rows: every rank its own; pref: once over the job)
            score_step = step

            def step():
                score_step()
                reg()
        if self.use_st_gumbel and self.guni is None:         # the stream position moves past this step's draws (a torch op: captured with the rest)
            launch_step, gstate, gadv = step, self.gstate, self.gadv

            def step():
                launch_step()
                gstate.add_(gadv)
        reduce_ = bind('ktup_shard_reduce_rows', _p(self.Gcat), d, d, 3 * B, 2 * B, _p(self.sort_ws), E, W, _p(self.Gwire), d, stream)
        if not self.multi:
            pack = bind('ktup_shard_pack_wire', T, tabs, lds, cap, d, _p(self.send_ids), 1, _p(X), d, stream)
            nl = [self.Gwire] + norm_list
            nptr, nsz = arr(_ptrs(nl)), arr(_i64s([t.numel() for t in nl]))
            gnorm = bind('ktup_optim_gradnorm_acc', len(nl), nptr, nsz, _p(self.acc), SLOTS, stream)
            apply_ = bind('ktup_shard_apply', kind, T, tabs, lds, states, slds, cap, d, _p(self.send_ids), 1, _p(self.Gwire), d, n_small, P,
                          sgp, sp0p, ss0p, sp1p, ss1p, None, self.lr, self.eps, _p(self.acc), SLOTS, self.max_norm,
                          self.counters.data_ptr() + 4 * (Wn * T), None, *close, adam, stream)
            count = [bind('ktup_shard_step_count', _p(self.opt_step), self.counters.data_ptr() + 4 * (Wn * T), None, self.betas[0], self.betas[1], stream)] if adam else []
            if self.fused_apply:
                nw = arr(_ptrs(sg_list))
                rep_p, rep_n = self._small_ws()
                rnorm = bind('ktup_shard_reduce_norm_fold', _p(self.Gcat), d, d, 3 * B, 2 * B, _p(self.sort_ws), E, W, _p(self.Gwire), d,
                             _p(self.xkeys), n_small, nw, P * d, small_weight, _p(self.acc), SLOTS, int(dup), *fold,
                             rep_p, 8 if rep_p else 0, arr(_ptrs([gP, gPn, gR, gRn])) if rep_p else None, stream)
                rapply = bind('ktup_shard_reduce_apply', kind, T, tabs, lds, states, slds, cap, _p(self.send_ids), 1, _p(self.Gcat), d, d,
                              3 * B, 2 * B, _p(self.sort_ws), E, _p(self.Gwire), d, _p(self.xkeys), n_small, P, sgp, sp0p, ss0p, sp1p, ss1p,
                              None, self.lr, self.eps, _p(self.acc), SLOTS, self.max_norm, self.counters.data_ptr() + 4 * (Wn * T), None,
                              *close, adam, stream)
                tail = [rnorm] + count + [rapply]
            else:
                tail = [reduce_, gnorm] + count + [apply_]
            if self._double():     # this step's route ran beside the previous step's walks (into this buffer set); the next step's runs beside these
                nxt = [self._route_launch(0, side if side is not None else stream, keep, 1 - self._bind_par)]
                head = ([self._catchup(self.send_ids, self.cap, adam, stream, arr)] if adam else []) + ([step] if self.direct else [pack, step])
                # the branch leaves at the step's first launch and is joined in front of its last one: a graph that ENDS in a join pays ~17 us
                # before the next replay starts (measured), and a route that starts beside the walks ends after them
                # (joined at the graph's end instead: 0.1217 against 0.1205 ms; the next route on the second stream as a graph of its own with
                #  events between the replays -- no branch inside the step's graph at all: 0.126; forked after the step kernel, beside the
                #  walks only: 0.143 -- the branch then ends after the walks and a graph that ends in a late join pays ~17 us)
                return [[('beside', head + tail[:-1], nxt), ('join',), tail[-1]] if side is not None else head + tail + nxt]
            if adam:                                         # the whole route, then the catch-up of the rows it named, then whoever reads them
                catch = self._catchup(self.send_ids, self.cap, adam, stream, arr)
                return [([route, catch, step] if self.direct else [route, catch, pack, step]) + tail]
            if beside:
                return [[('beside', [step], [route_phase(3, side)]), ('join',)] + tail]
            if self.direct and side is not None:
                return [[route_phase(1, stream), ('beside', [step], [route_phase(2, side)]), ('join',)] + tail]
            return [([route, step] if self.direct else [route, pack, step]) + tail]
        # ---- several ranks (or one in exchange form): five segments around the three all-to-alls and the all-reduce.  What does not lie on
        # the path of the data rides on the second stream: the counting sort of the entries (only the gradient reduction reads it) and the
        # zero-fill of the few wire rows that several entries share run beside the pack launch, the owner's route of the requested rows
        # (only the owner's reduction reads it) beside the step kernel.  The requester's reduction STORES its rows (ktup_shard_reduce_store):
        # no zero-filled 42 MB buffer, no read-modify-write.
        on = side if side is not None else stream

        def par(main, beside):
            return [('beside', main, beside), ('join',)] if side is not None else beside + main
        capo = arr(_i64s(self.cap_own))
        eoff_o = arr(_i64s([0, self.cap[0], self.cap[0] + self.cap[1], self.capsum][:T] + [self.capsum]))
        pack = bind('ktup_shard_pack_wire', T, tabs, lds, cap, d, _p(self.recv_ids), Wn, _p(self.Xsend), d, stream)
        sort_ = route_phase(5, on)
        zshared = bind('ktup_shard_zero_shared_rows', _p(self.sort_ws), E, W, _p(inv), _p(self.Gwire), d, d, on)
        rep_p, rep_n = self._small_ws()
        rstore = bind('ktup_shard_reduce_store_fold', _p(self.Gcat), d, d, 3 * B, 2 * B, _p(self.sort_ws), E, W, _p(self.Gwire), d,
                      rep_p, 8 if rep_p else 0, P * d, arr(_ptrs([gP, gPn, gR, gRn])) if rep_p else None, stream)

        def oroute_on(st_):
            return bind('ktup_shard_route', _p(self.recv_ids), W, self.capsum, T, eoff_o, 1, capo, 0, 0, _p(self.own_inverse),
                        _p(self.own_ids), None, _p(self.own_sort), _p(self.own_counters), None, 0, _p(self.own_ws), st_)
        oreduce = bind('ktup_shard_reduce_rows', _p(self.Grecv), d, d, W, 0, _p(self.own_sort), W, self.W_own, _p(self.Gown), d, stream)
        nptr, nsz = arr(_ptrs([self.Gown])), arr(_i64s([self.Gown.numel()]))
        gnorm = bind('ktup_optim_gradnorm_acc', 1, nptr, nsz, _p(self.acc), SLOTS, stream)
        N = n_small * P * d
        pack_b = bind('ktup_shard_bucket', 0, n_small, sgp, P * d, _p(self.bucket), _p(self.acc), SLOTS, self.counters.data_ptr() + 4 * (Wn * T),
                      None, 1.0, stream)
        fin_b = bind('ktup_shard_bucket', 1, n_small, None, P * d, _p(self.bucket), None, 0, None, self.acc.data_ptr() + 8 * SLOTS, small_weight, stream)
        apply_ = bind('ktup_shard_apply', kind, T, tabs, lds, states, slds, capo, d, _p(self.own_ids), 1, _p(self.Gown), d, n_small, P,
                      sgp, sp0p, ss0p, sp1p, ss1p, _p(self.bucket), self.lr, self.eps, self.acc.data_ptr() + 8 * SLOTS, 1, self.max_norm,
                      None, self.bucket.data_ptr() + 8 * (N + 1), *close, adam, stream)
        count = [bind('ktup_shard_step_count', _p(self.opt_step), None, self.bucket.data_ptr() + 8 * (N + 1), self.betas[0], self.betas[1], stream)] if adam else []
        if self.fused_apply:
            onorm = bind('ktup_shard_reduce_norm', _p(self.Grecv), d, d, W, 0, _p(self.own_sort), W, self.W_own, _p(self.Gown), d,
                         _p(self.own_xkeys), 0, None, 0, 1.0, _p(self.acc), SLOTS, 0, None, 0, None, stream)
            oapply = bind('ktup_shard_reduce_apply', kind, T, tabs, lds, states, slds, capo, _p(self.own_ids), 1, _p(self.Grecv), d, d, W, 0,
                          _p(self.own_sort), W, _p(self.Gown), d, _p(self.own_xkeys), n_small, P, sgp, sp0p, ss0p, sp1p, ss1p,
                          _p(self.bucket), self.lr, self.eps, self.acc.data_ptr() + 8 * SLOTS, 1, self.max_norm, None,
                          self.bucket.data_ptr() + 8 * (N + 1), *close, adam, stream)
            own_tail = [[onorm, pack_b], [fin_b] + count + [oapply]]
        else:
            own_tail = [[oreduce, gnorm, pack_b], [fin_b] + count + [apply_]]
        whole = side is not None and self._whole_step_graph() and self._open_branch()      # one graph: a branch may stay open across the exchanges
        head, nxt = [route_phase(4, stream)], []
        if self._pipelined():           # this step's route ran during the previous step (into this buffer set)
            head = []
            nxt = [self._route_launch(4, on, keep, 1 - self._bind_par)]

        def later(first):
            """Whole-step graph: the NEXT step's route leaves the main stream at the requester's reduction and is joined after the bucket
            launch (55 us of main-stream work -- reduction, gradient exchange, norm walk, bucket -- against ~30 of route).  On the branch
            that carries the sort and the owner's route it ended 23 us after the step kernel and the reduction waited for it
            (profiles/r06_config5_timeline.txt, first collection)."""
            if not nxt:
                return [first]
            own_tail[0] = own_tail[0] + [('join',)]
            return [('beside', [first], nxt)]
        if adam:    # the owner's route of the requested rows moves in front of the pack launch: the catch-up needs its DISTINCT rows
            catch = self._catchup(self.own_ids, self.cap_own, adam, stream, arr)
            if whole:
                return [head, [oroute_on(stream), catch, ('beside', [pack], [sort_, zshared])], [step, ('join',)] + later(rstore)] + own_tail
            return [head, [oroute_on(stream), catch] + par([pack], [sort_, zshared]), par([step, rstore], nxt) if nxt else [step, rstore]] + own_tail
        if whole:   # sort, zero-fill, the owner's route (and the next step's route): one branch from the id exchange to the end of the step kernel
            return [head, [('beside', [pack], [sort_, zshared, oroute_on(on)])], [step, ('join',)] + later(rstore)] + own_tail
        return [head, par([pack], [sort_, zshared]), par([step, rstore], [oroute_on(on)] + nxt)] + own_tail

    def _small_ws(self):
        """(pointer, bytes) of the step kernel's REPLICAS of the preference tables' gradients (ktup_train_rec_step_rows_ws: tile workgroup b
        adds its partial sums to replica b mod 8 instead of all 256 workgroups to one copy), or (None, 0): the launch that follows the step
        kernel folds them into gP / gPn (/ gR / gRn) -- the norm walk on one rank, the requester's reduction on several -- so the forms
        without such a launch (one rank through a gradient buffer) keep the plain flush.  KTUP_STEP_SMALL_WS=0: none."""
        import os
        if os.environ.get('KTUP_STEP_SMALL_WS', '1') == '0' or (not self.multi and not self.fused_apply):
            return None, 0
        if getattr(self, '_small_ws_buf', None) is None:
            n = int(L.load().ktup_train_rec_step_rows_ws_bytes(self.B, self.P, self.d))
            self._small_ws_buf = torch.zeros((n + 3) // 4, dtype=torch.float32, device=self.dev) if n else None
        if self._small_ws_buf is None:
            return None, 0
        return self._small_ws_buf.data_ptr(), self._small_ws_buf.numel() * 4

    def set_gumbel_uniforms(self, uniforms):
        """Parity hook: the ST-Gumbel gate of the NEXT steps reads its uniforms -- (2B, n_pref): one row per scored pair, positives then
        negatives, as transUP.py:159-162 draws them -- from a fixed buffer that this call fills, instead of drawing Philox numbers on
        the device.  None switches back."""
        if not self.use_st_gumbel:
            raise L.KtupError('the stepper was built without the ST-Gumbel gate (use_st_gumbel=True)')
        rebind = (uniforms is None) != (self.guni is None)
        if uniforms is None:
            self.guni = None
        else:
            if self.guni is None:
                self.guni = torch.empty(2 * self.B, self.P, dtype=torch.float32, device=self.dev)
            self.guni.copy_(uniforms)
        if rebind:                                            # the gate's arguments are baked into the bound launches
            torch.cuda.synchronize(self.dev)
            self._eager = None
            self._graphs = self._graphs1 = self._cycles = None

    def _route_launch(self, phase, on, keep, par=None):
        """ktup_shard_route_ktup, pre-bound (phases: include/ktup_hip.h), writing buffer set `par` (default: the set in use).  Several
        ranks: the launch clears the SLOTS partial sums of the norm but not the job-wide total behind them -- the owner's apply walk of
        step s reads that word while the route of step s + 1 may already run beside it (_pipelined)."""
        S = self._sets[self._bind_par if par is None else par]
        cap = _i64s(self.cap)
        keep.append(cap)
        fu, fp, fn, nb = self._feed
        return L.bind('ktup_shard_route_ktup', _p(fu), _p(fp), _p(fn), self.B, nb, _p(self.cursor), _p(self.item2ent), self.ent_pad,
                      _p(S['entries']), self.world, ctypes.addressof(cap), _p(S['inverse']), _p(S['send_ids']), _p(S['pair_map']), _p(S['sort_ws']),
                      _p(S['counters']), _p(S['acc']), SLOTS if self.multi else SLOTS + 1, _p(self.route_ws), phase, on)

    def _bind_route(self, stream):
        keep = []
        return self._route_launch(0 if not self.multi else 4, stream, keep), keep

    # ------------------------------------------------------------------------------------------------ the step
    def load_batch(self, u, pos_items, neg_items):
        """Copy a batch into the step's static id buffers (skip it by writing step.u / step.pi / step.ni in place, or by set_feed)."""
        if self._feed[0] is not self.u:
            self.set_feed(None)
        self.u.copy_(u, non_blocking=True); self.pi.copy_(pos_items, non_blocking=True); self.ni.copy_(neg_items, non_blocking=True)

    def set_feed(self, columns):
        """Device-fed batches: columns = (u, pos_items, neg_items), contiguous int64 device tensors of n_batches x B ids each (an epoch
        of pre-drawn batches, or whatever a device-side sampler refills in place).  Step s reads batch (cursor mod n_batches) and the
        step's own first launches move the device cursor on: `run()` then needs no per-step copy or argument.  None: back to the
        static one-batch buffers that load_batch fills."""
        if columns is None:
            self._feed = (self.u, self.pi, self.ni, 1)
        else:
            u, p, n = columns
            for c in (u, p, n):
                if c.dtype != torch.int64 or c.device != self.dev or not c.is_contiguous() or c.numel() % self.B or c.numel() != u.numel():
                    raise L.KtupError('feed columns are contiguous int64 device tensors of n_batches x B ids each')
            self._feed = (u, p, n, u.numel() // self.B)
        self._fed, self._routed, self._par = columns is not None, False, 0
        self.cursor.zero_()
        self._eager = None
        self._graphs = self._graphs1 = self._cycles = None                   # the column addresses are baked into the bound launches

    def __call__(self, u=None, pos_items=None, neg_items=None):
        if u is not None:
            self.load_batch(u, pos_items, neg_items)
        self.run()


class ShardedKgStepper(_ShardedStepBase):
    """KTUP's kg step (knowledgable_recommendation.py:345-383, 394-403) on the row-sharded entity table:
        kg_lambda * ( marginLoss(pos, neg, margin) + orthogonalLoss(rel[r], norm[r]) + normLoss(ent[h, t of pos and neg]) + normLoss(rel[r]) )
    with the scores of jTransUP.py:144-157 (TransH on the entity / rel / norm tables; transh=False: TransE, no norm table), backward,
    clip_grad_norm, optimizer.step.

    step = ShardedKgStepper(Et, rel, norm, batch=8192, kind='adagrad', lr=0.005, max_norm=5.0, margin=1.0, kg_lambda=1.0)
    step(ph, pt, pr, nh, nt, nr)       # int64 device tensors of `batch` ids each; returns nothing, syncs nothing
    step.loss_sum                      # running sums of [margin term, orthogonalLoss, normLoss(ent), normLoss(rel)] of THIS rank's triples,
                                       # un-weighted: kg_lambda x their sum is the reference's loss scalar

    Et: parallel.ShardedTable; rel, norm: replicated (R, d) parameters.  small_state: the Adagrad sums of [rel, norm] when another
    stepper (the rec step of the joint schedule) already owns them.  marginLoss and the row regularisers are SUMS over the batch:
    every rank contributes its own triples' terms as they are, the gradients add up across ranks."""

    def __init__(self, Et, rel, norm, batch, kind='adagrad', lr=0.005, eps=1e-10, max_norm=0.0, l1=False, margin=1.0, kg_lambda=1.0,
                 transh=True, regs=7, small_state=None, group=None, capacity_factor=1.25, use_graphs=True, force_exchange=False,
                 direct=None, overlap_route=True, betas=(0.9, 0.999), opt_step=None, exchange_graph=True, weight_decay=0.0):
        self.exchange_graph = bool(exchange_graph)
        self.weight_decay = float(weight_decay)
        self.lazy = is_lazy(kind, weight_decay)
        if kind not in KINDS:
            raise ValueError('row-sparse steps exist for plain SGD, Adagrad and Adam')
        self.betas = (float(betas[0]), float(betas[1]))
        self.has_state = kind != 'sgd' or self.lazy

        self.tables = [Et]
        self.transh = bool(transh)
        self.small = [rel, norm] if self.transh else [rel]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if Et.world != self.world or Et.rank != self.rank:
            raise ValueError('the entity table must be sharded over the stepper\'s process group')
        self.kind, self.lr, self.eps, self.max_norm = kind, float(lr), float(eps), float(max_norm)
        self.l1, self.margin, self.kg_lambda, self.regs = bool(l1), float(margin), float(kg_lambda), int(regs)
        self.B = B = int(batch)
        self.d = d = Et.d
        self.P = P = rel.shape[0]
        dev = self.dev = Et.weight.device
        if dev.type != 'cuda':
            raise L.KtupError('ShardedKgStepper runs the HIP kernels: the tables must live on the GPU (no CPU fallback)')
        if any(tuple(s.shape) != (P, d) or not s.is_contiguous() for s in self.small):
            raise ValueError('rel / norm are contiguous (R, d) tables of the entity table\'s width')
        lib = L.load()
        if not lib.ktup_train_step_supported(1 if self.transh else 2, d, P):
            raise L.KtupError('no fused kg step kernel for d=%d (ktup_train_step_supported)' % d)
        self.use_graphs = bool(use_graphs)
        self.overlap_route = bool(overlap_route)
        self._side = torch.cuda.Stream(device=dev) if self.overlap_route else None
        self.multi = self.world > 1 or bool(force_exchange)
        self.capacity_factor = float(capacity_factor)
        can_direct = not self.multi
        if direct and not can_direct:
            raise ValueError('direct gathers need a single rank')
        self.direct = can_direct if direct is None else bool(direct)
        W_ = self.world
        n_dist = 4 * B
        cap = [n_dist if W_ == 1 else min(n_dist, int(math.ceil(self.capacity_factor * n_dist / W_)) + 64)]
        self.cap, self.capsum = cap, cap[0]
        self.W = W = W_ * self.capsum
        self.E = E = 4 * B
        i64 = lambda n, fill=None: torch.empty(n, dtype=torch.int64, device=dev) if fill is None else torch.full((n,), fill, dtype=torch.int64, device=dev)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.cols = [i64(B, 0) for _ in range(6)]                        # ph, pt, pr, nh, nt, nr
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self._feed = tuple(self.cols) + (1,)
        # the route's output twice (ShardedKtupStepper.__init__: steps fed from device columns alternate between the sets)
        self._sets = [{'entries': i64(E, 0), 'rels': i64(2 * B, 0), 'inverse': i64(E, 0), 'send_ids': i64(W, -1),
                       'sort_ws': i32((lib.ktup_shard_route_sort_bytes(E, W) + 3) // 4), 'counters': i32(W_ + 1),
                       'acc': torch.zeros(SLOTS + 1, dtype=torch.float64, device=dev)} for _ in range(2)]
        self._par = 0
        self._use(0)
        self.order = i32(B)
        self.route_ws = torch.empty((lib.ktup_shard_route_workspace_bytes(E) + 7) // 8, dtype=torch.int64, device=dev)
        self.X = f32(W + 1, d)
        self.GE = f32(E, d)
        self.Gwire = f32(W, d)
        self.xkeys = i32(max(2, lib.ktup_shard_reduce_list_len(E, d)))
        self.loss_sum, self.loss_step = f32(4), f32(4)
        self.skipped = i32(1)
        self.small_g = [f32(P, d) for _ in self.small]
        if self.has_state:
            self.small_state = list(small_state) if small_state is not None else [row_state(s.data, kind, weight_decay) for s in self.small]
            if not _check_state(Et.state, Et.weight.data, kind, weight_decay):
                Et.state = row_state(Et.weight.data, kind, weight_decay)
        else:
            self.small_state = [None] * len(self.small)
        self.opt_step = opt_step if opt_step is not None else torch.zeros(2, dtype=torch.int64, device=dev)      # [step, two floats of bias corrections]
        self.steps = 0
        if self.multi:
            self.recv_ids = i64(W, -1)
            self.Xsend = f32(W, d)
            self.Grecv = f32(W, d)
            self.cap_own = [max(1, min(W_ * cap[0], Et.weight.shape[0]))]
            self.W_own = Wo = self.cap_own[0]
            self.own_inverse = i64(W, 0)
            self.own_ids = i64(Wo, -1)
            self.own_sort = i32((lib.ktup_shard_route_sort_bytes(W, Wo) + 3) // 4)
            self.own_counters = i32(1 + 1)
            self.own_ws = torch.empty((lib.ktup_shard_route_workspace_bytes(W) + 7) // 8, dtype=torch.int64, device=dev)
            self.Gown = f32(Wo, d)
            self.own_xkeys = i32(max(2, lib.ktup_shard_reduce_list_len(W, d)))
            self.bucket = torch.zeros(len(self.small) * P * d + 2, dtype=torch.float64, device=dev)
        self._eager = None
        self._graphs = self._graphs1 = self._cycles = None
        self._fed = self._routed = False

    def _bind(self, stream, side=None):
        B, d, P, W, E, Wn = self.B, self.d, self.P, self.W, self.E, self.world
        Et = self.tables[0]
        keep = self._keep = []

        def arr(x):
            keep.append(x)
            return ctypes.addressof(x)
        tabs = arr(_ptrs([Et.weight.data]))
        lds = arr(_i64s([Et.weight.data.stride(0)]))
        states = arr(_ptrs([Et.state])) if self.has_state else None
        slds = arr(_i64s([Et.state.stride(0)])) if self.has_state else lds
        adam = arr(self._rule()) if self.lazy else None                 # (the lazy rules: Adam, or any kind under weight decay)
        cap = arr(_i64s(self.cap))
        kind = KINDS['adam'] if self.lazy else KINDS[self.kind]      # (the C side: 'state rows [m | v | last] + the rule' or a plain form)
        n_small = len(self.small)
        sgp = arr(_ptrs(self.small_g))
        sp0p = arr(_ptrs([s.data for s in self.small]))
        ss0p = arr(_ptrs(self.small_state)) if self.has_state else None
        rel = self.small[0].data
        norm = self.small[1].data if self.transh else None
        gR, gN = self.small_g[0], (self.small_g[1] if self.transh else None)
        close = (_p(self.loss_step), 4, _p(self.loss_sum), _p(self.skipped))
        skip_i = self.counters.data_ptr() + 4 * Wn
        bind = L.bind
        f = self._feed

        def route_phase(phase, on):
            return self._route_launch(phase, on, keep)
        order = bind('ktup_shard_kg_rel_order', _p(self.rels), B, P, _p(self.order), stream)
        if self.direct:
            Esrc, lde, ent_ids, ent_pad = Et.weight.data, Et.weight.data.stride(0), self.entries, -1
        else:
            Esrc, lde, ent_ids, ent_pad = self.X, d, self.inverse, W                  # padding entries route to the zero row W
        step = bind('ktup_train_kg_step_rows', int(self.transh), _p(Esrc), lde, _p(rel), d, _p(norm), d, d, _p(ent_ids), ent_pad, _p(self.rels),
                    _p(self.order), B, int(self.l1), self.margin, self.kg_lambda, self.regs, _p(self.loss_step), _p(self.GE), _p(gR), _p(gN),
                    *((None, 0) if self.multi else (_p(self.acc), SLOTS)), stream)
        if not self.multi:
            pack = bind('ktup_shard_pack_wire', 1, tabs, lds, cap, d, _p(self.send_ids), 1, _p(self.X), d, stream)
            rnorm = bind('ktup_shard_reduce_norm', _p(self.GE), d, d, E, 0, _p(self.sort_ws), E, W, _p(self.Gwire), d, _p(self.xkeys), n_small,
                         sgp, P * d, 1.0, _p(self.acc), SLOTS, 1, None, 0, None, stream)
            rapply = bind('ktup_shard_reduce_apply', kind, 1, tabs, lds, states, slds, cap, _p(self.send_ids), 1, _p(self.GE), d, d, E, 0,
                          _p(self.sort_ws), E, _p(self.Gwire), d, _p(self.xkeys), n_small, P, sgp, sp0p, ss0p, None, None, None, self.lr,
                          self.eps, _p(self.acc), SLOTS, self.max_norm, skip_i, None, *close, adam, stream)
            count = [bind('ktup_shard_step_count', _p(self.opt_step), skip_i, None, self.betas[0], self.betas[1], stream)] if adam else []
            if self._double():     # routed beside the previous step's walks; the next step's route (the other buffer set) beside these
                nxt = [self._route_launch(0, side if side is not None else stream, keep, 1 - self._bind_par)]
                head = ([self._catchup(self.send_ids, self.cap, adam, stream, arr)] if adam else []) + ([] if self.direct else [pack]) + [order, step]
                walks = [rnorm] + count + [rapply]
                return [[('beside', head + walks[:-1], nxt), ('join',), walks[-1]] if side is not None else head + walks + nxt]
            if adam:
                catch = self._catchup(self.send_ids, self.cap, adam, stream, arr)
                return [[route_phase(0, stream), catch] + ([] if self.direct else [pack]) + [order, step, rnorm] + count + [rapply]]
            if self.direct and side is not None:
                return [[route_phase(1, stream), ('beside', [order, step], [route_phase(2, side)]), ('join',), rnorm] + count + [rapply]]
            return [[route_phase(0, stream)] + ([] if self.direct else [pack]) + [order, step, rnorm] + count + [rapply]]
        # several ranks: as ShardedKtupStepper._bind -- the sort and the shared rows' zero-fill beside the pack launch, the owner's route
        # beside the step kernel, the requester's reduction storing its rows
        on = side if side is not None else stream

        def par(main, beside):
            return [('beside', main, beside), ('join',)] if side is not None else beside + main
        capo = arr(_i64s(self.cap_own))
        eoff_o = arr(_i64s([0, self.capsum]))
        pack = bind('ktup_shard_pack_wire', 1, tabs, lds, cap, d, _p(self.recv_ids), Wn, _p(self.Xsend), d, stream)
        sort_ = route_phase(5, on)
        zshared = bind('ktup_shard_zero_shared_rows', _p(self.sort_ws), E, W, _p(self.inverse), _p(self.Gwire), d, d, on)
        rstore = bind('ktup_shard_reduce_store', _p(self.GE), d, d, E, 0, _p(self.sort_ws), E, W, _p(self.Gwire), d, stream)

        def oroute_on(st_):
            return bind('ktup_shard_route', _p(self.recv_ids), W, self.capsum, 1, eoff_o, 1, capo, 0, 0, _p(self.own_inverse),
                        _p(self.own_ids), None, _p(self.own_sort), _p(self.own_counters), None, 0, _p(self.own_ws), st_)
        N = n_small * P * d
        onorm = bind('ktup_shard_reduce_norm', _p(self.Grecv), d, d, W, 0, _p(self.own_sort), W, self.W_own, _p(self.Gown), d,
                     _p(self.own_xkeys), 0, None, 0, 1.0, _p(self.acc), SLOTS, 0, None, 0, None, stream)
        pack_b = bind('ktup_shard_bucket', 0, n_small, sgp, P * d, _p(self.bucket), _p(self.acc), SLOTS, skip_i, None, 1.0, stream)
        fin_b = bind('ktup_shard_bucket', 1, n_small, None, P * d, _p(self.bucket), None, 0, None, self.acc.data_ptr() + 8 * SLOTS, 1.0, stream)
        oapply = bind('ktup_shard_reduce_apply', kind, 1, tabs, lds, states, slds, capo, _p(self.own_ids), 1, _p(self.Grecv), d, d, W, 0,
                      _p(self.own_sort), W, _p(self.Gown), d, _p(self.own_xkeys), n_small, P, sgp, sp0p, ss0p, None, None, _p(self.bucket),
                      self.lr, self.eps, self.acc.data_ptr() + 8 * SLOTS, 1, self.max_norm, None, self.bucket.data_ptr() + 8 * (N + 1),
                      *close, adam, stream)
        count = [bind('ktup_shard_step_count', _p(self.opt_step), None, self.bucket.data_ptr() + 8 * (N + 1), self.betas[0], self.betas[1], stream)] if adam else []
        own_tail = [[onorm, pack_b], [fin_b] + count + [oapply]]
        whole = side is not None and self._whole_step_graph() and self._open_branch()
        head, nxt = [route_phase(4, stream)], []
        if self._pipelined():
            head = []
            nxt = [self._route_launch(4, on, keep, 1 - self._bind_par)]

        def later(first):           # (ShardedKtupStepper._bind: the next step's route from the requester's reduction to the bucket launch)
            if not nxt:
                return [first]
            own_tail[0] = own_tail[0] + [('join',)]
            return [('beside', [first], nxt)]
        if adam:
            catch = self._catchup(self.own_ids, self.cap_own, adam, stream, arr)
            if whole:
                return [head, [oroute_on(stream), catch, ('beside', [pack], [sort_, zshared])], [order, step, ('join',)] + later(rstore)] + own_tail
            return [head, [oroute_on(stream), catch] + par([pack], [sort_, zshared]), par([order, step, rstore], nxt) if nxt else [order, step, rstore]] + own_tail
        if whole:
            return [head, [('beside', [pack], [sort_, zshared, oroute_on(on)])], [order, step, ('join',)] + later(rstore)] + own_tail
        return [head, par([pack], [sort_, zshared]), par([order, step, rstore], [oroute_on(on)] + nxt)] + own_tail

    def _route_launch(self, phase, on, keep, par=None):
        S = self._sets[self._bind_par if par is None else par]
        cap = _i64s(self.cap)
        keep.append(cap)
        f = self._feed
        return L.bind('ktup_shard_route_kg', _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(f[4]), _p(f[5]), self.B, f[6], _p(self.cursor),
                      _p(S['entries']), _p(S['rels']), self.world, ctypes.addressof(cap), _p(S['inverse']), _p(S['send_ids']), _p(S['sort_ws']),
                      _p(S['counters']), _p(S['acc']), SLOTS if self.multi else SLOTS + 1, _p(self.route_ws), phase, on)

    def _bind_route(self, stream):
        keep = []
        return self._route_launch(0 if not self.multi else 4, stream, keep), keep

    def load_batch(self, ph, pt, pr, nh, nt, nr):
        if self._feed[0] is not self.cols[0]:
            self.set_feed(None)
        for dst, src in zip(self.cols, (ph, pt, pr, nh, nt, nr)):
            dst.copy_(src, non_blocking=True)

    def set_feed(self, columns):
        """columns = (ph, pt, pr, nh, nt, nr): contiguous int64 device tensors of n_batches x B ids each; step s reads batch
        (cursor mod n_batches) and moves the device cursor on.  None: back to the static buffers load_batch fills."""
        if columns is None:
            self._feed = tuple(self.cols) + (1,)
        else:
            cs = tuple(columns)
            for c in cs:
                if c.dtype != torch.int64 or c.device != self.dev or not c.is_contiguous() or c.numel() % self.B or c.numel() != cs[0].numel():
                    raise L.KtupError('feed columns are contiguous int64 device tensors of n_batches x B ids each')
            self._feed = cs + (cs[0].numel() // self.B,)
        self._fed, self._routed, self._par = columns is not None, False, 0
        self.cursor.zero_()
        self._eager = None
        self._graphs = self._graphs1 = self._cycles = None

    def __call__(self, *ids):
        if ids:
            self.load_batch(*ids)
        self.run()


class ShardedKtupJoint(object):
    """The joint schedule of knowledgable_recommendation.py:209,320: step s is a rec step iff s % 10 < 10 * joint_ratio, else a
    kg step -- each stepper replays its own graphs; the entity shard, the rel / norm tables and every Adagrad sum are shared.

    joint = ShardedKtupJoint.build(Ut, It, Et, pref, pref_norm, rel, norm, item2ent, batch=8192, joint_ratio=0.7, ...)
    joint.rec.set_feed((u, pos, neg)); joint.kg.set_feed((ph, pt, pr, nh, nt, nr)); joint.run() ...; joint.check()"""

    def __init__(self, rec, kg, joint_ratio=0.7):
        self.rec, self.kg = rec, kg
        self.switch = 10 * float(joint_ratio)
        self.steps = 0

    @classmethod
    def build(cls, Ut, It, Et, pref, pref_norm, rel, norm, item2ent, batch, joint_ratio=0.7, margin=1.0, kg_lambda=1.0, kg_batch=None,
              orth=True, **kw):
        rec = ShardedKtupStepper(Ut, It, Et, pref, pref_norm, rel, norm, item2ent, batch=batch, orth=orth, **kw)
        kw_kg = {k: v for k, v in kw.items() if k not in ('target', 'ent_pad', 'fused_apply', 'opt_step', 'use_st_gumbel', 'gumbel_seed')}
        kg = ShardedKgStepper(Et, rel, norm, batch=kg_batch or batch, margin=margin, kg_lambda=kg_lambda,
                              small_state=rec.small_state[2:4] if rec.has_state else None, opt_step=rec.opt_step, **kw_kg)
        return cls(rec, kg, joint_ratio)

    def flush(self):
        """Adam: every row of the three shards and of the four small tables up to the current step (the rec stepper holds them all)."""
        self.rec.flush()

    def close(self):
        self.rec.close()
        self.kg.close()

    def is_rec(self, step=None):
        return (self.steps if step is None else step) % 10 < self.switch

    def run(self):
        (self.rec if self.is_rec() else self.kg).run()
        self.steps += 1

    def run_cycle(self, n):
        """The next n steps of the schedule, every stretch of rec (kg) steps as one graph replay of its stepper (run_cycle: an even number
        of steps per replay, an odd stretch's last step by itself)."""
        left = int(n)
        while left > 0:
            rec, run = self.is_rec(), 0
            while run < left and self.is_rec(self.steps + run) == rec:
                run += 1
            st = self.rec if rec else self.kg
            st.run_cycle(run - (run & 1))
            if run & 1:
                st.run()
            self.steps += run
            left -= run

    def check(self):
        self.rec.check()
        self.kg.check()
