"""GPU-resident training steps of the three drivers: the arithmetic of their `train_loop` step bodies
(knowledgable_recommendation.py:330-401 -> JointStepper; item_recommendation.py:160-195 -> RecStepper;
knowledge_representation.py:176-211 -> KGStepper) through the C ABI and, in a single process, replayed from one HIP graph per
step kind.  Where a fused kernel exists (ktup_train_step_supported: TUP / KTUP at d in {64, 100, 128}, TransH / TransE at any
d % 4 == 0) a step is TWO launches: ktup_train_rec_step or ktup_train_kg_step (scores, loss, regularisers, every gradient) and
ktup_optim_clip_step (norm, clip, optimizer, zero-fill of the gradients).  Otherwise (and with KTUP_FUSED_STEP=0) the round-1 sequence of about a dozen
launches runs: same arithmetic, separate kernels.

The autograd route (`model(...)`, `bprLoss`, `.backward()`, `clip_and_step`) costs ~50 launches and ~0.5 ms of Python per
B=512 step: every Function allocates and zero-fills table-shaped gradients which autograd then adds into `.grad`, and
positives and negatives are scored by separate launches.  Here positives and negatives share one forward and one
backward launch (ids concatenated in fixed buffers), the backward kernels accumulate straight into the persistent,
zero-filled `.grad` tensors, and the optimizer pass leaves them zeroed for the next step.  Same kernels, same
arithmetic: tests/test_fast_train.py checks the tables against the autograd route after a mixed rec/kg schedule.

Data parallel (config 4, torchrun, one process per GPU): every rank keeps all tables, takes rows [rank*B/G, (rank+1)*B/G)
of the step's global batch, and ONE all-reduce (RCCL) of a flat bucket -- all gradients are views into it, the loss
scalars ride at its end -- precedes the clip + step, which is then identical on every rank.  Loss terms scale as in
jTransUP/parallel.py: batch means and whole-table regularisers by 1/G, batch sums unchanged.

rec step (knowledgable_recommendation.py:335-344):  bprLoss(pos, neg, target=-1) + orthogonalLoss(pref, pref_norm)
kg step  (:345-382):  kg_lambda * ( marginLoss(pos, neg, margin) + orthogonalLoss(rel, norm)[rel ids]
                                    + normLoss(ent)[h, t ids of pos and neg] + normLoss(rel)[rel ids] )
"""
import torch
import torch.distributed as dist

from jTransUP.hip import lib as L
from jTransUP.hip import ops


def _p(t):
    return None if t is None else t.data_ptr()


class _StepperBase(object):
    """Shared machinery: flat gradient bucket (grads are views, 8 loss scalars at its end), data-parallel slice + all-reduce,
    pre-bound launches, HIP-graph replay, the K20 clip + step."""

    KINDS = ()        # step kinds of the subclass, e.g. ('rec', 'kg'), with the number of id tensors each takes
    N_IDS = {}

    def __init__(self, model, trainer, FLAGS, batch_size, group=None, use_graphs=None):
        if trainer.fused is None:
            raise L.KtupError('the GPU-resident step needs the fused optimizer (KTUP_FUSED_OPTIM=0 disables it)')
        self.m, self.trainer = model, trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if int(batch_size) % self.world:
            raise L.KtupError('batch_size %d is not divisible by the %d data-parallel ranks' % (batch_size, self.world))
        self.GB = int(batch_size)                    # global batch (what the driver samples)
        self.B = self.GB // self.world               # rows this rank scores
        self.margin, self.max_norm = float(FLAGS.margin), float(FLAGS.clipping_max_value)
        self.target = float(trainer.model_target)
        self.l1 = int(bool(getattr(model, 'L1_flag', False)))
        self.tabs = tuple(trainer.parameters)
        dev = self.tabs[0].device
        self.dev = dev
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        # persistent zero-filled gradients (torch 0.3 zero_grad semantics) as views into one flat bucket with 8 loss scalars at
        # its end: parallel.ReplicaGradSync owns the bucket and its ONE all-reduce per step
        from jTransUP.parallel import ReplicaGradSync
        self.sync = ReplicaGradSync(trainer.parameters, group=group, extra=8)
        self.flat, self.loss = self.sync.flat, self.sync.extra
        if self.world > 1:                           # identical replicas to start from
            for p in trainer.parameters:
                dist.broadcast(p.data, src=0, group=group)
        B = self.B
        self.score, self.gscore = torch.zeros(2 * B, **f32), torch.zeros(2 * B, **f32)
        self.inv_world = torch.full((), 1.0 / self.world, **f32)                         # upstream gradient of 'mean' / 'replicated' terms
        self.one = torch.ones((), **f32)
        self._keys = None
        self._stream = None
        # HIP graphs: with a step-independent optimizer every launch argument of a step is static (the ST-Gumbel stream
        # position lives in device memory, KTUP_GUMBEL_PHILOX_DEV), so the launches of a step replay as ONE graph launch
        # (KTUP_TRAIN_GRAPHS=0 disables).  Inputs are copied into fixed buffers first.
        if use_graphs is None:
            import os
            use_graphs = os.environ.get('KTUP_TRAIN_GRAPHS', '1') != '0'
        # Data-parallel replicas: the step is still ONE graph -- fused step kernel, the all-reduce of the flat bucket (RCCL
        # collectives are capturable), clip + optimizer launch -- when the backend is nccl; under the gloo test hook the
        # all-reduce is staged through the host and the launches are issued one by one (KTUP_DP_GRAPHS=0 forces that too)
        import os as _os0
        # Several ranks: capturing the all-reduce inside the step's graph is exercised over RCCL at world 1 only (this repo's boxes
        # have one GPU), so it is opt-in there (KTUP_DP_GRAPHS=1) and the default issues the launches and the collective one by
        # one; every rank takes the same route (same environment, same step counts).
        dp_ok = self.world == 1 or (dist.get_backend(group) == 'nccl' and _os0.environ.get('KTUP_DP_GRAPHS', '0') == '1')
        if dist.is_initialized() and self.world == 1 and dist.get_backend(group) != 'nccl':
            from jTransUP import parallel as _par
            dp_ok = not _par._FORCE[0]              # forced collectives on gloo stage through the host: not capturable
        self.use_graphs = bool(use_graphs) and dp_ok
        import os as _os
        self.want_fused = _os.environ.get('KTUP_FUSED_STEP', '1') != '0'
        self.out = {k: torch.zeros((), **f32) for k in self.KINDS}                       # fused steps: where the step's loss is published
        # the gradient norm without a pass over the gradients (include/ktup_hip.h `gnorm`): the fused step kernels track the squared
        # norm of the buffers their atomics build, ktup_optim_clip_step reads it -- one process only (an all-reduce of the gradients
        # changes their norm), only with clipping on, and for d <= 128 (at d = 256 the rec step kernel has no registers left for the
        # returned values).  KTUP_TRACKED_NORM=0: the norm pass + grid barrier of round 2.
        self._gn = None
        if self.world == 1 and self.max_norm > 0 and self.tabs[0].shape[1] <= 128 and _os.environ.get('KTUP_TRACKED_NORM', '1') != '0':
            self._gn = torch.zeros(64, dtype=torch.float64, device=dev)                 # KTUP_GNORM_WS_DOUBLES
        self._graphs = {}
        self._eager_steps = {k: 0 for k in self.KINDS}
        self._feeds, self._sampler, self._feed_launch, self._feed_ok = {}, None, {}, {}
        self._acc_on = False
        self.acc = {k: torch.zeros((), **f32) for k in self.KINDS}                       # fed steps: running sum of the steps' losses
        self._setup(FLAGS, f32, i64)

    # ------------------------------------------------------------------------------------------------ device-fed steps
    def attach_feeds(self, sampler, **feeds):
        """-device_sampling: `feeds[kind]` (DeviceFeeder) + `sampler` (DeviceSampler) let a step of that kind build its own batch
        on the device (ktup_feed_rec / ktup_feed_kg, captured in the step's graph): `fed_step(kind)` then needs no id tensors
        and no host work besides the graph replay.  Single process only (replicas slice one global batch per step)."""
        self._sampler = sampler
        self._feeds = dict(feeds)
        self._feed_ok = {}
        self._keys = None                       # bind the feed launches with the next plan

    def can_feed(self, kind):
        ok = self._feed_ok.get(kind)
        if ok is None:
            ok = False
            if kind in self._feeds and self.world == 1:
                self._plans()                   # decides fused_step for this shape
                ok = bool(self.fused_step)      # the multi-launch routes take id arrays the feed launches do not fill
            self._feed_ok[kind] = ok
        return ok

    def _acc_ptr(self, kind):
        """Fed steps: the optimizer launch adds the step's loss to acc[kind] itself."""
        return _p(self.acc[kind]) if self._acc_on else None

    def _make_feed(self, kind, st):
        b, sm, feed = L.bind, self._sampler, self._feeds[kind]
        if kind == 'rec':
            return b('ktup_feed_rec', _p(feed.cols[0]), _p(feed.cols[1]), feed.n, self.B, _p(feed.cursor), _p(sm.offset_dev[0:]),
                     sm.n_items, _p(sm.bitmap), sm.words if sm.bitmap is not None else 0, sm.seed, 1, _p(self.u2), _p(self.i2),
                     _p(sm.rec_workspace()), _p(sm.fail), st)
        return b('ktup_feed_kg', _p(feed.cols[0]), _p(feed.cols[1]), _p(feed.cols[2]), feed.n, self.B, _p(feed.cursor),
                 _p(sm.offset_dev[1:]), sm.n_ent, sm.n_rel, _p(sm.keys), 0 if sm.keys is None else sm.keys.numel(), sm.seed,
                 _p(self.h2), _p(self.t2), _p(self.r2), _p(sm.fail), st)

    def _bind_feeds(self, st):
        self._feed_launch = {kind: self._make_feed(kind, st) for kind in self._feeds}

    def take_sums(self):
        """{kind: sum of the losses of the fed steps since the last call} (one sync; the fed steps accumulate on the device)."""
        out = {k: float(v.item()) for k, v in self.acc.items()}
        for v in self.acc.values():
            v.zero_()
        return out

    def fed_cycle(self, kinds):
        """len(kinds) consecutive fed steps as ONE graph replay (hipGraphLaunch + the Python around it cost ~5 us per replay --
        a tenth of a step).  Returns len(kinds), or 0 when the cycle cannot run right now: before every kind has its
        single-step graph (warm-up, lazy optimizer state), after the optimizer was re-created, or when a feeder's epoch ends
        inside the cycle (the host reshuffles between replays, never inside one)."""
        fused = self.trainer.fused
        if not (self.use_graphs and fused.graph_safe()):
            return 0
        count = {}
        for kind in kinds:
            count[kind] = count.get(kind, 0) + 1
        for kind, c in count.items():
            entry = self._graphs.get(kind + '+fed')
            feed = self._feeds.get(kind)
            if entry is None or entry[2] is not fused or feed.start + (c - 1) * self.B > feed.n - self.B:
                return 0
        for kind, c in count.items():
            for _ in range(c):
                self._feeds[kind].fed()
                self._sampler.fed(self.B, kind)
        gkey = 'cycle:' + ','.join(kinds)
        entry = self._graphs.get(gkey)
        if entry is not None and entry[2] is not fused:
            entry = None
        if entry is None:
            graph = torch.cuda.CUDAGraph()
            import os as _os
            ahead = _os.environ.get('KTUP_FEED_AHEAD', '0') != '0'      # measured, round 5: 0.0456 against 0.0434 ms per step -- off
            with L.capture(graph):
                self._acc_on = True
                try:
                    if ahead:
                        self._capture_cycle_feeds_ahead(kinds)
                    else:
                        for kind in kinds:
                            self._plans()
                            self._feed_launch[kind]()
                            (self._rec_eager if kind == 'rec' else self._kg_eager)(*((None,) * self.N_IDS[kind]))
                finally:
                    self._acc_on = False
            self._keys = None
            fused._plan = None
            entry = (graph, None, fused)
            self._graphs[gkey] = entry
            captured = True
        else:
            captured = False
        entry[0].replay()
        if not captured:
            fused.bump_steps(len(kinds))
        self.trainer.step += len(kinds)
        return len(kinds)

    def _capture_cycle_feeds_ahead(self, kinds):
        """The launches of a fed cycle with the FEEDS on a second branch of the graph: a feed depends on the sampler's state and the
        cursor only, never on a step's result, so all of the cycle's feeds run back to back beside the chain
        step kernel -> clip + optimizer -> step kernel -> ..., each into id buffers of its own (step k reads slot k), and step k waits
        for feed k alone -- which has long finished when the chain gets there (ten feeds take ~90 us, the chain ~350).  The feeds
        would leave the critical path but for the first one: ~8 of a step's ~43 us.  (Round 4 tried the fork / join PER STEP: the
        cross-queue hand-over cost more than the feed it hid; here the graph forks once.)  Same feed kernels in the same order on
        one queue, so the batches are the single-step route's (tests/test_fast_train.py runs under either setting).
        MEASURED (round 5, tools/step_time.py): 0.0456 ms per step against 0.0434 with the feeds in the chain -- ten cross-queue
        dependency edges cost more than the ten feeds they hide -- so this form is opt-in (KTUP_FEED_AHEAD=1), kept as the record
        of the experiment."""
        main = torch.cuda.current_stream(self.dev)
        if getattr(self, '_feed_side', None) is None:
            self._feed_side = torch.cuda.Stream(device=self.dev)
        side = self._feed_side
        i64 = dict(dtype=torch.int64, device=self.dev)
        saved = {k: getattr(self, k, None) for k in ('_ids_rec', 'u2', 'i2', '_ids_kg', 'h2', 't2', 'r2', 'ht4')}
        slots = []
        try:
            for kind in kinds:                                   # slot buffers with the layout of _id_buffers
                self._id_buffers(kind, i64)
                slots.append({k: getattr(self, k) for k in (('_ids_rec', 'u2', 'i2') if kind == 'rec' else ('_ids_kg', 'h2', 't2', 'r2', 'ht4'))})
            self._cycle_slots = getattr(self, '_cycle_slots', []) + [slots]        # keep them alive as long as the graphs
            side.wait_stream(main)
            events = []
            with torch.cuda.stream(side):
                for kind, slot in zip(kinds, slots):
                    for k, v in slot.items():
                        setattr(self, k, v)
                    self._make_feed(kind, side.cuda_stream)()
                    ev = torch.cuda.Event()
                    ev.record(side)
                    events.append(ev)
            for kind, slot, ev in zip(kinds, slots, events):
                for k, v in slot.items():
                    setattr(self, k, v)
                self._keys = None                                # bind the step's launches to this slot's buffers
                self._plans()
                main.wait_event(ev)
                (self._rec_eager if kind == 'rec' else self._kg_eager)(*((None,) * self.N_IDS[kind]))
            main.wait_stream(side)
        finally:
            for k, v in saved.items():
                if v is not None:
                    setattr(self, k, v)
            self._keys = None

    def fed_step(self, kind):
        """One step of `kind` on the feeder's next batch: the step's graph is  feed launch -> step kernel -> clip + optimizer
        launch, and the host does nothing but replay it.  The optimizer launch adds the step's loss to `acc[kind]` (take_sums).
        (Drawing the NEXT batch on a side stream beside the optimizer launch was built and measured: the fork / join inside the
        graph costs more than the 8 us it hides -- 56 vs 50 us per step.)"""
        feed, sm, fused = self._feeds[kind], self._sampler, self.trainer.fused
        eager = self._rec_eager if kind == 'rec' else self._kg_eager
        nones = (None,) * self.N_IDS[kind]

        def fed_eager():
            self._plans()
            self._feed_launch[kind]()
            self._acc_on = True
            try:
                return eager(*nones)
            finally:
                self._acc_on = False

        feed.fed()
        sm.fed(self.B, kind)
        if not (self.use_graphs and fused.graph_safe()):
            out = fed_eager()
            self.trainer.step += 1
            return out
        gkey = kind + '+fed'
        entry = self._graphs.get(gkey)
        if entry is not None and entry[2] is not fused:       # the trainer re-created its optimizer (LR decay): capture again
            entry = None
            self._eager_steps[kind] = 0
        if entry is None and self._eager_steps[kind] < 2:
            self._eager_steps[kind] += 1
            out = fed_eager()
            self.trainer.step += 1
            return out
        if entry is None:
            graph = torch.cuda.CUDAGraph()
            with L.capture(graph):
                out = fed_eager()
            self._keys = None                                 # plans were bound to the capture stream: rebind for eager use
            fused._plan = None
            entry = (graph, out, fused)
            self._graphs[gkey] = entry
            captured = True
        else:
            captured = False
        entry[0].replay()
        if not captured:
            fused.bump_steps()
        self.trainer.step += 1
        return entry[1]

    def _gumbel_stream(self, draws_per_step):
        """(mode, pointer argument) of the preference gate, and the device-side stream position for the hard gate:
        uint64[2] = {seed, offset}; `_gumbel_advance()` moves the offset past a step's draws (a captured launch)."""
        if not getattr(self.m, 'use_st_gumbel', False):
            self.gstate = None
            return
        seed = (int(self.m._gumbel.seed) * 6364136223846793005 + 1442695040888963407) % (1 << 62)   # own stream, apart from eval's
        self.gstate = torch.tensor([seed, 0], dtype=torch.int64, device=self.dev)
        self.gadv = torch.tensor([0, int(draws_per_step)], dtype=torch.int64, device=self.dev)

    def _gate_args(self):
        if self.gstate is not None and getattr(self, 'guni', None) is not None:
            return (ops.GUMBEL_INPUT, _p(self.guni))
        return (ops.GUMBEL_OFF, None) if self.gstate is None else (ops.GUMBEL_PHILOX_DEV, _p(self.gstate))

    def set_gumbel_uniforms(self, uniforms):
        """Parity hook (tests against the reference's goldens): the ST-Gumbel gate of the NEXT steps reads its uniforms -- one row
        of n_pref values per scored pair, positives then negatives, as transUP.py:159-162 drew them -- from a fixed buffer that
        this call fills, instead of drawing Philox numbers on the device.  None switches back."""
        if self.gstate is None:
            raise L.KtupError('the model has no ST-Gumbel gate')
        if uniforms is None:
            if getattr(self, 'guni', None) is not None:    # bound launches and captured graphs still point at the uniform buffer
                self._keys = None
                self._graphs = {}
                self._eager_steps = {k: 0 for k in self.KINDS}
            self.guni = None
        else:
            if getattr(self, 'guni', None) is None:
                self.guni = torch.empty(2 * self.B, uniforms.shape[1], dtype=torch.float32, device=self.dev)
                self._keys = None                      # re-bind the launches with the new gate arguments
                self._graphs = {}
            self.guni.copy_(uniforms)

    def _gumbel_advance(self):
        if self.gstate is not None:
            self.gstate.add_(self.gadv)

    def _mine(self, t):
        """This rank's rows of a global-batch id tensor."""
        return t if self.world == 1 else t[self.rank * self.B:(self.rank + 1) * self.B]

    def _id_buffers(self, kind, i64):
        """The persistent [pos ; neg] id arrays of a step kind as views of ONE buffer, so that `_pack` fills them with one
        torch.cat (one launch and one dispatch per step instead of two for a rec step, four for a kg step)."""
        B = self.B
        if kind == 'rec':
            self._ids_rec = torch.zeros(4 * B, **i64)
            self.u2, self.i2 = self._ids_rec[:2 * B], self._ids_rec[2 * B:]              # [u ; u], [pos ; neg]
        else:
            self._ids_kg = torch.zeros(10 * B, **i64)
            self.h2, self.t2, self.r2 = self._ids_kg[:2 * B], self._ids_kg[2 * B:4 * B], self._ids_kg[4 * B:6 * B]
            self.ht4 = self._ids_kg[6 * B:]                                              # ph, pt, nh, nt (normLoss rows)

    def _pack(self, kind, args):
        if kind == 'rec':
            u, pi, ni = (self._mine(x) for x in args)
            torch.cat((u, u, pi, ni), out=self._ids_rec)
        else:
            ph, pt, pr, nh, nt, nr = (self._mine(x) for x in args)
            torch.cat((ph, nh, pt, nt, pr, nr, ph, pt, nh, nt), out=self._ids_kg)

    def _plans(self):
        st = torch.cuda.current_stream(self.dev).cuda_stream
        if self._keys is None or st != self._stream or \
                self._keys != tuple(t.data_ptr() for t in self.tabs) + tuple(t.grad.data_ptr() for t in self.tabs):
            self._keys = tuple(t.data_ptr() for t in self.tabs) + tuple(t.grad.data_ptr() for t in self.tabs)
            self._stream = st
            self._bind(st)
            if self._feeds:
                self._bind_feeds(st)

    def _gn_ptr(self):
        return None if self._gn is None else _p(self._gn)

    def _optimizer_launches(self, loss=None, tracked=False):
        """`tracked`: the launch before this one was a fused step bound with the gradient-norm workspace."""
        self.sync.all_reduce_grads()       # world > 1: gradients of all tables + the loss scalars, one bucket, one collective
        self.trainer.fused.clip_and_step(self.max_norm, zero_grads=True, loss=loss, gnorm=self._gn_ptr() if tracked else None)

    def _fused_ok(self, kind, d, n_pref=0):
        ok = bool(self.want_fused and L.load().ktup_train_step_supported(kind, d, n_pref))
        if not ok and L.get_option('deterministic'):
            # KTUP_DETERMINISTIC=1 (include/ktup_hip.h option "deterministic"): only the fused step kernels have the one-workgroup
            # form whose gradient sums do not depend on the order atomics land in; the multi-launch route would silently not have it
            raise L.KtupError('KTUP_DETERMINISTIC=1 needs a fused step kernel; this shape (kind %d, d=%d, n_pref=%d) has none' % (kind, d, n_pref))
        return ok

    def _step(self, kind, eager, args):
        fused = self.trainer.fused
        if not (self.use_graphs and fused.graph_safe()):
            out = eager(*args)
            self.trainer.step += 1
            return out
        entry = self._graphs.get(kind)
        if entry is not None and entry[2] is not fused:       # the trainer re-created its optimizer (LR decay): capture again
            entry = None
            self._eager_steps[kind] = 0
        if entry is None and self._eager_steps[kind] < 2:       # the first steps run eagerly (allocations, lazy optimizer state)
            self._eager_steps[kind] += 1
            out = eager(*args)
            self.trainer.step += 1
            return out
        self._pack(kind, args)                                   # ids -> the persistent [pos ; neg] buffers (outside the graph)
        if entry is None:
            graph = torch.cuda.CUDAGraph()
            with L.capture(graph):
                out = eager(*([None] * len(args)))               # None: ids are already packed
            self._keys = None                                    # plans were bound to the capture stream: rebind for eager use
            fused._plan = None
            entry = (graph, out, fused)
            self._graphs[kind] = entry
            captured = True                                      # the capture pass already ran the host side of clip_and_step
        else:
            captured = False
        entry[0].replay()
        if not captured:
            fused.bump_steps()
        self.trainer.step += 1
        return entry[1]

class JointStepper(_StepperBase):
    """KTUP (jtransup, own tables): knowledgable_recommendation.py:330-401."""
    KINDS = ('rec', 'kg')
    N_IDS = {'rec': 3, 'kg': 6}

    def _setup(self, FLAGS, f32, i64):
        model, B = self.m, self.B
        self.kg_lambda = float(FLAGS.kg_lambda)
        U, I, E, P, Pn, R, Rn = model._rec_tables()
        self.tabs = (U, I, E, P, Pn, R, Rn)
        self._id_buffers('rec', i64); self._id_buffers('kg', i64)
        self.gAC = torch.zeros(2, P.shape[0], P.shape[1], **f32)                         # mixed-table gradients gA, gC
        self.lam = torch.full((), self.kg_lambda, **f32)                                 # upstream gradient of the KG 'sum' terms
        self.ws = ops.pref_workspace(P, Pn, R, Rn)
        self.ent_pad = model.ent_total - 1
        self.i2e = model._item2ent
        self._gumbel_stream(2 * B * P.shape[0])

    # ------------------------------------------------------------------------------------------------ launch plans
    def _bind(self, st):
        """Every launch of a step has fixed arguments (persistent buffers): marshal them once (lib.bind).  Re-done when a
        table's storage moves (load_state_dict keeps storages, so in practice never) or the stream changes (graph capture)."""
        U, I, E, P, Pn, R, Rn = self.tabs
        B = self.B
        n_pref, d = P.shape
        n_rel = min(R.shape[0], Rn.shape[0])
        pos, neg, gpos, gneg = self.score[:B], self.score[B:], self.gscore[:B], self.gscore[B:]
        gate, gptr = self._gate_args()
        b = L.bind
        self.fused_step = self._fused_ok(0, d, n_pref) and self._fused_ok(1, d)
        if self.fused_step:
            inv = 1.0 / self.world
            self._rec_fused = b('ktup_train_rec_step', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(self.i2e), self.ent_pad,
                                _p(P), _p(Pn), _p(R), _p(Rn), P.stride(0), n_pref, d, _p(self.u2), _p(self.i2), B, self.l1, gate, gptr, 0, 0,
                                self.target, inv, 1, _p(self.loss), _p(U.grad), _p(I.grad), _p(E.grad), _p(P.grad), _p(Pn.grad),
                                _p(R.grad), _p(Rn.grad), self._gn_ptr(), st)
            self._kg_fused = b('ktup_train_kg_step', 1, _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.h2),
                               _p(self.t2), _p(self.r2), B, self.l1, self.margin, self.kg_lambda, 7, _p(self.loss), _p(E.grad),
                               _p(R.grad), _p(Rn.grad), self._gn_ptr(), st)
        self._rec_head = [
            b('ktup_pref_prepare', _p(P), _p(Pn), _p(R), _p(Rn), P.stride(0), n_pref, d, _p(self.ws), st)]
        self._rec_soft = [
            b('ktup_score_ktup_fwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(self.i2e), _p(self.ws), n_pref, d,
              _p(self.u2), _p(self.i2), 2 * B, self.l1, gate, gptr, 0, 0, _p(self.score), st)]
        # value + gradient in one launch; the loss slots are zeroed once at the top of a step and accumulated into
        self._rec_loss = [
            b('ktup_loss_bpr_fused', _p(pos), _p(neg), B, self.target, _p(self.inv_world), _p(self.loss[0:]), _p(gpos), _p(gneg), st)]
        self._rec_soft_bwd = [
            b('ktup_score_ktup_bwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(self.i2e), self.ent_pad,
              _p(self.ws), n_pref, d, _p(self.u2), _p(self.i2), 2 * B, self.l1, gate, gptr, 0, 0,
              _p(self.gscore), _p(U.grad), _p(I.grad), _p(E.grad), _p(self.gAC[0]), _p(self.gAC[1]), st)]
        self._rec_tail = [
            b('ktup_reg_orth_fused', _p(P), P.stride(0), _p(Pn), Pn.stride(0), d, None, n_pref, _p(self.inv_world), _p(self.loss[1:]),
              _p(P.grad), _p(Pn.grad), st)]
        self._kg = [
            b('ktup_score_transh_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), n_rel, d, _p(self.h2),
              _p(self.t2), _p(self.r2), 2 * B, self.l1, _p(self.score), st),
            b('ktup_loss_margin_fused', _p(pos), _p(neg), B, self.margin, _p(self.lam), _p(self.loss[0:]), _p(gpos), _p(gneg), st),
            b('ktup_score_transh_bwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.h2), _p(self.t2),
              _p(self.r2), 2 * B, self.l1, _p(self.gscore), _p(E.grad), _p(R.grad), _p(Rn.grad), st),
            b('ktup_reg_orth_fused', _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.r2), 2 * B, _p(self.lam), _p(self.loss[1:]),
              _p(R.grad), _p(Rn.grad), st),
            b('ktup_reg_norm_fused', _p(E), E.stride(0), d, _p(self.ht4), 4 * B, _p(self.lam), _p(self.loss[2:]), _p(E.grad), st),
            b('ktup_reg_norm_fused', _p(R), R.stride(0), d, _p(self.r2), 2 * B, _p(self.lam), _p(self.loss[3:]), _p(R.grad), st)]

    # ------------------------------------------------------------------------------------------------ rec
    def _rec_eager(self, u, pi, ni):
        m, B = self.m, self.B
        U, I, E, P, Pn, R, Rn = self.tabs
        self._plans()
        if u is not None:
            self._pack('rec', (u, pi, ni))
        if self.fused_step:              # one launch: table mixing, both scores, bprLoss, backward, gradient fan-out, orthogonalLoss
            self._rec_fused()
            self._gumbel_advance()
            if self.world > 1:
                self.loss[:2].mul_(self.inv_world)
            self._optimizer_launches(loss=(_p(self.loss), 2, 1.0, _p(self.out['rec']), self._acc_ptr('rec')), tracked=True)
            return self.out['rec']
        self._rec_head[0]()
        self.gAC.zero_(); self.loss.zero_()
        self._rec_soft[0](); self._rec_loss[0](); self._rec_soft_bwd[0]()
        self._gumbel_advance()
        # A = pref + rel and C = pref_norm + norm: the mixed-table gradient goes to both summands
        torch._foreach_add_([P.grad, R.grad, Pn.grad, Rn.grad], [self.gAC[0], self.gAC[0], self.gAC[1], self.gAC[1]])
        self._rec_tail[0]()
        if self.world > 1:
            self.loss[:2].mul_(self.inv_world)
        self._optimizer_launches()
        return self.loss[0] + self.loss[1]

    # ------------------------------------------------------------------------------------------------ kg
    def _kg_eager(self, ph, pt, pr, nh, nt, nr):
        self._plans()
        if ph is not None:
            self._pack('kg', (ph, pt, pr, nh, nt, nr))
        if self.fused_step:              # one launch: both TransH scores, marginLoss, the three regularisers, every gradient
            self._kg_fused()
            self._optimizer_launches(loss=(_p(self.loss), 4, self.kg_lambda, _p(self.out['kg']), self._acc_ptr('kg')), tracked=True)
            return self.out['kg']
        self.loss.zero_()
        for launch in self._kg:
            launch()
        self._optimizer_launches()
        return self.kg_lambda * self.loss[:4].sum()

    # ------------------------------------------------------------------------------------------------ public steps
    def rec_step(self, u, pi, ni):
        """u, pi, ni: int64 device tensors of the GLOBAL batch.  Returns the step's loss (0-dim device tensor)."""
        return self._step('rec', self._rec_eager, (u, pi, ni))

    def kg_step(self, ph, pt, pr, nh, nt, nr):
        return self._step('kg', self._kg_eager, (ph, pt, pr, nh, nt, nr))


class RecStepper(_StepperBase):
    """Rec-only driver (item_recommendation.py:160-195): TUP (transup) with its regularisers (:177-180), or BPRMF (bpr only).
    loss slots: 0 bpr, 1 orthogonalLoss(pref, pref_norm), 2 normLoss(user rows), 3 normLoss(item rows), 4 normLoss(pref)."""
    KINDS = ('rec',)
    N_IDS = {'rec': 3}

    def _setup(self, FLAGS, f32, i64):
        model, B = self.m, self.B
        self.tup = hasattr(model, 'pref_embeddings')
        if self.tup:
            U, I, P, Pn = model._tables()
            self.tabs = (U, I, P, Pn)
            self.gAC = torch.zeros(2, P.shape[0], P.shape[1], **f32)
            self.ws = ops.pref_workspace(P, Pn)
        else:
            self.tabs = (model.user_embeddings.weight, model.item_embeddings.weight)
        self._id_buffers('rec', i64)
        self._gumbel_stream(2 * B * self.tabs[2].shape[0] if self.tup else 0)

    def _bind(self, st):
        B, b = self.B, L.bind
        U, I = self.tabs[0], self.tabs[1]
        d = U.shape[1]
        pos, neg, gpos, gneg = self.score[:B], self.score[B:], self.gscore[:B], self.gscore[B:]
        self._loss = [b('ktup_loss_bpr_fused', _p(pos), _p(neg), B, self.target, _p(self.inv_world), _p(self.loss[0:]), _p(gpos),
                        _p(gneg), st)]
        self.fused_step = False
        if not self.tup:
            self._fused_ok(-1, d)                          # (raises under KTUP_DETERMINISTIC=1: BPRMF steps are several launches with plain atomics)
            self._fwd = b('ktup_score_bprmf_fwd', _p(U), U.stride(0), _p(I), I.stride(0), d, _p(self.u2), _p(self.i2), 2 * B,
                          _p(self.score), st)
            self._bwd = b('ktup_score_bprmf_bwd', _p(U), U.stride(0), _p(I), I.stride(0), d, _p(self.u2), _p(self.i2), 2 * B,
                          _p(self.gscore), _p(U.grad), _p(I.grad), st)
            return
        P, Pn = self.tabs[2], self.tabs[3]
        n_pref = P.shape[0]
        gate, gptr = self._gate_args()
        self.fused_step = self._fused_ok(0, d, n_pref)
        if self.fused_step:
            self._rec_fused = b('ktup_train_rec_step', _p(U), U.stride(0), _p(I), I.stride(0), None, 0, None, -1, _p(P), _p(Pn), None, None,
                                P.stride(0), n_pref, d, _p(self.u2), _p(self.i2), B, self.l1, gate, gptr, 0, 0, self.target, 1.0 / self.world,
                                1, _p(self.loss), _p(U.grad), _p(I.grad), None, _p(P.grad), _p(Pn.grad), None, None, None, st)   # (no tracked
                                # norm: the row regularisers below add to the gradients after this launch)
        self._prep = b('ktup_pref_prepare', _p(P), _p(Pn), None, None, P.stride(0), n_pref, d, _p(self.ws), st)
        self._fwd = b('ktup_score_tup_fwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(self.ws), n_pref, d, _p(self.u2), _p(self.i2),
                      2 * B, self.l1, gate, gptr, 0, 0, _p(self.score), st)
        self._bwd = b('ktup_score_tup_bwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(self.ws), n_pref, d, _p(self.u2), _p(self.i2),
                      2 * B, self.l1, gate, gptr, 0, 0, _p(self.gscore), _p(U.grad), _p(I.grad), _p(self.gAC[0]), _p(self.gAC[1]), st)
        self._regs = [
            b('ktup_reg_orth_fused', _p(P), P.stride(0), _p(Pn), Pn.stride(0), d, None, n_pref, _p(self.inv_world), _p(self.loss[1:]),
              _p(P.grad), _p(Pn.grad), st),
            b('ktup_reg_norm_fused', _p(U), U.stride(0), d, _p(self.u2), B, _p(self.one), _p(self.loss[2:]), _p(U.grad), st),      # the step's users once
            b('ktup_reg_norm_fused', _p(I), I.stride(0), d, _p(self.i2), 2 * B, _p(self.one), _p(self.loss[3:]), _p(I.grad), st),  # pos and neg items
            b('ktup_reg_norm_fused', _p(P), P.stride(0), d, None, n_pref, _p(self.inv_world), _p(self.loss[4:]), _p(P.grad), st)]

    def _rec_eager(self, u, pi, ni):
        m, B = self.m, self.B
        self._plans()
        if u is not None:
            self._pack('rec', (u, pi, ni))
        self.loss.zero_()
        if not self.tup:
            self._fwd(); self._loss[0](); self._bwd()
            if self.world > 1:
                self.loss[:1].mul_(self.inv_world)
            self._optimizer_launches()
            return self.loss[0] + 0.0
        U, I, P, Pn = self.tabs
        if self.fused_step:              # scores, bprLoss, backward and orthogonalLoss in one launch; the row regularisers follow
            self._rec_fused()
            self._gumbel_advance()
            for launch in self._regs[1:]:
                launch()
            if self.world > 1:
                self.loss[:2].mul_(self.inv_world); self.loss[4:5].mul_(self.inv_world)
            self._optimizer_launches(loss=(_p(self.loss), 5, 1.0, _p(self.out['rec']), self._acc_ptr('rec')))
            return self.out['rec']
        self._prep()
        self.gAC.zero_()
        self._fwd(); self._loss[0](); self._bwd()
        self._gumbel_advance()
        torch._foreach_add_([P.grad, Pn.grad], [self.gAC[0], self.gAC[1]])
        for launch in self._regs:
            launch()
        if self.world > 1:       # batch mean and whole-table terms by 1/G, batch sums as they are
            self.loss[:2].mul_(self.inv_world); self.loss[4:5].mul_(self.inv_world)
        self._optimizer_launches()
        return self.loss[:5].sum()

    def rec_step(self, u, pi, ni):
        return self._step('rec', self._rec_eager, (u, pi, ni))


class KGStepper(_StepperBase):
    """KG-only driver (knowledge_representation.py:176-211) for TransE / TransH / TransR: marginLoss + normLoss(entity rows
    of the positive and negative triples) + normLoss(relation rows) (+ orthogonalLoss(rel, norm) rows for TransH)."""
    KINDS = ('kg',)
    N_IDS = {'kg': 6}

    def _setup(self, FLAGS, f32, i64):
        model, B = self.m, self.B
        self.transh = hasattr(model, 'norm_embeddings')
        self.transr = hasattr(model, 'proj_embeddings')
        E, R = model.ent_embeddings.weight, model.rel_embeddings.weight
        self.tabs = (E, R, model.norm_embeddings.weight) if self.transh else (E, R, model.proj_embeddings.weight) if self.transr else (E, R)
        if self.transr:             # scratch of the relation-bucketed forward (K4)
            nbytes = L.load().ktup_score_transr_workspace_bytes(2 * B, R.shape[0])
            self.rws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=self.dev)
        self._id_buffers('kg', i64)

    def _bind(self, st):
        B, b = self.B, L.bind
        E, R = self.tabs[0], self.tabs[1]
        d = E.shape[1]
        pos, neg, gpos, gneg = self.score[:B], self.score[B:], self.gscore[:B], self.gscore[B:]
        calls = []
        self.fused_step = self._fused_ok(-1 if self.transr else (1 if self.transh else 2), d)      # (-1: no fused kernel; TransR's step stays multi-launch)
        if self.fused_step:
            Rn_ = self.tabs[2] if self.transh else None
            self._kg_fused = b('ktup_train_kg_step', int(self.transh), _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn_),
                               Rn_.stride(0) if self.transh else 0, d, _p(self.h2), _p(self.t2), _p(self.r2), B, self.l1, self.margin, 1.0,
                               7 if self.transh else 6, _p(self.loss), _p(E.grad), _p(R.grad), _p(Rn_.grad) if self.transh else None,
                               self._gn_ptr(), st)
        if self.transh:
            Rn = self.tabs[2]
            n_rel = min(R.shape[0], Rn.shape[0])
            calls.append(b('ktup_score_transh_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), n_rel, d, _p(self.h2),
                           _p(self.t2), _p(self.r2), 2 * B, self.l1, _p(self.score), st))
        elif self.transr:
            M = self.tabs[2]
            calls.append(b('ktup_score_transr_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), min(R.shape[0], M.shape[0]),
                           d, _p(self.h2), _p(self.t2), _p(self.r2), 2 * B, self.l1, _p(self.score), _p(self.rws), st))
        else:
            calls.append(b('ktup_score_transe_fwd', _p(E), E.stride(0), _p(R), R.stride(0), R.shape[0], d, _p(self.h2), _p(self.t2), _p(self.r2),
                           2 * B, self.l1, _p(self.score), st))
        calls.append(b('ktup_loss_margin_fused', _p(pos), _p(neg), B, self.margin, _p(self.one), _p(self.loss[0:]), _p(gpos), _p(gneg), st))
        if self.transh:
            calls += [b('ktup_score_transh_bwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.h2), _p(self.t2),
                        _p(self.r2), 2 * B, self.l1, _p(self.gscore), _p(E.grad), _p(R.grad), _p(Rn.grad), st),
                      b('ktup_reg_orth_fused', _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.r2), 2 * B, _p(self.one),
                        _p(self.loss[1:]), _p(R.grad), _p(Rn.grad), st)]
        elif self.transr:
            calls.append(b('ktup_score_transr_bwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), d, _p(self.h2), _p(self.t2),
                           _p(self.r2), 2 * B, self.l1, _p(self.gscore), _p(E.grad), _p(R.grad), _p(M.grad), st))
        else:
            calls.append(b('ktup_score_transe_bwd', _p(E), E.stride(0), _p(R), R.stride(0), d, _p(self.h2), _p(self.t2), _p(self.r2),
                           2 * B, self.l1, _p(self.gscore), _p(E.grad), _p(R.grad), st))
        calls += [b('ktup_reg_norm_fused', _p(E), E.stride(0), d, _p(self.ht4), 4 * B, _p(self.one), _p(self.loss[2:]), _p(E.grad), st),
                  b('ktup_reg_norm_fused', _p(R), R.stride(0), d, _p(self.r2), 2 * B, _p(self.one), _p(self.loss[3:]), _p(R.grad), st)]
        self._calls = calls

    def _kg_eager(self, ph, pt, pr, nh, nt, nr):
        self._plans()
        if ph is not None:
            self._pack('kg', (ph, pt, pr, nh, nt, nr))
        if self.fused_step:
            self._kg_fused()
            self._optimizer_launches(loss=(_p(self.loss), 4, 1.0, _p(self.out['kg']), self._acc_ptr('kg')), tracked=True)
            return self.out['kg']
        self.loss.zero_()
        for launch in self._calls:
            launch()
        self._optimizer_launches()
        return self.loss[:4].sum()

    def kg_step(self, ph, pt, pr, nh, nt, nr):
        return self._step('kg', self._kg_eager, (ph, pt, pr, nh, nt, nr))


class DeviceFeeder(object):
    """Device-resident training data for the GPU-resident loop: the rating / triple lists as device tensors, an epoch
    permutation drawn on the device, batches as slices, negatives from the K19 samplers (utils/device_sampler.py).
    Iterator contract of utils/data.py MakeTrainIterator (data.py:87-110): endless, the order list holds `negtive_samples`
    copies of every example, and -- exactly like the reference -- it wraps (reshuffles) once `start > n - batch_size` with
    n = the number of DISTINCT examples, so only the first n entries of each shuffled order are consumed, and the tail
    partial batch is dropped.  -device_sampling and -nodevice_sampling therefore share one epoch / shuffle cadence."""

    def __init__(self, rows, batch_size, device, negtive_samples=1, seed=0):
        rows = torch.as_tensor(rows, dtype=torch.int64).reshape(len(rows), -1)[:, :3].to(device)
        self.B, self.dev, self.rep = int(batch_size), device, int(negtive_samples)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.n = rows.shape[0]
        if self.n < self.B:
            # the reference slices a batch out of the replicated order list even when n < B <= n * negtive_samples
            # (data.py:87-110); here an epoch's columns hold the n distinct examples only, so a full batch must fit in them
            raise ValueError('fewer distinct training examples (%d) than one batch (%d): lower -batch_size (negtive_samples does '
                             'not add rows to an epoch of the device-resident feeder)' % (self.n, self.B))
        # column-major: a batch is then a contiguous SLICE of this epoch's shuffled columns (no per-step gather launches), and
        # the columns live in fixed storage, reshuffled in place -- what the feed kernels' graph-static arguments point at
        self.src = [rows[:, c].contiguous() for c in range(rows.shape[1])]
        self.cols = [torch.empty(self.n, dtype=torch.int64, device=device) for _ in self.src]
        self.cursor = torch.zeros(1, dtype=torch.int64, device=device)      # device copy of `start` (ktup_feed_* advance it)
        self._shuffle()

    def _shuffle(self):
        self.order = order = (torch.randperm(self.n * self.rep, generator=self.gen, device=self.dev) % self.n)[:self.n]
        for src, col in zip(self.src, self.cols):
            torch.index_select(src, 0, order, out=col)
        self.start = 0
        self.cursor.zero_()
        self._cursor_ok = True

    def _take(self):
        if self.start > self.n - self.B:              # data.py:101-103: dataset_size, not len(order)
            self._shuffle()
        s = self.start
        self.start += self.B
        return s

    def next_cols(self):
        """The next batch as one contiguous 1-D view per column (no launches)."""
        s = self._take()
        self._cursor_ok = False
        return tuple(col[s:s + self.B] for col in self.cols)

    def next(self):
        return torch.stack(self.next_cols(), dim=1)

    def fed(self):
        """Account for one batch a ktup_feed_* launch is about to take (it moves the device cursor the same way)."""
        s = self._take()
        if not self._cursor_ok:
            self.cursor.fill_(s)
            self._cursor_ok = True
