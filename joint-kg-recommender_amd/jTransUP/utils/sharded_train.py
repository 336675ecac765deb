"""-shard_tables: KTUP's joint training loop (knowledgable_recommendation.py:319-403) on ROW-SHARDED user / item / entity tables
(BASELINE config 5): every rank owns rows {g : g % world == rank} of the three big tables and their optimizer state, the four
preference-side tables (a few dozen rows) stay replicated; a step is sharded_ktup.ShardedKtupJoint's -- fixed-shape exchange,
fused step kernels, row-sparse optimizer on exactly the touched rows, replayed as HIP graphs.  The reference is single-device:
this module only adapts that stepper to the driver's loop (same batches, same step counting, same logging, evaluation and
checkpointing as the replicated route of utils/fast_train.py).

    torchrun --nproc-per-node 8 run_knowledgable_recommendation.py -model_type jtransup -noshare_embeddings -shard_tables \
        -optimizer_type Adam -l2_lambda 0 -L1_flag -seed 7 ...                    # the recipe of the reference's ktup.sh

What the rows' owners do NOT keep is a dense gradient or a dense optimizer pass.  Under Adagrad / plain SGD without weight decay
(l2_lambda = 0) a row that no batch touches never moves, so updating the touched rows IS the reference's dense step.  Under Adam a dense
step moves every row that ever had a gradient, and under weight decay (-l2_lambda, 1e-5 by default: base.py:51) every optimizer moves
every row at every step (g = l2_lambda * p): the steppers replay the steps a row has missed before the row is read again
(sharded_ktup.py, include/ktup_hip.h ktup_adam_t: `rule` and `weight_decay`), `flush()` does it for all rows before an evaluation or a
checkpoint -- the same tables as torch.optim's dense optimizers over whole tables.  (The replay under weight decay is one optimizer step
per missed step and row: fine where rows come round every few steps -- ml1m --, the wrong tool for tables of millions of rows that a
batch visits every thousand steps -- there -l2_lambda 0, like the published recipe.)  The ST-Gumbel gate (-use_st_gumbel, the gate of
transup.sh) draws its noise on the device, every rank from its own Philox stream.  Other settings are refused by name.  Every
rank draws the same global batches (same -seed, like the replicated route) and takes its slice.

Memory: the shards are the ONLY resident copy of the three big tables.  Once they are built the model's own whole tables are released
(zero-row placeholders) and so is the dense optimizer's state for them; a rank holds rows/world x (table + optimizer state).
Evaluation runs ON the shards (`rec_shard` / `kg_shard`): a rank's candidates are the rows it owns -- the lattice rank + world * j of the
catalogue, ranked with the strided forms of parallel.sharded_topk / sharded_gold_ranks (order within a shard = global id order, so
ties fall as in the whole-table walk) -- the B query rows of a batch come from their owners by one all-reduce of a (B x d) buffer that
is zero elsewhere, and the entity rows of a rank's items are fetched once per evaluation.  Whole tables exist only transiently:
for a whole-table checkpoint (`-shard_whole_checkpoint`, on by default so that `-load_experiment_name` / `-eval_only_mode` keep working;
off: shard files only) and for the per-user report mode, gathered by `sync_model` and released right after.  The whole-table
checkpoint carries the gathered optimizer state (Adagrad sums / Adam moments) in torch.optim's layout; next to every checkpoint each
rank writes `<file>.shard<rank>of<world>` with its rows, their optimizer state and the step counter -- a run started with
`-load_experiment_name <file>` picks its shard file up when there is one (exact resume), else shards the whole tables it loaded."""
import os

import torch
import torch.distributed as dist

from jTransUP import parallel
from jTransUP.hip import lib as L
from jTransUP.sharded_ktup import ShardedKtupJoint

BIG = ('user_embeddings', 'item_embeddings', 'ent_embeddings')
SMALL = ('pref_embeddings', 'pref_norm_embeddings', 'rel_embeddings', 'norm_embeddings')
TUP_BIG, TUP_SMALL = BIG[:2], SMALL[:2]            # -model_type transup (run_item_recommendation.py): no entity side


def check_flags(FLAGS, model):
    """Everything -shard_tables cannot do is refused here, by the reference's flag names."""
    if FLAGS.model_type == 'transup':
        pass                                         # TUP: user / item tables sharded, the preference tables replicated
    elif FLAGS.model_type != 'jtransup' or getattr(FLAGS, 'share_embeddings', False):
        raise L.KtupError('-shard_tables trains transup, or jtransup with its own tables (-model_type jtransup -noshare_embeddings)')
    if FLAGS.optimizer_type not in ('Adagrad', 'SGD', 'Adam') or (FLAGS.optimizer_type == 'SGD' and FLAGS.momentum != 0):
        raise L.KtupError('-shard_tables updates only the rows a batch touches: exact for -optimizer_type Adagrad, SGD with -momentum 0, '
                          'or Adam (whose untouched steps are replayed when a row is touched again)')
    if FLAGS.l2_lambda < 0:
        raise L.KtupError('-l2_lambda must not be negative')
    tup = FLAGS.model_type == 'transup'
    d, P = model.embedding_size, (model.pref_embeddings.weight.shape[0] if tup else model.rel_total)
    if not L.load().ktup_train_step_supported(0, d, P) or not (tup or L.load().ktup_train_step_supported(1, d, P)):
        raise L.KtupError('-shard_tables: no fused step kernels for -embedding_size %d with %d preferences' % (d, P))


@torch.no_grad()
def gather_table(full, rows, total_rows, world, group=None):
    """Rows {g : g % world == r} of a table live on rank r (`rows`: this rank's, in increasing g): write every rank's rows into
    `full` (total_rows x d, the same on every rank afterwards).  One all-gather of equal-sized blocks -- the ranks with one row
    fewer pad theirs.  Plain torch + torch.distributed: runs on CPU tensors over gloo as well (tests/test_parallel_gloo.py)."""
    if world == 1:
        full.copy_(rows)
        return full
    n_max = (total_rows + world - 1) // world
    mine = torch.zeros(n_max, full.shape[1], dtype=full.dtype, device=full.device)
    mine[:rows.shape[0]] = rows
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    for r, part in enumerate(parts):
        n_r = (total_rows - r + world - 1) // world
        full[r::world] = part[:n_r]
    return full


class _RecOnly(object):
    """What the driver needs of ShardedKtupJoint when there is no kg half (TUP)."""

    def __init__(self, rec):
        self.rec, self.kg, self.steps = rec, None, 0

    def flush(self):
        self.rec.flush()

    def check(self):
        self.rec.check()

    def close(self):
        self.rec.close()


class ShardedJointDriver(object):
    """The stepper interface of utils/fast_train.JointStepper (GB, rec_step, kg_step, take_sums, can_feed) over ShardedKtupJoint."""

    def __init__(self, model, trainer, FLAGS, batch_size, logger=None, group=None):
        check_flags(FLAGS, model)
        self.m, self.trainer, self.FLAGS, self.group, self.logger = model, trainer, FLAGS, group, logger
        self.tup = FLAGS.model_type == 'transup'
        self.BIG, self.SMALL = (TUP_BIG, TUP_SMALL) if self.tup else (BIG, SMALL)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if int(batch_size) % self.world:
            raise L.KtupError('batch_size %d is not divisible by the %d ranks' % (batch_size, self.world))
        self.GB, self.B = int(batch_size), int(batch_size) // self.world
        dev = model.user_embeddings.weight.device
        self.dev = dev
        if self.world > 1:                           # identical tables to start from (the replicated route does the same)
            for p in model.parameters():
                dist.broadcast(p.data, src=0, group=group)
        self.tables = []
        for name in self.BIG:
            full = getattr(model, name).weight.data
            t = parallel.ShardedTable(full.shape[0], full.shape[1], rank=self.rank, world=self.world, group=group, device=dev,
                                      init=lambda g, full=full: full[g.to(full.device)])
            self.tables.append(t)
        self.small = [getattr(model, name).weight for name in self.SMALL]
        self.kind = FLAGS.optimizer_type.lower()
        self.weight_decay = float(getattr(FLAGS, 'l2_lambda', 0.0) or 0.0)
        from jTransUP.sharded_ktup import is_lazy
        self.lazy = is_lazy(self.kind, self.weight_decay)          # state rows [m | v | last] + catch-up of the untouched steps
        self._lr = None
        self._build(float(trainer.learning_rate))
        self.acc = {'rec': 0.0, 'kg': 0.0}
        self._import_optimizer()                     # a whole-table checkpoint was loaded: its sums / moments go to the shards
        self._whole = True                           # the model's own big tables are allocated ...
        self._dirty = False                          # ... and equal the shards
        self.release_model()
        if logger is not None:
            logger.info('The shards are the only resident copy of the big tables: rank %d holds %s of %s user / item / entity rows (and their '
                        'optimizer state); the model\'s own tables are released.'
                        % (self.rank, ' / '.join(str(t.weight.shape[0]) for t in self.tables), ' / '.join(str(t.total_rows) for t in self.tables)))
        self._item_ent = None                        # (n_my_items + 1, d): the entity rows of this rank's items, per evaluation
        model._shard_native = self                   # models/knowledgable_recommendation.py evaluates on the shards
        self._wrap_trainer()
        resume = getattr(FLAGS, 'load_experiment_name', None)
        if resume:
            # as given first -- what ModelTrainer does with the same flag (trainer.py:52) --, then under -log_path
            path = resume if (os.path.isabs(resume) or os.path.isfile(self.shard_file(resume)) or os.path.isfile(resume)) \
                else os.path.join(FLAGS.log_path, resume)
            if os.path.isfile(self.shard_file(path)):
                self.load_shards(path)
                if logger is not None:
                    logger.info('Restored rank %d\'s shard (rows, optimizer state, step counter) from %s.' % (self.rank, self.shard_file(path)))
            elif getattr(FLAGS, 'eval_only_mode', False) and not os.path.isfile(path) and not os.path.isfile(resume):
                raise L.KtupError('-eval_only_mode -load_experiment_name %s: neither that file nor %s exists -- randomly initialised tables would '
                                  'be evaluated' % (resume, self.shard_file(path)))

    def _build(self, lr):
        F = self.FLAGS
        if self.tup:     # TUP's rec step alone (item_recommendation.py:160-192), with its row regularisers
            from jTransUP.sharded_ktup import ShardedKtupStepper
            self.joint = _RecOnly(ShardedKtupStepper(self.tables[0], self.tables[1], None, self.small[0], self.small[1], None, None, None,
                                                     batch=self.B, kind=self.kind, lr=lr, max_norm=F.clipping_max_value,
                                                     eps=1e-8 if self.kind == 'adam' else 1e-10, l1=bool(F.L1_flag),
                                                     target=float(self.trainer.model_target), orth=True, row_regs=True, group=self.group,
                                                     capacity_factor=float(getattr(F, 'shard_capacity_factor', 1.25)),
                                                     weight_decay=self.weight_decay, use_st_gumbel=bool(getattr(F, 'use_st_gumbel', False)),
                                                     gumbel_seed=int(getattr(F, 'seed', 0) or 0)))
            self._lr = lr
            self._base = {'rec': torch.zeros(4), 'kg': torch.zeros(4)}
            return
        self.joint = ShardedKtupJoint.build(*self.tables, *self.small, self.m._item2ent, batch=self.B, joint_ratio=F.joint_ratio,
                                            margin=F.margin, kg_lambda=F.kg_lambda, kind=self.kind, lr=lr, max_norm=F.clipping_max_value,
                                            eps=1e-8 if self.kind == 'adam' else 1e-10,       # torch.optim's defaults (utils/trainer.py:63-77 passes none)
                                            l1=bool(F.L1_flag), target=float(self.trainer.model_target), orth=True,
                                            ent_pad=self.m.ent_total - 1, group=self.group,
                                            capacity_factor=float(getattr(F, 'shard_capacity_factor', 1.25)),
                                            weight_decay=self.weight_decay, use_st_gumbel=bool(getattr(F, 'use_st_gumbel', False)),
                                            gumbel_seed=int(getattr(F, 'seed', 0) or 0))
        self._lr = lr
        self._base = {'rec': torch.zeros(2), 'kg': torch.zeros(4)}

    def _follow_trainer(self):
        """ModelTrainer lowers the learning rate by building a NEW optimizer (utils/trainer.py:107-110): accumulators start over."""
        lr = float(self.trainer.learning_rate)
        if lr != self._lr:
            self._collect()
            self.joint.check()                       # the steppers about to be replaced carry the skipped-step counters
            self.joint.flush()                       # Adam: the moves the old optimizer still owes the rows it has not touched lately
            for t in self.tables:
                if t.state is not None:
                    t.state.zero_()
            self._build(lr)                          # fresh small-table sums, launches rebound with the new rate

    def _mine(self, t):
        return t if self.world == 1 else t[self.rank * self.B:(self.rank + 1) * self.B]

    def can_feed(self, kind):
        return False

    def rec_step(self, u, pi, ni):
        """u, pi, ni: int64 device tensors of the GLOBAL batch (every rank draws the same one); this rank scores its slice."""
        self._follow_trainer()
        self.joint.rec.load_batch(self._mine(u), self._mine(pi), self._mine(ni))
        self.joint.rec.run()
        self.joint.steps += 1
        self.trainer.step += 1
        self._dirty = True
        return None

    def kg_step(self, ph, pt, pr, nh, nt, nr):
        self._follow_trainer()
        self.joint.kg.load_batch(*(self._mine(x) for x in (ph, pt, pr, nh, nt, nr)))
        self.joint.kg.run()
        self.joint.steps += 1
        self.trainer.step += 1
        self._dirty = True
        return None

    def _collect(self):
        """Fold the steppers' device-side loss sums into the host totals (one sync).  rec: the batch-mean BPR term averaged over the
        ranks + orthogonalLoss(pref, pref_norm) (identical on every rank); kg: kg_lambda x the sum over all ranks' triples."""
        rec = self.joint.rec.loss_sum.detach().cpu() - self._base['rec']
        kg = torch.zeros(4) if self.joint.kg is None else self.joint.kg.loss_sum.detach().cpu() - self._base['kg']
        self._base['rec'] += rec; self._base['kg'] += kg
        # TUP's row regularisers: every rank's own rows' terms and its 1 / world share of normLoss(pref) -- both add up over the ranks
        rows = float(rec[2]) + float(rec[3]) if rec.numel() > 2 else 0.0
        vals = torch.tensor([float(rec[0]) / self.world + rows, float(kg.sum()) * getattr(self.FLAGS, 'kg_lambda', 1.0)], dtype=torch.float64)
        if self.world > 1:
            v = vals.to(self.dev)
            dist.all_reduce(v, group=self.group)
            vals = v.cpu()
        self.acc['rec'] += float(vals[0]) + float(rec[1])
        self.acc['kg'] += float(vals[1])

    def take_sums(self):
        self._collect()
        self.joint.check()                           # a skipped (overflowed) step raises here, where the loop syncs anyway
        out, self.acc = self.acc, {'rec': 0.0, 'kg': 0.0}
        return out

    # ------------------------------------------------------------------------------------------------ evaluation on the shards
    @torch.no_grad()
    def rows_everywhere(self, t, ids):
        """Rows `ids` (global, int64 device) of sharded table `t` on every rank: each rank fills the rows it owns into a zero buffer,
        one all-reduce sums them (x + 0 = x exactly)."""
        buf = torch.zeros(ids.numel(), t.d, dtype=torch.float32, device=self.dev)
        if self.world == 1:
            return t.weight.data.index_select(0, ids)
        mine = (ids % self.world) == self.rank
        buf[mine] = t.weight.data.index_select(0, torch.div(ids[mine], self.world, rounding_mode='floor'))
        parallel._all_reduce(buf, self.group)
        return buf

    @torch.no_grad()
    def begin_eval(self):
        """Before an evaluation reads the shards: Adam's pending zero-gradient steps (flush), and the entity rows of this rank's items
        -- E[item2ent[rank + world j]], the operand jTransUP.py:122-130 adds to the item row -- fetched from their owners once."""
        self.joint.flush()
        if self.tup:
            self._item_ent = True                    # (nothing to fetch: no entity side)
            return
        i2e = self.m._eval_item2ent.long()
        Et = self.tables[2]
        mine = None
        for q in range(self.world):                  # rank q's items need entity rows from everywhere: one summed buffer per rank
            rows = self.rows_everywhere(Et, i2e[q::self.world].contiguous())
            if q == self.rank:
                mine = rows
        self._item_ent = torch.cat([mine, torch.zeros(1, Et.d, dtype=torch.float32, device=self.dev)])

    def rec_shard(self):
        """(n_items, f(u_ids) -> (B, n_my_items) scores, ('lattice', rank, world)) for models/_driver.rec_eval_pass: transUP / jTransUP's
        evaluateRec (jTransUP.py:163-191) with this rank's item rows as the candidates."""
        from jTransUP.hip import ops
        m, Ut, It = self.m, self.tables[0], self.tables[1]
        P, Pn = self.small[0].data, self.small[1].data
        R, Rn = (None, None) if self.tup else (self.small[2].data, self.small[3].data)
        n_my = It.weight.shape[0]
        local_map = torch.arange(n_my, dtype=torch.int32, device=self.dev)       # item row j <-> row j of the fetched entity rows

        def f(u_ids):
            if self._item_ent is None:
                self.begin_eval()
            Uq = self.rows_everywhere(Ut, u_ids)
            q = torch.arange(u_ids.numel(), dtype=torch.int64, device=self.dev)
            # (-use_st_gumbel: the reference draws noise in evaluate too, transUP.py:92 -- here from the model's own Philox stream)
            mode, uni, seed, off = m._gumbel.mode_and_stream(m.use_st_gumbel, None, u_ids.numel() * n_my * P.shape[0])
            if self.tup:                             # transUP.py:84-102 with this rank's item rows as the candidates
                return ops.eval_tup(Uq, It.weight.data, P, Pn, q, m.L1_flag, mode, uni, seed, off)
            return ops.eval_ktup(Uq, It.weight.data, self._item_ent, P, Pn, R, Rn, local_map, q, m.L1_flag, mode, uni, seed, off)
        return It.total_rows, f, ('lattice', self.rank, self.world)

    def kg_shard(self, head):
        """(n_entities, f(q_ids, r_ids) -> (B, n_my_entities) scores, lattice): evaluateHead / evaluateTail (jTransUP.py:193-247: TransH
        over the entity table INCLUDING its pad row) with this rank's entity rows as the candidates."""
        from jTransUP.hip import ops
        m, Et = self.m, self.tables[2]
        R, N = self.small[2].data, self.small[3].data

        def f(q_ids, r_ids):
            if self._item_ent is None:
                self.begin_eval()
            Eq = self.rows_everywhere(Et, q_ids)
            q = torch.arange(q_ids.numel(), dtype=torch.int64, device=self.dev)
            return ops.eval_transh(Eq, R, N, q, r_ids, m.L1_flag, head, candidates=Et.weight.data)
        return Et.total_rows, f, ('lattice', self.rank, self.world)

    def end_eval(self):
        self._item_ent = None

    # ------------------------------------------------------------------------------------------------ whole tables <-> shards
    @torch.no_grad()
    def sync_model(self):
        """Gather every shard into the model's whole tables -- TRANSIENTLY, for a whole-table checkpoint or the per-user report mode;
        `release_model()` frees them again (the training steps and the evaluation never need them)."""
        if self._whole and not self._dirty:
            return
        self.joint.flush()                           # Adam: every row up to the current step, as the dense optimizer would hold it
        for name, t in zip(self.BIG, self.tables):
            w = getattr(self.m, name).weight
            if w.data.shape[0] != t.total_rows:
                w.data = torch.empty(t.total_rows, t.d, dtype=torch.float32, device=self.dev)
            gather_table(w.data, t.weight.data, t.total_rows, self.world, self.group)
        self._whole, self._dirty = True, False

    @torch.no_grad()
    def release_model(self):
        """Drop the model's whole big tables (zero-row placeholders keep the module structure) and the dense optimizer's state for them."""
        for name in self.BIG:
            w = getattr(self.m, name).weight
            w.data = torch.empty(0, w.data.shape[1], dtype=torch.float32, device=self.dev)
            w.grad = None
            opt = getattr(self.trainer, 'optimizer', None)
            st = opt.state.get(w) if opt is not None else None
            if st:
                for k, v in list(st.items()):
                    if torch.is_tensor(v) and v.dim() == 2:
                        st[k] = torch.empty(0, v.shape[1], dtype=v.dtype, device=v.device)
        self._whole = False

    # ---- optimizer state in torch.optim's layout (what ModelTrainer.save writes and .load reads)
    _KEYS = {'adagrad': ('sum',), 'adam': ('exp_avg', 'exp_avg_sq'), 'sgd': ()}

    @torch.no_grad()
    def _export_optimizer(self):
        """Shards' optimizer state -> trainer.optimizer.state as WHOLE tensors (called with the whole tables materialised, before
        ModelTrainer.save): the whole-table checkpoint then holds what a dense run's would."""
        keys = self._KEYS[self.kind]
        opt, d = getattr(self.trainer, 'optimizer', None), self.tables[0].d
        if not keys or opt is None:
            return
        step = float(self.joint.rec.opt_step[0].item()) if self.lazy else float(self.trainer.step)
        block = {'adam': (0, 1), 'adagrad': (1,)}.get(self.kind, ())          # which d-wide block of a lazy state row [m | v | last] a key is
        pieces = [(getattr(self.m, n).weight, t.state, t.total_rows, True) for n, t in zip(self.BIG, self.tables)] + \
                 [(p, s, p.shape[0], False) for p, s in zip(self.small, self.joint.rec.small_state)]
        for p, st, rows, big in pieces:
            entry = opt.state[p]
            entry['step'] = torch.tensor(step)
            for k, key in enumerate(keys):
                part = st[:, block[k] * d:(block[k] + 1) * d].contiguous() if self.lazy else st
                full = torch.empty(rows, d, dtype=torch.float32, device=self.dev)
                entry[key] = gather_table(full, part, rows, self.world, self.group) if big else full.copy_(part)

    @torch.no_grad()
    def _import_optimizer(self):
        """trainer.optimizer.state (a whole-table checkpoint was loaded through ModelTrainer.load) -> the shards' state."""
        keys = self._KEYS[self.kind]
        opt, d = getattr(self.trainer, 'optimizer', None), self.tables[0].d
        if opt is None:
            return
        pieces = [(getattr(self.m, n).weight, t.state, True) for n, t in zip(self.BIG, self.tables)] + \
                 [(p, s, False) for p, s in zip(self.small, self.joint.rec.small_state)]
        step = 0
        block = {'adam': (0, 1), 'adagrad': (1,)}.get(self.kind, ())
        for p, st, big in pieces:
            entry = opt.state.get(p) or {}
            if not keys or any(key not in entry or entry[key].shape != p.shape for key in keys):
                continue
            step = max(step, int(float(entry.get('step', 0))))
            for k, key in enumerate(keys):
                src = entry[key].to(self.dev)
                src = src[self.rank::self.world] if big else src
                if self.lazy:
                    st[:, block[k] * d:(block[k] + 1) * d] = src
                else:
                    st.copy_(src)
            if self.lazy and step > 0:                   # every row is as the dense optimizer left it at `step`
                st[:, 2 * d] = torch.full((st.shape[0],), step, dtype=torch.int32, device=self.dev).view(torch.float32)
        if self.lazy and step > 0:
            self.joint.rec.opt_step[:1].fill_(step)

    @torch.no_grad()
    def load_from_model(self):
        """The model's whole tables -> this rank's rows (after pre-trained tables or a whole-table checkpoint were loaded)."""
        for name, t in zip(self.BIG, self.tables):
            t.weight.data.copy_(getattr(self.m, name).weight.data[self.rank::self.world])
        self._dirty = False

    def shard_file(self, filename):
        return '%s.shard%dof%d' % (filename, self.rank, self.world)

    def save_shards(self, filename):
        j = self.joint
        j.flush()
        tr = self.trainer
        torch.save({'rank': self.rank, 'world': self.world, 'step': self.trainer.step, 'joint_steps': j.steps, 'lr': self._lr,
                    'best_step': getattr(tr, 'best_step', 0), 'best_dev_performance': getattr(tr, 'best_dev_performance', 0.0),
                    'best_performances': getattr(tr, 'best_performances', None),
                    'opt_step': int(j.rec.opt_step[0].item()),
                    'rows': {n: t.weight.data.cpu() for n, t in zip(self.BIG, self.tables)},
                    'row_state': {n: (None if t.state is None else t.state.cpu()) for n, t in zip(self.BIG, self.tables)},
                    'small': {n: p.data.cpu() for n, p in zip(self.SMALL, self.small)},
                    'small_state': [None if s is None else s.cpu() for s in j.rec.small_state]}, self.shard_file(filename))

    @torch.no_grad()
    def load_shards(self, filename):
        ck = torch.load(self.shard_file(filename), map_location='cpu', weights_only=False)
        if ck['rank'] != self.rank or ck['world'] != self.world:
            raise L.KtupError('%s was written by rank %d of %d' % (self.shard_file(filename), ck['rank'], ck['world']))
        if ck['lr'] != self._lr:
            # the run had lowered its learning rate: the trainer follows FIRST (a fresh torch optimizer at that rate, like the run's own
            # decay), so that _follow_trainer sees no change on the next step and leaves the restored state alone
            if hasattr(self.trainer, 'optimizer_reset'):
                self.trainer.optimizer_reset(ck['lr'])
                self.release_model()                 # (the new dense optimizer holds no state yet; keep it that way for the big tables)
            else:
                self.trainer.learning_rate = ck['lr']
            self._build(ck['lr'])
        for n, t in zip(self.BIG, self.tables):
            t.weight.data.copy_(ck['rows'][n])
            if t.state is not None and ck['row_state'][n] is not None:
                t.state.copy_(ck['row_state'][n])
        for n, p in zip(self.SMALL, self.small):
            p.data.copy_(ck['small'][n])
        for s, v in zip(self.joint.rec.small_state, ck['small_state']):
            if s is not None and v is not None:
                s.copy_(v)                           # (the kg stepper shares the rel / norm sums)
        self.joint.rec.opt_step[:1].fill_(int(ck.get('opt_step', 0)))
        self.joint.steps = ck['joint_steps']
        self.trainer.step = ck['step']
        for key in ('best_step', 'best_dev_performance', 'best_performances'):     # what ModelTrainer.load restores (trainer.py:146-150):
            if key in ck and hasattr(self.trainer, key):                        # without it the first evaluation of a resumed run
                setattr(self.trainer, key, ck[key])                            # overwrites the best checkpoint unconditionally
        self._dirty = True

    def _wrap_trainer(self):
        """ModelTrainer.save writes the whole tables (gathered first) and, beside them, this rank's shard file."""
        trainer, inner = self.trainer, self.trainer.save

        def save(filename):
            if getattr(self.FLAGS, 'shard_whole_checkpoint', True):
                self.sync_model()                    # transient whole tables + the gathered optimizer state, in the reference's layout
                self._export_optimizer()
                inner(filename)
                self.release_model()
            self.save_shards(filename)
        trainer.save = save
