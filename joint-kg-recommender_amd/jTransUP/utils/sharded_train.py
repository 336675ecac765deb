"""-shard_tables: KTUP's joint training loop (knowledgable_recommendation.py:319-403) on ROW-SHARDED user / item / entity tables
(BASELINE config 5): every rank owns rows {g : g % world == rank} of the three big tables and their Adagrad sums, the four
preference-side tables (a few dozen rows) stay replicated; a step is sharded_ktup.ShardedKtupJoint's -- fixed-shape exchange,
fused step kernels, row-sparse optimizer on exactly the touched rows, replayed as HIP graphs.  The reference is single-device:
this module only adapts that stepper to the driver's loop (same batches, same step counting, same logging, evaluation and
checkpointing as the replicated route of utils/fast_train.py).

    torchrun --nproc-per-node 8 run_knowledgable_recommendation.py -model_type jtransup -noshare_embeddings -shard_tables \
        -optimizer_type Adagrad -l2_lambda 0 -seed 7 -shard_eval_candidates ...

What the rows' owners do NOT keep is a dense gradient or a dense optimizer pass: a row that no batch touches never moves, which
equals the reference's dense step exactly for Adagrad / plain SGD without weight decay (l2_lambda = 0) -- other settings are
refused by name.  Every rank draws the same global batches (same -seed, like the replicated route) and takes its slice.

Evaluation and checkpoints see whole tables: before either, the shards are gathered into the model's own parameters
(`sync_model`, one all-gather per table), so evaluateRec / evaluateKG (with -shard_eval_candidates: every rank scores its slice of
the catalogue) and ModelTrainer.save run unchanged; next to every checkpoint each rank also writes `<file>.shard<rank>of<world>`
with its rows AND their Adagrad sums (load_shards restores a run exactly; the whole-table file alone restarts the sums)."""
import os

import torch
import torch.distributed as dist

from jTransUP import parallel
from jTransUP.hip import lib as L
from jTransUP.sharded_ktup import ShardedKtupJoint

BIG = ('user_embeddings', 'item_embeddings', 'ent_embeddings')
SMALL = ('pref_embeddings', 'pref_norm_embeddings', 'rel_embeddings', 'norm_embeddings')


def check_flags(FLAGS, model):
    """Everything -shard_tables cannot do is refused here, by the reference's flag names."""
    if FLAGS.model_type != 'jtransup' or FLAGS.share_embeddings:
        raise L.KtupError('-shard_tables trains jtransup with its own tables (-model_type jtransup -noshare_embeddings)')
    if FLAGS.optimizer_type not in ('Adagrad', 'SGD', 'Adam') or (FLAGS.optimizer_type == 'SGD' and FLAGS.momentum != 0):
        raise L.KtupError('-shard_tables updates only the rows a batch touches: exact for -optimizer_type Adagrad, SGD with -momentum 0, '
                          'or Adam (whose untouched steps are replayed when a row is touched again)')
    if FLAGS.l2_lambda != 0:
        raise L.KtupError('-shard_tables needs -l2_lambda 0 (weight decay moves every row of every table on every step)')
    if FLAGS.use_st_gumbel:
        raise L.KtupError('-shard_tables has no ST-Gumbel step yet (-nouse_st_gumbel)')
    d, P = model.embedding_size, model.rel_total
    if not L.load().ktup_train_step_supported(0, d, P) or not L.load().ktup_train_step_supported(1, d, P):
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


class ShardedJointDriver(object):
    """The stepper interface of utils/fast_train.JointStepper (GB, rec_step, kg_step, take_sums, can_feed) over ShardedKtupJoint."""

    def __init__(self, model, trainer, FLAGS, batch_size, logger=None, group=None):
        check_flags(FLAGS, model)
        self.m, self.trainer, self.FLAGS, self.group, self.logger = model, trainer, FLAGS, group, logger
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
        for name in BIG:
            full = getattr(model, name).weight.data
            t = parallel.ShardedTable(full.shape[0], full.shape[1], rank=self.rank, world=self.world, group=group, device=dev,
                                      init=lambda g, full=full: full[g.to(full.device)])
            self.tables.append(t)
        self.small = [getattr(model, name).weight for name in SMALL]
        self.kind = FLAGS.optimizer_type.lower()
        self._lr = None
        self._build(float(trainer.learning_rate))
        self.acc = {'rec': 0.0, 'kg': 0.0}
        self._dirty = False                          # the model's whole tables lag behind the shards
        self._wrap_trainer()

    def _build(self, lr):
        F = self.FLAGS
        self.joint = ShardedKtupJoint.build(*self.tables, *self.small, self.m._item2ent, batch=self.B, joint_ratio=F.joint_ratio,
                                            margin=F.margin, kg_lambda=F.kg_lambda, kind=self.kind, lr=lr, max_norm=F.clipping_max_value,
                                            eps=1e-8 if self.kind == 'adam' else 1e-10,       # torch.optim's defaults (utils/trainer.py:63-77 passes none)
                                            l1=bool(F.L1_flag), target=float(self.trainer.model_target), orth=True,
                                            ent_pad=self.m.ent_total - 1, group=self.group)
        self._lr = lr
        self._base = {'rec': torch.zeros(2), 'kg': torch.zeros(4)}

    def _follow_trainer(self):
        """ModelTrainer lowers the learning rate by building a NEW optimizer (utils/trainer.py:107-110): accumulators start over."""
        lr = float(self.trainer.learning_rate)
        if lr != self._lr:
            self._collect()
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
        kg = self.joint.kg.loss_sum.detach().cpu() - self._base['kg']
        self._base['rec'] += rec; self._base['kg'] += kg
        vals = torch.tensor([float(rec[0]) / self.world, float(kg.sum()) * self.FLAGS.kg_lambda], dtype=torch.float64)
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

    # ------------------------------------------------------------------------------------------------ whole tables <-> shards
    @torch.no_grad()
    def sync_model(self):
        """Gather every shard into the model's whole tables (before an evaluation or a checkpoint)."""
        if not self._dirty:
            return
        self.joint.flush()                           # Adam: every row up to the current step, as the dense optimizer would hold it
        for name, t in zip(BIG, self.tables):
            gather_table(getattr(self.m, name).weight.data, t.weight.data, t.total_rows, self.world, self.group)
        self._dirty = False

    @torch.no_grad()
    def load_from_model(self):
        """The model's whole tables -> this rank's rows (after pre-trained tables or a whole-table checkpoint were loaded)."""
        for name, t in zip(BIG, self.tables):
            t.weight.data.copy_(getattr(self.m, name).weight.data[self.rank::self.world])
        self._dirty = False

    def shard_file(self, filename):
        return '%s.shard%dof%d' % (filename, self.rank, self.world)

    def save_shards(self, filename):
        j = self.joint
        j.flush()
        torch.save({'rank': self.rank, 'world': self.world, 'step': self.trainer.step, 'joint_steps': j.steps, 'lr': self._lr,
                    'opt_step': int(j.rec.opt_step.item()),
                    'rows': {n: t.weight.data.cpu() for n, t in zip(BIG, self.tables)},
                    'row_state': {n: (None if t.state is None else t.state.cpu()) for n, t in zip(BIG, self.tables)},
                    'small': {n: p.data.cpu() for n, p in zip(SMALL, self.small)},
                    'small_state': [None if s is None else s.cpu() for s in j.rec.small_state]}, self.shard_file(filename))

    @torch.no_grad()
    def load_shards(self, filename):
        ck = torch.load(self.shard_file(filename), map_location='cpu', weights_only=False)
        if ck['rank'] != self.rank or ck['world'] != self.world:
            raise L.KtupError('%s was written by rank %d of %d' % (self.shard_file(filename), ck['rank'], ck['world']))
        if ck['lr'] != self._lr:
            self._build(ck['lr'])
        for n, t in zip(BIG, self.tables):
            t.weight.data.copy_(ck['rows'][n])
            if t.state is not None and ck['row_state'][n] is not None:
                t.state.copy_(ck['row_state'][n])
        for n, p in zip(SMALL, self.small):
            p.data.copy_(ck['small'][n])
        for s, v in zip(self.joint.rec.small_state, ck['small_state']):
            if s is not None and v is not None:
                s.copy_(v)                           # (the kg stepper shares the rel / norm sums)
        self.joint.rec.opt_step.fill_(int(ck.get('opt_step', 0)))
        self.joint.steps = ck['joint_steps']
        self.trainer.step = ck['step']
        self._dirty = True
        self.sync_model()

    def _wrap_trainer(self):
        """ModelTrainer.save writes the whole tables (gathered first) and, beside them, this rank's shard file."""
        trainer, inner = self.trainer, self.trainer.save

        def save(filename):
            self.sync_model()
            inner(filename)
            self.save_shards(filename)
        trainer.save = save
