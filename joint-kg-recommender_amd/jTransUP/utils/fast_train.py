"""GPU-resident training step of the joint driver (KTUP, -noshare_embeddings): the arithmetic of
knowledgable_recommendation.py:330-401 (`train_loop`'s step body) issued as ~a dozen launches through the C ABI.

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


class JointStepper(object):
    def __init__(self, model, trainer, FLAGS, batch_size, group=None):
        if trainer.fused is None:
            raise L.KtupError('JointStepper needs the fused optimizer (KTUP_FUSED_OPTIM=0 disables it)')
        self.m, self.trainer = model, trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if int(batch_size) % self.world:
            raise L.KtupError('batch_size %d is not divisible by the %d data-parallel ranks' % (batch_size, self.world))
        self.GB = int(batch_size)                    # global batch (what the driver samples)
        self.B = self.GB // self.world               # rows this rank scores
        self.margin, self.kg_lambda, self.max_norm = float(FLAGS.margin), float(FLAGS.kg_lambda), float(FLAGS.clipping_max_value)
        self.target = float(trainer.model_target)
        self.l1 = int(bool(model.L1_flag))
        U, I, E, P, Pn, R, Rn = model._rec_tables()
        self.tabs = (U, I, E, P, Pn, R, Rn)
        dev = U.device
        self.dev = dev
        f32 = dict(dtype=torch.float32, device=dev)
        # persistent zero-filled gradients (torch 0.3 zero_grad semantics) as views into one flat bucket; 4 loss scalars at its end
        sizes = [p.numel() for p in trainer.parameters]
        pad = [(-n) % 4 for n in sizes]              # keep every view 16-byte aligned for the float4 kernels
        self.flat = torch.zeros(sum(sizes) + sum(pad) + 4, **f32)
        off = 0
        for p, n, q in zip(trainer.parameters, sizes, pad):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n + q
        self.loss = self.flat[off:off + 4]
        if self.world > 1:                           # identical replicas to start from
            for p in trainer.parameters:
                dist.broadcast(p.data, src=0, group=group)
        B = self.B
        i64 = dict(dtype=torch.int64, device=dev)
        self.u2, self.i2 = torch.zeros(2 * B, **i64), torch.zeros(2 * B, **i64)          # [pos ; neg]
        self.h2, self.t2, self.r2 = torch.zeros(2 * B, **i64), torch.zeros(2 * B, **i64), torch.zeros(2 * B, **i64)
        self.ht4 = torch.zeros(4 * B, **i64)                                             # ph, pt, nh, nt (normLoss rows)
        self.score, self.gscore = torch.zeros(2 * B, **f32), torch.zeros(2 * B, **f32)
        self.gAC = torch.zeros(2, P.shape[0], P.shape[1], **f32)                         # mixed-table gradients gA, gC
        self.inv_world = torch.full((), 1.0 / self.world, **f32)                         # upstream gradient of 'mean' / 'replicated' terms
        self.lam = torch.full((), self.kg_lambda, **f32)                                 # ... of the KG 'sum' terms
        self.ws = ops.pref_workspace(P, Pn, R, Rn)
        self.ent_pad = model.ent_total - 1
        self.i2e = model._item2ent

    def _mine(self, t):
        """This rank's rows of a global-batch id tensor."""
        return t if self.world == 1 else t[self.rank * self.B:(self.rank + 1) * self.B]

    # ------------------------------------------------------------------------------------------------ rec
    def rec_step(self, u, pi, ni):
        """u, pi, ni: int64 device tensors of B ids.  Returns the step's loss (0-dim device tensor)."""
        m, B = self.m, self.B
        U, I, E, P, Pn, R, Rn = self.tabs
        st = torch.cuda.current_stream(self.dev).cuda_stream
        u, pi, ni = self._mine(u), self._mine(pi), self._mine(ni)
        self.u2[:B].copy_(u); self.u2[B:].copy_(u); self.i2[:B].copy_(pi); self.i2[B:].copy_(ni)
        n_pref, d = P.shape
        mode, uni, seed, off = m._gumbel.mode_and_stream(m.use_st_gumbel, None, 2 * B * n_pref)
        L.call('ktup_pref_prepare', _p(P), _p(Pn), _p(R), _p(Rn), P.stride(0), n_pref, d, _p(self.ws), st)
        L.call('ktup_score_ktup_fwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(self.i2e), _p(self.ws), n_pref, d,
               _p(self.u2), _p(self.i2), 2 * B, self.l1, int(mode), _p(uni), int(seed), int(off), _p(self.score), st)
        pos, neg, gpos, gneg = self.score[:B], self.score[B:], self.gscore[:B], self.gscore[B:]
        L.call('ktup_loss_bpr_fwd', _p(pos), _p(neg), B, self.target, _p(self.loss[0:]), st)
        L.call('ktup_loss_bpr_bwd', _p(pos), _p(neg), B, self.target, _p(self.inv_world), _p(gpos), _p(gneg), st)
        self.gAC.zero_()
        L.call('ktup_score_ktup_bwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(self.i2e), self.ent_pad,
               _p(self.ws), n_pref, d, _p(self.u2), _p(self.i2), 2 * B, self.l1, int(mode), _p(uni), int(seed), int(off),
               _p(self.gscore), _p(U.grad), _p(I.grad), _p(E.grad), _p(self.gAC[0]), _p(self.gAC[1]), st)
        # A = pref + rel and C = pref_norm + norm: the mixed-table gradient goes to both summands
        torch._foreach_add_([P.grad, R.grad, Pn.grad, Rn.grad], [self.gAC[0], self.gAC[0], self.gAC[1], self.gAC[1]])
        L.call('ktup_reg_orth_fwd', _p(P), P.stride(0), _p(Pn), Pn.stride(0), d, None, n_pref, _p(self.loss[1:]), st)
        L.call('ktup_reg_orth_bwd', _p(P), P.stride(0), _p(Pn), Pn.stride(0), d, None, n_pref, _p(self.inv_world), _p(P.grad), _p(Pn.grad), st)
        if self.world > 1:
            self.loss[:2].mul_(self.inv_world); self.loss[2:].zero_()
        self._optimizer_step()
        return self.loss[0] + self.loss[1]

    # ------------------------------------------------------------------------------------------------ kg
    def kg_step(self, ph, pt, pr, nh, nt, nr):
        B = self.B
        _, _, E, _, _, R, Rn = self.tabs
        st = torch.cuda.current_stream(self.dev).cuda_stream
        ph, pt, pr, nh, nt, nr = (self._mine(x) for x in (ph, pt, pr, nh, nt, nr))
        self.h2[:B].copy_(ph); self.h2[B:].copy_(nh); self.t2[:B].copy_(pt); self.t2[B:].copy_(nt)
        self.r2[:B].copy_(pr); self.r2[B:].copy_(nr)
        self.ht4[:B].copy_(ph); self.ht4[B:2 * B].copy_(pt); self.ht4[2 * B:3 * B].copy_(nh); self.ht4[3 * B:].copy_(nt)
        d, n_rel = E.shape[1], min(R.shape[0], Rn.shape[0])
        L.call('ktup_score_transh_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), n_rel, d, _p(self.h2),
               _p(self.t2), _p(self.r2), 2 * B, self.l1, _p(self.score), st)
        pos, neg, gpos, gneg = self.score[:B], self.score[B:], self.gscore[:B], self.gscore[B:]
        L.call('ktup_loss_margin_fwd', _p(pos), _p(neg), B, self.margin, _p(self.loss[0:]), st)
        L.call('ktup_loss_margin_bwd', _p(pos), _p(neg), B, self.margin, _p(self.lam), _p(gpos), _p(gneg), st)
        L.call('ktup_score_transh_bwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.h2), _p(self.t2),
               _p(self.r2), 2 * B, self.l1, _p(self.gscore), _p(E.grad), _p(R.grad), _p(Rn.grad), st)
        L.call('ktup_reg_orth_fwd', _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.r2), 2 * B, _p(self.loss[1:]), st)
        L.call('ktup_reg_orth_bwd', _p(R), R.stride(0), _p(Rn), Rn.stride(0), d, _p(self.r2), 2 * B, _p(self.lam), _p(R.grad), _p(Rn.grad), st)
        L.call('ktup_reg_norm_fwd', _p(E), E.stride(0), d, _p(self.ht4), 4 * B, _p(self.loss[2:]), st)
        L.call('ktup_reg_norm_bwd', _p(E), E.stride(0), d, _p(self.ht4), 4 * B, _p(self.lam), _p(E.grad), st)
        L.call('ktup_reg_norm_fwd', _p(R), R.stride(0), d, _p(self.r2), 2 * B, _p(self.loss[3:]), st)
        L.call('ktup_reg_norm_bwd', _p(R), R.stride(0), d, _p(self.r2), 2 * B, _p(self.lam), _p(R.grad), st)
        self._optimizer_step()
        return self.kg_lambda * self.loss.sum()

    def _optimizer_step(self):
        if self.world > 1:       # gradients of all tables + the loss scalars, one bucket
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.trainer.fused.clip_and_step(self.max_norm, zero_grads=True)
        self.trainer.step += 1


class DeviceFeeder(object):
    """Device-resident training data for the GPU-resident loop: the rating / triple lists as device tensors, an epoch
    permutation drawn on the device, batches as slices, negatives from the K19 samplers (utils/device_sampler.py).
    Iterator contract of utils/data.py MakeTrainIterator (data.py:87-110): endless, reshuffled every epoch, the tail
    partial batch dropped, `negtive_samples` copies of every example per epoch."""

    def __init__(self, rows, batch_size, device, negtive_samples=1, seed=0):
        self.rows = torch.as_tensor(rows, dtype=torch.int64).reshape(len(rows), -1)[:, :3].contiguous().to(device)
        self.B, self.dev, self.rep = int(batch_size), device, int(negtive_samples)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.n = self.rows.shape[0]
        if self.n * self.rep < self.B:
            raise ValueError('fewer training examples (%d) than one batch (%d)' % (self.n * self.rep, self.B))
        self._shuffle()

    def _shuffle(self):
        self.order = torch.randperm(self.n * self.rep, generator=self.gen, device=self.dev) % self.n
        self.start = 0

    def next(self):
        if self.start > self.order.numel() - self.B:
            self._shuffle()
        idx = self.order[self.start:self.start + self.B]
        self.start += self.B
        return self.rows.index_select(0, idx)
