"""Optimizer ownership, step counting, best-metric checkpointing and pre-trained table loading behind the reference's
ModelTrainer interface and checkpoint layout (jTransUP/utils/trainer.py:20-217).

The optimizer is a torch.optim object (dense update of every table each step, weight_decay = l2_lambda, a FRESH optimizer
on every LR decay -- the reference's semantics).  On the GPU its clip_grad_norm + step() arithmetic runs as two HIP launches
(utils/fused_optim.py, K20) on the same state tensors, so checkpoints do not change; KTUP_FUSED_OPTIM=0 keeps torch's own
multi-tensor kernels."""
import os

import torch
import torch.optim as optim

from jTransUP.utils.misc import USE_CUDA, to_gpu

check_rho = 1.0     # a new best must exceed check_rho * previous best


def get_checkpoint_path(FLAGS, suffix='.ckpt'):
    """-ckpt_path names either the checkpoint file itself (*.ckpt) or the directory it goes to."""
    return FLAGS.ckpt_path if FLAGS.ckpt_path.endswith('.ckpt') else os.path.join(FLAGS.ckpt_path, FLAGS.experiment_name + suffix)


def get_model_target(model_type):
    """+1: higher score = better (dot-product scorers); -1: translation distances, lower = better."""
    return {True: 1, False: -1}[model_type in ('bprmf', 'cofm', 'fm')]


def _make_optimizer(kind, parameters, lr, weight_decay, momentum):
    if kind == 'Adam':
        return optim.Adam(parameters, lr=lr, weight_decay=weight_decay)
    if kind == 'Adagrad':
        return optim.Adagrad(parameters, lr=lr, weight_decay=weight_decay)
    if kind == 'SGD':
        return optim.SGD(parameters, lr=lr, weight_decay=weight_decay, momentum=momentum)
    if kind == 'Rmsprop':
        return optim.RMSprop(parameters, lr=lr, weight_decay=weight_decay, momentum=momentum)
    raise ValueError('unknown optimizer_type %r' % (kind,))


class ModelTrainer(object):
    def __init__(self, model, logger, epoch_length, FLAGS):
        self.model, self.logger, self.epoch_length = model, logger, epoch_length
        logger.info('One epoch is ' + str(epoch_length) + ' steps.')
        self.model_target = get_model_target(FLAGS.model_type)
        self.optimizer_type, self.l2_lambda, self.momentum = FLAGS.optimizer_type, FLAGS.l2_lambda, FLAGS.momentum
        self.learning_rate_decay_when_no_progress = FLAGS.learning_rate_decay_when_no_progress
        self.eval_interval_steps = FLAGS.eval_interval_steps
        self.step = self.best_step = 0
        self.best_dev_performance, self.best_performances = 0.0, None
        to_gpu(model)
        self.parameters = [p for _, p in model.named_parameters()]
        self.optimizer_reset(FLAGS.learning_rate)
        self.checkpoint_path = get_checkpoint_path(FLAGS)
        resume = FLAGS.load_experiment_name
        if FLAGS.eval_only_mode and resume and os.path.isfile(resume):
            logger.info('Found checkpoint, restoring.')
            self.load(resume, cpu=not USE_CUDA)
            logger.info('Resuming at step: {} with best dev performance: {} and test performance : {}.'.format(
                self.best_step, self.best_dev_performance, self.best_performances))

    # ---- optimizer
    def reset(self):
        self.step = self.best_step = 0

    def optimizer_reset(self, learning_rate):
        """A new optimizer at `learning_rate`: moments / accumulators start over, as in the reference."""
        self.learning_rate = learning_rate
        self.optimizer = _make_optimizer(self.optimizer_type, self.parameters, learning_rate, self.l2_lambda, self.momentum)
        self.fused = None
        if USE_CUDA and os.environ.get('KTUP_FUSED_OPTIM', '1') != '0':
            from jTransUP.utils.fused_optim import FusedOptimizer
            self.fused = FusedOptimizer(self.optimizer)

    def optimizer_zero_grad(self):
        """Zero-fill (torch 0.3 semantics): tables that have had a gradient keep receiving weight decay / moment decay
        on steps that do not touch them (KTUP alternates rec and KG steps)."""
        self.optimizer.zero_grad(set_to_none=False)

    def optimizer_step(self):
        self.optimizer.step()
        self.step += 1

    def clip_and_step(self, max_norm):
        """Global-norm clip over ALL tables, then the optimizer step (what every driver's step ends with)."""
        if self.fused is None:
            torch.nn.utils.clip_grad_norm_(self.parameters, max_norm)
            self.optimizer_step()
        else:
            self.fused.clip_and_step(max_norm)
            self.step += 1

    # ---- model selection
    def new_performance(self, dev_performance, performances):
        """Checkpoint when metric[0] of the first evaluation file improves; multiply the LR by
        learning_rate_decay_when_no_progress once a whole epoch has gone by without an improvement."""
        improved = dev_performance[0] > check_rho * self.best_dev_performance
        if improved:
            # order of trainer.py:94-99: the file is written BEFORE best_dev_performance moves, so a checkpoint carries the best value
            # before this improvement (0.0 for a run's first one) -- what a run resumed from it prints and compares against
            self.best_step = self.step
            self.logger.info('Checkpointing ...')
            self.save(self.checkpoint_path)
            self.best_performances, self.best_dev_performance = performances, dev_performance[0]
        if self.learning_rate_decay_when_no_progress != 1.0:
            epoch_start = self.step - self.step % self.epoch_length
            first_eval_of_epoch = self.step - epoch_start <= self.eval_interval_steps
            if first_eval_of_epoch and self.best_step < epoch_start - self.epoch_length:
                self.logger.info('No improvement after one epoch. Lowering learning rate.')
                self.optimizer_reset(self.learning_rate * self.learning_rate_decay_when_no_progress)
        return improved

    def checkpoint(self):
        self.logger.info('Checkpointing.')
        self.save(self.checkpoint_path)

    # ---- persistence (same dict keys as the reference's checkpoints)
    def save(self, filename):
        tables = {name: tensor.detach().cpu() for name, tensor in self.model.state_dict().items()}
        torch.save({'step': self.step, 'best_step': self.best_step, 'best_dev_performance': self.best_dev_performance,
                    'model_state_dict': tables, 'optimizer_state_dict': (self.fused or self.optimizer).state_dict()}, filename)

    def load(self, filename, cpu=False):
        ck = torch.load(filename, map_location='cpu' if cpu else None, weights_only=False)
        self.model.load_state_dict(ck['model_state_dict'], strict=False)
        (self.fused or self.optimizer).load_state_dict(ck['optimizer_state_dict'])
        self.step, self.best_step, self.best_dev_performance = ck['step'], ck['best_step'], ck['best_dev_performance']

    def loadEmbedding(self, filename, embedding_names, cpu=False, e_remap=None, i_remap=None):
        """Copy pre-trained tables out of another run's checkpoint (TUP + TransH checkpoints feed KTUP).  Tables named in
        `embedding_names` are taken over; an entity / relation table with one row fewer than the model's goes into its
        leading rows (KTUP's padded entity table); with e_remap / i_remap (shared-embedding mode) rows move to remapped ids."""
        assert os.path.isfile(filename), 'Checkpoint file not found!'
        self.logger.info('Found checkpoint, restoring pre-trained embeddings.')
        theirs = torch.load(filename, map_location='cpu', weights_only=False)['model_state_dict']
        wanted = {name: table for name, table in theirs.items() if name in embedding_names}
        merged = self.model.state_dict()
        merged.update(wanted)
        for key, what in (('ent_embeddings', 'entities'), ('rel_embeddings', 'relations')):
            full = key + '.weight'
            module = getattr(self.model, key, None)
            if module is None or full not in theirs or full not in merged:
                continue
            rows = theirs[full]
            if rows.shape[0] + 1 == module.weight.shape[0]:
                merged.pop(full)
                module.weight.data[:rows.shape[0]] = rows.to(module.weight.device)
                self.logger.info('Restored ' + str(len(rows)) + ' ' + what + ' from checkpoint.')
        for remap, key, what in ((e_remap, 'ent_embeddings', 'entities'), (i_remap, 'item_embeddings', 'items')):
            full = key + '.weight'
            if remap is None or full not in merged or full not in embedding_names:
                continue
            rows = merged.pop(full)
            table = getattr(self.model, key).weight.data
            old_ids = torch.tensor(list(remap.keys()), dtype=torch.long)
            new_ids = torch.tensor(list(remap.values()), dtype=torch.long)
            if rows.data_ptr() == table.data_ptr():
                # the checkpoint holds no such table, so `rows` IS the model's own (state_dict aliases the parameters): the reference's
                # loop (trainer.py:176-200) then moves rows one at a time inside the table it reads from, later moves seeing earlier
                # ones -- what happens to the item table when the second of two checkpoints is loaded with the remaps
                # (knowledgable_recommendation.py:478).  Same order, same result.
                for old, new in zip(old_ids.tolist(), new_ids.tolist()):
                    table[new] = table[old]
            else:
                table[new_ids.to(table.device)] = rows[old_ids].to(table.device)
            self.logger.info('Restored ' + str(len(remap)) + ' ' + what + ' from checkpoint.')
            if key == 'item_embeddings' and 'item_bias.weight' in merged and 'item_bias.weight' in wanted:      # coFM (trainer.py:202-211)
                bias = merged.pop('item_bias.weight')
                mine = self.model.item_bias.weight.data
                mine[new_ids.to(mine.device)] = bias[old_ids].to(mine.device)
                self.logger.info('Restored ' + str(len(remap)) + ' items bias from checkpoint.')
        self.model.load_state_dict(merged, strict=False)
        self.logger.info('Load Embeddings of {} from {}.'.format(', '.join(wanted), filename))
