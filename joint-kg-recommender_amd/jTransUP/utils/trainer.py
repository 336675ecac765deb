"""ModelTrainer with the reference's interface and checkpoint layout (jTransUP/utils/trainer.py:20-217).

The optimizer is a torch.optim object (dense update of every table each step, weight_decay = l2_lambda, re-created on
LR decay -- exactly the reference's semantics).  On the GPU its clip_grad_norm + step() arithmetic runs as two HIP
launches (utils/fused_optim.py, K20) on the same state tensors, so checkpoints do not change; KTUP_FUSED_OPTIM=0 keeps
torch's own multi-tensor kernels."""
import os

import torch
import torch.optim as optim

from jTransUP.utils.misc import USE_CUDA, to_gpu


def get_checkpoint_path(FLAGS, suffix='.ckpt'):
    """trainer.py:7-13."""
    if FLAGS.ckpt_path.endswith('.ckpt'):
        return FLAGS.ckpt_path
    return os.path.join(FLAGS.ckpt_path, FLAGS.experiment_name + suffix)


def get_model_target(model_type):
    """trainer.py:15-17: +1 for dot-product scorers (higher = better), -1 for translation distances."""
    return 1 if model_type in ('bprmf', 'cofm', 'fm') else -1


check_rho = 1.0


class ModelTrainer(object):
    def __init__(self, model, logger, epoch_length, FLAGS):
        self.model, self.logger, self.epoch_length = model, logger, epoch_length
        self.model_target = get_model_target(FLAGS.model_type)
        logger.info('One epoch is ' + str(epoch_length) + ' steps.')
        self.parameters = [p for _, p in model.named_parameters()]
        self.optimizer_type = FLAGS.optimizer_type
        self.l2_lambda = FLAGS.l2_lambda
        self.learning_rate_decay_when_no_progress = FLAGS.learning_rate_decay_when_no_progress
        self.momentum = FLAGS.momentum
        self.eval_interval_steps = FLAGS.eval_interval_steps
        self.step = self.best_step = 0
        self.best_dev_performance = 0.0
        self.best_performances = None
        to_gpu(model)
        self.optimizer_reset(FLAGS.learning_rate)
        self.checkpoint_path = get_checkpoint_path(FLAGS)
        if FLAGS.eval_only_mode and FLAGS.load_experiment_name and os.path.isfile(FLAGS.load_experiment_name):
            logger.info('Found checkpoint, restoring.')
            self.load(FLAGS.load_experiment_name, cpu=not USE_CUDA)
            logger.info('Resuming at step: {} with best dev performance: {} and test performance : {}.'.format(
                self.best_step, self.best_dev_performance, self.best_performances))

    def reset(self):
        self.step = self.best_step = 0

    def optimizer_reset(self, learning_rate):
        """trainer.py:63-77: a FRESH optimizer (state is dropped) at the given rate."""
        self.learning_rate = learning_rate
        kw = dict(lr=learning_rate, weight_decay=self.l2_lambda)
        if self.optimizer_type == 'Adam':
            self.optimizer = optim.Adam(self.parameters, **kw)
        elif self.optimizer_type == 'SGD':
            self.optimizer = optim.SGD(self.parameters, momentum=self.momentum, **kw)
        elif self.optimizer_type == 'Adagrad':
            self.optimizer = optim.Adagrad(self.parameters, **kw)
        elif self.optimizer_type == 'Rmsprop':
            self.optimizer = optim.RMSprop(self.parameters, momentum=self.momentum, **kw)
        self.fused = None
        if USE_CUDA and os.environ.get('KTUP_FUSED_OPTIM', '1') != '0':
            from jTransUP.utils.fused_optim import FusedOptimizer
            self.fused = FusedOptimizer(self.optimizer)

    def optimizer_step(self):
        self.optimizer.step()
        self.step += 1

    def clip_and_step(self, max_norm):
        """clip_grad_norm over ALL tables, then the optimizer step (e.g. item_recommendation.py:189-192)."""
        if self.fused is not None:
            self.fused.clip_and_step(max_norm)
            self.step += 1
        else:
            torch.nn.utils.clip_grad_norm_(self.parameters, max_norm)
            self.optimizer_step()

    def optimizer_zero_grad(self):
        """Zero-fill (torch 0.3 semantics): tables that have had a gradient keep receiving weight decay / moment decay
        on steps that do not touch them (KTUP alternates rec and KG steps)."""
        self.optimizer.zero_grad(set_to_none=False)

    def new_performance(self, dev_performance, performances):
        """trainer.py:86-103: checkpoint on a new best of metric[0]; halve the LR after an epoch without one."""
        is_best = False
        care = dev_performance[0]
        if care > check_rho * self.best_dev_performance:
            self.best_step = self.step
            self.logger.info('Checkpointing ...')
            self.save(self.checkpoint_path)
            self.best_performances = performances
            self.best_dev_performance = care
            is_best = True
        if self.learning_rate_decay_when_no_progress != 1.0:
            last_epoch_start = self.step - (self.step % self.epoch_length)
            if self.step - last_epoch_start <= self.eval_interval_steps and self.best_step < (last_epoch_start - self.epoch_length):
                self.logger.info('No improvement after one epoch. Lowering learning rate.')
                self.optimizer_reset(self.learning_rate * self.learning_rate_decay_when_no_progress)
        return is_best

    def checkpoint(self):
        self.logger.info('Checkpointing.')
        self.save(self.checkpoint_path)

    def save(self, filename):
        """trainer.py:109-126: same dict keys; tensors are written from CPU copies."""
        cpu_state = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        torch.save({'step': self.step, 'best_step': self.best_step, 'best_dev_performance': self.best_dev_performance,
                    'model_state_dict': cpu_state, 'optimizer_state_dict': self.optimizer.state_dict()}, filename)

    def load(self, filename, cpu=False):
        ck = torch.load(filename, map_location='cpu' if cpu else None, weights_only=False)
        self.model.load_state_dict(ck['model_state_dict'], strict=False)
        self.optimizer.load_state_dict(ck['optimizer_state_dict'])
        self.step, self.best_step = ck['step'], ck['best_step']
        self.best_dev_performance = ck['best_dev_performance']

    def loadEmbedding(self, filename, embedding_names, cpu=False, e_remap=None, i_remap=None):
        """trainer.py:144-217: copy pre-trained tables (TUP + TransH checkpoints feed KTUP, ktup.sh:1), including the
        E -> E+1 row case of the padded entity table and the id remaps of the shared-embedding mode."""
        assert os.path.isfile(filename), 'Checkpoint file not found!'
        self.logger.info('Found checkpoint, restoring pre-trained embeddings.')
        old = torch.load(filename, map_location='cpu', weights_only=False)['model_state_dict']
        model_dict = self.model.state_dict()
        pretrained = {k: v for k, v in old.items() if k in embedding_names}
        model_dict.update(pretrained)
        for key, attr, what in (('ent_embeddings.weight', 'ent_embeddings', 'entities'),
                                ('rel_embeddings.weight', 'rel_embeddings', 'relations')):
            if key in old and key in model_dict and hasattr(self.model, attr) and \
                    len(old[key]) + 1 == len(getattr(self.model, attr).weight.data):
                loaded = old[key]
                del model_dict[key]
                getattr(self.model, attr).weight.data[:len(loaded), :] = loaded.to(getattr(self.model, attr).weight.device)
                self.logger.info('Restored ' + str(len(loaded)) + ' ' + what + ' from checkpoint.')
        for remap, key, attr, what in ((e_remap, 'ent_embeddings.weight', 'ent_embeddings', 'entities'),
                                       (i_remap, 'item_embeddings.weight', 'item_embeddings', 'items')):
            if remap is not None and key in model_dict and key in embedding_names:
                loaded = model_dict.pop(key)
                table = getattr(self.model, attr).weight.data
                src = torch.tensor(list(remap.keys()), dtype=torch.long)
                dst = torch.tensor([remap[k] for k in remap], dtype=torch.long)
                table[dst.to(table.device)] = loaded[src].to(table.device)
                self.logger.info('Restored ' + str(len(remap)) + ' ' + what + ' from checkpoint.')
        self.model.load_state_dict(model_dict, strict=False)
        self.logger.info('Load Embeddings of {} from {}.'.format(', '.join(list(pretrained.keys())), filename))
