"""Flag definitions, defaults and the model factory, name-for-name with the reference's jTransUP/models/base.py
(flags :22-98, defaults :100-125, init_model :128-174), on the local gflags-compatible registry."""
import time
from functools import reduce

import torch

from jTransUP.utils import flags as gflags

_IN_SCOPE = ('transup', 'bprmf', 'transe', 'transh', 'transr', 'jtransup')
_OTHER = ('fm', 'transd', 'cfkg', 'cke', 'cofm')      # reference baselines outside the accelerated path


def get_flags():
    if gflags.FLAGS.is_defined('model_type'):
        return
    g = gflags
    g.DEFINE_enum('model_type', 'transup', ['transup', 'bprmf', 'fm', 'transe', 'transh', 'transr', 'transd', 'cfkg', 'cke',
                                            'cofm', 'jtransup'], '')
    g.DEFINE_enum('dataset', 'ml1m', ['ml1m', 'dbbook2014', 'amazon-book', 'last-fm', 'yelp2018'], 'dataset directory name')
    g.DEFINE_bool('filter_wrong_corrupted', True, 'filter test samples from train and validations')
    g.DEFINE_bool('share_embeddings', False, '')
    g.DEFINE_bool('use_st_gumbel', False, '')
    g.DEFINE_integer('max_queue', 10, 'accepted for compatibility (no process fan-out here)')
    g.DEFINE_integer('num_processes', 4, 'accepted for compatibility (ranking runs on the device)')
    g.DEFINE_float('learning_rate', 0.001, 'Used in optimizer.')
    g.DEFINE_float('norm_lambda', 1.0, 'decay of joint model.')
    g.DEFINE_float('kg_lambda', 1.0, 'decay of kg model.')
    g.DEFINE_integer('early_stopping_steps_to_wait', 70000, 'stop after this many steps without a new best (0 = never)')
    g.DEFINE_bool('L1_flag', False, 'L1 distance as dissimilarity; else squared L2')
    g.DEFINE_bool('is_report', False, 'log per-user top-n and induced preferences')
    g.DEFINE_float('l2_lambda', 1e-5, '')
    g.DEFINE_integer('embedding_size', 64, '')
    g.DEFINE_integer('negtive_samples', 1, '')
    g.DEFINE_integer('batch_size', 512, 'Minibatch size.')
    g.DEFINE_enum('optimizer_type', 'Adagrad', ['Adam', 'SGD', 'Adagrad', 'Rmsprop'], '')
    g.DEFINE_float('learning_rate_decay_when_no_progress', 0.5, 'LR multiplier when an epoch passes without a new best')
    g.DEFINE_integer('eval_interval_steps', 14000, 'Evaluate at this interval.')
    g.DEFINE_integer('training_steps', 1400000, 'Stop training after this point.')
    g.DEFINE_float('clipping_max_value', 5.0, '')
    g.DEFINE_float('margin', 1.0, 'Used in margin loss.')
    g.DEFINE_float('momentum', 0.9, 'The momentum of the optimizer.')
    g.DEFINE_bool('device_sampling', False, '(this build) keep the training data on the GPU and draw negatives with the '
                  'on-device samplers (K19) instead of the host samplers of utils/data.py')
    g.DEFINE_integer('seed', 0, 'Fix the random seed. 0 means no seeding.')
    g.DEFINE_integer('topn', 10, '')
    g.DEFINE_integer('num_preferences', 4, '')
    g.DEFINE_float('joint_ratio', 0.5, '(0 - 1) share of recommendation steps; kg gets 1 - joint_ratio')
    g.DEFINE_string('experiment_name', None, '')
    g.DEFINE_string('data_path', None, '')
    g.DEFINE_string('rec_test_files', None, "multiple filenames separated by ':'")
    g.DEFINE_string('kg_test_files', None, "multiple filenames separated by ':'")
    g.DEFINE_string('log_path', None, '')
    g.DEFINE_enum('log_level', 'debug', ['debug', 'info'], '')
    g.DEFINE_string('ckpt_path', None, 'Where to save/load checkpoints. If not set, the same as log_path')
    g.DEFINE_string('load_ckpt_file', None, "pretrained checkpoints under log path, separated by ':'")
    g.DEFINE_boolean('has_visualization', True, 'visdom curves (ignored with a warning when visdom is missing)')
    g.DEFINE_integer('visualization_port', 8097, '')
    g.DEFINE_boolean('eval_only_mode', False, 'load a checkpoint and only evaluate')
    g.DEFINE_string('load_experiment_name', None, '')


def flag_defaults(FLAGS):
    """base.py:100-125."""
    if not FLAGS.experiment_name:
        FLAGS.experiment_name = '{}-{}-{}'.format(FLAGS.dataset, FLAGS.model_type, int(time.time()))
    if not FLAGS.data_path:
        FLAGS.data_path = '../datasets/'
    if not FLAGS.log_path:
        FLAGS.log_path = '../log/'
    if not FLAGS.ckpt_path:
        FLAGS.ckpt_path = FLAGS.log_path
    if FLAGS.seed != 0:
        torch.manual_seed(FLAGS.seed)
    if FLAGS.model_type in ('cke', 'jtransup'):
        FLAGS.share_embeddings = False
    elif FLAGS.model_type == 'cfkg':
        FLAGS.share_embeddings = True


def init_model(FLAGS, user_total, item_total, entity_total, relation_total, logger, i_map=None, e_map=None, new_map=None):
    """base.py:128-174."""
    logger.info('Building model.')
    mt = FLAGS.model_type
    if mt == 'transup':
        from jTransUP.models import transUP as mod
    elif mt == 'bprmf':
        from jTransUP.models import bprmf as mod
    elif mt == 'transe':
        from jTransUP.models import transE as mod
    elif mt == 'transh':
        from jTransUP.models import transH as mod
    elif mt == 'transr':
        from jTransUP.models import transR as mod
    elif mt == 'jtransup':
        from jTransUP.models import jTransUP as mod
    elif mt in _OTHER:
        raise NotImplementedError('model_type %r is a reference baseline outside the MI355X-accelerated scoring path '
                                  '(in scope: %s)' % (mt, ', '.join(_IN_SCOPE)))
    else:
        raise NotImplementedError
    model = mod.build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=i_map, e_map=e_map,
                            new_map=new_map)
    logger.info('Architecture: {}'.format(model))
    total_params = sum(reduce(lambda x, y: x * y, w.size(), 1.0) for w in model.parameters())
    logger.info('Total params: {}'.format(total_params))
    return model
