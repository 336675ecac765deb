"""Command-line surface and model factory of the drop-in package.

The three task drivers share one flag set (names, types and defaults are the reference's: jTransUP/models/base.py:22-98,
defaults filled in :100-125) and one factory keyed on -model_type (:128-174).  Flags live in a declarative table here and
are registered on the local gflags-compatible registry (utils/flags.py)."""
import importlib
import time

import torch

from jTransUP.utils import flags as gflags

# -model_type -> module under jTransUP.models that provides build_model(...)
ACCELERATED = {'transup': 'transUP', 'bprmf': 'bprmf', 'transe': 'transE', 'transh': 'transH', 'transr': 'transR',
               'jtransup': 'jTransUP', 'cke': 'CKE', 'cfkg': 'CFKG', 'fm': 'fm', 'cofm': 'cofm'}
REFERENCE_ONLY = ('transd',)      # TransD: outside SURVEY.md section 8 (and its evaluateTail has a NameError in the reference, transD.py:127)
MODEL_TYPES = ['transup', 'bprmf', 'fm', 'transe', 'transh', 'transr', 'transd', 'cfkg', 'cke', 'cofm', 'jtransup']
DATASETS = ['ml1m', 'dbbook2014', 'amazon-book', 'last-fm', 'yelp2018']

# (name, kind, default, help[, choices])
FLAG_TABLE = [
    ('model_type', 'enum', 'transup', 'which scorer to train / evaluate', MODEL_TYPES),
    ('dataset', 'enum', 'ml1m', 'dataset directory under -data_path', DATASETS),
    ('optimizer_type', 'enum', 'Adagrad', 'torch.optim class', ['Adam', 'SGD', 'Adagrad', 'Rmsprop']),
    ('log_level', 'enum', 'debug', 'logging level', ['debug', 'info']),
    # model
    ('embedding_size', 'int', 64, 'd'),
    ('num_preferences', 'int', 4, 'P (TUP)'),
    ('L1_flag', 'bool', False, 'L1 distance; otherwise squared L2'),
    ('use_st_gumbel', 'bool', False, 'straight-through Gumbel preference gate'),
    ('share_embeddings', 'bool', False, 'one table for aligned items and entities (joint models)'),
    # optimisation
    ('learning_rate', 'float', 0.001, 'initial learning rate'),
    ('learning_rate_decay_when_no_progress', 'float', 0.5, 'multiplier after an epoch without a new best'),
    ('l2_lambda', 'float', 1e-5, 'weight decay of the optimizer'),
    ('momentum', 'float', 0.9, 'SGD / RMSprop momentum'),
    ('clipping_max_value', 'float', 5.0, 'global gradient-norm clip'),
    ('margin', 'float', 1.0, 'margin of the KG ranking loss'),
    ('norm_lambda', 'float', 1.0, 'weight of the alignment term (coFM / CFKG)'),
    ('kg_lambda', 'float', 1.0, 'weight of the KG loss in joint training'),
    ('joint_ratio', 'float', 0.5, 'share of recommendation steps in joint training (KG gets the rest)'),
    ('batch_size', 'int', 512, 'examples per step'),
    ('negtive_samples', 'int', 1, 'epoch multiplier of the training iterator'),
    ('training_steps', 'int', 1400000, 'stop after this many steps'),
    ('eval_interval_steps', 'int', 14000, 'evaluate every this many steps'),
    ('early_stopping_steps_to_wait', 'int', 70000, 'stop when the best step is this old (0 = never)'),
    ('seed', 'int', 0, '0 = do not seed'),
    # evaluation
    ('topn', 'int', 10, 'cut-off of the ranking metrics'),
    ('filter_wrong_corrupted', 'bool', True, 'filter known positives out of negatives and rankings'),
    ('is_report', 'bool', False, 'log per-user top-n lists and induced preferences'),
    ('eval_only_mode', 'bool', False, 'load -load_experiment_name and evaluate only'),
    ('num_processes', 'int', 4, 'accepted for compatibility: ranking runs on the device'),
    ('max_queue', 'int', 10, 'accepted for compatibility: no worker processes here'),
    # this build
    ('device_sampling', 'bool', True, 'keep training data and negative sampling on the GPU (K19); -nodevice_sampling runs the python samplers'),
    ('shard_eval_candidates', 'bool', False, 'torchrun only: every rank scores its slice of the item / entity catalogue for ALL queries '
                                             '(top-n lists merged, KG rank counts all-reduced) instead of whole batches being dealt to the ranks'),
    ('shard_tables', 'bool', False, 'transup, or jtransup with its own tables: the user / item (/ entity) tables and their optimizer state are partitioned by row over '
                                    'the ranks (row % world) and a step exchanges only the rows its batch touches (BASELINE config 5; one process: the '
                                    'same row-sparse step without an exchange); needs -optimizer_type Adagrad, Adam or SGD -momentum 0 (-l2_lambda > 0: the steps a row was not '
                                    'touched for are replayed one by one when it is touched again -- fine at ml1m size, use -l2_lambda 0 for tables of millions of rows); '
                                    'the shards are the only resident copy: evaluation runs on them'),
    ('shard_capacity_factor', 'float', 1.25, '-shard_tables under torchrun: distinct rows a rank may ask ONE owner for per step = factor x entries / world '
                                             '+ 64; a step that needs more is skipped on every rank and the run stops at the next check (factor = world never overflows)'),
    ('shard_whole_checkpoint', 'bool', True, '-shard_tables: besides the per-rank shard files write the reference-layout whole-table checkpoint (tables '
                                             'gathered transiently); -noshard_whole_checkpoint: shard files only'),
    # files
    ('data_path', 'str', None, 'root of the datasets'),
    ('log_path', 'str', None, 'logs (and, by default, checkpoints)'),
    ('ckpt_path', 'str', None, 'checkpoints; defaults to -log_path'),
    ('experiment_name', 'str', None, 'names the log and checkpoint files'),
    ('load_experiment_name', 'str', None, 'checkpoint to restore in -eval_only_mode'),
    ('load_ckpt_file', 'str', None, "pre-trained checkpoints under -log_path, ':'-separated"),
    ('rec_test_files', 'str', None, "rating evaluation files, ':'-separated"),
    ('kg_test_files', 'str', None, "triple evaluation files, ':'-separated"),
    ('has_visualization', 'bool', True, 'visdom curves (no-op when visdom is missing)'),
    ('visualization_port', 'int', 8097, 'visdom port'),
]

_DEFINE = {'enum': lambda n, d, h, c: gflags.DEFINE_enum(n, d, c, h), 'int': lambda n, d, h, c: gflags.DEFINE_integer(n, d, h),
           'float': lambda n, d, h, c: gflags.DEFINE_float(n, d, h), 'bool': lambda n, d, h, c: gflags.DEFINE_bool(n, d, h),
           'str': lambda n, d, h, c: gflags.DEFINE_string(n, d, h)}


def get_flags():
    """Register the flag table once (idempotent, like calling the reference's get_flags twice would not be)."""
    if gflags.FLAGS.is_defined('model_type'):
        return
    for entry in FLAG_TABLE:
        name, kind, default, text = entry[:4]
        _DEFINE[kind](name, default, text, entry[4] if len(entry) > 4 else None)


def flag_defaults(FLAGS):
    """Derived defaults: names from the clock, paths relative to the package, sharing forced by the model family."""
    FLAGS.experiment_name = FLAGS.experiment_name or '%s-%s-%d' % (FLAGS.dataset, FLAGS.model_type, int(time.time()))
    FLAGS.data_path = FLAGS.data_path or '../datasets/'
    FLAGS.log_path = FLAGS.log_path or '../log/'
    FLAGS.ckpt_path = FLAGS.ckpt_path or FLAGS.log_path
    if FLAGS.seed:
        torch.manual_seed(FLAGS.seed)
    forced = {'cke': False, 'jtransup': False, 'cfkg': True}.get(FLAGS.model_type)
    if forced is not None:
        FLAGS.share_embeddings = forced


def init_model(FLAGS, user_total, item_total, entity_total, relation_total, logger, i_map=None, e_map=None, new_map=None):
    logger.info('Building model.')
    kind = FLAGS.model_type
    if kind in REFERENCE_ONLY:
        raise NotImplementedError('model_type %r is a reference baseline outside the MI355X-accelerated scoring path '
                                  '(in scope: %s)' % (kind, ', '.join(sorted(ACCELERATED))))
    if kind not in ACCELERATED:
        raise NotImplementedError
    module = importlib.import_module('jTransUP.models.' + ACCELERATED[kind])
    model = module.build_model(FLAGS, user_total, item_total, entity_total, relation_total, i_map=i_map, e_map=e_map, new_map=new_map)
    logger.info('Architecture: {}'.format(model))
    logger.info('Total params: {}'.format(float(sum(w.numel() for w in model.parameters()))))
    return model
