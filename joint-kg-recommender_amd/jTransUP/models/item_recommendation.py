"""Item-recommendation task driver (BPRMF / TUP) with the reference's entry points
(jTransUP/models/item_recommendation.py: evaluate :27-75, train_loop :77-194, run :196-273)."""
import math
import os
import random

import torch

from jTransUP.data.load_rating_data import load_data
from jTransUP.models import _driver as D
from jTransUP.models.base import flag_defaults, get_flags, init_model
from jTransUP.utils import flags as gflags
from jTransUP.utils.data import getNegRatings
from jTransUP.utils.loss import bprLoss, normLoss, orthogonalLoss
from jTransUP.utils.trainer import ModelTrainer

FLAGS = gflags.FLAGS


def evaluate(FLAGS, model, eval_iter, eval_dict, all_dicts, logger, eval_descending=True, is_report=False):
    model.eval(); model.disable_grad()
    score_fn = model.evaluate
    if hasattr(model, 'prepare_items'):            # TUP: the item side of the gate once per pass (weights are frozen here)
        lazy = []                                  # ... and only if the batch walk runs at all

        def score_fn(u):
            if not lazy:
                lazy.append(model.prepare_items())
            return model.evaluate(u, items=lazy[0])
    from jTransUP.models._shard_eval import rec_shard_fn
    # the whole-pass route prepares its own item side, so that a captured pass (D._rec_eval_fused) recomputes it from the tables
    pass_fn = (lambda u, fo, fi, n: model.evaluate_topk(u, model.prepare_items(), n, fo, fi)) if hasattr(model, 'evaluate_topk') else None
    native = getattr(model, '_shard_native', None)             # -shard_tables: the candidates are the item rows this rank owns
    if native is not None and not is_report:
        results = D.rec_eval_pass(FLAGS, None, eval_iter, eval_dict, all_dicts, eval_descending, want_rows=False, shard=native.rec_shard())
        perf = D.summarize_rec(FLAGS, results, logger)
        model.enable_grad()
        return perf
    results = D.rec_eval_pass(FLAGS, score_fn, eval_iter, eval_dict, all_dicts, eval_descending, want_rows=is_report,
                              shard=rec_shard_fn(model), pass_fn=pass_fn, graph_key=D.model_graph_key(model) if pass_fn else None)
    perf = D.summarize_rec(FLAGS, results, logger)
    if is_report:
        D.report_rec(FLAGS, model, results, all_dicts, eval_dict, logger, FLAGS.model_type in ('transup', 'jtransup', 'cjtransup'))
    model.enable_grad()
    return perf


def train_loop(FLAGS, model, trainer, train_dataset, eval_datasets, user_total, item_total, logger, vis=None, is_report=False):
    train_iter, train_total, train_list, train_dict = train_dataset
    all_dicts = [train_dict] + [d[3] for d in eval_datasets] if FLAGS.filter_wrong_corrupted else None
    # TUP / BPRMF: the step body below as a handful of C-ABI launches (utils/fast_train.py RecStepper), optionally with the
    # training data and the negative sampling on the device (-device_sampling)
    stepper = feed = sampler = None
    sharded = bool(getattr(FLAGS, 'shard_tables', False))
    if sharded:
        # config 3 at scale: TUP's user / item tables row-sharded over the ranks, fixed-shape exchange, row-sparse optimizer on the touched
        # rows (utils/sharded_train.py; the stepper is sharded_ktup.ShardedKtupStepper without an entity table)
        from jTransUP.utils.sharded_train import ShardedJointDriver
        stepper = ShardedJointDriver(model, trainer, FLAGS, FLAGS.batch_size, logger)
        logger.info('Row-sharded training step enabled (-shard_tables): rank %d of %d owns rows r %% %d == %d of the user / item tables.'
                    % (stepper.rank, stepper.world, stepper.world, stepper.rank))
        if FLAGS.device_sampling:
            from jTransUP.utils.device_sampler import DeviceSampler
            from jTransUP.utils.fast_train import DeviceFeeder
            sampler = DeviceSampler(D.DEV, seed=FLAGS.seed)
            sampler.set_rating_dicts(user_total, item_total, all_dicts)
            feed = DeviceFeeder(train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed)
            logger.info('Training data and negative sampling are device-resident (-device_sampling).')
    elif D.USE_CUDA and FLAGS.model_type in ('transup', 'bprmf') and trainer.fused is not None \
            and (FLAGS.model_type == 'bprmf' or FLAGS.embedding_size % 4 == 0) \
            and os.environ.get('KTUP_FAST_TRAIN', '1') != '0':             # (a TUP width that is not a multiple of 4: the autograd route)
        from jTransUP.utils.fast_train import DeviceFeeder, RecStepper
        stepper = RecStepper(model, trainer, FLAGS, FLAGS.batch_size)
        logger.info('GPU-resident training step enabled (KTUP_FAST_TRAIN=0 selects the autograd route).')
        if FLAGS.device_sampling:
            from jTransUP.utils.device_sampler import DeviceSampler
            sampler = DeviceSampler(D.DEV, seed=FLAGS.seed)
            sampler.set_rating_dicts(user_total, item_total, all_dicts)
            feed = DeviceFeeder(train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed)
            stepper.attach_feeds(sampler, rec=feed)
            logger.info('Training data and negative sampling are device-resident (-device_sampling).')
    D.require_stepper_for_replicas(stepper, 'transup, bprmf')
    logger.info('Training.')

    def do_eval(totals):
        if sharded:
            stepper.sync_model() if is_report else stepper.begin_eval()      # (report mode walks whole tables: gathered for this pass only)
        logger.info('train loss:{:.4f}!'.format(totals['rec'] / FLAGS.eval_interval_steps))
        perfs = []
        for i, ed in enumerate(eval_datasets):
            others = [train_dict] + [d[3] for j, d in enumerate(eval_datasets) if j != i] if FLAGS.filter_wrong_corrupted else None
            perfs.append(evaluate(FLAGS, model, ed[0], ed[3], others, logger, eval_descending=trainer.model_target == 1,
                                  is_report=is_report))
        if trainer.step > 0 and perfs:
            trainer.new_performance(perfs[0], perfs)
            if vis is not None:
                vis.plot_many_stack({'Rec Train Loss': totals['rec'] / FLAGS.eval_interval_steps}, win_name='Loss Curve')
                for name, col in (('F1', 0), ('Precision', 1), ('Recall', 2), ('Hit Ratio', 3), ('NDCG', 4)):
                    vis.plot_many_stack({'Rec Eval {} {}'.format(i, name): p[col] for i, p in enumerate(perfs)},
                                        win_name='Rec {}@{}'.format(name, FLAGS.topn))
        if sharded:
            stepper.end_eval()
            if is_report:
                stepper.release_model()
        return perfs

    def do_step(step):
        if feed is not None:
            if stepper.can_feed('rec'):                    # batch + negatives drawn inside the step's own graph
                if D.steps_before_pause(FLAGS, step, trainer.best_step) >= 10 and stepper.fed_cycle(('rec',) * 10):
                    return 'rec', None                     # ten steps in one replay; losses are summed on the device
                stepper.fed_step('rec')
                return 'rec', None                         # (stepper.take_sums)
            u_d, pi_d = feed.next_cols()
            return 'rec', stepper.rec_step(u_d, pi_d, sampler.sample_rec(u_d, pi_d))
        u, pi, ni = getNegRatings(next(train_iter), item_total, all_dicts=all_dicts)
        u_var, pi_var, ni_var = D.ids(u), D.ids(pi), D.ids(ni)
        if stepper is not None and len(u) == stepper.GB:
            return 'rec', stepper.rec_step(u_var, pi_var, ni_var)
        trainer.optimizer_zero_grad()
        pos_score, neg_score = model(u_var, pi_var), model(u_var, ni_var)
        losses = bprLoss(pos_score, neg_score, target=trainer.model_target)
        if FLAGS.model_type in ('transup', 'transupb'):       # item_recommendation.py:177-180, gathers fused
            losses = losses + orthogonalLoss(model.pref_embeddings.weight, model.pref_norm_embeddings.weight) \
                + normLoss(model.user_embeddings.weight, ids=u_var) \
                + normLoss(model.item_embeddings.weight, ids=torch.cat([pi_var, ni_var])) \
                + normLoss(model.pref_embeddings.weight)
        losses.backward()
        D.clip_and_step(FLAGS, model, trainer)
        return 'rec', losses

    D.training_loop(FLAGS, model, trainer, logger, do_step, do_eval, ['rec'], sampler=sampler, stepper=stepper)


def run(only_forward=False):
    D.setup_replicas(FLAGS)
    if FLAGS.seed != 0:
        random.seed(FLAGS.seed)
        torch.manual_seed(FLAGS.seed)
    vis = D.make_visualizer(FLAGS)
    logger = D.setup_logger(FLAGS)
    dataset_path = os.path.join(FLAGS.data_path, FLAGS.dataset)
    train_dataset, eval_datasets, u_map, i_map = load_data(dataset_path, FLAGS.rec_test_files.split(':'), FLAGS.batch_size,
                                                           logger=logger, negtive_samples=FLAGS.negtive_samples)
    train_iter, train_total, train_list, train_dict = train_dataset
    user_total = max(len(u_map), max(u_map.values()))
    item_total = max(len(i_map), max(i_map.values()))
    D.freeze_heap()
    model = init_model(FLAGS, user_total, item_total, 0, 0, logger)
    trainer = ModelTrainer(model, logger, math.ceil(train_total / FLAGS.batch_size), FLAGS)
    if FLAGS.load_ckpt_file is not None:
        trainer.loadEmbedding(os.path.join(FLAGS.log_path, FLAGS.load_ckpt_file), model.state_dict(), cpu=not D.USE_CUDA)
        model.is_pretrained = True
    if only_forward:
        shards = None
        if getattr(FLAGS, 'shard_tables', False):      # -eval_only_mode under -shard_tables: shard what was loaded, evaluate ON the shards
            from jTransUP.utils.sharded_train import ShardedJointDriver
            shards = ShardedJointDriver(model, trainer, FLAGS, FLAGS.batch_size, logger)
            logger.info('Row-sharded evaluation (-shard_tables): rank %d of %d.' % (shards.rank, shards.world))
            shards.sync_model() if FLAGS.is_report else shards.begin_eval()
        for i, ed in enumerate(eval_datasets):
            others = [train_dict] + [d[3] for j, d in enumerate(eval_datasets) if j != i] if FLAGS.filter_wrong_corrupted else None
            evaluate(FLAGS, model, ed[0], ed[3], others, logger, eval_descending=trainer.model_target == 1,
                     is_report=FLAGS.is_report)
        if shards is not None:
            shards.end_eval()
    else:
        train_loop(FLAGS, model, trainer, train_dataset, eval_datasets, user_total, item_total, logger, vis=vis, is_report=False)
    if vis is not None:
        vis.log('Finish!', win_name='Best Performances')


if __name__ == '__main__':
    import sys
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    run(only_forward=FLAGS.eval_only_mode)
