"""Knowledge-representation task driver (TransE / TransH / TransR) with the reference's entry points
(jTransUP/models/knowledge_representation.py: evaluate :28-105, train_loop :107-219, run :221-310)."""
import math
import os
import random

import torch

from jTransUP.data.load_triple_data import load_data
from jTransUP.models import _driver as D
from jTransUP.models.base import flag_defaults, get_flags, init_model
from jTransUP.utils import flags as gflags
from jTransUP.utils import loss
from jTransUP.utils.data import getTrainTripleBatch
from jTransUP.utils.trainer import ModelTrainer

FLAGS = gflags.FLAGS


def evaluate(FLAGS, model, entity_total, relation_total, eval_head_iter, eval_tail_iter, eval_head_dict, eval_tail_dict,
             all_head_dicts, all_tail_dicts, logger, eval_descending=True, is_report=False):
    model.eval(); model.disable_grad()
    from jTransUP.models._shard_eval import kg_shard_fn
    head_fn, tail_fn = model.evaluateHead, model.evaluateTail
    kw = {}
    if hasattr(model, 'prepare_entities'):           # TransR: the entity side of both passes, once (it does not depend on the queries)
        ents = model.prepare_entities()
        head_fn, tail_fn = (lambda t, r: model.evaluateHead(t, r, ents=ents)), (lambda h, r: model.evaluateTail(h, r, ents=ents))
        kw = {'ents': ents}
    # models with rank_entities (TransE, TransH, TransR): the whole pass -- scores and filtered gold ranks -- behind one call per direction
    rank = (lambda head: (lambda q, r, desc, go, gi, fo, fi: model.rank_entities(q, r, head, desc, go, gi, fo, fi, **kw))) \
        if hasattr(model, 'rank_entities') else (lambda head: None)
    head_results = D.kg_eval_pass(FLAGS, head_fn, eval_head_iter, eval_head_dict, all_head_dicts, eval_descending,
                                  want_rows=is_report, shard=kg_shard_fn(model, True), rank_fn=rank(True))
    tail_results = D.kg_eval_pass(FLAGS, tail_fn, eval_tail_iter, eval_tail_dict, all_tail_dicts, eval_descending,
                                  want_rows=is_report, shard=kg_shard_fn(model, False), rank_fn=rank(False))
    perf = D.summarize_kg(FLAGS, head_results, tail_results, logger)
    if is_report:
        D.report_kg(head_results, tail_results, logger)
    model.enable_grad()
    return perf


def kg_step_loss(FLAGS, model, triple_batch, entity_total, all_head_dicts, all_tail_dicts, with_orth):
    """Margin loss + regularisers of one triple batch (knowledge_representation.py:176-204); the embedding gathers the
    reference repeats for the regularisers are fused into the regulariser kernels."""
    ph, pt, pr, nh, nt, nr = getTrainTripleBatch(triple_batch, entity_total, all_head_dicts=all_head_dicts,
                                                 all_tail_dicts=all_tail_dicts)
    ph_v, pt_v, pr_v, nh_v, nt_v, nr_v = (D.ids(x) for x in (ph, pt, pr, nh, nt, nr))
    return (ph_v, pt_v, pr_v), (nh_v, nt_v, nr_v), torch.cat([ph_v, pt_v, nh_v, nt_v]), torch.cat([pr_v, nr_v])


def train_loop(FLAGS, model, trainer, train_dataset, eval_datasets, entity_total, relation_total, logger, vis=None,
               is_report=False):
    train_iter, train_total, train_list, train_head_dict, train_tail_dict = train_dataset
    all_head_dicts = all_tail_dicts = None
    if FLAGS.filter_wrong_corrupted:
        all_head_dicts = [train_head_dict] + [d[4] for d in eval_datasets]
        all_tail_dicts = [train_tail_dict] + [d[5] for d in eval_datasets]
    # TransE / TransH / TransR: the step body below as a handful of C-ABI launches (utils/fast_train.py KGStepper), optionally with the
    # triples and the corruption sampling on the device (-device_sampling)
    stepper = feed = sampler = None
    if D.USE_CUDA and FLAGS.model_type in ('transe', 'transh', 'transr') and trainer.fused is not None \
            and os.environ.get('KTUP_FAST_TRAIN', '1') != '0':
        from jTransUP.utils.fast_train import DeviceFeeder, KGStepper
        stepper = KGStepper(model, trainer, FLAGS, FLAGS.batch_size)
        logger.info('GPU-resident training step enabled (KTUP_FAST_TRAIN=0 selects the autograd route).')
        if FLAGS.device_sampling:
            from jTransUP.utils.device_sampler import DeviceSampler
            sampler = DeviceSampler(D.DEV, seed=FLAGS.seed)
            known = None
            if FLAGS.filter_wrong_corrupted:
                known = [train_list] + [[(h, t, r) for (t, r), hs in d[4].items() for h in hs] for d in eval_datasets]
            sampler.set_triples(entity_total, relation_total, known)
            feed = DeviceFeeder(train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed)
            stepper.attach_feeds(sampler, kg=feed)
            logger.info('Training data and negative sampling are device-resident (-device_sampling).')
    D.require_stepper_for_replicas(stepper, 'transe, transh, transr')
    logger.info('Training.')

    def do_eval(totals):
        logger.info('train loss:{:.4f}!'.format(totals['kg'] / FLAGS.eval_interval_steps))
        perfs = []
        for i, ed in enumerate(eval_datasets):
            hd = td = None
            if FLAGS.filter_wrong_corrupted:
                hd = [train_head_dict] + [d[4] for j, d in enumerate(eval_datasets) if j != i]
                td = [train_tail_dict] + [d[5] for j, d in enumerate(eval_datasets) if j != i]
            perfs.append(evaluate(FLAGS, model, entity_total, relation_total, ed[0], ed[1], ed[4], ed[5], hd, td, logger,
                                  eval_descending=False, is_report=is_report))
        if trainer.step > 0 and perfs:
            trainer.new_performance(perfs[0], perfs)
            if vis is not None:
                vis.plot_many_stack({'KG Train Loss': totals['kg'] / FLAGS.eval_interval_steps}, win_name='Loss Curve')
                vis.plot_many_stack({'KG Eval {} Hit'.format(i): p[0] for i, p in enumerate(perfs)},
                                    win_name='KG Hit Ratio@{}'.format(FLAGS.topn))
                vis.plot_many_stack({'KG Eval {} MeanRank'.format(i): p[1] for i, p in enumerate(perfs)}, win_name='KG MeanRank')
        return perfs

    def do_step(step):
        if feed is not None:
            if stepper.can_feed('kg'):                     # batch + negatives drawn inside the step's own graph
                if D.steps_before_pause(FLAGS, step, trainer.best_step) >= 10 and stepper.fed_cycle(('kg',) * 10):
                    return 'kg', None                     # ten steps in one replay; losses are summed on the device
                stepper.fed_step('kg')
                return 'kg', None                         # (stepper.take_sums)
            ph, pt, pr = feed.next_cols()                  # (h, t, r): tail before relation, like the files
            nh, nt = sampler.sample_kg(ph, pt, pr)
            return 'kg', stepper.kg_step(ph, pt, pr, nh, nt, pr)
        if stepper is not None:
            ph, pt, pr, nh, nt, nr = getTrainTripleBatch(next(train_iter), entity_total, all_head_dicts=all_head_dicts,
                                                         all_tail_dicts=all_tail_dicts)
            if len(ph) != stepper.GB:
                raise RuntimeError('training batch of unexpected size (MakeTrainIterator yields full batches)')
            return 'kg', stepper.kg_step(*(D.ids(x) for x in (ph, pt, pr, nh, nt, nr)))
        pos, neg, ent_ids, rel_ids = kg_step_loss(FLAGS, model, next(train_iter), entity_total, all_head_dicts, all_tail_dicts,
                                                  FLAGS.model_type == 'transh')
        trainer.optimizer_zero_grad()
        pos_score, neg_score = model(*pos), model(*neg)
        losses = loss.marginLoss()(pos_score, neg_score, FLAGS.margin)
        if FLAGS.model_type == 'transh':
            losses = losses + loss.orthogonalLoss(model.rel_embeddings.weight, model.norm_embeddings.weight, ids=rel_ids)
        losses = losses + loss.normLoss(model.ent_embeddings.weight, ids=ent_ids) \
            + loss.normLoss(model.rel_embeddings.weight, ids=rel_ids)
        losses.backward()
        D.clip_and_step(FLAGS, model, trainer)
        return 'kg', losses

    D.training_loop(FLAGS, model, trainer, logger, do_step, do_eval, ['kg'], sampler=sampler, stepper=stepper)
    trainer.save(trainer.checkpoint_path + '_final')      # knowledge_representation.py:219


def run(only_forward=False):
    D.setup_replicas(FLAGS)
    if FLAGS.seed != 0:
        random.seed(FLAGS.seed)
        torch.manual_seed(FLAGS.seed)
    vis = D.make_visualizer(FLAGS)
    logger = D.setup_logger(FLAGS)
    kg_path = os.path.join(os.path.join(FLAGS.data_path, FLAGS.dataset), 'kg')
    eval_files = FLAGS.kg_test_files.split(':') if FLAGS.kg_test_files else []
    train_dataset, eval_datasets, e_map, r_map = load_data(kg_path, eval_files, FLAGS.batch_size, logger=logger,
                                                           negtive_samples=FLAGS.negtive_samples)
    entity_total = max(len(e_map), max(e_map.values()))
    relation_total = max(len(r_map), max(r_map.values()))
    train_iter, train_total, train_list, train_head_dict, train_tail_dict = train_dataset
    D.freeze_heap()
    model = init_model(FLAGS, 0, 0, entity_total, relation_total, logger)
    trainer = ModelTrainer(model, logger, math.ceil(train_total / FLAGS.batch_size), FLAGS)
    if FLAGS.load_ckpt_file is not None:
        trainer.loadEmbedding(os.path.join(FLAGS.log_path, FLAGS.load_ckpt_file), model.state_dict(), cpu=not D.USE_CUDA)
        model.is_pretrained = True
    if only_forward:
        for i, ed in enumerate(eval_datasets):
            hd = td = None
            if FLAGS.filter_wrong_corrupted:
                hd = [train_head_dict] + [d[4] for j, d in enumerate(eval_datasets) if j != i]
                td = [train_tail_dict] + [d[5] for j, d in enumerate(eval_datasets) if j != i]
            evaluate(FLAGS, model, entity_total, relation_total, ed[0], ed[1], ed[4], ed[5], hd, td, logger,
                     eval_descending=False, is_report=FLAGS.is_report)
    else:
        train_loop(FLAGS, model, trainer, train_dataset, eval_datasets, entity_total, relation_total, logger, vis=vis,
                   is_report=False)
    if vis is not None:
        vis.log('Finish!', win_name='Best Performances')


if __name__ == '__main__':
    import sys
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    run(only_forward=FLAGS.eval_only_mode)
