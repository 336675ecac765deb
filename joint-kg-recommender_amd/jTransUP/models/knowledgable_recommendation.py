"""Joint KG + recommendation task driver (KTUP) with the reference's entry points
(jTransUP/models/knowledgable_recommendation.py: evaluateRec :50-104, evaluateKG :106-190, train_loop :192-409,
run :411-538)."""
import math
import os
import random

import torch

from jTransUP.data.load_kg_rating_data import load_data
from jTransUP.models import _driver as D
from jTransUP.models.base import flag_defaults, get_flags, init_model
from jTransUP.utils import flags as gflags
from jTransUP.utils import loss
from jTransUP.utils.data import getNegRatings, getTrainTripleBatch
from jTransUP.utils.loss import bprLoss, orthogonalLoss
from jTransUP.utils.trainer import ModelTrainer

FLAGS = gflags.FLAGS


def getMappedEntities(i_ids, i_remap, new_map):
    """:28-37 -- aligned (entity, item) index pairs of a set of items (alignment loss of the non-KTUP joint models)."""
    e_ids, new_i_ids = [], []
    for i in set(i_ids):
        if i not in i_remap:
            continue
        pair = new_map[i_remap[i]]
        if pair[0] != -1:
            e_ids.append(pair[0]); new_i_ids.append(pair[1])
    return e_ids, new_i_ids


def getMappedItems(e_ids, e_remap, new_map):
    """:39-48."""
    i_ids, new_e_ids = [], []
    for e in set(e_ids):
        if e not in e_remap:
            continue
        pair = new_map[e_remap[e]]
        if pair[1] != -1:
            i_ids.append(pair[1]); new_e_ids.append(pair[0])
    return new_e_ids, i_ids


def evaluateRec(FLAGS, model, eval_iter, eval_dict, all_dicts, i_map, logger, eval_descending=True, is_report=False):
    all_i_var = D.ids([i_map[i] for i in range(len(i_map))]) if FLAGS.share_embeddings else None
    model.eval(); model.disable_grad()
    has_items = hasattr(model, 'prepare_items')
    lazy = []                                      # item side once per pass, and only if the batch walk runs at all

    def score_fn(u):
        if not has_items:
            return model.evaluateRec(u, all_i_ids=all_i_var)
        if not lazy:
            lazy.append(model.prepare_items(all_i_var))
        return model.evaluateRec(u, all_i_ids=all_i_var, items=lazy[0])
    from jTransUP.models._shard_eval import rec_shard_fn
    # the whole-pass route prepares its own item side, so that a captured pass (D._rec_eval_fused) recomputes it from the tables
    pass_fn = (lambda u, fo, fi, n: model.evaluate_topk(u, model.prepare_items(), n, fo, fi)) \
        if has_items and hasattr(model, 'evaluate_topk') and not FLAGS.share_embeddings else None
    native = getattr(model, '_shard_native', None)             # -shard_tables: the candidates are the rows this rank owns
    if native is not None and not is_report:
        results = D.rec_eval_pass(FLAGS, None, eval_iter, eval_dict, all_dicts, eval_descending, want_rows=False, shard=native.rec_shard())
    else:
        results = D.rec_eval_pass(FLAGS, score_fn, eval_iter, eval_dict, all_dicts, eval_descending, want_rows=is_report,
                                  shard=rec_shard_fn(model), pass_fn=pass_fn, graph_key=D.model_graph_key(model) if pass_fn else None)
    perf = D.summarize_rec(FLAGS, results, logger)
    if is_report:
        D.report_rec(FLAGS, model, results, all_dicts, eval_dict, logger, FLAGS.model_type in ('transup', 'jtransup'))
    model.enable_grad()
    return perf


def evaluateKG(FLAGS, model, eval_head_iter, eval_tail_iter, eval_head_dict, eval_tail_dict, all_head_dicts, all_tail_dicts, e_map,
               logger, eval_descending=True, is_report=False):
    model.eval(); model.disable_grad()
    all_e_var = D.ids([e_map[e] for e in range(len(e_map))]) if FLAGS.share_embeddings else None
    remap = None if FLAGS.share_embeddings else e_map      # :125,140 (identity for id-ordered map files)
    from jTransUP.models._shard_eval import kg_shard_fn
    kw = {'ents': model.prepare_entities()} if hasattr(model, 'prepare_entities') else {}       # CKE: TransR's entity side, once
    # jTransUP / CFKG: the whole pass -- scores and filtered gold ranks -- behind one call per direction (model.rank_entities)
    rank = (lambda head: (lambda q, r, desc, go, gi, fo, fi: model.rank_entities(q, r, head, desc, go, gi, fo, fi, all_e_ids=all_e_var))) \
        if hasattr(model, 'rank_entities') else (lambda head: None)
    native = getattr(model, '_shard_native', None)
    if native is not None and not is_report:
        head_results = D.kg_eval_pass(FLAGS, None, eval_head_iter, eval_head_dict, all_head_dicts, eval_descending, remap=remap, want_rows=False,
                                      shard=native.kg_shard(True))
        tail_results = D.kg_eval_pass(FLAGS, None, eval_tail_iter, eval_tail_dict, all_tail_dicts, eval_descending, remap=remap, want_rows=False,
                                      shard=native.kg_shard(False))
    else:
        head_results = D.kg_eval_pass(FLAGS, lambda t, r: model.evaluateHead(t, r, all_e_ids=all_e_var, **kw), eval_head_iter, eval_head_dict,
                                      all_head_dicts, eval_descending, remap=remap, want_rows=is_report, shard=kg_shard_fn(model, True),
                                      rank_fn=rank(True))
        tail_results = D.kg_eval_pass(FLAGS, lambda h, r: model.evaluateTail(h, r, all_e_ids=all_e_var, **kw), eval_tail_iter, eval_tail_dict,
                                      all_tail_dicts, eval_descending, remap=remap, want_rows=is_report, shard=kg_shard_fn(model, False),
                                      rank_fn=rank(False))
    perf = D.summarize_kg(FLAGS, head_results, tail_results, logger)
    if is_report:
        D.report_kg(head_results, tail_results, logger)
    model.enable_grad()
    return perf


def train_loop(FLAGS, model, trainer, rating_train_dataset, triple_train_dataset, rating_eval_datasets, triple_eval_datasets, e_map,
               i_map, ikg_map, logger, vis=None, is_report=False):
    rating_train_iter, rating_train_total, rating_train_list, rating_train_dict = rating_train_dataset
    triple_train_iter, triple_train_total, triple_train_list, head_train_dict, tail_train_dict = triple_train_dataset
    all_rating_dicts = all_head_dicts = all_tail_dicts = None
    if FLAGS.filter_wrong_corrupted:
        all_rating_dicts = [rating_train_dict] + [d[3] for d in rating_eval_datasets]
        all_head_dicts = [head_train_dict] + [d[4] for d in triple_eval_datasets]
        all_tail_dicts = [tail_train_dict] + [d[5] for d in triple_eval_datasets]
    item_total, entity_total = len(i_map), len(e_map)
    step_to_switch = 10 * FLAGS.joint_ratio          # :209 -- rec step iff step % 10 < 10 * joint_ratio
    # KTUP with its own tables: the step body below runs as ~a dozen C-ABI launches (utils/fast_train.py) instead of through
    # autograd; with -device_sampling the batches and their negatives never leave the GPU either.
    stepper = rec_feed = kg_feed = sampler = None
    sharded = bool(getattr(FLAGS, 'shard_tables', False))
    if sharded:
        # BASELINE config 5: row-sharded user / item / entity tables, fixed-shape exchange, row-sparse Adagrad (utils/sharded_train.py)
        from jTransUP.utils.sharded_train import ShardedJointDriver
        stepper = ShardedJointDriver(model, trainer, FLAGS, FLAGS.batch_size, logger)
        logger.info('Row-sharded training step enabled (-shard_tables): rank %d of %d owns rows r %% %d == %d of the user / item / entity tables.'
                    % (stepper.rank, stepper.world, stepper.world, stepper.rank))
        if FLAGS.device_sampling:                          # batches and negatives drawn on the device (every rank draws the same global batch)
            from jTransUP.utils.device_sampler import DeviceSampler
            from jTransUP.utils.fast_train import DeviceFeeder
            sampler = DeviceSampler(D.DEV, seed=FLAGS.seed)
            sampler.set_rating_dicts(model.user_total, item_total, all_rating_dicts)
            known = None
            if FLAGS.filter_wrong_corrupted:
                known = [triple_train_list] + [[(h, t, r) for (t, r), hs in d[4].items() for h in hs] for d in triple_eval_datasets]
            sampler.set_triples(entity_total, model.rel_total, known)
            rec_feed = DeviceFeeder(rating_train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed)
            kg_feed = DeviceFeeder(triple_train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed + 1)
            logger.info('Training data and negative sampling are device-resident (-device_sampling).')
    elif D.USE_CUDA and FLAGS.model_type == 'jtransup' and not FLAGS.share_embeddings and trainer.fused is not None \
            and FLAGS.embedding_size % 4 == 0 \
            and os.environ.get('KTUP_FAST_TRAIN', '1') != '0':
        from jTransUP.utils.fast_train import DeviceFeeder, JointStepper
        stepper = JointStepper(model, trainer, FLAGS, FLAGS.batch_size)
        logger.info('GPU-resident training step enabled (KTUP_FAST_TRAIN=0 selects the autograd route).')
        if FLAGS.device_sampling:
            from jTransUP.utils.device_sampler import DeviceSampler
            sampler = DeviceSampler(D.DEV, seed=FLAGS.seed)
            sampler.set_rating_dicts(model.user_total, item_total, all_rating_dicts)
            known = None
            if FLAGS.filter_wrong_corrupted:
                known = [triple_train_list] + [[(h, t, r) for (t, r), hs in d[4].items() for h in hs] for d in triple_eval_datasets]
            sampler.set_triples(entity_total, model.rel_total, known)
            rec_feed = DeviceFeeder(rating_train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed)
            kg_feed = DeviceFeeder(triple_train_list, FLAGS.batch_size, D.DEV, FLAGS.negtive_samples, seed=FLAGS.seed + 1)
            stepper.attach_feeds(sampler, rec=rec_feed, kg=kg_feed)
            logger.info('Training data and negative sampling are device-resident (-device_sampling).')
    D.require_stepper_for_replicas(stepper, 'jtransup, -noshare_embeddings')
    logger.info('Training.')

    def do_eval(totals):
        if sharded:
            if is_report:
                stepper.sync_model()                       # the per-user report walks whole tables: gathered for this pass only
            else:
                stepper.begin_eval()                       # evaluation ON the shards: flush (Adam), the items' entity rows
        rec_loss = totals['rec'] / (FLAGS.eval_interval_steps * FLAGS.joint_ratio)
        kg_loss = totals['kg'] / (FLAGS.eval_interval_steps * (1 - FLAGS.joint_ratio)) if FLAGS.joint_ratio < 1 else 0.0
        logger.info('rec train loss:{:.4f}, kg train loss:{:.4f}!'.format(rec_loss, kg_loss))
        rec_perfs, kg_perfs = [], []
        for i, ed in enumerate(rating_eval_datasets):
            others = [rating_train_dict] + [d[3] for j, d in enumerate(rating_eval_datasets) if j != i] \
                if FLAGS.filter_wrong_corrupted else None
            rec_perfs.append(evaluateRec(FLAGS, model, ed[0], ed[3], others, i_map, logger,
                                         eval_descending=trainer.model_target == 1, is_report=is_report))
        for i, ed in enumerate(triple_eval_datasets):
            hd = td = None
            if FLAGS.filter_wrong_corrupted:
                hd = [head_train_dict] + [d[4] for j, d in enumerate(triple_eval_datasets) if j != i]
                td = [tail_train_dict] + [d[5] for j, d in enumerate(triple_eval_datasets) if j != i]
            kg_perfs.append(evaluateKG(FLAGS, model, ed[0], ed[1], ed[4], ed[5], hd, td, e_map, logger, eval_descending=False,
                                       is_report=is_report))
        if trainer.step > 0 and rec_perfs:
            trainer.new_performance(rec_perfs[0], rec_perfs)
            if vis is not None:
                vis.plot_many_stack({'Rec Train Loss': rec_loss, 'KG Train Loss': kg_loss}, win_name='Loss Curve')
                for name, col in (('F1 Score', 0), ('Precision', 1), ('Recall', 2), ('Hit Ratio', 3), ('NDCG', 4)):
                    vis.plot_many_stack({'Rec Eval {} {}'.format(i, name): p[col] for i, p in enumerate(rec_perfs)},
                                        win_name='Rec {}@{}'.format(name, FLAGS.topn))
                if kg_perfs:
                    vis.plot_many_stack({'KG Eval {} Hit'.format(i): p[0] for i, p in enumerate(kg_perfs)},
                                        win_name='KG Hit Ratio@{}'.format(FLAGS.topn))
                    vis.plot_many_stack({'KG Eval {} MeanRank'.format(i): p[1] for i, p in enumerate(kg_perfs)}, win_name='KG MeanRank')
        if sharded:
            stepper.end_eval()
            if is_report:
                stepper.release_model()
        return rec_perfs

    cycle10 = tuple('rec' if k < step_to_switch else 'kg' for k in range(10))

    def do_step(step):
        is_rec = step % 10 < step_to_switch
        e_ids = i_ids = None
        if stepper is not None and rec_feed is not None:
            kind = 'rec' if is_rec else 'kg'
            if stepper.can_feed(kind):                     # batch + negatives drawn inside the step's own graph
                if step % 10 == 0 and D.steps_before_pause(FLAGS, step, trainer.best_step) >= 10 and stepper.fed_cycle(cycle10):
                    return kind, None                      # ten steps in one replay; losses are summed on the device
                stepper.fed_step(kind)
                return kind, None                          # (stepper.take_sums)
            if is_rec:
                u, pi = rec_feed.next_cols()
                return 'rec', stepper.rec_step(u, pi, sampler.sample_rec(u, pi))
            ph, pt, pr = kg_feed.next_cols()               # (h, t, r): tail before relation, like the files
            nh, nt = sampler.sample_kg(ph, pt, pr)
            return 'kg', stepper.kg_step(ph, pt, pr, nh, nt, pr)
        if stepper is not None:
            if is_rec:
                u, pi, ni = getNegRatings(next(rating_train_iter), item_total, all_dicts=all_rating_dicts)
                if len(u) == stepper.GB:
                    return 'rec', stepper.rec_step(D.ids(u), D.ids(pi), D.ids(ni))
            else:
                ph, pt, pr, nh, nt, nr = getTrainTripleBatch(next(triple_train_iter), entity_total, all_head_dicts=all_head_dicts,
                                                             all_tail_dicts=all_tail_dicts)
                if len(ph) == stepper.GB:
                    return 'kg', stepper.kg_step(*(D.ids(x) for x in (ph, pt, pr, nh, nt, nr)))
            raise RuntimeError('training batch of unexpected size (MakeTrainIterator yields full batches)')
        if is_rec:
            u, pi, ni = getNegRatings(next(rating_train_iter), item_total, all_dicts=all_rating_dicts)
            e_ids, i_ids = getMappedEntities(pi + ni, i_map, ikg_map)
            if FLAGS.share_embeddings:
                ni = [i_map[i] for i in ni]; pi = [i_map[i] for i in pi]
            u_var, pi_var, ni_var = D.ids(u), D.ids(pi), D.ids(ni)
            trainer.optimizer_zero_grad()
            pos_score = model((u_var, pi_var), None, is_rec=True)
            neg_score = model((u_var, ni_var), None, is_rec=True)
            losses = bprLoss(pos_score, neg_score, target=trainer.model_target)
            if FLAGS.model_type in ('transup', 'jtransup'):
                losses = losses + orthogonalLoss(model.pref_embeddings.weight, model.pref_norm_embeddings.weight)
        else:
            ph, pt, pr, nh, nt, nr = getTrainTripleBatch(next(triple_train_iter), entity_total, all_head_dicts=all_head_dicts,
                                                         all_tail_dicts=all_tail_dicts)
            e_ids, i_ids = getMappedItems(ph + pt + nh + nt, e_map, ikg_map)
            if FLAGS.share_embeddings:
                ph, pt, nh, nt = ([e_map[e] for e in x] for x in (ph, pt, nh, nt))
            ph_v, pt_v, pr_v, nh_v, nt_v, nr_v = (D.ids(x) for x in (ph, pt, pr, nh, nt, nr))
            trainer.optimizer_zero_grad()
            pos_score = model(None, (ph_v, pt_v, pr_v), is_rec=False)
            neg_score = model(None, (nh_v, nt_v, nr_v), is_rec=False)
            losses = loss.marginLoss()(pos_score, neg_score, FLAGS.margin)
            rel_ids = torch.cat([pr_v, nr_v])
            if FLAGS.model_type in ('jtransup',):
                losses = losses + loss.orthogonalLoss(model.rel_embeddings.weight, model.norm_embeddings.weight, ids=rel_ids)
            losses = losses + loss.normLoss(model.ent_embeddings.weight, ids=torch.cat([ph_v, pt_v, nh_v, nt_v])) \
                + loss.normLoss(model.rel_embeddings.weight, ids=rel_ids)
            losses = FLAGS.kg_lambda * losses
        if not FLAGS.share_embeddings and FLAGS.model_type not in ('cke', 'jtransup'):      # :385-390 (cofm / cfkg only)
            losses = losses + FLAGS.norm_lambda * loss.pNormLoss(model.ent_embeddings(D.ids(e_ids)), model.item_embeddings(D.ids(i_ids)),
                                                                 L1_flag=FLAGS.L1_flag)
        losses.backward()
        D.clip_and_step(FLAGS, model, trainer)
        return ('rec' if is_rec else 'kg'), losses

    D.training_loop(FLAGS, model, trainer, logger, do_step, do_eval, ['rec', 'kg'], sampler=sampler, stepper=stepper)


def run(only_forward=False):
    D.setup_replicas(FLAGS)
    if FLAGS.seed != 0:
        random.seed(FLAGS.seed)
        torch.manual_seed(FLAGS.seed)
    vis = D.make_visualizer(FLAGS)
    logger = D.setup_logger(FLAGS)
    dataset_path = os.path.join(FLAGS.data_path, FLAGS.dataset)
    rec_eval_files = FLAGS.rec_test_files.split(':') if FLAGS.rec_test_files is not None else []
    kg_eval_files = FLAGS.kg_test_files.split(':') if FLAGS.kg_test_files is not None else []
    (rating_train_dataset, rating_eval_datasets, u_map, i_map, triple_train_dataset, triple_eval_datasets, e_map, r_map,
     ikg_map) = load_data(dataset_path, rec_eval_files, kg_eval_files, FLAGS.batch_size, negtive_samples=FLAGS.negtive_samples,
                          logger=logger)
    rating_train_total, rating_train_dict = rating_train_dataset[1], rating_train_dataset[3]
    triple_train_total, head_dict, tail_dict = triple_train_dataset[1], triple_train_dataset[3], triple_train_dataset[4]
    user_total = max(len(u_map), max(u_map.values()))
    item_total = max(len(i_map), max(i_map.keys()))
    entity_total = max(len(e_map), max(e_map.keys()))
    relation_total = max(len(r_map), max(r_map.values()))
    if FLAGS.share_embeddings:
        item_total = entity_total = len(ikg_map)
    D.freeze_heap()
    model = init_model(FLAGS, user_total, item_total, entity_total, relation_total, logger, i_map=i_map, e_map=e_map, new_map=ikg_map)
    triple_epoch = math.ceil(float(triple_train_total) / (1 - FLAGS.joint_ratio) / FLAGS.batch_size) if FLAGS.joint_ratio < 1 else 0
    rating_epoch = math.ceil(float(rating_train_total) / FLAGS.joint_ratio / FLAGS.batch_size)
    trainer = ModelTrainer(model, logger, max(triple_epoch, rating_epoch), FLAGS)
    if FLAGS.load_ckpt_file is not None:
        for filename in FLAGS.load_ckpt_file.split(':'):
            if FLAGS.share_embeddings:
                trainer.loadEmbedding(os.path.join(FLAGS.log_path, filename), model.state_dict(), e_remap=e_map, i_remap=i_map)
            else:
                trainer.loadEmbedding(os.path.join(FLAGS.log_path, filename), model.state_dict())
        model.is_pretrained = True
    if only_forward:
        shards = None
        if getattr(FLAGS, 'shard_tables', False):      # -eval_only_mode under -shard_tables: shard what was loaded (or pick this rank's
            from jTransUP.utils.sharded_train import ShardedJointDriver      # shard file up) and evaluate ON the shards, like the training loop
            shards = ShardedJointDriver(model, trainer, FLAGS, FLAGS.batch_size, logger)
            logger.info('Row-sharded evaluation (-shard_tables): rank %d of %d.' % (shards.rank, shards.world))
            shards.sync_model() if FLAGS.is_report else shards.begin_eval()
        for i, ed in enumerate(rating_eval_datasets):
            others = [rating_train_dict] + [d[3] for j, d in enumerate(rating_eval_datasets) if j != i] \
                if FLAGS.filter_wrong_corrupted else None
            evaluateRec(FLAGS, model, ed[0], ed[3], others, i_map, logger, eval_descending=trainer.model_target == 1,
                        is_report=FLAGS.is_report)
        for i, ed in enumerate(triple_eval_datasets):
            hd = td = None
            if FLAGS.filter_wrong_corrupted:
                hd = [head_dict] + [d[4] for j, d in enumerate(triple_eval_datasets) if j != i]
                td = [tail_dict] + [d[5] for j, d in enumerate(triple_eval_datasets) if j != i]
            evaluateKG(FLAGS, model, ed[0], ed[1], ed[4], ed[5], hd, td, e_map, logger, eval_descending=False,
                       is_report=FLAGS.is_report)
        if shards is not None:
            shards.end_eval()
    else:
        train_loop(FLAGS, model, trainer, rating_train_dataset, triple_train_dataset, rating_eval_datasets, triple_eval_datasets,
                   e_map, i_map, ikg_map, logger, vis=vis, is_report=False)
    if vis is not None:
        vis.log('Finish!', win_name='Best Performances')


if __name__ == '__main__':
    import sys
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    run(only_forward=FLAGS.eval_only_mode)
