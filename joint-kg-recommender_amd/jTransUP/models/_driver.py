"""Shared plumbing of the three task drivers: logging, the periodic-evaluation training loop, and the
evaluation passes (device scores -> device ranking -> host metric means).

What differs from the reference's loops (jTransUP/models/{item_recommendation,knowledge_representation,
knowledgable_recommendation}.py) is mechanical only: losses stay on the device and are read back once per
evaluation interval (the reference syncs with `losses.data[0]` every step), index tensors are created directly on the
device, ranking runs on the device, and 0.3-era APIs (`Variable`, `clip_grad_norm`) are the modern equivalents."""
import json
import logging
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from tqdm import tqdm

from jTransUP.utils.misc import USE_CUDA
from jTransUP.utils.ranking import RankIndex, evalKGProcess, evalRecProcess

DEV = torch.device('cuda') if USE_CUDA else torch.device('cpu')


def ids(values):
    return torch.tensor(values, dtype=torch.long, device=DEV)


_BATCH_IDS = {}
_KG_BATCH = {}


def batch_ids(batch, values=None):
    """Device ids of an evaluation batch, uploaded once: the iterator hands out the same batch objects in every pass."""
    hit = _BATCH_IDS.get(id(batch))
    if hit is None or hit[0] is not batch:
        hit = _BATCH_IDS[id(batch)] = (batch, ids(batch if values is None else values))
    return hit[1]


def setup_logger(FLAGS):
    os.makedirs(FLAGS.log_path, exist_ok=True)
    logger = logging.getLogger()
    logger.setLevel(logging.DEBUG if FLAGS.log_level == 'debug' else logging.INFO)
    fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    for h in (logging.FileHandler(os.path.join(FLAGS.log_path, FLAGS.experiment_name + '.log')), logging.StreamHandler()):
        h.setFormatter(fmt)
        logger.addHandler(h)
    logger.info('Flag Values:\n' + json.dumps(FLAGS.FlagValuesDict(), indent=4, sort_keys=True))
    return logger


def make_visualizer(FLAGS):
    if not FLAGS.has_visualization:
        return None
    from jTransUP.utils.visuliazer import Visualizer
    vis = Visualizer(env=FLAGS.experiment_name, port=FLAGS.visualization_port)
    vis.log(json.dumps(FLAGS.FlagValuesDict(), indent=4, sort_keys=True), win_name='Parameter')
    return vis


def freeze_heap():
    """Called once after the dataset is loaded: the rating / triple lists and filter dicts (millions of small objects at ml1m size)
    move to the collector's permanent generation, so that no later collection -- Python's own, every few hundred container
    allocations of the training loop -- walks them again (a full pass over that heap is ~0.1 s, a thousand B = 512 steps)."""
    import gc
    gc.collect()
    gc.freeze()


def flat_keys(eval_iter):
    return [k if not isinstance(k, list) else tuple(k) for batch in eval_iter for k in batch]


_INDEX_CACHE = {}


def setup_replicas(FLAGS):
    """torchrun (WORLD_SIZE > 1): one process per GPU, replicas with ONE gradient all-reduce per step (config 4, see
    utils/fast_train.py).  Every rank draws the same global batches (same -seed) and scores its slice; evaluation batches
    are dealt to the ranks; ranks > 0 log and checkpoint under their own experiment name.  -> (rank, world)."""
    from jTransUP import parallel
    rank, world = parallel.init_distributed()
    if world > 1:
        if rank > 0:
            FLAGS.experiment_name = '%s.rank%d' % (FLAGS.experiment_name, rank)
        if FLAGS.seed == 0:
            raise ValueError('data-parallel runs need a fixed -seed (every rank must draw the same batches)')
    return rank, world


def require_stepper_for_replicas(stepper, which):
    if dist.is_initialized() and dist.get_world_size() > 1 and stepper is None:
        raise NotImplementedError('data-parallel training runs through the GPU-resident step (%s)' % which)


def rank_index(eval_iter, eval_dict, all_dicts):
    """The CSR filter / gold sets of an evaluation pass are a function of the datasets only: built once per run and reused by
    every periodic evaluation (building them costs ~20x the device pass at ml1m size)."""
    key = (id(eval_iter), id(eval_dict), None if all_dicts is None else tuple(id(d) for d in all_dicts))
    hit = _INDEX_CACHE.get(key)
    if hit is None or hit[0] is not eval_iter or hit[1] is not eval_dict:
        hit = (eval_iter, eval_dict, all_dicts, RankIndex(flat_keys(eval_iter), eval_dict, all_dicts, DEV))
        if len(_INDEX_CACHE) > 16:                    # bounded like the graph / pinned-buffer caches: long-lived processes build many iterators
            _INDEX_CACHE.clear()
        _INDEX_CACHE[key] = hit
    return hit[3]


def _my_batches(n_batches):
    """Evaluation batches of this rank under torchrun (batch b goes to rank b % world); every batch in a single process."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        return range(dist.get_rank(), n_batches, dist.get_world_size()), dist.get_world_size()
    return range(n_batches), 1


def _gather_batches(per_batch, n_batches, world):
    """{batch index: result} of every rank -> list in batch order on every rank."""
    if world == 1:
        return [per_batch[b] for b in range(n_batches)]
    parts = [None] * world
    dist.all_gather_object(parts, per_batch)
    merged = {}
    for part in parts:
        merged.update(part)
    return [merged[b] for b in range(n_batches)]


def _shard_mode(FLAGS, shard, want_rows):
    """Candidate-sharded evaluation applies under torchrun with -shard_eval_candidates, for models with a sliceable candidate
    table, on the training-time path (metric columns only; the per-user report rows keep the whole-batch route)."""
    if shard is not None and len(shard) == 3 and not want_rows:
        return True                                  # -shard_tables: the candidates are the rows a rank owns, whatever the world size
    return shard is not None and not want_rows and getattr(FLAGS, 'shard_eval_candidates', False) \
        and dist.is_initialized() and dist.get_world_size() > 1


def _shard_layout(shard):
    """(candidate scores of a batch, global id of local candidate 0, id stride) -- a contiguous block [lo, hi) of the catalogue
    (models/_shard_eval.py: whole tables on every rank), or the lattice rank + world * j (utils/sharded_train.py: row-sharded tables)."""
    from jTransUP import parallel
    if len(shard) == 3:
        n_cand, fn, (_, rank, world) = shard
        return fn, rank, world
    n_cand, fn = shard
    lo, hi = parallel.shard_bounds(n_cand, dist.get_rank(), dist.get_world_size())
    return (lambda *a: fn(*a, lo, hi)), lo, 1


def _rec_eval_sharded(FLAGS, shard, eval_iter, index, descending):
    """rec_eval_pass with the catalogue split over the ranks: every rank walks ALL batches, scores its candidate slice, and the
    filtered top-n lists are merged (parallel.sharded_topk); the metric columns are then identical on every rank."""
    from jTransUP import parallel
    from jTransUP.hip import ops
    fn, lo, stride = _shard_layout(shard)
    cols = []
    for u_ids in eval_iter:
        s, e = index.rows_of(u_ids)
        f_off, f_ids = index.filter_slice(s, e)
        top, _ = parallel.sharded_topk(fn(batch_ids(u_ids)), lo, FLAGS.topn, descending, f_off, f_ids, stride=stride)
        g_off, g_ids = index.gold_slice(s, e)[:2]
        cols.append(ops.rec_metrics(top.to(torch.int32).contiguous(), g_off, g_ids))
    host = torch.cat(cols).cpu().numpy() if cols else np.zeros((0, 5))
    keep = np.concatenate([index.present_h[slice(*index.rows_of(b))] for b in eval_iter]) if cols else np.zeros(0, bool)
    return host[keep]


def _kg_eval_sharded(FLAGS, shard, eval_iter, index, descending, remap):
    """kg_eval_pass with the entity catalogue split over the ranks: per-shard rank counts, all-reduced (parallel.sharded_gold_ranks)."""
    from jTransUP import parallel
    fn, lo, stride = _shard_layout(shard)
    out = []
    for batch in eval_iter:
        q = ids([k[0] if remap is None else remap[k[0]] for k in batch])
        r = ids([k[1] for k in batch])
        keys = [tuple(k) for k in batch]
        s, e = index.rows_of(keys)
        f_off, f_ids = index.filter_slice(s, e)
        g_off, g_ids, g_off_h, _ = index.gold_slice(s, e)
        n = int(g_off_h[-1])
        if n == 0:
            continue
        g_rows = torch.repeat_interleave(torch.arange(len(keys), device=DEV), (g_off[1:] - g_off[:-1]))
        out.append(parallel.sharded_gold_ranks(fn(q, r), lo, descending, g_off, g_ids, g_rows, f_off, f_ids, stride=stride))
    ranks = torch.cat(out).cpu().numpy() if out else np.zeros(0, np.int32)
    ranks = ranks[ranks >= 0]
    return np.stack([(ranks < FLAGS.topn).astype(np.float64), ranks.astype(np.float64)], axis=1)


_PASS_IDS = {}
_KG_PASS_IDS = {}


_PINNED = {}


def _to_host(t, copy=True):
    """Device -> host through a cached pinned buffer (a pageable destination costs a staging copy: 24 vs 14 us for the 240 KB of
    metric columns of an ml1m pass).  The returned array is a copy (copy=False: a view of the pinned buffer, valid until the
    next call with this shape -- for callers that index or copy it right away)."""
    key = (tuple(t.shape), t.dtype)
    pin = _PINNED.get(key)
    if pin is None:
        if len(_PINNED) > 8:
            _PINNED.clear()
        pin = _PINNED[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
    pin.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return pin.numpy().copy() if copy else pin.numpy()


_EVAL_GRAPHS = {}


def _present_rows(host, index):
    """The metric rows of the keys that have gold items, as an array of their own (`host` is a view of a reused buffer)."""
    return host.copy() if index.all_present else host[index.present_h]


def _single_process():
    """Graph capture of the evaluation pass is kept to single-process runs: under torchrun a collective backend's own threads
    may touch the device while a capture is open."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def model_graph_key(model):
    """What a captured evaluation pass of `model` depends on besides the pass's own inputs: the addresses of its tables.  None for a
    model whose pass draws noise (ST-Gumbel gate): a replay would repeat the captured stream position."""
    if getattr(model, 'use_st_gumbel', False):
        return None
    return (id(model),) + tuple(p.data_ptr() for p in model.parameters())


def _rec_eval_fused(FLAGS, pass_fn, eval_iter, index, graph_key=None):
    """The whole pass in one sweep: every evaluation user at once through the fused score + filtered top-n kernel
    (model.evaluate_topk), the per-user metrics on the device (K18b), one (users x 5) copy back.  None if the model declines.
    With a `graph_key` (model_graph_key: the pass reads the tables in place, so the same launches serve every periodic
    evaluation of a run) the second pass is captured -- item side, sweep, merge, metrics --
    and every later one is ONE graph replay + the copy back: the ~6 launches of a 0.22 ms pass otherwise leave ~0.03 ms of gaps between them.
    KTUP_EVAL_GRAPH=0 switches the replay off; runs with more than one process never capture."""
    from jTransUP.hip import ops
    hit = _PASS_IDS.get(id(eval_iter))
    if hit is None or hit[0] is not eval_iter:
        if len(_PASS_IDS) > 16:
            _PASS_IDS.clear()
        hit = _PASS_IDS[id(eval_iter)] = (eval_iter, ids([u for batch in eval_iter for u in batch]))
    users = hit[1]
    if users.numel() == 0:
        return np.zeros((0, 5))
    fo, fi = (index.f_off, index.f_ids) if index.has_filter else (None, None)

    def body():
        top = pass_fn(users, fo, fi, FLAGS.topn)
        return None if top is None else ops.rec_metrics(top, index.g_off, index.g_ids)

    use_graph = graph_key is not None and os.environ.get('KTUP_EVAL_GRAPH', '1') != '0' and _single_process()
    key = (id(eval_iter), id(index), FLAGS.topn, graph_key)
    entry = _EVAL_GRAPHS.get(key) if use_graph else None
    if entry is not None and (entry[2] is not eval_iter or entry[3] is not index):
        entry = None
    if entry is None:
        cols = body()                                   # eager: the first pass of a run, or no replay wanted
        if cols is None:
            return None
        if use_graph:
            if len(_EVAL_GRAPHS) > 16:
                _EVAL_GRAPHS.clear()
            _EVAL_GRAPHS[key] = (None, None, eval_iter, index)           # seen once: the next pass captures
        return _present_rows(_to_host(cols, copy=False), index)
    if entry[0] is None:                                # second pass: capture (allocations land in the graph's own pool), then replay
        graph = torch.cuda.CUDAGraph()
        from jTransUP.hip.lib import capture as _capture
        with _capture(graph):
            cols = body()
        entry = _EVAL_GRAPHS[key] = (graph, cols, eval_iter, index)
    entry[0].replay()
    return _present_rows(_to_host(entry[1], copy=False), index)     # the copy back stays outside the graph (a copy node: 0.5 ms)


def rec_eval_pass(FLAGS, score_fn, eval_iter, eval_dict, all_dicts, descending, want_rows=True, shard=None, pass_fn=None, graph_key=None):
    """One pass over the evaluation users: all-item scores, filtered top-n, metric rows (misc.py:148-248 semantics).
    want_rows=False returns the (n x 5) metric array only (no per-user report rows).  Under torchrun the batches are
    dealt round-robin to the ranks and the results gathered, so every rank reports the same numbers."""
    index = rank_index(eval_iter, eval_dict, all_dicts)
    if _shard_mode(FLAGS, shard, want_rows):
        return _rec_eval_sharded(FLAGS, shard, eval_iter, index, descending)
    if pass_fn is not None and not want_rows and not descending and os.environ.get('KTUP_EVAL_PASS', '1') != '0':
        fused = _rec_eval_fused(FLAGS, pass_fn, eval_iter, index, graph_key)      # every rank runs the whole pass: ~0.3 ms at ml1m size
        if fused is not None:
            return fused
    mine, world = _my_batches(len(eval_iter))
    per_batch = {}
    pbar = tqdm(total=len(mine), desc='Run Eval')
    for b in mine:
        u_ids = eval_iter[b]
        scores = score_fn(batch_ids(u_ids))
        per_batch[b] = evalRecProcess((u_ids, scores), eval_dict, all_dicts=all_dicts, descending=descending, topn=FLAGS.topn,
                                      index=index, as_array=False if want_rows else 'device')
        pbar.update(1)
    pbar.close()
    if not want_rows and per_batch:               # metric columns stayed on the device: ONE copy back for the whole pass
        order = list(per_batch)
        host = torch.cat([per_batch[b] for b in order]).cpu().numpy()
        at = 0
        for b in order:
            s, e = index.rows_of(eval_iter[b])
            per_batch[b] = host[at:at + e - s][index.present_h[s:e]]
            at += e - s
    parts = _gather_batches(per_batch, len(eval_iter), world)
    if want_rows:
        return [row for part in parts for row in part]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 5))


def _kg_eval_fused(FLAGS, rank_fn, eval_iter, index, descending, remap, want_rows):
    """The whole pass in one call: every (entity, relation) key of the iterator at once through model.rank_entities (scores +
    filtered gold ranks per chunk of 512 keys under the C ABI), ONE copy back of the ranks.  Same rows / (hit, rank) array as the
    per-batch walk below."""
    hit = _KG_PASS_IDS.get(id(eval_iter))
    if hit is None or hit[0] is not eval_iter or hit[2] is not remap:
        keys = [k for batch in eval_iter for k in batch]
        if len(_KG_PASS_IDS) > 16:
            _KG_PASS_IDS.clear()
        hit = _KG_PASS_IDS[id(eval_iter)] = (eval_iter, (ids([k[0] if remap is None else remap[k[0]] for k in keys]), ids([k[1] for k in keys])), remap)
    q_dev, r_dev = hit[1]
    if q_dev.numel() == 0 or len(index.g_ids_h) == 0:
        return [] if want_rows else np.zeros((0, 2))
    ranks = rank_fn(q_dev, r_dev, descending, index.g_off, index.g_ids, index.f_off if index.has_filter else None,
                    index.f_ids if index.has_filter else None)
    if ranks is None:
        return None
    ranks = _to_host(ranks[:len(index.g_ids_h)])
    if not want_rows:
        ranks = ranks[ranks >= 0]
        return np.stack([(ranks < FLAGS.topn).astype(np.float64), ranks.astype(np.float64)], axis=1)
    out = []
    for b, key in enumerate(index.keys):
        lo, hi = int(index.g_off_h[b]), int(index.g_off_h[b + 1])
        rows = sorted((int(ranks[i]), int(index.g_ids_h[i])) for i in range(lo, hi) if ranks[i] >= 0)
        out.extend((1 if rk < FLAGS.topn else 0, rk, key, gid) for rk, gid in rows)
    return out


def kg_eval_pass(FLAGS, score_fn, eval_iter, eval_dict, all_dicts, descending, remap=None, want_rows=True, shard=None, rank_fn=None):
    """One pass over (t, r) or (h, r) keys: all-entity scores, filtered gold ranks (misc.py:61-146 semantics); batches are
    dealt to the ranks like in rec_eval_pass.  want_rows=False returns the (n x 2) array of (hit, rank) only: the ranks stay
    on the device until ONE copy back at the end of the pass."""
    index = rank_index(eval_iter, eval_dict, all_dicts)
    if _shard_mode(FLAGS, shard, want_rows):
        return _kg_eval_sharded(FLAGS, shard, eval_iter, index, descending, remap)
    if rank_fn is not None and os.environ.get('KTUP_EVAL_PASS', '1') != '0':
        fused = _kg_eval_fused(FLAGS, rank_fn, eval_iter, index, descending, remap, want_rows)   # every rank runs the whole pass
        if fused is not None:
            return fused
    mine, world = _my_batches(len(eval_iter))
    per_batch = {}
    pbar = tqdm(total=len(mine), desc='Run Eval')
    for b in mine:
        batch = eval_iter[b]
        hit = _KG_BATCH.get(id(batch))
        if hit is None or hit[0] is not batch or hit[1] is not remap:      # per batch object: device ids and key tuples, once
            q = [k[0] if remap is None else remap[k[0]] for k in batch]
            hit = _KG_BATCH[id(batch)] = (batch, remap, ids(q), ids([k[1] for k in batch]), [tuple(k) for k in batch])
        _, _, q_dev, r_dev, keys = hit
        scores = score_fn(q_dev, r_dev)
        per_batch[b] = evalKGProcess((keys, scores), eval_dict, all_dicts=all_dicts, descending=descending, topn=FLAGS.topn,
                                     index=index, as_array=False if want_rows else 'device')
        pbar.update(1)
    pbar.close()
    if not want_rows:
        order = [b for b in per_batch if torch.is_tensor(per_batch[b])]
        host = torch.cat([per_batch[b] for b in order]).cpu().numpy() if order else np.zeros(0, np.int32)
        at = 0
        for b in per_batch:
            n = per_batch[b].numel() if torch.is_tensor(per_batch[b]) else 0
            ranks = host[at:at + n]
            ranks = ranks[ranks >= 0]
            per_batch[b] = np.stack([(ranks < FLAGS.topn).astype(np.float64), ranks.astype(np.float64)], axis=1)
            at += n
        parts = _gather_batches(per_batch, len(eval_iter), world)
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, 2))
    return [row for part in _gather_batches(per_batch, len(eval_iter), world) for row in part]


def summarize_rec(FLAGS, results, logger):
    f1, p, r, hit, ndcg = (results if isinstance(results, np.ndarray) else np.array([row[:5] for row in results])).mean(axis=0)
    logger.info('f1:{:.4f}, p:{:.4f}, r:{:.4f}, hit:{:.4f}, ndcg:{:.4f}, topn:{}.'.format(f1, p, r, hit, ndcg, FLAGS.topn))
    return f1, p, r, hit, ndcg


def summarize_kg(FLAGS, head_results, tail_results, logger):
    as_cols = lambda res: res if isinstance(res, np.ndarray) else np.array([row[:2] for row in res])
    head_hit, head_rank = as_cols(head_results).mean(axis=0)
    tail_hit, tail_rank = as_cols(tail_results).mean(axis=0)
    logger.info('head hit:{:.4f}, head mean rank:{:.4f}, topn:{}.'.format(head_hit, head_rank, FLAGS.topn))
    logger.info('tail hit:{:.4f}, tail mean rank:{:.4f}, topn:{}.'.format(tail_hit, tail_rank, FLAGS.topn))
    hn, tn = len(head_results), len(tail_results)
    avg_hit = float(head_hit * hn + tail_hit * tn) / (hn + tn)
    avg_rank = float(head_rank * hn + tail_rank * tn) / (hn + tn)
    logger.info('avg hit:{:.4f}, avg mean rank:{:.4f}, topn:{}.'.format(avg_hit, avg_rank, FLAGS.topn))
    # MRR is not a metric of the reference (it reports hit@n and mean rank only): mean 1 / (rank + 1) over the same 0-based
    # filtered ranks, logged on its own line so that the reference's log lines and return value stay as they are
    logger.info('avg mrr:{:.4f} (head {:.4f}, tail {:.4f}).'.format(*kg_mrr(head_results, tail_results)))
    return avg_hit, avg_rank


def kg_mrr(head_results, tail_results):
    """(avg, head, tail) mean reciprocal rank from kg_eval_pass results (rows or (n x 2) arrays of (hit, 0-based rank))."""
    ranks = lambda res: (res[:, 1] if isinstance(res, np.ndarray) else np.array([row[1] for row in res], dtype=np.float64)).astype(np.float64)
    h, t = ranks(head_results), ranks(tail_results)
    rr = lambda r: float((1.0 / (r + 1.0)).mean()) if r.size else 0.0
    both = np.concatenate([h, t])
    return rr(both), rr(h), rr(t)


def report_kg(head_results, tail_results, logger):
    for hit, _, (t, r), gold_h in head_results:
        logger.info('H\t{}\t{}\t{}\t{}'.format(gold_h, t, r, hit))
    for hit, _, (h, r), gold_t in tail_results:
        logger.info('T\t{}\t{}\t{}\t{}'.format(h, gold_t, r, hit))


def report_rec(FLAGS, model, results, all_dicts, eval_dict, logger, with_preferences):
    """item_recommendation.py:55-73: per-user gold / top lists, with the induced preference of every gold item."""
    for row in results:
        u_id, top_ids, gold = row[-1]
        gold_ids = list(gold)
        if with_preferences:
            for d in all_dicts or []:
                gold_ids += list(d.get(u_id, set()))
            gold_ids += list(eval_dict.get(u_id, set()))
            probs, _, _ = model.reportPreference(ids([u_id]), ids(gold_ids))
            best = torch.max(probs, 1)[1].tolist()
            gold_strs = ','.join('{}({})'.format(i, p) for i, p in zip(gold_ids, best))
        else:
            gold_strs = ','.join(str(i) for i in gold_ids)
        logger.info('user:{}\tgold:{}\ttop:{}'.format(u_id, gold_strs, ','.join(str(i) for i in top_ids)))


def clip_and_step(FLAGS, model, trainer):
    """Global-norm clip over ALL tables, then the dense optimizer step (e.g. item_recommendation.py:189-192)."""
    trainer.clip_and_step(FLAGS.clipping_max_value)


def steps_before_pause(FLAGS, step, best_step=None):
    """How many consecutive steps may run from `step` before the loop has to look again: the next evaluation, the end of
    training, or (best_step given) the step at which the reference loop's early-stopping test first fires
    (step - best_step > early_stopping_steps_to_wait, e.g. knowledgable_recommendation.py:215-217)."""
    n = min(FLAGS.eval_interval_steps - step % FLAGS.eval_interval_steps, FLAGS.training_steps - step)
    if best_step is not None and FLAGS.early_stopping_steps_to_wait > 0:
        n = min(n, max(1, best_step + FLAGS.early_stopping_steps_to_wait + 1 - step))
    return n


def _check_clip_barrier(trainer):
    """The one-launch clip + optimizer kernel waits at a hand-rolled grid barrier; if its workgroups were ever not all resident
    (a GPU shared with another job), the poll timed out and that step went unclipped -- not the reference's step.  Checked where
    the loop syncs anyway; raises rather than training on."""
    fused = getattr(trainer, 'fused', None)
    if fused is not None and fused.barrier_timeouts():
        from jTransUP.hip.lib import KtupError
        raise KtupError('ktup_optim_clip_step timed out at its grid barrier (the GPU is shared or partitioned): at least one step was '
                        'applied without the global-norm clip.  Re-run with KTUP_CLIP_STEP=0 (two launches, no barrier).')


def training_loop(FLAGS, model, trainer, logger, do_step, do_eval, loss_names, on_train_mode=None, sampler=None, stepper=None):
    """The reference's loop skeleton: early stopping, evaluation every eval_interval_steps (including step 0, where
    only the metrics are logged), otherwise one optimisation step.  `do_step(step)` returns (name, loss_tensor);
    `do_eval(mean_losses)` returns the performance list whose first entry drives checkpointing / LR decay.  `sampler`: the
    on-device negative sampler, whose failure counter is checked where the loop syncs anyway (before every evaluation and
    at the end): a draw without an admissible candidate raises instead of training on a stand-in.  `stepper`: device-fed steps
    (fast_train fed_step) return no loss tensor -- they sum their losses on the device, collected here with the other totals."""
    pbar = None
    sums = {k: torch.zeros((), device=DEV) for k in loss_names}
    model.train(); model.enable_grad()
    while trainer.step < FLAGS.training_steps:                 # a device-fed do_step may run several steps per call (fed_cycle)
        if FLAGS.early_stopping_steps_to_wait > 0 and (trainer.step - trainer.best_step) > FLAGS.early_stopping_steps_to_wait:
            logger.info('No improvement after ' + str(FLAGS.early_stopping_steps_to_wait) + ' steps. Stopping training.')
            break
        if trainer.step % FLAGS.eval_interval_steps == 0:
            if pbar is not None:
                pbar.close()
            totals = {k: float(v.item()) for k, v in sums.items()}      # the only loss read-back
            if stepper is not None:
                for k, v in stepper.take_sums().items():
                    totals[k] = totals.get(k, 0.0) + v
            if sampler is not None:
                sampler.check()
            _check_clip_barrier(trainer)
            do_eval(totals)
            pbar = tqdm(total=FLAGS.eval_interval_steps, desc='Training')
            for v in sums.values():
                v.zero_()
            model.train(); model.enable_grad()
        before = trainer.step
        name, loss = do_step(before)
        if loss is not None:
            sums[name] += loss.detach()
        pbar.update(trainer.step - before)
    if pbar is not None:
        pbar.close()
    if sampler is not None:
        sampler.check()
    if stepper is not None and hasattr(stepper, 'joint'):      # -shard_tables: steps after the last evaluation are checked here
        stepper.joint.check()
    _check_clip_barrier(trainer)
