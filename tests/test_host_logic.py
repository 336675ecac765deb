"""CPU tests of the host side of the boundary: flags, file formats, samplers, iterators, joint schedule, metrics,
trainer checkpoints.  (G7-G9 of SURVEY.md section 8c: constraint / structure checks, no RNG parity with the reference.)"""
import collections
import logging
import os
import random

import numpy as np
import pytest
import torch

from tests.synth import make_dataset


def test_flags_gflags_compatible_surface():
    from jTransUP.models.base import flag_defaults, get_flags
    from jTransUP.utils.flags import FLAGS, FlagError
    get_flags(); FLAGS.reset()
    rest = FLAGS(['prog', '-model_type', 'jtransup', '-noshare_embeddings', '-L1_flag', '-embedding_size', '100', '-joint_ratio=0.7',
                  '--seed', '3', '-nohas_visualization', '-rec_test_files', 'valid.dat:test.dat', 'positional'])
    assert rest == ['prog', 'positional']
    assert (FLAGS.model_type, FLAGS.L1_flag, FLAGS.embedding_size, FLAGS.joint_ratio, FLAGS.seed, FLAGS.has_visualization) == \
        ('jtransup', True, 100, 0.7, 3, False)
    # reference defaults (base.py:22-98)
    assert (FLAGS.batch_size, FLAGS.num_preferences, FLAGS.topn, FLAGS.margin, FLAGS.optimizer_type, FLAGS.l2_lambda,
            FLAGS.clipping_max_value, FLAGS.eval_interval_steps, FLAGS.negtive_samples) == (512, 4, 10, 1.0, 'Adagrad', 1e-5, 5.0, 14000, 1)
    FLAGS.share_embeddings = True
    flag_defaults(FLAGS)
    assert FLAGS.share_embeddings is False                 # forced off for jtransup (base.py:120-123)
    assert FLAGS.data_path == '../datasets/' and FLAGS.ckpt_path == FLAGS.log_path
    with pytest.raises(FlagError):
        FLAGS(['prog', '-model_type', 'bpr'])               # the enum spells it `bprmf`
    with pytest.raises(FlagError):
        FLAGS(['prog', '-no_such_flag', '1'])
    assert 'model_type' in FLAGS.FlagValuesDict()
    FLAGS.reset()


def test_file_formats_and_totals(tmp_path):
    from jTransUP.data import load_kg_rating_data, load_rating_data, load_triple_data
    d = make_dataset(str(tmp_path))
    train, evals, u_map, i_map = load_rating_data.load_data(d, ['valid.dat', 'test.dat'], 16)
    assert train[1] == len(train[2]) == 720 and len(evals) == 2 and evals[0][1] == 90
    assert all(i in train[3][u] for u, i in train[2])
    assert max(len(u_map), max(u_map.values())) == 40 and max(len(i_map), max(i_map.values())) == 50
    assert sum(len(b) for b in evals[0][0]) == len(evals[0][3])            # eval iterator keeps the tail batch
    ttrain, tevals, e_map, r_map = load_triple_data.load_data(os.path.join(d, 'kg'), ['valid.dat'], 16)
    h, t, r = ttrain[2][0]
    first = open(os.path.join(d, 'kg', 'train.dat')).readline().split('\t')
    assert (h, t, r) == (int(first[0]), int(first[1]), int(first[2]))       # head, TAIL, relation order
    assert h in ttrain[3][(t, r)] and t in ttrain[4][(h, r)]
    out = load_kg_rating_data.load_data(d, ['valid.dat'], ['valid.dat'], 16)
    i_remap, e_remap, ikg_map = out[3], out[6], out[8]
    assert len(i_remap) == 50 and len(e_remap) == 60 and len(ikg_map) == 50 + 60 - 35
    for i_id, idx in i_remap.items():
        assert ikg_map[idx][1] == i_id
    assert sum(1 for v in ikg_map.values() if v[0] != -1 and v[1] != -1) == 35


def test_negative_samplers_constraints(tmp_path):
    from jTransUP.data import load_rating_data, load_triple_data
    from jTransUP.utils.data import getNegRatings, getTrainTripleBatch
    d = make_dataset(str(tmp_path))
    train, evals, _, _ = load_rating_data.load_data(d, ['valid.dat', 'test.dat'], 32)
    all_dicts = [train[3]] + [e[3] for e in evals]
    random.seed(1)
    for _ in range(20):
        batch = next(train[0])
        u, pi, ni = getNegRatings(batch, 50, all_dicts=all_dicts)
        assert len(ni) == len(set(ni)) == len(batch)                         # unique inside the batch (data.py:77-82)
        for uu, p, n in zip(u, pi, ni):
            assert n != p and all(n not in dic.get(uu, ()) for dic in all_dicts)
    with pytest.raises(TypeError):
        getNegRatings(next(train[0]), 50, all_dicts=None)                    # the reference crashes here too (data.py:79)
    ttrain, tevals, _, _ = load_triple_data.load_data(os.path.join(d, 'kg'), ['valid.dat'], 64)
    hd, td = [ttrain[3], tevals[0][4]], [ttrain[4], tevals[0][5]]
    heads = tails = 0
    for _ in range(30):
        batch = [tuple(x) for x in next(ttrain[0])]
        ph, pt, pr, nh, nt, nr = getTrainTripleBatch(batch, 60, all_head_dicts=hd, all_tail_dicts=td)
        assert pr == nr
        for a, b, c, e, f in zip(ph, pt, pr, nh, nt):
            assert (a != e) != (b != f)                                      # exactly one side is corrupted
            if a != e:
                heads += 1
                assert all(e not in dic.get((b, c), ()) for dic in hd)
            else:
                tails += 1
                assert all(f not in dic.get((a, c), ()) for dic in td)
    assert 0.4 < heads / float(heads + tails) < 0.6                          # fair coin (data.py:13-14)


def test_train_iterator_drops_tail_and_reshuffles():
    from jTransUP.utils.data import MakeEvalIterator, MakeTrainIterator
    random.seed(0)
    data = [(i, i) for i in range(10)]
    it = MakeTrainIterator(data, 4)
    epoch1 = [tuple(x) for _ in range(2) for x in next(it)]
    epoch2 = [tuple(x) for _ in range(2) for x in next(it)]
    assert len(set(epoch1)) == 8 and len(set(epoch2)) == 8                   # 2 full batches per epoch, tail of 2 dropped
    ev = MakeEvalIterator(list(range(10)), np.dtype('int'), 4)
    assert [len(b) for b in ev] == [4, 4, 2]


def test_joint_schedule_matches_reference_rule():
    for ratio, nrec in ((0.5, 5), (0.7, 7), (0.9, 9)):
        assert [s % 10 < 10 * ratio for s in range(10)] == [s < nrec for s in range(10)]


def test_ndcg_known_answers_and_metrics():
    from jTransUP.utils.evaluation import dcg_at_k, ndcg_at_k
    from jTransUP.utils.ranking import rec_metrics
    assert ndcg_at_k([2, 1, 2, 0], 4) == pytest.approx(0.9203032077642922, rel=1e-15)
    assert ndcg_at_k([2, 1, 2, 0], 4, method=1) == pytest.approx(0.96519546960144276, rel=1e-15)
    assert ndcg_at_k([0], 1) == 0.0 and ndcg_at_k([1], 2) == 1.0
    assert dcg_at_k([3, 2, 3, 0, 0, 1, 2, 2, 3, 0], 10, method=0) == pytest.approx(9.6051177391888114)
    f1, p, r, hit, ndcg = rec_metrics([5, 9, 1, 7], {9, 7, 33})
    assert (p, r, hit) == (0.5, 2 / 3.0, 1) and f1 == pytest.approx(2 * 0.5 * (2 / 3.0) / (0.5 + 2 / 3.0))
    assert rec_metrics([1, 2], {3}) == (0.0, 0.0, 0.0, 0, 0.0)


def test_trainer_checkpoint_layout_and_pretrain_load(tmp_path):
    """State-dict keys, save/load round trip, and loadEmbedding's E -> E+1 padded-entity case (trainer.py:164-169)."""
    from jTransUP.models import jTransUP as jt, transH
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    from jTransUP.utils.trainer import ModelTrainer, get_model_target
    get_flags(); FLAGS.reset()
    FLAGS(['prog', '-model_type', 'transh', '-log_path', str(tmp_path), '-experiment_name', 'th', '-optimizer_type', 'Adam'])
    FLAGS.ckpt_path = str(tmp_path)
    logger = logging.getLogger('t')
    torch.manual_seed(0)
    th = transH.TransHModel(False, 8, 11, 3)
    tr = ModelTrainer(th, logger, 10, FLAGS)
    tr.step, tr.best_step, tr.best_dev_performance = 7, 5, 0.25
    tr.save(tr.checkpoint_path)
    ck = torch.load(tr.checkpoint_path, weights_only=False)
    assert sorted(ck) == ['best_dev_performance', 'best_step', 'model_state_dict', 'optimizer_state_dict', 'step']
    assert sorted(ck['model_state_dict']) == ['ent_embeddings.weight', 'norm_embeddings.weight', 'rel_embeddings.weight']
    th2 = transH.TransHModel(False, 8, 11, 3)
    tr2 = ModelTrainer(th2, logger, 10, FLAGS)
    tr2.load(tr.checkpoint_path, cpu=True)
    assert (tr2.step, tr2.best_step, tr2.best_dev_performance) == (7, 5, 0.25)
    assert torch.equal(th2.ent_embeddings.weight.cpu(), th.ent_embeddings.weight.cpu())
    im = {i: i for i in range(6)}
    nm = {i: (i, i) for i in range(6)}
    k = jt.jTransUPModel(False, 8, 4, 6, 11, 3, im, nm, False, False)
    FLAGS.model_type = 'jtransup'
    trk = ModelTrainer(k, logger, 10, FLAGS)
    trk.loadEmbedding(tr.checkpoint_path, k.state_dict())
    assert torch.equal(k.ent_embeddings.weight[:11].cpu(), th.ent_embeddings.weight.cpu())
    assert float(k.ent_embeddings.weight[11].abs().sum()) == 0.0            # pad row untouched
    assert torch.equal(k.rel_embeddings.weight.cpu(), th.rel_embeddings.weight.cpu())
    assert get_model_target('bprmf') == 1 and get_model_target('jtransup') == -1
    FLAGS.reset()


def test_binary_dataset_cache_round_trip_and_staleness(tmp_path, monkeypatch):
    """data/cache.py: the second load comes from the .npz cache (same structures, same dict order), an edited source file
    is re-parsed, KTUP_DATA_CACHE=0 bypasses the cache; cache files are numpy-only data (no pickle: a planted file cannot run
    code and is simply ignored)."""
    import os
    import time
    from jTransUP.data import cache, load_kg_rating_data
    make_dataset(str(tmp_path))
    root = os.path.join(str(tmp_path), 'ml1m')
    monkeypatch.delenv('KTUP_DATA_CACHE', raising=False)
    calls = []
    real = cache._read
    monkeypatch.setattr(cache, '_read', lambda f: (real(f), calls.append(1))[0])     # counts successful reads
    assert not hasattr(cache, 'pickle')
    first = load_kg_rating_data.load_data(root, ['valid.dat'], ['valid.dat'], 16)
    assert not calls and os.path.isdir(os.path.join(root, '.ktup_cache'))
    second = load_kg_rating_data.load_data(root, ['valid.dat'], ['valid.dat'], 16)
    assert len(calls) >= 8                                          # every parsed file came from its cache file
    for a, b in ((first[0], second[0]), (first[4], second[4])):     # train datasets: totals, lists, dicts
        assert a[1:] == b[1:]
    assert list(first[3].items()) == list(second[3].items()) and first[8] == second[8]     # i_remap order, joint map
    # stale: append a rating -> the train file is parsed again and the new pair shows up
    train = os.path.join(root, 'train.dat')
    with open(train, 'a') as f:
        f.write('0\t1\t5\n')
    os.utime(train, ns=(time.time_ns(), time.time_ns()))
    third = load_kg_rating_data.load_data(root, ['valid.dat'], ['valid.dat'], 16)
    assert third[0][1] == first[0][1] + 1
    # a planted / corrupt cache file is data, not code: it fails to load as an .npz and the source is parsed instead
    import pickle
    planted = cache._cache_file(train)
    with open(planted, 'wb') as f:
        pickle.dump(('boom',), f)
    fourth = load_kg_rating_data.load_data(root, ['valid.dat'], ['valid.dat'], 16)
    assert fourth[0][1:] == third[0][1:]
    monkeypatch.setenv('KTUP_DATA_CACHE', '0')
    calls.clear()
    load_kg_rating_data.load_data(root, ['valid.dat'], ['valid.dat'], 16)
    assert not calls


def test_kg_summary_and_mrr_from_the_reference_rank_lists():
    """summarize_kg keeps the reference's numbers (hit@n, mean 0-based rank) and adds MRR = mean 1 / (rank + 1) over the same
    filtered ranks -- rows (report mode) and (n x 2) arrays (fast path) agree with the oracle's walk."""
    import logging
    import types
    import numpy as np
    from jTransUP.models import _driver as D
    from oracle import cpu_ref as O
    rng = np.random.RandomState(0)
    FL = types.SimpleNamespace(topn=10)
    head, tail, all_ranks = [], [], {'h': [], 't': []}
    for side, out in (('h', head), ('t', tail)):
        for q in range(40):
            pred = rng.rand(300).astype(np.float32)
            gold = set(rng.choice(300, size=rng.randint(1, 5), replace=False).tolist())
            filt = set(rng.choice(300, size=30, replace=False).tolist()) - gold
            hits, ranks, ids = O.kg_performance(pred, gold, filt, topn=10)
            out.extend((h, r, (q, 0), g) for h, r, g in zip(hits, ranks, ids))
            all_ranks[side].extend(ranks)
    log = logging.getLogger('mrr-test')
    avg_hit, avg_rank = D.summarize_kg(FL, head, tail, log)
    n = len(head) + len(tail)
    assert abs(avg_rank - (sum(all_ranks['h']) + sum(all_ranks['t'])) / n) < 1e-9
    assert abs(avg_hit - (sum(r[0] for r in head) + sum(r[0] for r in tail)) / n) < 1e-12
    want = (O.mrr_from_ranks(all_ranks['h'] + all_ranks['t']), O.mrr_from_ranks(all_ranks['h']), O.mrr_from_ranks(all_ranks['t']))
    np.testing.assert_allclose(D.kg_mrr(head, tail), want, rtol=1e-12)
    cols = lambda rows: np.array([[r[0], r[1]] for r in rows], dtype=np.float64)
    np.testing.assert_allclose(D.kg_mrr(cols(head), cols(tail)), want, rtol=1e-12)
    assert O.mrr_from_ranks([0, 1, 3]) == (1 + 0.5 + 0.25) / 3


def test_device_feeder_iterator_contract_on_cpu():
    """DeviceFeeder mirrors MakeTrainIterator (utils/data.py:87-110): endless, the order holds every example
    `negtive_samples` times, and -- the reference's cadence -- it reshuffles once start > n - batch_size with n the number
    of DISTINCT examples (so with negtive_samples > 1 only the first n order entries are consumed per shuffle), tail partial
    batch dropped.  (The class only needs a torch device, so it runs on CPU here.)"""
    import torch
    from jTransUP.utils.fast_train import DeviceFeeder
    rows = [(i, i * 2 % 17, 1) for i in range(50)]
    f = DeviceFeeder(rows, 16, torch.device('cpu'), negtive_samples=2, seed=5)
    per_epoch = 50 // 16                                         # starts 0, 16, 32; 48 > 50 - 16 wraps (data.py:101-103)
    epochs = []
    for _ in range(3):
        got = torch.cat([f.next() for _ in range(per_epoch)])
        assert got.shape == (per_epoch * 16, 3)
        assert all(tuple(r) in set(rows) for r in got.tolist())
        counts = torch.bincount(got[:, 0], minlength=50)
        assert int(counts.max()) <= 2 and int(counts.sum()) == per_epoch * 16      # each example at most `negtive_samples` times
        epochs.append((got, f.order))
    assert epochs[0][1] is not epochs[1][1] and not torch.equal(epochs[0][1], epochs[1][1])   # a new permutation every epoch
    # same cadence as the host iterator the reference uses: count its reshuffles over the same number of batches
    import random
    from jTransUP.utils import data as host_data
    shuffles = []
    real = random.shuffle
    try:
        random.shuffle = lambda x: (shuffles.append(1), real(x))[1]
        it = host_data.MakeTrainIterator(rows, 16, negtive_samples=2)
        for _ in range(3 * per_epoch):
            next(it)
    finally:
        random.shuffle = real
    assert len(shuffles) == 3                                    # initial shuffle + one per wrap, like the feeder's 3 orders
    with pytest.raises(ValueError):
        DeviceFeeder(rows[:5], 16, torch.device('cpu'))


def test_rank_index_csr_layout_on_cpu():
    """RankIndex: CSR filter / gold sets of an evaluation pass (filters = union over all_dicts, golds ascending, keys absent
    from eval_dict flagged), contiguous-slice look-ups and their memoisation."""
    import torch
    from jTransUP.utils.ranking import RankIndex
    keys = [3, 7, 9, 11]
    eval_dict = {3: {5, 1}, 9: {2}, 11: {8, 0, 4}}
    a, b = {3: {1, 9}, 7: {4}}, {3: {2}, 11: {6}}
    idx = RankIndex(keys, eval_dict, [a, b], torch.device('cpu'))
    assert idx.f_off.tolist() == [0, 3, 4, 4, 5] and idx.f_ids.tolist() == [1, 2, 9, 4, 6]
    assert idx.g_off.tolist() == [0, 2, 2, 3, 6] and idx.g_ids.tolist()[:6] == [1, 5, 2, 0, 4, 8]
    assert idx.present_h.tolist() == [True, False, True, True]
    batch = [7, 9]
    assert idx.rows_of(batch) == (1, 3) and idx.rows_of(batch) == (1, 3)
    f_off, f_ids = idx.filter_slice(1, 3)
    assert f_off.tolist() == [0, 1, 1] and f_ids.tolist()[:1] == [4]
    assert idx.filter_slice(1, 3)[0] is f_off                      # memoised
    g_off, g_ids, g_off_h, g_ids_h = idx.gold_slice(1, 3)
    assert g_off.tolist() == [0, 0, 1] and g_ids_h.tolist() == [2]
    with pytest.raises(KeyError):
        idx.rows_of([9, 7])
    none = RankIndex(keys, eval_dict, None, torch.device('cpu'))
    assert none.filter_slice(0, 2) == (None, None)


def test_wave_model_counts_the_waits_a_lone_wave_cannot_hide():
    """tools/wave_model.py (the one-wave issue model used to compare schedules of the K5-K7 backward tile): an LDS read waited for
    right away costs its latency; the same read issued ahead of eight independent MFMAs costs nothing; dependent MFMAs serialise on
    the matrix pipe either way; a VALU read of an MFMA result waits for the pipe."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('wave_model', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'wave_model.py'))
    W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
    mf = ['v_mfma_f32_16x16x4_f32 a[%d:%d], v1, v2, a[%d:%d]' % (4 * i, 4 * i + 3, 4 * i, 4 * i + 3) for i in range(8)]
    late = W.model(mf + ['ds_read_b128 v[10:13], v0', 's_waitcnt lgkmcnt(0)', 'v_mfma_f32_16x16x4_f32 a[0:3], v10, v2, a[0:3]'])
    early = W.model(['ds_read_b128 v[10:13], v0'] + mf + ['s_waitcnt lgkmcnt(0)', 'v_mfma_f32_16x16x4_f32 a[0:3], v10, v2, a[0:3]'])
    assert late['mfma'] == early['mfma'] == 9
    assert early['stall_lgkm'] == 0 and late['stall_lgkm'] > 0
    assert late['clocks'] > early['clocks']
    chain = W.model(['v_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]'] * 4 + ['v_accvgpr_read_b32 v5, a0'])
    assert chain['clocks'] >= 4 * 32 and chain['stall_mfma_dep'] > 0
    thin = W.model(['v_mfma_f32_4x4x1_16b_f32 a[0:3], v1, v2, a[0:3]'] * 4)
    assert thin['clocks'] < W.model(['v_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]'] * 4)['clocks']


def test_gate_helper_methods_of_the_preference_models():
    """transUP.py:118-170 / jTransUP.py:262-314: convert_to_one_hot, masked_softmax, st_gumbel_softmax exist on both preference
    models with the reference's semantics (one-hot forward value, softmax gradient, noise from torch's global generator)."""
    import torch
    import torch.nn.functional as F
    from oracle import cpu_ref as O
    from jTransUP.models.transUP import TransUPModel
    from jTransUP.models.jTransUP import jTransUPModel
    tup = TransUPModel(False, 8, 5, 6, 3, True)
    ktup = jTransUPModel(False, 8, 5, 6, 7, 3, {i: i for i in range(6)}, {i: (i, i) for i in range(6)}, False, True)
    for m in (tup, ktup):
        idx = torch.tensor([[2, 0], [1, 3]])
        assert torch.equal(m.convert_to_one_hot(idx, 4), F.one_hot(idx, 4)) and m.convert_to_one_hot(idx, 4).dtype == idx.dtype
        logits = torch.randn(4, 7, 5, requires_grad=True)
        assert torch.allclose(m.masked_softmax(logits), F.softmax(logits, dim=2))
        torch.manual_seed(11)
        y = m.st_gumbel_softmax(logits, temperature=0.7)
        torch.manual_seed(11)
        uni = torch.empty(4, 7, 5).uniform_()                       # the draw the method makes (shim 4 of tests/golden/make_goldens.py)
        eps = 1e-20
        soft = F.softmax((logits + (-torch.log(-torch.log(uni + eps) + eps))) / 0.7, dim=2)
        assert torch.equal(y.detach(), F.one_hot(soft.argmax(2), 5).float())          # forward value: the one-hot
        g, = torch.autograd.grad((y * torch.arange(5.)).sum(), logits)
        g_soft, = torch.autograd.grad((soft * torch.arange(5.)).sum(), logits)
        assert torch.allclose(g, g_soft)                            # backward: the softmax's
        torch.manual_seed(11)
        assert torch.equal(m.st_gumbel_softmax(logits).detach(), O.st_gumbel_softmax(logits.detach(), uni))


def test_shard_tables_refuses_what_it_cannot_train_by_flag_name():
    """utils/sharded_train.check_flags: -shard_tables updates only the rows a batch touches, which equals the reference's dense step
    (knowledgable_recommendation.py:394-403) for Adagrad / plain SGD / Adam -- with the steps a row was not touched for replayed where the
    dense step moves untouched rows (Adam; any of them under -l2_lambda > 0) -- everything else is refused before any table is sharded,
    naming the reference's flag."""
    import types
    from jTransUP.hip import lib as L
    from jTransUP.utils import sharded_train as S
    model = types.SimpleNamespace(embedding_size=100, rel_total=20)
    ok = dict(model_type='jtransup', share_embeddings=False, optimizer_type='Adagrad', momentum=0.0, l2_lambda=0.0, use_st_gumbel=False)
    S.check_flags(types.SimpleNamespace(**ok), model)                                  # the supported combination passes
    S.check_flags(types.SimpleNamespace(**dict(ok, optimizer_type='SGD')), model)
    S.check_flags(types.SimpleNamespace(**dict(ok, optimizer_type='Adam')), model)       # ktup.sh's optimizer: row-sparse with catch-up
    S.check_flags(types.SimpleNamespace(**dict(ok, l2_lambda=1e-5)), model)              # the reference's default weight decay (base.py:51)
    S.check_flags(types.SimpleNamespace(**dict(ok, use_st_gumbel=True)), model)          # transup.sh's gate
    tup_model = types.SimpleNamespace(embedding_size=100, pref_embeddings=types.SimpleNamespace(weight=torch.zeros(20, 100)))
    S.check_flags(types.SimpleNamespace(**dict(ok, model_type='transup')), tup_model)    # TUP: the user / item tables alone
    for change, word in ((dict(model_type='transe'), 'jtransup'), (dict(share_embeddings=True), 'noshare_embeddings'),
                         (dict(optimizer_type='Rmsprop'), 'optimizer_type'), (dict(optimizer_type='SGD', momentum=0.9), 'momentum'),
                         (dict(l2_lambda=-1e-5), 'l2_lambda')):
        with pytest.raises(L.KtupError) as e:
            S.check_flags(types.SimpleNamespace(**dict(ok, **change)), model)
        assert word in str(e.value), (change, str(e.value))
    with pytest.raises(L.KtupError) as e:                                              # a width without fused step kernels
        S.check_flags(types.SimpleNamespace(**ok), types.SimpleNamespace(embedding_size=36, rel_total=20))
    assert 'embedding_size 36' in str(e.value)


def test_sparse_adam_host_side_layouts():
    """jTransUP/sharded_ktup.py: what the host hands the row-sparse Adam kernels (include/ktup_hip.h ktup_adam_t) -- the struct's C layout,
    the state row [m | v | last], how many missed steps are replayed one by one for the betas in use, and which state a table needs."""
    import ctypes
    import math
    from jTransUP.sharded_ktup import KINDS, RULES, AdamRule, _check_state, adam_replay, adam_state_pitch, is_lazy, row_state
    assert KINDS == {'sgd': 0, 'adagrad': 1, 'adam': 2}                                   # KTUP_OPT_SGD / _ADAGRAD / _ADAM
    assert RULES == {'adam': 0, 'adagrad': 1, 'sgd': 2}                                   # ktup_adam_t.rule
    # {float, float, int32 replay, int32 rule, pointer, float weight_decay, float}
    assert ctypes.sizeof(AdamRule) == 32 and AdamRule.step.offset == 16 and AdamRule.replay.offset == 8 and AdamRule.rule.offset == 12
    assert AdamRule.weight_decay.offset == 24
    assert is_lazy('adam') and is_lazy('adagrad', 1e-5) and is_lazy('sgd', 1e-5) and not is_lazy('adagrad') and not is_lazy('sgd', 0.0)
    assert adam_state_pitch(256) == 516 and adam_state_pitch(100) % 4 == 0                # rows stay 16-byte aligned
    for betas in ((0.9, 0.999), (0.8, 0.99), (0.95, 0.999)):
        k, r = adam_replay(betas), betas[0] / math.sqrt(betas[1])
        assert r ** k <= 1e-5 < r ** (k - 1)                                              # the first k whose increment is below 1e-5 of the first
    assert adam_replay((0.9, 0.999)) == 110 and adam_replay((0.0, 0.999)) == 0
    w = torch.zeros(7, 12)
    assert row_state(w, 'sgd') is None and row_state(w, 'adagrad').shape == (7, 12) and row_state(w, 'adam').shape == (7, 28)
    assert _check_state(None, w, 'sgd') and _check_state(row_state(w, 'adam'), w, 'adam') and not _check_state(row_state(w, 'adagrad'), w, 'adam')
    assert not _check_state(None, w, 'adagrad')                                           # a stepper of another kind re-creates the state
    # under weight decay every kind keeps the lazy rows [m | v | last] (Adagrad's sum in the v half)
    assert row_state(w, 'adagrad', 1e-5).shape == (7, 28) and row_state(w, 'sgd', 1e-5).shape == (7, 28)
    assert _check_state(row_state(w, 'sgd', 1e-5), w, 'sgd', 1e-5) and not _check_state(row_state(w, 'adagrad'), w, 'adagrad', 1e-5)


def test_which_tables_take_their_gradient_directly():
    """hip/ops.py _grad_targets: the score Functions add into an existing `.grad` only for dense contiguous fp32 LEAVES without hooks, outside
    create_graph, and while set_direct_grad is on; everything else gets zero-filled buffers out of one allocation and the autograd
    hand-over (the host-side decision, on CPU tensors: no kernel runs here)."""
    from jTransUP.hip import ops
    p = torch.nn.Parameter(torch.randn(6, 8))
    assert not ops._takes_grad_directly(p)                                   # no .grad yet: autograd allocates (and steals) it
    p.grad = torch.zeros(6, 8)
    assert ops._takes_grad_directly(p)
    assert not ops._takes_grad_directly(p * 2.0)                             # a non-leaf (the zero-tail staged tables)
    frozen = torch.randn(6, 8)
    assert not ops._takes_grad_directly(frozen)                              # requires_grad False
    wide = torch.nn.Parameter(torch.randn(6, 16)[:, :8])                     # a view with a row pitch: the kernels' gradient pitch is the table's
    wide.grad = torch.zeros(6, 8)
    assert not ops._takes_grad_directly(wide)
    half = torch.nn.Parameter(torch.randn(6, 8))
    half.grad = torch.zeros(6, 8)
    half.grad.data = half.grad.data.double()
    assert not ops._takes_grad_directly(half)
    hooked = torch.nn.Parameter(torch.randn(6, 8)); hooked.grad = torch.zeros(6, 8)
    hooked.register_hook(lambda g: g)
    assert not ops._takes_grad_directly(hooked)                              # a hook must see the gradient: hand-over
    q = torch.nn.Parameter(torch.randn(3, 8)); q.grad = torch.ones(3, 8)
    with torch.no_grad():                                                    # (backward without create_graph runs with grad mode off)
        bufs, rets = ops._grad_targets(p, None, q, frozen)
        assert bufs[0] is p.grad and rets[0] is None and bufs[1] is None and rets[1] is None and bufs[2] is q.grad and rets[2] is None
        assert bufs[3] is rets[3] and bufs[3].shape == frozen.shape and float(bufs[3].abs().sum()) == 0.0
        was = ops.set_direct_grad(False)
        try:
            bufs, rets = ops._grad_targets(p, q)
            assert bufs[0] is rets[0] and bufs[0] is not p.grad and bufs[1] is rets[1] and float(bufs[0].abs().sum() + bufs[1].abs().sum()) == 0.0
            assert bufs[0].data_ptr() % 16 == 0 and bufs[1].data_ptr() % 16 == 0     # one allocation, 16-byte aligned parts
        finally:
            assert ops.set_direct_grad(was) is False
        bufs, rets = ops._grad_targets(p, p)                                 # one table in two roles: one direct buffer, one of its own
        assert bufs[0] is p.grad and rets[0] is None and bufs[1] is rets[1] and bufs[1] is not p.grad
    bufs, rets = ops._grad_targets(p)                                        # grad mode on (create_graph): hand-over
    assert bufs[0] is rets[0] and bufs[0] is not p.grad


def test_bench_leg_steps_hook(monkeypatch):
    """bench.py _leg_steps: KTUP_BENCH_LEG_STEPS shortens the N-GPU legs for the two-ranks-on-one-GPU contract test only."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    monkeypatch.delenv('KTUP_BENCH_LEG_STEPS', raising=False)
    assert bench._leg_steps(100, 20) == (100, 20)
    monkeypatch.setenv('KTUP_BENCH_LEG_STEPS', '12')
    assert bench._leg_steps(100, 20) == (12, 4)
    monkeypatch.setenv('KTUP_BENCH_LEG_STEPS', '0')
    assert bench._leg_steps(200, 20) == (200, 20)
