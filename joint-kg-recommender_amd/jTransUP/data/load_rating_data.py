"""Rating files -> lists / dicts / iterators (jTransUP/data/load_rating_data.py).  Formats: `u\\ti\\trating` per line
in train.dat / valid.dat / test.dat (rating parsed as int, then ignored); `mapped_id\\toriginal` in u_map.dat, i_map.dat."""
import os

import numpy as np

from jTransUP.data.cache import cached
from jTransUP.utils.data import MakeEvalIterator, MakeTrainIterator


def loadVocab(filename):
    """original id -> mapped id."""
    vocab = {}
    with open(filename, 'r', encoding='utf-8') as fin:
        for line in fin:
            parts = line.strip().split('\t')
            if len(parts) == 2:
                vocab[parts[1]] = int(parts[0])
    return vocab


def loadRatings(filename):
    """-> (count, [(u, i), ...], {u: set(items)})."""
    pairs, by_user = [], {}
    with open(filename, 'r', encoding='utf-8') as fin:
        for line in fin:
            parts = line.strip().split('\t')
            if len(parts) != 3:
                continue
            u, i, _ = int(parts[0]), int(parts[1]), int(parts[2])
            pairs.append((u, i))
            by_user.setdefault(u, set()).add(i)
    return len(pairs), pairs, by_user


def load_data(data_path, eval_filenames, batch_size, negtive_samples=1, logger=None):
    train_total, train_list, train_dict = cached(os.path.join(data_path, 'train.dat'), loadRatings)
    eval_files = [os.path.join(data_path, f) for f in eval_filenames]
    evals = [cached(f, loadRatings) for f in eval_files]
    if logger is not None:
        logger.info('Totally {} train ratings, {} eval ratings in files: {}!'.format(
            train_total, ','.join(str(e[0]) for e in evals), ';'.join(eval_files)))
    u_map = cached(os.path.join(data_path, 'u_map.dat'), loadVocab)
    i_map = cached(os.path.join(data_path, 'i_map.dat'), loadVocab)
    if logger is not None:
        logger.info('successfully load {} users and {} items!'.format(len(u_map), len(i_map)))
    train_iter = MakeTrainIterator(train_list, batch_size, negtive_samples=negtive_samples)
    eval_datasets = [[MakeEvalIterator(list(e[2].keys()), np.dtype('int'), batch_size), e[0], e[1], e[2]] for e in evals]
    return (train_iter, train_total, train_list, train_dict), eval_datasets, u_map, i_map
