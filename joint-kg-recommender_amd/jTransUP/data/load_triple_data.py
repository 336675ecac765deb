"""Triple files -> lists / dicts / iterators (jTransUP/data/load_triple_data.py).  Format of kg/*.dat: `h\\tt\\tr`
(TAIL BEFORE RELATION); `mapped_id\\toriginal` in e_map.dat, r_map.dat."""
import os

import numpy as np

from jTransUP.data.cache import cached
from jTransUP.utils.data import MakeEvalIterator, MakeTrainIterator


def loadTriples(filename):
    """-> (count, [(h, t, r)], {(t, r): set(heads)}, {(h, r): set(tails)})."""
    triples, heads, tails = [], {}, {}
    with open(filename, 'r', encoding='utf-8') as fin:
        for line in fin:
            parts = line.strip().split('\t')
            if len(parts) != 3:
                continue
            h, t, r = int(parts[0]), int(parts[1]), int(parts[2])
            triples.append((h, t, r))
            heads.setdefault((t, r), set()).add(h)
            tails.setdefault((h, r), set()).add(t)
    return len(triples), triples, heads, tails


def loadVocab(filename):
    vocab = {}
    with open(filename, 'r', encoding='utf-8') as fin:
        for line in fin:
            parts = line.strip().split('\t')
            if len(parts) == 2:
                vocab[parts[1]] = int(parts[0])
    return vocab


def load_data(kg_path, eval_filenames, batch_size, negtive_samples=1, logger=None):
    train_total, train_list, train_head_dict, train_tail_dict = cached(os.path.join(kg_path, 'train.dat'), loadTriples)
    eval_files = [os.path.join(kg_path, f) for f in eval_filenames]
    evals = [cached(f, loadTriples) for f in eval_files]
    if logger is not None:
        logger.info('Totally {} train triples, {} eval triples in files: {}!'.format(
            train_total, ','.join(str(e[0]) for e in evals), ';'.join(eval_files)))
    e_map = cached(os.path.join(kg_path, 'e_map.dat'), loadVocab)
    r_map = cached(os.path.join(kg_path, 'r_map.dat'), loadVocab)
    if logger is not None:
        logger.info('successfully load {} entities and {} relations!'.format(len(e_map), len(r_map)))
    train_iter = MakeTrainIterator(train_list, batch_size, negtive_samples=negtive_samples)
    dt = np.dtype('int,int')
    eval_datasets = [[MakeEvalIterator(list(e[2].keys()), dt, batch_size), MakeEvalIterator(list(e[3].keys()), dt, batch_size),
                      e[0], e[1], e[2], e[3]] for e in evals]
    return (train_iter, train_total, train_list, train_head_dict, train_tail_dict), eval_datasets, e_map, r_map
