"""Joint dataset: ratings + triples + the item <-> entity alignment of `i2kg_map.tsv` (lines `orig_item\\ttitle\\turi`).
Entry points and return shapes follow jTransUP/data/load_kg_rating_data.py (loadR2KgMap :5-19, rebuildEntityItemVocab :21-48,
load_data :50-65)."""
import os

from jTransUP.data import load_rating_data, load_triple_data


def _three_column_lines(filename):
    with open(filename, 'r', encoding='utf-8') as fin:
        for raw in fin:
            cols = raw.strip().split('\t')
            if len(cols) == 3:
                yield cols


def loadR2KgMap(filename):
    """-> (original item -> entity uri, entity uri -> original item); malformed lines are skipped."""
    pairs = [(item, uri) for item, _, uri in _three_column_lines(filename)]
    i2kg = dict(pairs)
    kg2i = {uri: item for item, uri in pairs}
    print('successful load {} item and {} entity pairs!'.format(len(i2kg), len(kg2i)))
    return i2kg, kg2i


def rebuildEntityItemVocab(map1, map2, links):
    """Joint vocabulary over two id spaces.  map1 / map2: original name -> id; links: name in space 1 -> name in space 2.
    Index k of the joint vocabulary is the k-th entry of map1 (with its linked partner from map2, if any), followed by the
    entries of map2 that no entry of map1 links to, in map2's order.
    -> new_map[index] = (id1 or -1, id2 or -1), remap1[id1] = index, remap2[id2] = index, number of linked pairs."""
    joint = []                                  # index -> [id1, id2]
    partner_index = {}                          # name in space 2 -> index of the entry it is linked from
    for name1, id1 in map1.items():
        name2 = links.get(name1)
        linked = name2 is not None and name2 in map2
        if linked:
            partner_index[name2] = len(joint)
        joint.append([id1, map2[name2] if linked else -1])
    remap2 = {}
    for name2, id2 in map2.items():
        at = partner_index.get(name2)
        if at is None:
            at = len(joint)
            joint.append([-1, id2])
        remap2[id2] = at
    new_map = {k: (a, b) for k, (a, b) in enumerate(joint)}
    remap1 = {joint[k][0]: k for k in range(len(map1))}
    return new_map, remap1, remap2, len(partner_index)


def load_data(data_path, rec_eval_files, kg_eval_files, batch_size, negtive_samples=1, logger=None):
    """-> rating train / eval datasets, u_map, i_remap, triple train / eval datasets, e_remap, r_map, ikg_map."""
    ratings = load_rating_data.load_data(data_path, rec_eval_files, batch_size, logger=logger, negtive_samples=negtive_samples)
    triples = load_triple_data.load_data(os.path.join(data_path, 'kg'), kg_eval_files, batch_size, logger=logger,
                                         negtive_samples=negtive_samples)
    rating_train, rating_evals, u_map, i_map = ratings
    triple_train, triple_evals, e_map, r_map = triples
    uri_to_item = loadR2KgMap(os.path.join(data_path, 'i2kg_map.tsv'))[1]
    ikg_map, e_remap, i_remap, aligned = rebuildEntityItemVocab(e_map, i_map, uri_to_item)
    if logger is not None:
        logger.info('Find {} aligned items and entities!'.format(aligned))
    return rating_train, rating_evals, u_map, i_remap, triple_train, triple_evals, e_remap, r_map, ikg_map
