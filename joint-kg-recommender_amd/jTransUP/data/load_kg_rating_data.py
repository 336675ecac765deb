"""Joint ratings + triples + item<->entity alignment (jTransUP/data/load_kg_rating_data.py).
i2kg_map.tsv: `orig_item\\ttitle\\turi` per line."""
import os

from jTransUP.data import load_rating_data, load_triple_data


def loadR2KgMap(filename):
    i2kg, kg2i = {}, {}
    with open(filename, 'r', encoding='utf-8') as fin:
        for line in fin:
            parts = line.strip().split('\t')
            if len(parts) == 3:
                i2kg[parts[0]] = parts[2]
                kg2i[parts[2]] = parts[0]
    print('successful load {} item and {} entity pairs!'.format(len(i2kg), len(kg2i)))
    return i2kg, kg2i


def rebuildEntityItemVocab(map1, map2, links):
    """load_kg_rating_data.py:21-48: joint vocabulary.  map1/map2: original -> id; links: original1 -> original2.
    Returns new_map[index] = (id1 | -1, id2 | -1), remap1[id1] = index, remap2[id2] = index, #aligned."""
    new_map, remap1, aligned_at = {}, {}, {}
    for index, (org1, id1) in enumerate(map1.items()):
        id2 = -1
        org2 = links.get(org1)
        if org2 is not None and org2 in map2:
            id2 = map2[org2]
            aligned_at[org2] = index
        new_map[index] = (id1, id2)
        remap1[id1] = index
    index = len(map1)
    remap2 = {}
    for org2, id2 in map2.items():
        if org2 in aligned_at:
            remap2[id2] = aligned_at[org2]
        else:
            new_map[index] = (-1, id2)
            remap2[id2] = index
            index += 1
    return new_map, remap1, remap2, len(aligned_at)


def load_data(data_path, rec_eval_files, kg_eval_files, batch_size, negtive_samples=1, logger=None):
    kg_path = os.path.join(data_path, 'kg')
    rating_train, rating_evals, u_map, i_map = load_rating_data.load_data(data_path, rec_eval_files, batch_size, logger=logger,
                                                                          negtive_samples=negtive_samples)
    triple_train, triple_evals, e_map, r_map = load_triple_data.load_data(kg_path, kg_eval_files, batch_size, logger=logger,
                                                                          negtive_samples=negtive_samples)
    _, kg2i_map = loadR2KgMap(os.path.join(data_path, 'i2kg_map.tsv'))
    ikg_map, e_remap, i_remap, aligned = rebuildEntityItemVocab(e_map, i_map, kg2i_map)
    if logger is not None:
        logger.info('Find {} aligned items and entities!'.format(aligned))
    return rating_train, rating_evals, u_map, i_remap, triple_train, triple_evals, e_remap, r_map, ikg_map
