"""Synthetic dataset in the reference's on-disk formats for host-logic and end-to-end tests."""
import os

import numpy as np


def make_dataset(root, n_users=40, n_items=50, n_ent=60, n_rel=5, n_ratings=900, n_triples=700, aligned=35, seed=0):
    rng = np.random.RandomState(seed)
    d = os.path.join(root, 'ml1m')
    os.makedirs(os.path.join(d, 'kg'), exist_ok=True)
    pairs = set()
    while len(pairs) < n_ratings:
        pairs.add((int(rng.randint(n_users)), int(rng.randint(n_items))))
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    cut1, cut2 = int(0.8 * len(pairs)), int(0.9 * len(pairs))
    for name, part in (('train.dat', pairs[:cut1]), ('valid.dat', pairs[cut1:cut2]), ('test.dat', pairs[cut2:])):
        with open(os.path.join(d, name), 'w') as f:
            for u, i in part:
                f.write('%d\t%d\t1\n' % (u, i))
    with open(os.path.join(d, 'u_map.dat'), 'w') as f:
        for u in range(n_users):
            f.write('%d\tuser%d\n' % (u, u))
    with open(os.path.join(d, 'i_map.dat'), 'w') as f:
        for i in range(n_items):
            f.write('%d\titem%d\n' % (i, i))
    triples = set()
    while len(triples) < n_triples:
        triples.add((int(rng.randint(n_ent)), int(rng.randint(n_ent)), int(rng.randint(n_rel))))
    triples = sorted(triples)
    rng.shuffle(triples)
    cut1, cut2 = int(0.8 * len(triples)), int(0.9 * len(triples))
    for name, part in (('train.dat', triples[:cut1]), ('valid.dat', triples[cut1:cut2]), ('test.dat', triples[cut2:])):
        with open(os.path.join(d, 'kg', name), 'w') as f:
            for h, t, r in part:
                f.write('%d\t%d\t%d\n' % (h, t, r))          # head, TAIL, relation
    with open(os.path.join(d, 'kg', 'e_map.dat'), 'w') as f:
        for e in range(n_ent):
            f.write('%d\thttp://kg/e%d\n' % (e, e))
    with open(os.path.join(d, 'kg', 'r_map.dat'), 'w') as f:
        for r in range(n_rel):
            f.write('%d\trel%d\n' % (r, r))
    items = rng.permutation(n_items)[:aligned]
    ents = rng.permutation(n_ent)[:aligned]
    with open(os.path.join(d, 'i2kg_map.tsv'), 'w') as f:
        for i, e in zip(items, ents):
            f.write('item%d\ttitle %d\thttp://kg/e%d\n' % (i, i, e))
    return d
