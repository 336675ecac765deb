import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import numpy as np, torch
from jTransUP.hip import ops
DEV = 'cuda'
for d in (36, 64, 100):
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'score_d%d.npz' % d)))
    dv = lambda k: torch.from_numpy(g[k]).to(DEV)
    E, R, M = dv('transr.ent_embeddings.weight'), dv('transr.rel_embeddings.weight'), dv('transr.proj_embeddings.weight')
    for l1 in (False, True):
        tag = 'transr.%s.' % ('L1' if l1 else 'L2')
        for a, (h, t) in (('pos', ('ph', 'pt')), ('neg', ('nh', 'nt'))):
            got = ops.score_transr(E, R, M, dv(h), dv(t), dv('pr'), l1).cpu().numpy()
            want = g[tag + a]
            err = np.abs(got - want)
            print('d', d, 'L1' if l1 else 'L2', a, 'max abs', float(err.max()), 'max rel', float((err / np.maximum(np.abs(want), 1e-30)).max()),
                  'viol(1e-4,1e-5)', int((err > 1e-5 + 1e-4 * np.abs(want)).sum()), 'of', want.size, 'scale', float(np.abs(want).max()))
