"""Times of the one-wave-per-pair kernels (csrc/ktup_score_pref_row.hip) at -embedding_size > 256: KTUP score + backward of a B = 512
step's 1,024 pairs, and the all-item scores of a 512-user evaluation batch against 3,240 items.
    python tools/wide_row_time.py [d ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch

from jTransUP.hip import ops

dev = torch.device('cuda', 0)
NU, NI, NE, P = 6040, 3240, 14708, 20
for d in [int(x) for x in sys.argv[1:]] or [256, 320, 512, 1024]:
    g = torch.Generator().manual_seed(1)
    mk = lambda r: torch.nn.functional.normalize(torch.randn(r, d, generator=g), dim=1).to(dev).requires_grad_(True)
    U, I, E, A, C, R, Rn = mk(NU), mk(NI), mk(NE + 1), mk(P), mk(P), mk(P), mk(P)
    i2e = torch.randint(0, NE, (NI,), generator=g).to(dev, torch.int32)
    n = 1024
    u = torch.randint(0, NU, (n,), generator=g).to(dev); i = torch.randint(0, NI, (n,), generator=g).to(dev)
    gs = torch.randn(n, generator=g).to(dev)
    for t in (U, I, E, A, C, R, Rn):
        t.grad = torch.zeros_like(t)

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    out = {}
    for l1 in (False, True):
        for hard in (False, True):
            uni = torch.rand(n, P, generator=g).to(dev) if hard else None
            mode = ops.GUMBEL_INPUT if hard else ops.GUMBEL_OFF
            f = lambda: ops.score_ktup(U, I, E, A, C, R, Rn, i2e, u, i, l1, mode, uni, ent_pad=NE)
            with torch.no_grad():
                tf = timed(f)
            tb = timed(lambda: f().backward(gs)) - tf
            out[('l1' if l1 else 'l2') + ('_hard' if hard else '_soft')] = (round(tf, 4), round(tb, 4))
    uq = torch.arange(512, device=dev)
    with torch.no_grad():
        te = timed(lambda: ops.eval_ktup(U, I, E, A, C, R, Rn, i2e, uq, False), reps=5)
    print('WIDEROW d=%d  fwd/bwd ms of 1024 pairs %s  eval 512 x 3240 all-item scores %.3f ms' % (d, out, te), flush=True)
