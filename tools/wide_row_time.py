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
    # a whole ml1m-size pass (6,040 users, filtered top-10): one sweep (d <= 256) against per-batch scores + topk_filtered
    allu = torch.arange(NU, device=dev)
    f_off = (torch.arange(NU + 1, device=dev) * 20)
    f_ids = torch.randint(0, NI, (NU * 20,), generator=g).to(dev, torch.int32).view(NU, 20).sort(1).values.reshape(-1).contiguous()
    ps = {}
    with torch.no_grad():
        for l1 in (True, False):
            items = ops.eval_pref_items(I, E, A, C, R, Rn, i2e)
            if ops.eval_pref_topk(U, allu[:64], items, l1, 10) is not None:
                ps[('l1' if l1 else 'l2') + '_sweep'] = round(timed(lambda: ops.eval_pref_topk(U, allu, items, l1, 10, f_off, f_ids), reps=5), 3)

            def batched():
                for b0 in range(0, NU, 512):
                    ub = allu[b0:b0 + 512]
                    m = ops.eval_ktup(U, I, E, A, C, R, Rn, i2e, ub, l1, items=items)
                    ops.topk_filtered(m, False, 10, f_off[b0:b0 + ub.numel() + 1] - f_off[b0], f_ids[int(f_off[b0]):int(f_off[min(b0 + 512, NU)])])
            ps[('l1' if l1 else 'l2') + '_batched'] = round(timed(batched, reps=3), 3)
    print('WIDEPASS d=%d  6040 x 3240 filtered top-10 pass ms %s' % (d, ps), flush=True)
    print('WIDEROW d=%d  fwd/bwd ms of 1024 pairs %s  eval 512 x 3240 all-item scores %.3f ms' % (d, out, te), flush=True)
