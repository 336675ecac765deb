"""Per-kernel timings at ml1m shape (d=100) through the ops wrappers, large batches so that launch overhead vanishes:
    python tools/kernel_times.py            (on an MI355X)
Prints microseconds per call (HIP events, 20 calls) and algorithmic TB/s where SURVEY.md 8(d) defines bytes per row."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import bench                                   # noqa: E402
from jTransUP.hip import ops                   # noqa: E402

dev = torch.device('cuda', 0)
W, i2e, idx = bench.build_world(3, dev)
D_ = {k: v.to(dev) for k, v in W.items()}
i2e = i2e.to(dev, torch.int32)
X = {k: v.to(dev) for k, v in idx.items()}
d = D_['U'].shape[1]
gen = torch.Generator(device=dev); gen.manual_seed(0)
M = torch.nn.functional.normalize(torch.randn(D_['R'].shape[0], d * d, generator=gen, device=dev), dim=1)   # TransR projections
NT = X['h'].numel()
NP = X['u'].numel()


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def leaf(t):
    return t.detach().clone().requires_grad_(True)


rows = []
with torch.no_grad():
    rows.append(('K1 BPRMF fwd', NP, 8 * d + 20, timed(lambda: ops.score_bprmf(D_['U'], D_['I'], X['u'], X['i']))))
    rows.append(('K2 TransE fwd L2', NT, 8 * d + 28, timed(lambda: ops.score_transe(D_['E'], D_['R'], X['h'], X['t'], X['r'], False))))
    rows.append(('K2 TransE fwd L1', NT, 8 * d + 28, timed(lambda: ops.score_transe(D_['E'], D_['R'], X['h'], X['t'], X['r'], True))))
    rows.append(('K3 TransH fwd', NT, 8 * d + 28, timed(lambda: ops.score_transh(D_['E'], D_['R'], D_['Rn'], X['h'], X['t'], X['r'], False))))
    rows.append(('K4 TransR fwd', NT, 8 * d + 28, timed(lambda: ops.score_transr(D_['E'], D_['R'], M, X['h'], X['t'], X['r'], False))))
    rows.append(('K5 TUP fwd soft', NP, 8 * d + 20, timed(lambda: ops.score_tup(D_['U'], D_['I'], D_['P'], D_['Pn'], X['u'], X['i'], False))))
    rows.append(('K7 TUP fwd hard (Philox)', NP, 8 * d + 20, timed(lambda: ops.score_tup(D_['U'], D_['I'], D_['P'], D_['Pn'], X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 1, 0))))
    rows.append(('K6 KTUP fwd soft', NP, 12 * d + 24, timed(lambda: ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, X['u'], X['i'], False))))
    rows.append(('K6 KTUP fwd hard (Philox)', NP, 12 * d + 24, timed(lambda: ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e, X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 1, 0))))


def fwd_bwd(make):
    def run():
        tabs, f = make()
        f(*tabs).sum().backward()
    return run


E, R, Rn, U, I, P, Pn = (D_[k] for k in ('E', 'R', 'Rn', 'U', 'I', 'P', 'Pn'))
rows.append(('K2 TransE fwd+bwd', NT, None, timed(fwd_bwd(lambda: ((leaf(E), leaf(R)), lambda e, r: ops.score_transe(e, r, X['h'], X['t'], X['r'], False))), 10)))
rows.append(('K3 TransH fwd+bwd', NT, None, timed(fwd_bwd(lambda: ((leaf(E), leaf(R), leaf(Rn)), lambda e, r, n: ops.score_transh(e, r, n, X['h'], X['t'], X['r'], False))), 10)))
rows.append(('K4 TransR fwd+bwd', NT, None, timed(fwd_bwd(lambda: ((leaf(E), leaf(R), leaf(M)), lambda e, r, m: ops.score_transr(e, r, m, X['h'], X['t'], X['r'], False))), 10)))
rows.append(('K6 KTUP fwd+bwd soft', NP, None, timed(fwd_bwd(lambda: ((leaf(U), leaf(I), leaf(E), leaf(P), leaf(Pn), leaf(R), leaf(Rn)),
                                                                   lambda u, i, e, p, pn, r, rn: ops.score_ktup(u, i, e, p, pn, r, rn, i2e, X['u'], X['i'], False, ent_pad=E.shape[0] - 1))), 10)))
nq = 512
q, rq = X['h'][:nq], X['r'][:nq]
with torch.no_grad():
    rows.append(('K12 TransE eval 512 x E L2', nq, None, timed(lambda: ops.eval_transe(E, R, q, rq, False, False))))
    rows.append(('K12 TransE eval 512 x E L1', nq, None, timed(lambda: ops.eval_transe(E, R, q, rq, True, False))))
    rows.append(('K13 TransH eval 512 x E', nq, None, timed(lambda: ops.eval_transh(E, R, Rn, q, rq, False, False))))
    rows.append(('K14 TransR eval 512 x E L2', nq, None, timed(lambda: ops.eval_transr(E, R, M, q, rq, False, False), 5)))
    rows.append(('K14 TransR eval 512 x E L1', nq, None, timed(lambda: ops.eval_transr(E, R, M, q, rq, True, False), 5)))
    for l1 in (False, True):        # entity side prepared once per pass (ktup_eval_transr_prepare)
        ents = ops.eval_transr_entities(E, M, R.shape[0], l1)
        rows.append(('K14 TransR eval %s, prepared' % ('L1' if l1 else 'L2'), nq, None,
                     timed(lambda: ops.eval_transr(E, R, M, q, rq, l1, False, ents=ents), 5)))
        rows.append(('K14 TransR entity side %s' % ('L1' if l1 else 'L2'), E.shape[0] * R.shape[0], None,
                     timed(lambda: ops.eval_transr_entities(E, M, R.shape[0], l1), 5)))
    rows.append(('K15 TUP eval 512 x I', nq, None, timed(lambda: ops.eval_tup(U, I, P, Pn, X['u'][:nq], False))))
    rows.append(('K16 KTUP eval 512 x I', nq, None, timed(lambda: ops.eval_ktup(U, I, E, P, Pn, R, Rn, i2e, X['u'][:nq], False))))
    rows.append(('K11 BPRMF eval 512 x I', nq, None, timed(lambda: ops.eval_bprmf(U, I, X['u'][:nq]))))
print('%-30s %10s %10s %10s' % ('kernel (incl. wrapper launches)', 'rows', 'us/call', 'TB/s'))
for name, n, bpr, us in rows:
    print('%-30s %10d %10.1f %10s' % (name, n, us, '' if bpr is None else '%.2f' % (n * bpr / us / 1e6)))
