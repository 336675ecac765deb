import sys, os, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/joint-kg-recommender_amd')
import bench
from jTransUP.hip import ops
dev = torch.device('cuda', 0)
W, i2e, idx = bench.build_world(3, dev)
D_ = {k: v.to(dev) for k, v in W.items()}
i2e_d = i2e.to(dev, torch.int32)
X = {k: v.to(dev) for k, v in idx.items()}
def run(f, n=30):
    with torch.no_grad():
        for _ in range(5): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for var in ('2', '7'):
    os.environ['KTUP_PREF_FWD'] = var
    t_soft = run(lambda: ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e_d, X['u'], X['i'], False))
    t_hard = run(lambda: ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e_d, X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 7, 0))
    t_tup = run(lambda: ops.score_tup(D_['U'], D_['I'], D_['P'], D_['Pn'], X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 7, 0))
    print('variant', var, 'ktup soft %.3f ms  ktup hard(philox) %.3f ms  tup hard(philox) %.3f ms (incl. prepare + python)' % (t_soft, t_hard, t_tup))
os.environ['KTUP_PREF_FWD'] = '2'
a = ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e_d, X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 7, 123)
os.environ['KTUP_PREF_FWD'] = '7'
b = ops.score_ktup(D_['U'], D_['I'], D_['E'], D_['P'], D_['Pn'], D_['R'], D_['Rn'], i2e_d, X['u'], X['i'], False, ops.GUMBEL_PHILOX, None, 7, 123)
print('philox hard: max abs diff vs pref_fwd2', float((a - b).abs().max()), 'rel', float(((a - b).abs() / a.abs().clamp_min(1e-6)).max()), 'mismatch>1e-3:', int(((a-b).abs() > 1e-3 * a.abs()).sum()))
