import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/joint-kg-recommender_amd')
from jTransUP.hip import ops
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
d, P, n = 256, 20, 262144
mk = lambda r: torch.nn.functional.normalize(torch.randn(r, d, generator=g, device=dev), dim=1)
U, I, E = mk(200000), mk(100000), mk(300001)
Pm, Pn, R, Rn = mk(P), mk(P), mk(P), mk(P)
i2e = torch.randint(0, 300000, (100000,), generator=g, device=dev).to(torch.int32)
u = torch.randint(0, 200000, (n,), generator=g, device=dev); i = torch.randint(0, 100000, (n,), generator=g, device=dev)
def run(f, reps=20):
    with torch.no_grad():
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ws = ops.pref_workspace(Pm, Pn, R, Rn)
outs = {}
for var in ('7', '2'):
    os.environ['KTUP_PREF_FWD'] = var
    for l1 in (False, True):
        f = lambda: ops.score_ktup(U, I, E, Pm, Pn, R, Rn, i2e, u, i, l1, ws=ws)
        ms = run(f)
        outs[(var, l1)] = f()
        print('d=256 P=20 n=%d KTUP_PREF_FWD=%s l1=%d: %.3f ms  (%.2f G rows/s, %.2f TB/s of row bytes)' % (n, var, l1, ms, n / ms / 1e6, n * 3096 / ms / 1e9))
    hard = lambda: ops.score_ktup(U, I, E, Pm, Pn, R, Rn, i2e, u, i, False, ops.GUMBEL_PHILOX, None, 5, 0, ws=ws)
    print('   hard gate: %.3f ms' % run(hard))
for l1 in (False, True):
    a, b = outs[('7', l1)], outs[('2', l1)]
    print('max rel diff mc vs generic l1=%d: %.3g' % (l1, float(((a - b).abs() / b.abs().clamp_min(1e-6)).max())))
