"""K6 forward at config 5's width on HBM-resident tables (bench.py gather_stress_bench's d = 256 leg on its own), for A/B runs:

    KTUP_FWD_WIDE=0 python tools/fwd_d256_time.py      # one wave per 16-pair tile (rounds 1-4)
    python tools/fwd_d256_time.py

Prints ms per 716,800 pairs, algorithmic TB/s and the fraction of the 8 TB/s HBM peak.  Needs a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch

import bench as B
from jTransUP.hip import lib as L
from jTransUP.hip import ops

dev = torch.device('cuda')
gen = torch.Generator(device=dev); gen.manual_seed(3)
d2, sc2, reps = 256, 400, 40
nu, ni, ne = B.NU * sc2, B.NI * sc2, B.NE * sc2
mk2 = lambda rows: torch.randn(rows, d2, generator=gen, device=dev).mul_(1.0 / 16.0)
U, I, E = mk2(nu), mk2(ni), mk2(ne + 1)
P2 = [torch.nn.functional.normalize(torch.randn(B.NR, d2, generator=gen, device=dev), dim=1) for _ in range(4)]
i2e = torch.randint(0, ne, (ni,), generator=gen, device=dev).to(torch.int32)
u = torch.randint(0, nu, (B.REC_ROWS,), generator=gen, device=dev)
i = torch.randint(0, ni, (B.REC_ROWS,), generator=gen, device=dev)
ws2 = ops.pref_workspace(*P2)
s_rec = torch.empty(B.REC_ROWS, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for l1 in (0, 1):
    for nt in (0, 1):
        L.set_option('nt_gather', nt)
        rec2 = L.bind('ktup_score_ktup_fwd', U.data_ptr(), U.stride(0), I.data_ptr(), I.stride(0), E.data_ptr(), E.stride(0), i2e.data_ptr(),
                      ws2.data_ptr(), B.NR, d2, u.data_ptr(), i.data_ptr(), B.REC_ROWS, l1, ops.GUMBEL_OFF, None, 0, 0, s_rec.data_ptr(), st)
        for _ in range(25):
            rec2()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize(dev)
        for a, b in ev:
            a.record(); rec2(); b.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
        bpr2 = 12 * d2 + 24
        print('FWD256', {'fwd_wide': L.get_option('fwd_wide'), 'l1': l1, 'nt_gather': nt, 'ms': round(ms, 4),
                         'TBs': round(B.REC_ROWS * bpr2 / (ms * 1e-3) / 1e12, 3), 'frac_hbm': round(B.REC_ROWS * bpr2 / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBS, 3)})
L.set_option('nt_gather', 0)
