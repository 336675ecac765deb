"""Shape sweep of the loss / regulariser kernels (K8-K10) and the baseline scorers (run by hand on a GPU box): value and gradients
against the oracle over batch sizes 0 ... 4099, widths 1 ... 400 and id lists with repeats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import numpy as np
import torch

from oracle import cpu_ref as O
from jTransUP.hip import ops

DEV = 'cuda'
bad, ran = [], [0]


def close(a, b, scale=1.0):
    a = a.detach().cpu().double().numpy(); b = b.detach().cpu().double().numpy()
    return a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=1e-6 * max(1.0, scale))


def run(tag, fn):
    ran[0] += 1
    try:
        msg = fn()
    except Exception as e:                                    # noqa: BLE001
        msg = '%s: %s' % (type(e).__name__, str(e)[:200])
    if msg:
        bad.append((tag, msg))


gen = torch.Generator().manual_seed(3)
for n in (0, 1, 2, 63, 64, 65, 511, 512, 513, 4099):
    for param in (1.0, -1.0, 0.3, 3.0):
        def pair():
            pos = torch.randn(n, generator=gen) * 3; neg = torch.randn(n, generator=gen) * 3
            for name, fo, fd in (('bpr', O.bpr_loss, ops.bpr_loss), ('margin', O.margin_loss, ops.margin_loss)):
                if n == 0 and name == 'bpr':
                    continue                                   # a mean over nothing is NaN in the reference too
                pc, nc = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
                pd, nd = pos.to(DEV).requires_grad_(True), neg.to(DEV).requires_grad_(True)
                lo, ld = fo(pc, nc, param), fd(pd, nd, param)
                if not close(ld, lo, float(lo.abs())):
                    return '%s value %.7g vs %.7g' % (name, float(ld), float(lo))
                lo.backward(); ld.backward()
                if n and (not close(pd.grad, pc.grad) or not close(nd.grad, nc.grad)):
                    return '%s gradient' % name
            return None
        run('pair losses n=%d param=%g' % (n, param), pair)
for d in (1, 3, 4, 7, 20, 64, 100, 128, 130, 256, 300, 400):
    for rows, nid in ((1, 0), (5, 1), (70, 64), (70, 513), (3000, 4099), (33, None)):
        def regs():
            T = O.make_table(rows, d, gen) * 1.5; N = O.make_table(rows, d, gen)
            ids = None if nid is None else torch.randint(0, rows, (nid,), generator=gen)
            Tc, Nc = T.clone().requires_grad_(True), N.clone().requires_grad_(True)
            Td, Nd = T.to(DEV).requires_grad_(True), N.to(DEV).requires_grad_(True)
            idd = None if ids is None else ids.to(DEV)
            sel = (lambda x: x) if ids is None else (lambda x: x[ids])
            lo = O.norm_loss(sel(Tc)) + 0.5 * O.orthogonal_loss(sel(Tc), sel(Nc)) if (ids is None or nid) else None
            if lo is None:
                ld = ops.norm_loss(Td, idd) + 0.5 * ops.orthogonal_loss(Td, Nd, idd)
                return None if float(ld) == 0.0 else 'empty id list gives %g' % float(ld)
            ld = ops.norm_loss(Td, idd) + 0.5 * ops.orthogonal_loss(Td, Nd, idd)
            if not close(ld, lo, float(lo.abs())):
                return 'value %.7g vs %.7g' % (float(ld), float(lo))
            lo.backward(); ld.backward()
            s = float(Tc.grad.abs().max())
            if not close(Td.grad, Tc.grad, s) or not close(Nd.grad, Nc.grad, s):
                return 'gradient (max %.3g off)' % float((Td.grad.cpu() - Tc.grad).abs().max())
            return None
        run('regularisers d=%d rows=%d ids=%s' % (d, rows, nid), regs)
    print('d=%d done: %d cases, %d problems' % (d, ran[0], len(bad)), flush=True)
for b in bad:
    print('PROBLEM %s: %s' % b)
sys.exit(1 if bad else 0)
