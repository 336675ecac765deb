"""CPU checks of two numerical claims the HIP kernels rest on (the second at the end of the file).  The first: the error window of
the exact-order link-prediction pass (csrc/ktup_eval_kg_fused.hip: kg_list_scores_kernel writes
T = 2 kappa0 (1 + |w|^2)^2 (|c|^2 + max|e|^2), kappa0 = 2 (d + 8) 2^-24; the sweep only counts a comparison outside [gold - T, gold + T]
and hands the rest to the fp64 resolve) rests on one claim: the fp32 score of the expansion |c|^2 - 2 c.e + |e|^2 (TransH: + w.e (2 c.w +
w.e (|w|^2 - 2))), summed in ANY order, is within T / 2 of the exact score of the same fp32 vectors.  Checked here on the CPU in numpy for
several summation orders (sequential, reversed, pairwise, 4-wide blocks like the matrix cores' k steps) and for inputs chosen to stress
it (large norms, nearly equal vectors, cancelling coordinates).  No GPU, no library call: the formula is restated from the kernel source,
and the test fails if those source lines change."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'joint-kg-recommender_amd', 'csrc', 'ktup_eval_kg_fused.hip')


def test_the_window_formula_is_the_one_in_the_kernel_source():
    src = open(SRC).read()
    assert re.search(r'a\.kappa0\s*=\s*2\.f\s*\*\s*\(float\)\(d \+ 8\)\s*\*\s*5\.9604645e-8f', src)
    assert 'a.ktol[u0 + tid] = 2.f * a.kappa0 * (1.f + ww) * (1.f + ww) * (qs[tid * 4 + 0] + __uint_as_float(*a.enmax));' in src


def _sum32(x, order):
    """fp32 sum of the fp32 products x in one of the orders a kernel might use."""
    x = x.astype(np.float32)
    if order == 'seq':
        acc = np.float32(0)
        for v in x:
            acc = np.float32(acc + v)
        return acc
    if order == 'rev':
        return _sum32(x[::-1], 'seq')
    if order == 'pair':
        while x.size > 1:
            if x.size % 2:
                x = np.concatenate([x, np.zeros(1, np.float32)])
            x = (x[0::2] + x[1::2]).astype(np.float32)
        return x[0]
    if order == 'block4':                               # four partial sums advanced together, combined at the end
        pad = (-x.size) % 4
        x = np.concatenate([x, np.zeros(pad, np.float32)]).reshape(-1, 4)
        acc = np.zeros(4, np.float32)
        for row in x:
            acc = (acc + row).astype(np.float32)
        return np.float32(np.float32(acc[0] + acc[1]) + np.float32(acc[2] + acc[3]))
    raise ValueError(order)


def _dot32(a, b, order):
    return _sum32((a.astype(np.float32) * b.astype(np.float32)).astype(np.float32), order)


def _device_score(c, e, w, order):
    cc, en, ce = _dot32(c, c, order), _dot32(e, e, order), _dot32(c, e, order)
    s = np.float32(np.float32(cc + en) + np.float32(-2) * ce)
    if w is not None:
        we, cw, ww = _dot32(w, e, order), _dot32(c, w, order), _dot32(w, w, order)
        s = np.float32(s + we * np.float32(we * np.float32(ww - np.float32(2)) + np.float32(2) * cw))
    return s


def _exact_score(c, e, w):
    c, e = c.astype(np.float64), e.astype(np.float64)
    if w is None:
        return float(((c - e) ** 2).sum())
    w = w.astype(np.float64)
    # the expansion's own meaning: |c - (e - (w.e) w)|^2 with c.w, |w|^2 as they are (transH.py:58-71 projects e onto w's hyperplane)
    we, cw, ww = float(w @ e), float(c @ w), float(w @ w)
    return float(c @ c + e @ e - 2 * (c @ e) + we * (we * (ww - 2) + 2 * cw))


def _cases(rng, d):
    unit = lambda v: v / np.linalg.norm(v)
    for scale in (1.0, 0.05, 30.0):
        c = (rng.standard_normal(d) * scale).astype(np.float32)
        yield c, (rng.standard_normal(d) * scale).astype(np.float32)
        yield c, (c * np.float32(1 + 1e-4) + rng.standard_normal(d).astype(np.float32) * np.float32(1e-5 * scale))   # near ties
        yield c, -c                                                                                                      # largest score
    alt = (np.arange(d) % 2 * 2 - 1).astype(np.float32)
    yield alt * np.float32(3.0), alt * np.float32(-2.9999)                    # every product has the same sign and size
    yield unit(rng.standard_normal(d)).astype(np.float32), unit(rng.standard_normal(d)).astype(np.float32)   # TransE's unit rows


@pytest.mark.parametrize('d', [20, 36, 64, 100, 128])
@pytest.mark.parametrize('transh', [False, True])
def test_fp32_expansion_stays_inside_half_the_window(d, transh):
    rng = np.random.RandomState(1000 * d + transh)
    kappa0 = 2.0 * (d + 8) * 2.0 ** -24
    worst = 0.0
    for c, e in _cases(rng, d):
        w = None
        if transh:
            w = rng.standard_normal(d).astype(np.float32)
            w = (w / np.float32(np.linalg.norm(w) * rng.choice([1.0, 0.7, 1.3]))).astype(np.float32)    # norm rows are near unit, not exactly
        cc = float(c.astype(np.float64) @ c.astype(np.float64)); en = float(e.astype(np.float64) @ e.astype(np.float64))
        ww = 0.0 if w is None else float(w.astype(np.float64) @ w.astype(np.float64))
        T = 2.0 * kappa0 * (1.0 + ww) ** 2 * (cc + en)              # (the kernel uses max |e|^2 >= this candidate's: a wider window)
        exact = _exact_score(c, e, w)
        for order in ('seq', 'rev', 'pair', 'block4'):
            err = abs(float(_device_score(c, e, w, order)) - exact)
            assert err <= 0.5 * T, (d, transh, order, err, T)
            worst = max(worst, err / T)
    assert worst < 0.25                                              # the bound is a worst case: observed errors sit far inside it


def test_tracked_norm_identity_and_its_fp32_error():
    """The B = 512 step's gradient norm (include/ktup_hip.h `gnorm`; csrc/ktup_common.h sq_gain): adding v onto a cell that held `old`
    raises the buffer's squared norm by (2 old + v) v, so the sum of those terms over every add into zero-filled buffers IS the squared
    norm of what was built -- whatever the order the atomics land in.  Restated in numpy with fp32 cells and fp32 terms (per-lane
    partial sums in fp32, totals in fp64 as the kernels keep them): a BPR-like step where rows are shared by many examples and
    contributions partly cancel stays within 2e-6 of the norm of the final buffer, ten times inside the GPU test's 2e-5."""
    rng = np.random.RandomState(5)
    rows, d, adds = 50, 100, 640                                     # 640 row contributions into 50 rows: every row shared ~13 times
    for trial in range(3):
        buf = np.zeros((rows, d), np.float32)
        ids = rng.randint(0, rows, adds)
        vals = (rng.standard_normal((adds, d)) * rng.choice([1.0, -1.0], (adds, 1)) * 0.05).astype(np.float32)
        total = 0.0
        for k in rng.permutation(adds):                              # any landing order
            old = buf[ids[k]].copy()
            buf[ids[k]] = (old + vals[k]).astype(np.float32)
            term = ((np.float32(2) * old + vals[k]).astype(np.float32) * vals[k]).astype(np.float32)
            lane = np.float32(0)
            for t in term.reshape(-1, 4).sum(1, dtype=np.float32):  # a lane's four coordinates, then its running fp32 sum
                lane = np.float32(lane + t)
            total += float(lane)                                     # (workgroup totals are fp64)
        exact = float((buf.astype(np.float64) ** 2).sum())
        assert abs(total - exact) <= 2e-6 * exact, (trial, total, exact)


def test_sparse_adam_with_catch_up_is_the_dense_adam():
    """The third claim (round 5; csrc/ktup_shard_step.hip adam_row, jTransUP/sharded_ktup.py): a row-sparse Adam that, BEFORE a row is read,
    replays the zero-gradient steps the row has missed -- m <- beta1 m, sqrt(v) <- sqrt(beta2) sqrt(v), p <- p - lr / (1 - beta1^s) m /
    (sqrt(v) / sqrt(1 - beta2^s) + eps), at most `replay` of them one by one, then m and v in closed form -- and only then applies the
    step, holds the same tables as torch.optim.Adam stepping EVERY row with zero-filled gradients (what utils/trainer.py:63-66 builds).
    Restated in numpy (fp32 state, the kernel's order of operations), on a loss whose gradient depends on the weights (so a row read
    stale would show), rows touched at random with gaps from 1 to beyond the replay cap.  Also pins the cap the host computes."""
    import math

    import torch

    import sys
    sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
    from jTransUP.sharded_ktup import adam_replay, adam_state_pitch
    b1, b2, lr, eps = 0.9, 0.999, 0.001, 1e-8
    K = adam_replay((b1, b2))
    r = b1 / math.sqrt(b2)
    assert K == 110 and r ** K / (1 - r) < 1e-4 and adam_state_pitch(100) == 204        # the dropped tail < 1e-4 of the first increment
    rng = np.random.RandomState(7)
    rows, d, T = 40, 12, 400
    w0 = rng.standard_normal((rows, d)).astype(np.float32)
    c = (0.5 + rng.rand(rows, d)).astype(np.float32)                   # loss = sum over touched rows of 0.5 c w^2 + b w  ->  g = c w + b
    b = (rng.choice([-1.0, 1.0], (rows, d)) * (2.5 + rng.rand(rows, d))).astype(np.float32)   # |b| > |c w| over the run: no gradient near 0
    touch_p = np.concatenate([np.full(10, 0.9), np.full(10, 0.2), np.full(10, 0.02), np.full(10, 0.004)])   # gaps of ~1, ~5, ~50, ~250 steps
    touched = rng.rand(T, rows) < touch_p[None, :]
    touched[0] = True                                                  # (every table gets a gradient at step 1, like the rec step)
    # ---- dense reference
    W = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    opt = torch.optim.Adam([W], lr=lr, betas=(b1, b2), eps=eps)
    C, Bt = torch.from_numpy(c), torch.from_numpy(b)
    for t in range(T):
        opt.zero_grad(set_to_none=False)
        mask = torch.from_numpy(touched[t].astype(np.float32))[:, None]
        (mask * (0.5 * C * W * W + Bt * W)).sum().backward()
        opt.step()
    dense = W.detach().numpy()
    # ---- the lazy form
    f = np.float32
    p, m, v = w0.copy(), np.zeros_like(w0), np.zeros_like(w0)
    last = np.zeros(rows, np.int64)

    def catch_up(i, upto):
        miss = upto - last[i]
        if last[i] > 0 and miss > 0:
            k_run = min(miss, K)
            b1p, b2p = b1 ** int(last[i]), b2 ** int(last[i])
            sv = np.sqrt(v[i]).astype(f)
            sb2 = f(math.sqrt(b2))
            for _ in range(k_run):
                b1p *= b1; b2p *= b2
                c1 = f(lr / (1.0 - b1p)); ib = f(1.0 / math.sqrt(1.0 - b2p))
                m[i] = (m[i] - m[i] * f(1 - b1)).astype(f)
                sv = (sb2 * sv).astype(f)
                p[i] = (p[i] - (c1 * m[i]) / (sv * ib + f(eps))).astype(f)
            v[i] = (f(b2 ** int(miss)) * v[i]).astype(f)
            if miss > k_run:
                m[i] = (f(b1 ** int(miss - k_run)) * m[i]).astype(f)
        if upto > last[i] and last[i] > 0:
            last[i] = upto
    for t in range(1, T + 1):
        for i in np.nonzero(touched[t - 1])[0]:
            catch_up(i, t - 1)                                         # BEFORE the row is read
            g = (c[i] * p[i] + b[i]).astype(f)
            m[i] = (m[i] + (g - m[i]) * f(1 - b1)).astype(f)
            v[i] = (f(1 - b2) * g * g + f(b2) * v[i]).astype(f)
            p[i] = (p[i] - f(lr / (1 - b1 ** t)) * (m[i] / (np.sqrt(v[i]) / f(math.sqrt(1 - b2 ** t)) + f(eps)))).astype(f)
            last[i] = t
    for i in range(rows):
        catch_up(i, T)                                                 # the flush before an evaluation
    np.testing.assert_allclose(p, dense, rtol=2e-5, atol=5e-6)
    # and what a stale read costs (no catch-up before the gradient): far outside that band -- the test would see it
    assert r ** 5 > 0.5                                                # (a row resting five steps still owes more than half its first move)
