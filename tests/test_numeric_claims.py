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
