"""The series form of the row-sparse Adam's replay (csrc/ktup_shard_step.hip adam_zero_series), restated in numpy float32 and held against
the zero-gradient recurrence of torch.optim.Adam (adam.py _single_tensor_adam; utils/trainer.py:63-66 builds it) summed in float64.

K zero-gradient steps after step `last` move an element with moments (m0, v0) by  lr * m0 * sum_i c_i / (s + e_i),  s = sqrt(v0),
c_i = beta1^i sqrt(bc2_i) / (bc1_i beta2^(i/2)),  e_i = eps sqrt(bc2_i) / beta2^(i/2).  Around E = e_(i*) the sum is
1 / (s + E) * sum_n (-U)^n R_n with U = E / (s + E) and R_n = sum_i c_i (e_i / E - 1)^n; the kernel takes it (degree 4) only where
R_5 = sum_i c_i |e_i / E - 1|^5 <= 1e-6 R_0 and replays step by step otherwise.  No GPU needed: this is the arithmetic, not the kernel."""
import numpy as np
import pytest

F = np.float32
# (the betas as the C ABI carries them -- ktup_adam_t holds floats; 0.999f is 0.99900001287: over t steps that is a relative 1.3e-8 t on
#  beta2^t, the same in the kernel's step-by-step replay and in its series, and not what this file is about)
B1, B2, EPS = float(F(0.9)), float(F(0.999)), 1e-8


def exact_sum(last, K, s):
    i = np.arange(1, K + 1, dtype=np.float64)
    bc1, bc2 = 1 - B1 ** (last + i), 1 - B2 ** (last + i)
    return np.sum((B1 ** i / bc1)[None, :] / ((B2 ** (i / 2) / np.sqrt(bc2))[None, :] * s[:, None] + EPS), axis=1)


def one_minus_exp(x):
    """1 - e^x, x <= 0, float32: the exponential below -1/4, the degree-7 Taylor polynomial above (no cancellation)."""
    x = x.astype(F)
    t = F(1 / 5040)
    for c in (1 / 720, 1 / 120, 1 / 24, 1 / 6, 0.5, 1.0):
        t = x * t + F(c)
    return np.where(x > F(-0.25), -x * t, F(1) - np.exp(x, dtype=F)).astype(F)


def series_sum(last, K, s):
    ln1, ln2 = F(np.log(F(B1))), F(np.log(F(B2)))
    rho = float(B1) / np.sqrt(float(B2))
    istar = int(np.floor(1.0 / (1.0 - rho) + 0.5))
    tl, fs = F(last), F(min(K, istar))
    E = F(EPS) * np.sqrt(one_minus_exp(np.array((tl + fs) * ln2))) * np.exp(F(-0.5) * fs * ln2, dtype=F)
    i = np.arange(1, K + 1).astype(F)
    bc2s = np.sqrt(one_minus_exp((tl + i) * ln2))
    c = np.exp(i * (ln1 - F(0.5) * ln2), dtype=F) * bc2s / one_minus_exp((tl + i) * ln1)
    q = F(EPS) * bc2s * np.exp(F(-0.5) * i * ln2, dtype=F) / E - F(1)
    R = [F(np.sum(c * q ** n, dtype=F)) for n in range(5)]
    R5 = F(np.sum(c * np.abs(q) ** 5, dtype=F))
    s = s.astype(F)
    u = F(1) / (np.sqrt(s * s, dtype=F) + E)
    U = E * u
    h = R[4]
    for n in (3, 2, 1, 0):
        h = R[n] - U * h
    return (u * h).astype(np.float64), bool(R5 <= F(1e-6) * R[0])


@pytest.mark.parametrize('last', [1, 10, 64, 200, 400, 1000, 20000, 1000000])
@pytest.mark.parametrize('K', [8, 20, 60, 110])
def test_series_equals_the_recurrence_where_the_kernel_takes_it(last, K):
    s = np.concatenate([[0.0], np.logspace(-14, 0, 120)])              # from "eps carries the denominator" to sqrt(v) = 1
    got, taken = series_sum(last, K, s)
    want = exact_sum(last, K, s)
    if taken:
        assert np.max(np.abs(got - want) / want) < 3e-6                # the bound (1e-6 of the displacement) + float32 sums of <= 110 terms
    else:
        assert last <= 200                                             # only young states are left to the step-by-step replay


def test_settled_states_always_take_the_series():
    s = np.logspace(-12, 0, 7)
    for last in (250, 500, 5000, 10 ** 5, 10 ** 6):
        for K in (8, 33, 64, 65, 110):
            assert series_sum(last, K, s)[1]
