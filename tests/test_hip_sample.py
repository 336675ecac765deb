"""GPU checks of the on-device negative samplers (K19): the constraints of utils/data.py:12-85 hold for every draw,
draws are reproducible from (seed, offset), and the distributions are uniform / fair (no RNG parity with python's MT)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_rec_sampler_constraints_and_uniformity():
    from jTransUP.utils.device_sampler import DeviceSampler
    rng = np.random.RandomState(0)
    nu, ni, n = 200, 3240, 512
    train = {u: set(rng.randint(0, ni, size=165).tolist()) for u in range(nu)}
    valid = {u: set(rng.randint(0, ni, size=10).tolist()) for u in range(0, nu, 2)}
    s = DeviceSampler(DEV, seed=3)
    s.set_rating_dicts(nu, ni, [train, valid])
    counts = np.zeros(ni)
    for rep in range(40):
        u = torch.from_numpy(rng.randint(0, nu, size=n)).to(DEV)
        pos = torch.from_numpy(np.array([rng.choice(sorted(train[int(x)])) for x in u.cpu()])).to(DEV)
        neg = s.sample_rec(u, pos).cpu().numpy()
        assert (neg >= 0).all() and (neg < ni).all()
        assert len(set(neg.tolist())) == n                                  # unique inside the batch
        for uu, p, g in zip(u.cpu().tolist(), pos.cpu().tolist(), neg.tolist()):
            assert g != p and g not in train[uu] and g not in valid.get(uu, ())
        counts[neg] += 1
    assert counts.std() / counts.mean() < 0.6                               # ~Poisson(6.3): relative spread ~0.4
    s2 = DeviceSampler(DEV, seed=3)
    s2.set_rating_dicts(nu, ni, [train, valid])
    u = torch.arange(n, device=DEV) % nu; pos = torch.zeros(n, dtype=torch.long, device=DEV)
    s.offset = s2.offset = 777
    a, b = s.sample_rec(u, pos, unique_in_batch=False), s2.sample_rec(u, pos, unique_in_batch=False)
    assert torch.equal(a, b)                                                # (seed, offset) reproduces the batch
    # ... and so does the batch-unique mode: collisions are resolved by deterministic rounds, not by an atomic race
    u = torch.from_numpy(rng.randint(0, nu, size=n)).to(DEV); pos = torch.from_numpy(rng.randint(0, ni, size=n)).to(DEV)
    ref = None
    for _ in range(5):
        s.offset = s2.offset = 4242
        a, b = s.sample_rec(u, pos), s2.sample_rec(u, pos)
        assert torch.equal(a, b) and (ref is None or torch.equal(a, ref))
        ref = a
    s.check(); s2.check()                                                   # nothing failed so far
    # impossible request: more rows than admissible items -> still valid ids everywhere (never a -1 into the scoring kernels),
    # the admissible ones each used once, and check() raises
    from jTransUP.hip.lib import KtupError
    tiny = DeviceSampler(DEV, seed=1); tiny.set_rating_dicts(2, 40, [{0: set(range(30))}])
    out = tiny.sample_rec(torch.zeros(64, dtype=torch.long, device=DEV), torch.zeros(64, dtype=torch.long, device=DEV)).cpu()
    assert int(out.min()) >= 0 and int(out.max()) < 40
    ok = out[out >= 30]
    assert sorted(ok.tolist()) == list(range(30, 40))                       # the 10 admissible items, each exactly once
    with pytest.raises(KtupError):
        tiny.check()
    tiny.check()                                                            # the counter was reset
    # a user who rated everything but one item: the scan fallback finds it (no failure)
    full = DeviceSampler(DEV, seed=2); full.set_rating_dicts(1, 5000, [{0: set(range(5000)) - {4321}}])
    z = torch.zeros(3, dtype=torch.long, device=DEV)
    assert full.sample_rec(z, z, unique_in_batch=False).cpu().tolist() == [4321] * 3
    full.check()


def test_kg_sampler_constraints_and_fair_coin():
    from jTransUP.utils.device_sampler import DeviceSampler
    rng = np.random.RandomState(1)
    ne, nr, n = 500, 7, 4096
    train = [(int(rng.randint(ne)), int(rng.randint(ne)), int(rng.randint(nr))) for _ in range(20000)]
    test = [(int(rng.randint(ne)), int(rng.randint(ne)), int(rng.randint(nr))) for _ in range(2000)]
    known = set(train) | set(test)
    s = DeviceSampler(DEV, seed=9)
    s.set_triples(ne, nr, [train, test])
    batch = [train[i] for i in rng.randint(0, len(train), size=n)]
    h = torch.tensor([x[0] for x in batch], device=DEV); t = torch.tensor([x[1] for x in batch], device=DEV)
    r = torch.tensor([x[2] for x in batch], device=DEV)
    nh, nt = s.sample_kg(h, t, r)
    heads = 0
    for (a, b, c), x, y in zip(batch, nh.cpu().tolist(), nt.cpu().tolist()):
        assert (x != a) != (y != b)                                         # exactly one side corrupted
        assert (x, y, c) not in known                                        # never a known-true triple
        heads += x != a
    assert 0.45 < heads / float(n) < 0.55                                    # fair coin (data.py:13-14)
    nh2, nt2 = s.sample_kg(h, t, r)
    assert not (torch.equal(nh, nh2) and torch.equal(nt, nt2))              # the offset advanced
