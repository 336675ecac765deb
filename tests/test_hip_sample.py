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
    s.offsets['rec'] = s2.offsets['rec'] = 777
    a, b = s.sample_rec(u, pos, unique_in_batch=False), s2.sample_rec(u, pos, unique_in_batch=False)
    assert torch.equal(a, b)                                                # (seed, offset) reproduces the batch
    # ... and so does the batch-unique mode: collisions are resolved by deterministic rounds, not by an atomic race
    u = torch.from_numpy(rng.randint(0, nu, size=n)).to(DEV); pos = torch.from_numpy(rng.randint(0, ni, size=n)).to(DEV)
    ref = None
    for _ in range(5):
        s.offsets['rec'] = s2.offsets['rec'] = 4242
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


def test_feed_launches_equal_slice_plus_sampler():
    """ktup_feed_rec / ktup_feed_kg (batch slice + negatives + the steppers' [pos ; neg] layout from a device-side cursor and
    Philox counter) against the host-driven route: DeviceFeeder.next_cols + DeviceSampler.sample_* with the same seeds -- the
    same ids bit for bit, launch after launch, across an epoch wrap; cursor and counter end where the host mirrors say."""
    from jTransUP.hip import lib as L
    from jTransUP.hip.ops import _p, _stream
    from jTransUP.utils.device_sampler import DeviceSampler, TRIES
    from jTransUP.utils.fast_train import DeviceFeeder
    rng = np.random.RandomState(7)
    nu, ni, ne, nr, B = 60, 400, 300, 5, 32
    ratings = [(int(u), int(i)) for u, i in zip(rng.randint(0, nu, 150), rng.randint(0, ni, 150))]
    triples = [(int(h), int(t), int(r)) for h, t, r in zip(rng.randint(0, ne, 200), rng.randint(0, ne, 200), rng.randint(0, nr, 200))]
    rated = {}
    for u, i in ratings:
        rated.setdefault(u, set()).add(i)

    def make():
        s = DeviceSampler(DEV, seed=11)
        s.set_rating_dicts(nu, ni, [rated]); s.set_triples(ne, nr, [triples])
        return s, DeviceFeeder(ratings, B, DEV, seed=1), DeviceFeeder(triples, B, DEV, seed=2)
    (sa, ra, ka), (sb, rb, kb) = make(), make()
    i64 = dict(dtype=torch.int64, device=DEV)
    u2, i2, h2, t2, r2 = (torch.zeros(2 * B, **i64) for _ in range(5))
    st = _stream(DEV)
    for step in range(12):                                                  # 150 / 32: the rating columns wrap after 4 batches
        if step % 3 < 2:
            u, pi = ra.next_cols()
            ni_ = sa.sample_rec(u, pi)
            rb.fed(); sb.fed(B, 'rec')
            L.call('ktup_feed_rec', _p(rb.cols[0]), _p(rb.cols[1]), rb.n, B, _p(rb.cursor), _p(sb.offset_dev[0:]), ni, _p(sb.bitmap), sb.words,
                   sb.seed, 1, _p(u2), _p(i2), _p(sb.rec_workspace()), _p(sb.fail), st)
            assert torch.equal(u2, torch.cat([u, u])) and torch.equal(i2, torch.cat([pi, ni_])), step
            assert int(rb.cursor.item()) == rb.start == ra.start
        else:
            ph, pt, pr = ka.next_cols()
            nh, nt = sa.sample_kg(ph, pt, pr)
            kb.fed(); sb.fed(B, 'kg')
            L.call('ktup_feed_kg', _p(kb.cols[0]), _p(kb.cols[1]), _p(kb.cols[2]), kb.n, B, _p(kb.cursor), _p(sb.offset_dev[1:]), ne, nr,
                   _p(sb.keys), sb.keys.numel(), sb.seed, _p(h2), _p(t2), _p(r2), _p(sb.fail), st)
            assert torch.equal(h2, torch.cat([ph, nh])) and torch.equal(t2, torch.cat([pt, nt])) and torch.equal(r2, torch.cat([pr, pr])), step
    assert sb.offset_dev.tolist() == [sb.offsets['rec'], sb.offsets['kg']] == [sa.offsets['rec'], sa.offsets['kg']]
    assert sa.offset == 12 * B * TRIES
    assert bool((sb.rec_workspace() == -1).all())                          # the uniqueness scratch is left all-ones
    sa.check(); sb.check()
