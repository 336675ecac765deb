"""The oracle against the reference ITSELF on random worlds (tests/golden/live_check.py): runs only where /root/reference exists
-- the build container -- in a subprocess of its own, because the reference's package and this build's mirror of its interface
are both called jTransUP.  Skipped on the GPU box, where the reference cannot be (the committed goldens are what travels)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'jTransUP')), reason='the reference is present in the build container only')
@pytest.mark.parametrize('seed', [0, 1])
def test_oracle_matches_the_live_reference_on_random_worlds(seed):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'live_check.py'), '--worlds', '5', '--seed', str(seed), '--ref', REF],
                         capture_output=True, text=True, timeout=600, env=env, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    worst = rep['worst_excess_over_bar']
    assert worst.pop('ranking.mismatches') == 0
    assert len(worst) == 24                                      # every family ran: scores / evaluation matrices of eight model kinds, step loss + gradients of four
    # 1.0 = the goldens' bars (|diff| <= 1e-6 + 1e-5 |reference|; TransR 1e-5 + 1e-4; gradients 1e-6 + 1e-4)
    assert all(v <= 1.0 for v in worst.values()), worst
