"""bench.py's output contract, single process and as the driver launches it for N > 1 (torch.distributed.run, one rank per
GPU).  On a 1-GPU box the two ranks share the device through the KTUP_BENCH_BACKEND=gloo hook (RCCL refuses that)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
        'data', 'config', 'roofline', 'cpu_baseline'}


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, stdout[-2000:]            # ONE JSON line, rank 0 only
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--no-extras'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _line(r.stdout)
    assert KEYS <= set(out)
    assert out['n_gpus'] == 1 and out['steps'] == 4 and out['warmup'] == 2 and out['higher_is_better'] is True
    assert out['scaling'] == 'weak' and out['vs_baseline'] is None and out['dtype'] == 'f32' and out['data'] == 'synthetic'
    assert 'workload' in out['config'] and 'model' not in out['config']
    rows = out['config']['rows_per_step_per_gpu']
    assert abs(out['value'] - rows * 1e3 / out['ms_per_step']) <= 1e-6 * out['value']
    rf = out['roofline']
    # the line names the ceiling that binds the kernel: HBM (GB/s) only if the counters say so, else the shared fp32 pipe ("mfma")
    assert (rf['bound'], rf['unit'], rf['peak']) in (('hbm', 'GB/s', 8000.0), ('mfma', 'TFLOP/s', 157.3))
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0.05 < rf['frac'] < 1.5
    alg = rf['hbm_algorithmic']                                                     # SURVEY 8(d)'s figure stays in the line
    assert abs(alg['achieved'] - rf['rows_per_launch'] * rf['bytes_per_row'] / (rf['ms_per_launch'] * 1e-3) / 1e9) < 1e-6 * alg['achieved']
    assert abs(alg['frac_algorithmic'] - alg['achieved'] / 8000.0) < 1e-9 and rf['step_algorithmic_over_peak'] > 0


def test_two_ranks_as_the_driver_launches_them():
    env = dict(os.environ, KTUP_BENCH_BACKEND='gloo', KTUP_BENCH_LEG_STEPS='12')       # (short N-GPU legs: two ranks on one GPU stage every exchange through the host)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29571', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = _line(r.stdout)
    assert KEYS <= set(out) and out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['cpu_baseline'] is None
    rows = out['config']['rows_per_step_per_gpu']
    assert abs(out['value'] - 2 * rows * 1e3 / out['ms_per_step']) <= 1e-6 * out['value']     # whole-job rows / max-over-ranks time
    # the N-GPU legs: config 4's data-parallel training step (weak + strong, with its comm / compute split) and config 5's
    # row-sharded step
    dp = out['dp_train_step']
    assert 'error' not in dp and dp['world'] == 2 and dp['backend'] == 'gloo'
    for leg, gb in (('weak', 1024), ('strong', 512)):
        assert dp[leg]['global_batch'] == gb and dp[leg]['ms_per_step'] > 0 and dp[leg]['ms_allreduce_only'] > 0
        assert 0 < dp[leg]['ms_per_step_compute_only'] < dp[leg]['ms_per_step']
    c4 = out['config4_sharded_step']            # (an item -> entity map whose ids all fell on one owner overflowed this leg's wire rows until round 6)
    assert 'error' not in c4 and (c4.get('skipped') or (c4['world'] == 2 and c4['ms_per_step'] > 0))
    c5 = out['config5_step']
    assert 'error' not in c5 and (c5.get('skipped') or (c5['world'] == 2 and c5['ms_per_step'] > 0 and c5['batch_per_rank'] == 8192))
