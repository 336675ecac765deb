"""End-to-end throughput of the drop-in command line: run_knowledgable_recommendation.py (jtransup, d=100, B=512,
joint_ratio 0.7) on an ml1m-SHAPED synthetic dataset written in the reference's file formats, steps per second between
two periodic evaluations (so data loading, model construction and the step-0 evaluation are excluded; one evaluation pass
of the small validation files is included).  Host-side sampling (the reference's python samplers) vs -device_sampling."""
import datetime
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.synth import make_dataset

PKG = os.path.join(ROOT, 'joint-kg-recommender_amd')


def run(data, name, steps, extra, intervals=4):
    logs = os.path.join(data, 'log')
    os.makedirs(logs, exist_ok=True)
    cmd = [sys.executable, os.path.join(PKG, 'run_knowledgable_recommendation.py'), '-data_path', data, '-log_path', logs,
           '-dataset', 'ml1m', '-experiment_name', name, '-nohas_visualization', '-batch_size', '512', '-embedding_size', '100',
           '-seed', '3', '-eval_interval_steps', str(steps), '-training_steps', str((intervals + 1) * steps + 1), '-early_stopping_steps_to_wait', '0',
           '-learning_rate', '0.005', '-topn', '10', '-model_type', 'jtransup', '-rec_test_files', 'valid.dat', '-kg_test_files',
           'valid.dat', '-joint_ratio', '0.7', '-noshare_embeddings', '-num_preferences', '20', '-optimizer_type', 'Adagrad',
           '-log_level', 'info'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1700)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-3000:])
    stamps, marks = [], []
    for line in open(os.path.join(logs, name + '.log')):
        for tag in ('rec train loss', 'f1:', 'avg mrr'):
            if tag in line:
                t = datetime.datetime.strptime(line[:23], '%Y-%m-%d %H:%M:%S,%f')
                marks.append((tag, t))
                if tag == 'rec train loss':
                    stamps.append(t)
    assert len(stamps) >= intervals + 1, 'expected evaluations at steps 0, N, 2N, ...'
    # per interval k (evaluation at step k*N, then N steps): rec pass, kg pass, the rest (checkpoint + the steps)
    rows = []
    for k in range(1, intervals + 1):
        seg = [m for m in marks if stamps[k] <= m[1] and (k + 1 >= len(stamps) or m[1] < stamps[k + 1])]
        t0 = stamps[k]
        t_rec = next((m[1] for m in seg if m[0] == 'f1:'), t0)
        t_kg = next((m[1] for m in seg if m[0] == 'avg mrr'), t_rec)
        t1 = stamps[k + 1] if k + 1 < len(stamps) else None
        if t1 is not None:
            rows.append(((t1 - t0).total_seconds(), (t_rec - t0).total_seconds(), (t_kg - t_rec).total_seconds(), (t1 - t_kg).total_seconds()))
    dt = rows[0][0]                                   # the interval the earlier rounds quoted: evaluation at step N + steps N..2N
    best = min(r[0] for r in rows)
    return steps / dt, dt, steps / best, rows


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    only = sys.argv[2] if len(sys.argv) > 2 else None          # 'dev' / 'host': that sampling mode alone (bench.py's leg runs 'dev')
    with tempfile.TemporaryDirectory() as tmp:
        make_dataset(tmp, n_users=6040, n_items=3240, n_ent=14708, n_rel=20, n_ratings=120000, n_triples=60000, aligned=2934)
        for name, extra in (('dev', ['-device_sampling']), ('host', ['-nodevice_sampling'])):
            if only and name != only:
                continue
            n = steps if name == 'dev' else max(200, steps // 10)
            sps, dt, sps_best, rows = run(tmp, name, n, extra)
            print('%-5s sampling: %8.0f steps/s  (%d steps of B=512 in %.2f s incl. one evaluation pass) = %.2f M scored rows/s'
                  % (name, sps, n, dt, sps * 1024 / 1e6))
            print('%-5s sampling: fastest of %d intervals %8.0f steps/s; per interval [total, rec pass, kg pass, checkpoint + steps] s: %s'
                  % (name, len(rows), sps_best, ' '.join('[%.3f %.3f %.3f %.3f]' % r for r in rows)))


if __name__ == '__main__':
    main()
