"""Flag sweep of the three drop-in command lines (run by hand on a GPU box: `python tests/flag_sweep_cli.py [workers]`): every model
type x the flags the reference defines (models/base.py:23-98: optimizer, topn, negtive_samples, batch size, distance, gate, preference
count, width, weight decay, momentum, sharing, joint ratio, host / device sampling, -shard_tables) on the synthetic dataset of the
end-to-end tests, 25 training steps with two evaluations each.  A run is reported when it exits non-zero, logs a non-finite loss or
prints no metric row."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.synth import make_dataset

PKG = os.path.join(ROOT, 'joint-kg-recommender_amd')
REC, KG, JOINT = 'run_item_recommendation.py', 'run_knowledge_representation.py', 'run_knowledgable_recommendation.py'
RT, KT = ['-rec_test_files', 'valid.dat:test.dat'], ['-kg_test_files', 'valid.dat:test.dat']
cases = []


def add(script, model, *flags):
    base = ['-model_type', model] + (RT if script != KG else []) + (KT if script != REC else [])
    cases.append((script, base + list(flags)))


for opt in ('SGD', 'Adam', 'Adagrad', 'Rmsprop'):
    for m in ('bprmf', 'fm', 'transup'):
        add(REC, m, '-optimizer_type', opt)
    for m in ('transe', 'transh', 'transr'):
        add(KG, m, '-optimizer_type', opt)
    for m in ('jtransup', 'cke', 'cfkg', 'cofm'):
        add(JOINT, m, '-optimizer_type', opt)
for f in (['-topn', '1'], ['-topn', '20'], ['-topn', '60'], ['-negtive_samples', '3'], ['-batch_size', '7'], ['-batch_size', '2000'], ['-l2_lambda', '0'],
          ['-momentum', '0', '-optimizer_type', 'SGD'], ['-nodevice_sampling'], ['-seed', '0'], ['-clipping_max_value', '0.01'],
          ['-learning_rate_decay_when_no_progress', '0.1', '-early_stopping_steps_to_wait', '10']):
    add(REC, 'transup', *f); add(REC, 'bprmf', *f); add(KG, 'transh', *f); add(KG, 'transe', *f); add(KG, 'transr', *f)
    add(JOINT, 'jtransup', '-noshare_embeddings', *f); add(JOINT, 'cke', *f)
for f in (['-L1_flag'], ['-use_st_gumbel'], ['-L1_flag', '-use_st_gumbel'], ['-num_preferences', '1'], ['-num_preferences', '40'],
          ['-num_preferences', '70', '-L1_flag'], ['-embedding_size', '7'], ['-embedding_size', '300'], ['-embedding_size', '256', '-L1_flag'],
          ['-embedding_size', '216'], ['-embedding_size', '128', '-use_st_gumbel']):
    add(REC, 'transup', *f)
    add(JOINT, 'jtransup', '-noshare_embeddings', *f)
for f in (['-L1_flag'], ['-embedding_size', '7'], ['-embedding_size', '300'], ['-embedding_size', '256', '-L1_flag'], ['-margin', '3']):
    for m in ('transe', 'transh', 'transr'):
        if m == 'transr' and '300' in f:
            continue
        add(KG, m, *f)
for f in (['-share_embeddings'], ['-noshare_embeddings'], ['-joint_ratio', '0.1'], ['-joint_ratio', '0.9'], ['-kg_lambda', '0.1', '-norm_lambda', '0.3']):
    for m in ('jtransup', 'cke', 'cfkg', 'cofm'):
        add(JOINT, m, *f)
for f in (['-shard_tables'], ['-shard_tables', '-optimizer_type', 'Adam'], ['-shard_tables', '-use_st_gumbel'], ['-shard_tables', '-L1_flag'],
          ['-shard_tables', '-embedding_size', '100'], ['-shard_eval_candidates']):
    add(JOINT, 'jtransup', '-noshare_embeddings', *f)
    if '-shard_eval_candidates' not in f:
        add(REC, 'transup', *f)

tmp = tempfile.mkdtemp()
make_dataset(tmp)
logs = os.path.join(tmp, 'log')
os.makedirs(logs, exist_ok=True)


def run(k):
    script, flags = cases[k]
    name = 'c%03d' % k
    cmd = [sys.executable, os.path.join(PKG, script), '-data_path', tmp, '-log_path', logs, '-dataset', 'ml1m', '-experiment_name', name,
           '-nohas_visualization', '-batch_size', '32', '-embedding_size', '20', '-seed', '3', '-eval_interval_steps', '10', '-training_steps', '25',
           '-early_stopping_steps_to_wait', '0', '-learning_rate', '0.05', '-topn', '10'] + flags
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return k, 'timeout'
    if r.returncode != 0:
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()][-1:]
        return k, 'exit %d: %s' % (r.returncode, ' '.join(tail)[:220])
    log = open(os.path.join(logs, name + '.log')).read()
    losses = re.findall(r'loss:\s*(\S+?)[,\s]', log)
    if any(x.lower().startswith(('nan', 'inf')) for x in losses):
        return k, 'non-finite loss'
    if not re.search(r'(f1:\d|avg hit:\d)', log):
        return k, 'no metric row'
    return k, None


with ThreadPoolExecutor(int(sys.argv[1]) if len(sys.argv) > 1 else 4) as ex:
    res = list(ex.map(run, range(len(cases))))
nbad = 0
for k, msg in res:
    if msg:
        nbad += 1
        print('PROBLEM %s %s: %s' % (cases[k][0], ' '.join(cases[k][1]), msg))
print('%d runs, %d problems' % (len(cases), nbad))
sys.exit(1 if nbad else 0)
