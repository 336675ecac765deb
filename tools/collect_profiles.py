"""Collect the rocprofv3 evidence for profiles/ on a GPU box (run from the repo root through gpurun):

    python tools/collect_profiles.py r01

Three separate passes over the same bench command, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (PMC counters in
their own runs, kernel trace + stats only): (1) --kernel-trace --stats, (2) --pmc FETCH_SIZE, (3) --pmc WRITE_SIZE.
Writes gpurun_out/profiles/<tag>_kernel_stats.csv, <tag>_hbm_counters.csv, <tag>_hbm_traffic.json.
FETCH_SIZE / WRITE_SIZE are in KB; the read side is doubled (gfx950 correction of the guide's HBM section)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r01'
OUT = os.path.join(ROOT, 'gpurun_out', 'profiles')
BENCH = ['python', os.path.join(ROOT, 'bench.py'), '--no-extras']
BENCH_GS = ['python', os.path.join(ROOT, 'bench.py'), '--only', 'gather_stress']
sys.path.insert(0, ROOT)
KERNELS = {'ktup_rec_forward': 'pref_fwd_mc_kernel', 'ktup_kg_forward': 'transh_fwd_tile_kernel'}


def rocprof(name, flags, bench_args, bench=None):
    d = os.path.join('/tmp', 'ktup_prof_' + name)
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3'] + flags + ['--output-format', 'csv', '-d', d, '--'] + (bench or BENCH) + bench_args
    r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit('rocprofv3 failed: ' + ' '.join(cmd))
    return d


def find(d, suffix):
    hits = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    if not hits:
        raise SystemExit('no %s under %s' % (suffix, d))
    return hits[0]


def counter_per_launch(path, counter):
    """{kernel substring: average counter value per dispatch} from a rocprofv3 counter_collection.csv."""
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            key = (row['Kernel_Name'], row['Dispatch_Id'])
            per[key] = per.get(key, 0.0) + float(row['Counter_Value'])
    out = {}
    for tag, sub in KERNELS.items():
        vals = [v for (k, _), v in per.items() if sub in k]
        names = sorted({k for (k, _) in per if sub in k})
        out[tag] = (sum(vals) / max(len(vals), 1), len(vals), names[0] if names else None)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    d = rocprof('stats', ['--kernel-trace', '--stats'], ['--steps', '100', '--warmup', '10'])
    shutil.copy(find(d, 'kernel_stats.csv'), os.path.join(OUT, TAG + '_kernel_stats.csv'))
    rows = []
    res = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = rocprof(counter.lower(), ['--kernel-trace', '--pmc', counter], ['--steps', '5', '--warmup', '2'])
        path = find(d, 'counter_collection.csv')
        with open(path) as f:
            rd = csv.DictReader(f)
            for row in rd:
                if any(sub in row['Kernel_Name'] for sub in KERNELS.values()):
                    rows.append({k: row[k] for k in ('Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size',
                                                     'VGPR_Count', 'Counter_Name', 'Counter_Value')})
        res[counter] = counter_per_launch(path, counter)
    with open(os.path.join(OUT, TAG + '_hbm_counters.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    traffic = {}
    for tag in KERNELS:
        rd_kb, n, name = res['FETCH_SIZE'][tag]
        wr_kb, _, _ = res['WRITE_SIZE'][tag]
        read, write = 2.0 * rd_kb * 1024.0, wr_kb * 1024.0
        traffic[tag] = {'kernel': name, 'launches': n, 'hbm_bytes_per_launch': int(read + write), 'read': int(read),
                        'write': int(write)}
    # gather-stress companion (tables x1000 rows, HBM-resident): the same two counters over `bench.py --only gather_stress`
    gs = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = rocprof('gs_' + counter.lower(), ['--kernel-trace', '--pmc', counter], [], bench=BENCH_GS)
        gs[counter] = counter_per_launch(find(d, 'counter_collection.csv'), counter)
    for tag in KERNELS:
        rd_kb, n, name = gs['FETCH_SIZE'][tag]
        wr_kb, _, _ = gs['WRITE_SIZE'][tag]
        read, write = 2.0 * rd_kb * 1024.0, wr_kb * 1024.0
        traffic['gather_stress_' + tag] = {'kernel': name, 'launches': n, 'hbm_bytes_per_launch': int(read + write), 'read': int(read),
                                           'write': int(write)}
    import bench as B
    traffic['kernel_src_sha16'] = B.kernel_src_sha16()          # bench.py drops `traffic` when the kernel sources have changed since
    traffic['_method'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB units); read side doubled per '
                          'MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports half of wide coalesced reads); average over all '
                          'launches of `bench.py --steps 5 --warmup 2 --no-extras`')
    with open(os.path.join(OUT, TAG + '_hbm_traffic.json'), 'w') as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))
    with open(os.path.join(OUT, TAG + '_kernel_stats.csv')) as f:
        print(''.join(f.readlines()[:6]))


if __name__ == '__main__':
    main()
