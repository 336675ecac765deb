#!/bin/bash
# SQ counters of the all-candidate evaluation kernels (two PMC passes, kernel trace only): run from the repo root via gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_a /tmp/pmc_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_a -- python $R/tools/eval_variants.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_b -- python $R/tools/eval_variants.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ('/tmp/pmc_a', '/tmp/pmc_b'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, set()])
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'pairs_kernel' not in k and 'pairs_hard' not in k:
                continue
            k = k[k.index('pairs'):][:40]
            a = acc[(k, row['Counter_Name'])]
            a[0] += float(row['Counter_Value']); a[1].add(row['Dispatch_Id'])
        for (k, c), (v, ids) in sorted(acc.items()):
            print('%-42s %-22s %.4g  (per launch, %d launches)' % (k, c, v / len(ids), len(ids)))
PY
