"""Per-kernel averages of a rocprofv3 counter_collection.csv:  python tools/pmc_summary.py <csv> [kernel substring ...]"""
import csv
import sys
from collections import defaultdict

per = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row['Kernel_Name']
        if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
            continue
        per[k][row['Counter_Name']] += float(row['Counter_Value'])
        calls[k].add(row['Dispatch_Id'])
for k, c in per.items():
    n = max(len(calls[k]), 1)
    print(k[:120])
    print('   launches %d  ' % n + '  '.join('%s=%.3g' % (name, v / n) for name, v in sorted(c.items())))
