"""Timeline of the last replayed step(s) out of a `rocprofv3 --kernel-trace --output-format csv` run: for every kernel between two
occurrences of an anchor kernel (default: route_init_kernel, the first launch of a sharded step) its start relative to the anchor, its
duration and its hardware queue -- what runs beside what, and where the gaps are.
    python tools/trace_timeline.py <dir or *_kernel_trace.csv> [anchor substring] [steps back from the end, default 3]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else 'route_init_kernel'
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if (anchor in r['Kernel_Name'] and 'count' not in r['Kernel_Name'])]
    # an anchor occurrence that follows another within 40 us belongs to the same step (the owner's route has an init launch too)
    firsts = [i for k, i in enumerate(idx) if k == 0 or int(rows[i]['Start_Timestamp']) - int(rows[idx[k - 1]]['Start_Timestamp']) > 40000]
    if len(firsts) < back + 1:
        raise SystemExit('anchor %r found %d times' % (anchor, len(firsts)))
    a, b = firsts[-back - 1], firsts[-back]
    t0 = int(rows[a]['Start_Timestamp'])
    print('step of %.1f us (%d launches)' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
    last_end = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name']
        for junk in ('void ', '(anonymous namespace)::', 'ktup::'):
            name = name.replace(junk, '')
        print('%8.1f  +%6.1f us  gap %6.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, r.get('Queue_Id', '?'), name[:100]))
        last_end = max(last_end, e)


if __name__ == '__main__':
    main()
