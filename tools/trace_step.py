#!/usr/bin/env python
"""Timeline of ONE training step out of a rocprofv3 --kernel-trace CSV (the last complete step between two
adam_clip_kernel launches): start (us), duration, gap to the previous kernel, name; then totals per kernel name.

    python tools/trace_step.py gpurun_out/<tag>_kernel_trace.csv [--summary]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('nabu::', '')
    return n[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_clip' in r['Kernel_Name']]
    step = rows[idx[-2] + 1:idx[-1] + 1]
    t0 = int(step[0]['Start_Timestamp'])
    prev = None
    tot = collections.OrderedDict()
    gaps = 0.0
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0.0
        gaps += max(gap, 0.0)
        k = short(r['Kernel_Name'])
        c = tot.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
        if '--summary' not in sys.argv:
            print('%9.1f %8.1f gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, k))
        prev = e
    print('step span %.1f us, %d launches, gaps %.1f us' % ((prev - t0) / 1e3, len(step), gaps))
    for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print('%9.1f us %4d x  %s' % (us, n, k))


if __name__ == '__main__':
    main()
