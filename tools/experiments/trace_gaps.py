"""Idle time between the kernels of a training step from a rocprofv3 --kernel-trace CSV (steps delimited by adam_clip_kernel).
usage: rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline; python tools/experiments/trace_gaps.py /tmp/tr/*/*kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find step boundaries: adam_clip_kernel ends a step
ends = [i for i, r in enumerate(rows) if 'adam_clip' in r['Kernel_Name']]
print('kernels', len(rows), 'adam launches', len(ends))
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
    busy = 0; cur_end = t0; gaps = []
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur_end: gaps.append((s - cur_end, r['Kernel_Name'][:40]))
        busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    tot_gap = sum(g for g, _ in gaps)
    print('step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps (median %.2f us, >10us: %d)' % (
        len(seg), (t1 - t0) / 1e6, busy / 1e6, tot_gap / 1e6, len(gaps), sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3,
        sum(1 for g, _ in gaps if g > 10000)))
    big = sorted(gaps, reverse=True)[:8]
    print('   largest gaps before:', [(round(g / 1e3, 1), n) for g, n in big])
    by = collections.defaultdict(float)
    for g, n in gaps: by[n] += g
    print('   idle by following kernel:', [(n, round(v / 1e3)) for n, v in sorted(by.items(), key=lambda x: -x[1])[:8]])
