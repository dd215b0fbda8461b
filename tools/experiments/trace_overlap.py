import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name']
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', '?')))
ev.sort()
# decoder region = between first and last attn kernel of the LAST step in the trace
att = [e for e in ev if 'attn_' in e[2]]
t0, t1 = att[0][0], att[-1][1]
# split into contiguous regions (fwd decoder, bwd decoder per step) by gaps > 1 ms between attention kernels
regions, cur = [], [att[0]]
for e in att[1:]:
    if e[0] - cur[-1][1] > 1e6: regions.append(cur); cur = [e]
    else: cur.append(e)
regions.append(cur)
for reg in regions:
    a, b = reg[0][0], reg[-1][1]
    inside = [e for e in ev if e[0] >= a - 50000 and e[1] <= b + 50000]
    # union busy time and sum of durations
    pts = sorted([(e[0], 1) for e in inside] + [(e[1], -1) for e in inside])
    busy, depth, last, hist = 0, 0, None, collections.Counter()
    for t, d in pts:
        if last is not None and depth > 0: busy += t - last; hist[min(depth, 6)] += t - last
        depth += d; last = t
    tot = sum(e[1] - e[0] for e in inside)
    byk = collections.Counter()
    for e in inside: byk[e[2][:40]] += e[1] - e[0]
    print('region %.2f ms: %d kernels, sum of durations %.2f ms, busy (union) %.2f ms, idle %.2f ms, queues %d' % (
        (b - a) / 1e6, len(inside), tot / 1e6, busy / 1e6, (b - a - busy) / 1e6, len(set(e[3] for e in inside))))
    print('   time at concurrency depth', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
    print('   top kernels (ms of duration):', [(k, round(v / 1e6, 2)) for k, v in byk.most_common(6)])
# per-queue gaps inside the first region
reg = regions[0]
a, b = reg[0][0], reg[-1][1]
inside = [e for e in ev if e[0] >= a and e[1] <= b]
byq = collections.defaultdict(list)
for e in inside: byq[e[3]].append(e)
for q, lst in byq.items():
    lst.sort()
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    durs = [e[1] - e[0] for e in lst]
    gaps.sort()
    print('queue', q, 'kernels', len(lst), 'mean dur %.1f us' % (sum(durs) / len(durs) / 1e3), 'gap median %.1f us, p10 %.1f, p90 %.1f, mean %.1f' % (
        gaps[len(gaps) // 2] / 1e3, gaps[len(gaps) // 10] / 1e3, gaps[9 * len(gaps) // 10] / 1e3, sum(gaps) / len(gaps) / 1e3))
q0 = sorted(byq.values(), key=len)[-1][:14]
print([(e[2][11:30], round((e[0] - q0[0][0]) / 1e3, 1), round((e[1] - e[0]) / 1e3, 1)) for e in q0])
