"""Per-launch durations of the recurrent kernels inside training steps with idle gaps of 0 / 2 / 10 ms between the steps
(what the first forward launch pays for what ran before it; LABNOTES section 9).  python tools/experiments/launch_context.py"""
import sys, time; sys.path.insert(0, '.')
import torch, bench
args = bench.parse_args(['--no-cpu-baseline', '--no-alt', '--no-gemm-roofline', '--no-other-configs'])
wl = bench.make_workload(args, bench.make_server())
for i in range(5): wl.step(i)
wl.sync()
for gap in (0.0, 0.002, 0.01, 0.0):
    wl.start_timed_region()
    for i in range(6):
        wl.prof.enabled = True
        wl.timing = False
        wl.loss = wl.tr.step(wl.batches[i % 2])
        if gap:
            wl.sync(); time.sleep(gap)
    wl.sync()
    wl.end_timed_region()
    recs = wl.recs
    # records: (fwd?, B, T, H, ms) presumably; print per-launch ms grouped by T
    byT = {}
    for r in recs[8:]:
        byT.setdefault((r[0], r[2]), []).append(r[4])
    print('gap %.0f ms:' % (gap * 1e3), '  '.join('%s T=%d %.3f' % ('fwd' if k[0] else 'bwd', k[1], sum(v) / len(v)) for k, v in sorted(byT.items(), key=lambda kv: (not kv[0][0], -kv[0][1]))))
