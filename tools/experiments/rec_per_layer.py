"""Per-layer duration of the recurrent launches INSIDE a cfg2 training step (HIP events of the library), for
GEMM precision f32 / bf16x6 and the deferred weight-gradient order on / off:
  python tools/experiments/rec_per_layer.py"""
import os
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from nabu_amd import ops  # noqa: E402
from nabu_amd.neuralnetworks.components import layer  # noqa: E402

for prec, defer in (('f32', True), ('bf16x6', True), ('bf16x6', False), ('f32', False)):
    layer.DEFER_WEIGHT_GRADS[0] = defer
    args = bench.parse_args(['--gemm-precision', prec, '--no-cpu-baseline', '--no-alt'])
    wl = bench.HipWorkload(args, bench.make_server())
    for i in range(3):
        wl.step(i)
    wl.sync()
    wl.prof.collect()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 8
    for i in range(n):
        wl.step(i)
    e1.record()
    wl.sync()
    recs = wl.prof.collect()
    agg = {}
    for r in recs:
        agg.setdefault((r[0], r[2]), []).append(r[4] * 1e3 / r[2])
    print('%-7s defer %-5s step %.2f ms | us per sequential step by (pass, T): %s' % (
        prec, defer, e0.elapsed_time(e1) / n,
        '  '.join('%s/%d %.3f' % (k[0], k[1], sum(v) / len(v)) for k, v in sorted(agg.items()))), flush=True)
    ops.PROFILER = None
