"""Soak run of the persistent kernels: N training steps of a workload, status word checked at the end (a protocol
error in the exchange rings shows as a bounded-spin time-out = a loud failure, or as a NaN / diverging loss).
usage: python tools/experiments/soak.py cfg2 3000 [td]      (td: the reference's regularisation defaults switched on)"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench

wl_name, steps = sys.argv[1], int(sys.argv[2])
args = bench.parse_args(['--workload', wl_name, '--no-cpu-baseline', '--no-alt', '--no-gemm-roofline'] +
                        (['--training-defaults'] if 'td' in sys.argv[3:] else []))
server = bench.make_server()
wl = bench.make_workload(args, server)
t0 = time.time()
for i in range(steps):
    wl.step(i)
    if i % 500 == 499:
        wl.check()
wl.sync()
wl.check()
loss = float(wl.loss.item())
assert loss == loss, 'NaN loss'
print('SOAK OK %s: %d steps in %.1f s, %.3f ms/step, final loss %.4f' % (wl_name, steps, time.time() - t0, (time.time() - t0) * 1e3 / steps, loss))
