import os, sys; sys.path.insert(0, '.')
from nabu_amd import _hip
if os.environ.get('NABU_LIB'): _hip.LIB_PATH = os.path.abspath(os.environ['NABU_LIB'])
import torch, bench
from nabu_amd.neuralnetworks.trainers import loss_functions
args = bench.parse_args(['--workload', 'cfg2', '--no-cpu-baseline'])
w = bench.make_workload(args, bench.make_server())
losses = []
for i in range(3):
    w.step(i); losses.append(float(w.loss.item()))
loss_functions.check_status()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = int(os.environ.get('NSTEPS', '10'))
e0.record()
for i in range(n): w.step(i)
e1.record(); torch.cuda.synchronize()
loss_functions.check_status()
print('LIB', os.environ.get('NABU_LIB', 'default'), 'ms/step %.3f' % (e0.elapsed_time(e1) / n), 'losses', losses, 'last', float(w.loss.item()))
