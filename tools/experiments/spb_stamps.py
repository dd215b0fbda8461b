import sys, os; sys.path.insert(0, '.')
os.environ.setdefault('NABU_PERSIST_DEBUG', '4')
os.environ.setdefault('NABU_SPELLER_PERSIST_BWD_LOC', '2')     # cfg5: the location-aware backward kernel (not its default there)
import torch, numpy as np
import bench
from nabu_amd import _hip
args = bench.parse_args(['--workload', sys.argv[1] if len(sys.argv) > 1 else 'cfg3', '--no-cpu-baseline'])
w = bench.make_workload(args, bench.make_server())
for i in range(3): w.step(i)
torch.cuda.synchronize()
buf = [v for (d,t),v in _hip.Workspace._bufs.items() if t=='speller'][0]
st = buf[:1024].view(torch.int32).cpu().numpy().astype(np.int64)
x = st[48:59]
names = ['D1 gathers, saves, dot', 'D1 (loc) features + carry / (vanilla) -', 'D1 d-align', 'D1 tanh/dq', 'D1 publish (+ d features, d conv kernel)', 'D1b gather + sum + publish', 'D2 gather', 'D2 product + cell + publish', 'D3 gather + products', 'D3 reduce + publish']
for i in range(10): print('%-44s %6.2f us' % (names[i], ((x[i + 1] - x[i]) & 0xffffffff) / 100.0))
print('stamps 0..10 span %.2f us' % (((x[10] - x[0]) & 0xffffffff) / 100.0))
