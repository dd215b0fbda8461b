import sys, os; sys.path.insert(0, '.')
os.environ.setdefault('NABU_PERSIST_DEBUG', '4')
import torch, numpy as np
import bench
from nabu_amd import _hip
args = bench.parse_args(['--workload', os.environ.get('WL', 'cfg3'), '--no-cpu-baseline'])
w = bench.make_workload(args, bench.make_server())
for i in range(3): w.step(i)
torch.cuda.synchronize()
buf = _hip.Workspace._bufs[(str(torch.device('cuda', 0)), 'speller')] if (str(torch.device('cuda',0)),'speller') in _hip.Workspace._bufs else [v for (d,t),v in _hip.Workspace._bufs.items() if t=='speller'][0]
st = buf[:1024].view(torch.int32).cpu().numpy().astype(np.int64)
x = st[16:27]
names = ["A gather", "A mfma", "A red+barrier+gates", "barrier + B gather", "B compute+publish", "C gather", "C scores", "C stats+partial+publish", "D gather", "D combine"]
print('status', st[0])
for i, n in enumerate(names): print('%-22s %6.2f us' % (n, ((x[i + 1] - x[i]) & 0xffffffff) / 100.0))
print('total step %.2f us' % (((x[10] - x[0]) & 0xffffffff) / 100.0))

base = x[0]
for nm, o in (('step start', 128), ('h publish', 64), ('q publish', 96)):
    v = ((st[o:o + 32] - base) & 0xffffffff).astype(np.int64)
    v = np.where(v > 2**31, v - 2**32, v) / 100.0
    print(nm, 'per slot (us rel. to block 0 step start):', np.round(v, 2))
print('xcc*2+coloc of blocks 0..31', st[160:192])
it = ((st[193:223] - base) & 0xffffffff) / 100.0; print('B gather poll iteration end times (us rel. step start):', [round(float(v), 2) for v in it if v < 100])
print('barrier after gates: %.2f us' % (((st[16+11]-st[16+3]) & 0xffffffff)/100.0))
print('B gather: iterations', st[224], 'loop end at %.2f us' % (((st[225]-base) & 0xffffffff)/100.0))
print('save_step: %.2f us; B loop only: %.2f us' % (((st[16+13]-st[16+1]) & 0xffffffff)/100.0, ((st[16+12]-st[16+4]) & 0xffffffff)/100.0))
dw=(st[232]-st[230]) & 0xffffffff; dc=(st[233]-st[231]) & 0xffffffff; print('8 steps: %.2f us, %d shader clocks -> %.2f GHz' % (dw/100.0, dc, dc/(dw*10.0)))
