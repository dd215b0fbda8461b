"""Phase stamps of one persistent-decoder step (NABU_PERSIST_DEBUG=4) with recipe overrides from the environment:
OVER='{"decoder.dropout": "0.5"}' python tools/experiments/sp_stamps_td.py"""
import sys, os, json; sys.path.insert(0, '.')
os.environ.setdefault('NABU_PERSIST_DEBUG', '4')
import torch, numpy as np
import bench
from nabu_amd import _hip, recipes
over = json.loads(os.environ.get('OVER', '{}'))
orig = recipes.load_recipe
recipes.load_recipe = lambda r, **kw: orig(r, **dict(kw, **over))
args = bench.parse_args(['--workload', 'cfg3', '--no-cpu-baseline'])
w = bench.make_workload(args, bench.make_server())
for i in range(3): w.step(i)
torch.cuda.synchronize()
buf = [v for (d,t),v in _hip.Workspace._bufs.items() if t=='speller'][0]
st = buf[:1024].view(torch.int32).cpu().numpy().astype(np.int64)
x = st[16:27]
names = ["A gather", "A mfma", "A red+barrier+gates", "barrier + B gather", "B compute+publish", "C gather", "C scores", "C stats+partial+publish", "D gather", "D combine"]
print(over, 'status', st[0], 'paths', w.rank_paths())
print('  '.join('%s %.2f' % (n, ((x[i + 1] - x[i]) & 0xffffffff) / 100.0) for i, n in enumerate(names)))
print('total step %.2f us' % (((x[10] - x[0]) & 0xffffffff) / 100.0))
dw=(st[232]-st[230]) & 0xffffffff; dc=(st[233]-st[231]) & 0xffffffff; print('8 steps: %.2f us, %d shader clocks -> %.2f GHz' % (dw/100.0, dc, dc/(dw*10.0)))
