"""cfg3 with the reference's regularisation defaults switched on ONE AT A TIME: what each costs the step (ms per step).
usage: python tools/experiments/cfg3_td_parts.py [cfg3|cfg5]"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
CASES = [('plain', {}), ('encoder dropout + noise', {'encoder.input_noise': '0.6', 'encoder.dropout': '0.5'}),
         ('decoder dropout', {'decoder.dropout': '0.5'}), ('sample_prob 1e-6', {'decoder.sample_prob': '0.000001'}),
         ('sample_prob 0.1', {'decoder.sample_prob': '0.1'}), ('decoder dropout + sampling', {'decoder.dropout': '0.5', 'decoder.sample_prob': '0.1'})]
from nabu_amd import recipes
orig = recipes.load_recipe
for name, over in CASES:
    recipes.load_recipe = lambda r, **kw: orig(r, **dict(kw, **over))
    args = bench.parse_args(['--workload', w, '--no-cpu-baseline', '--no-alt', '--no-gemm-roofline'])
    wl = bench.make_workload(args, bench.make_server())
    for i in range(3):
        wl.step(i)
    wl.sync()
    t0 = time.perf_counter()
    for i in range(10):
        wl.step(i)
    wl.sync()
    wl.check()
    print('%-28s %.3f ms/step   decoder persistent (fwd, bwd bits): %s' % (name, (time.perf_counter() - t0) * 100, wl.rank_paths()[1]))
    del wl
    torch.cuda.empty_cache()
