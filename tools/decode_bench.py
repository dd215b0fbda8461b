"""Times the inference decoders (decode.hip) at the BASELINE shapes: CTC prefix beam search on
[32,125,40] / [8,200,40] posteriors, edit distance, attention beam search on the cfg3 decoder.
    python tools/decode_bench.py          (profiles/r01_decode_kernel_stats.csv comes from this command
                                           under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nabu_amd import ops, recipes
from nabu_amd import variables as vs
from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory, rnn_decoder
rng = np.random.default_rng(0)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B, T, C in ((32, 125, 40), (8, 200, 40), (32, 777, 40)):
    lg = rng.normal(0, 1, (B, T, C)).astype(np.float32)
    path = rng.integers(0, C, (B, T)); lg[np.arange(B)[:, None], np.arange(T)[None, :], path] += 4
    lg = torch.tensor(lg, device='cuda'); ln = torch.full((B,), T, dtype=torch.int32, device='cuda')
    print('ctc beam search B=%d T=%d C=%d beam 100: %.2f ms (%.1f us/frame)' % (B, T, C, timed(lambda: ops.ctc_beam_search(lg, ln)), timed(lambda: ops.ctc_beam_search(lg, ln)) * 1e3 / T))
h = torch.randint(0, 39, (32, 60), dtype=torch.int32, device='cuda'); r = torch.randint(0, 39, (32, 60), dtype=torch.int32, device='cuda')
l = torch.full((32,), 60, dtype=torch.int32, device='cuda')
print('edit distance 32 x (60,60): %.3f ms' % timed(lambda: ops.edit_distance(h, l, r, l), 20))
B, Te, E, U, C, W, S = 32, 125, 1024, 512, 40, 16, 100
mc, _, _ = recipes.load_recipe('cfg3_las_vanilla')
dec = ed_decoder_factory.factory('speller')(mc, {'text': C}, None)
store = vs.VariableStore(seed=1)
enc = torch.randn(B, Te, E, device='cuda'); el = SeqLen(np.full(B, Te, np.int32), 'cuda')
with torch.no_grad(), vs.as_default(store), vs.variable_scope(dec.scope):
    cell = dec.create_cell({'features': enc}, {'features': el}, False)
    def run():
        return rnn_decoder.beam_search(cell, enc, el, W, S, 1.0, 1.0)
    ms = timed(run, 3)
    seqs = run()[0]
print('attention beam search cfg3 decoder B=32 beam 16: %.1f ms for %d steps (%.0f us/step)' % (ms, seqs.shape[2], ms * 1e3 / seqs.shape[2]))
