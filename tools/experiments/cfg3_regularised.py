"""cfg3 training step with the reference's training defaults for the Speller (sample_prob 0.1, output dropout 0.9):
persistent decoder (round 3) against the step chain (NABU_SPELLER_PERSIST=0).  python tools/experiments/cfg3_regularised.py"""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from nabu_amd import recipes  # noqa: E402
from nabu_amd.computing import dist  # noqa: E402
from nabu_amd.neuralnetworks.trainers import trainer_factory  # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData  # noqa: E402

over = {'decoder.sample_prob': os.environ.get('SAMPLE', '0.1'), 'decoder.dropout': os.environ.get('KEEP', '0.9')}
mc, tc, ec = recipes.load_recipe('cfg3_las_vanilla', **over)
data = SyntheticData(32, 1000, 40, min_frames=1000, min_labels=20, max_labels=79, eos=True, time_reduction=8, seed=3234)
tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec, expdir=None,
                                         server=dist.create_server(), task_index=0)
bs = [tr.to_device(data.batch(i)) for i in range(2)]
for i in range(3):
    tr.step(bs[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for i in range(n):
    loss = tr.step(bs[i % 2])
torch.cuda.synchronize()
print('cfg3 sample_prob %s keep %s persist %s: %.2f ms/step, loss %.4f' % (over['decoder.sample_prob'], over['decoder.dropout'],
      os.environ.get('NABU_SPELLER_PERSIST', '1'), (time.perf_counter() - t0) / n * 1e3, float(loss.item())))
