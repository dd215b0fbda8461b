"""cfg3 geometry (32 x 125 encoder frames x 1024, 512 units: keys AND values of a slice fit the LDS) with location-aware
attention (10 filters of 101 taps): persistent backward kernel (NABU_SPELLER_PERSIST_BWD_LOC=1, round 3) against the
step chain for the backward pass (=0).  python tools/experiments/cfg3_location_aware.py"""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from nabu_amd import recipes  # noqa: E402
from nabu_amd.computing import dist  # noqa: E402
from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder  # noqa: E402
from nabu_amd.neuralnetworks.trainers import trainer_factory  # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData  # noqa: E402

over = {'decoder.attention': 'location_aware', 'decoder.numfilt': '10', 'decoder.filtersize': '101', 'decoder.sample_prob': '0'}
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
print('cfg3 + location-aware attention, persistent backward %s, paths %s: %.2f ms/step, loss %.4f' % (
    os.environ.get('NABU_SPELLER_PERSIST_BWD_LOC', '1'), rnn_decoder.dynamic_decode.last_paths,
    (time.perf_counter() - t0) / n * 1e3, float(loss.item())))
