"""The 13 dense products of a cfg2 training step on the REAL operands of a model that has trained for N steps
(VERDICT r03 item 1a): error against float64 of f16x3 / bf16x6 relative to the exact-fp32 MFMA kernel's, rms and max.
usage: python tools/experiments/real_operands.py [steps=50]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from nabu_amd import recipes, ops
from nabu_amd.neuralnetworks.components import layer
from nabu_amd.neuralnetworks.trainers import trainer_factory
from nabu_amd.processing.synthetic import SyntheticData
from real_operand_products import layer_products, product_errors

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.cuda.set_device(0)
ops.set_gemm_precision('bf16x6')
mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc')
data = SyntheticData(32, 1000, 40, min_frames=600, min_labels=20, max_labels=60, time_reduction=8, seed=2234)
tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec, expdir=None, server=None, task_index=0)
for i in range(steps):
    loss = tr.step(tr.to_device(data.batch(i)))
print('trained %d steps, loss %.4f' % (steps, float(loss)))
cap = []
layer.CAPTURE[0] = cap
tr.step(tr.to_device(data.batch(steps)))
layer.CAPTURE[0] = None
torch.cuda.synchronize()
print('%-28s %8s | %21s | %21s' % ('product', 'K', 'f16x3 / fp32 rms  max', 'bf16x6 / fp32 rms  max'))
for li, c in enumerate(reversed(cap)):          # captured in backward order
    for name, a, b in layer_products(c):
        e = product_errors(a, b)
        print('layer %d %-20s %8d | %9.3f %9.3f | %9.3f %9.3f' % (li, name, a.shape[1], e['f16x3'][0], e['f16x3'][1], e['bf16x6'][0], e['bf16x6'][1]))
