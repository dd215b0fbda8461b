"""One-time independent cross-check of the FULL-SIZE golden fixtures (cfg2_exact, cfg3_exact,
cfg5_exact): the same step recomputed by PyTorch-CPU float64 autograd — the batched per-frame
restatement of oracle/cpu_baseline.py for the encoder and CTC, tests/torch_ref.py for the Speller
(both written from the TF op semantics, not from oracle/nabu_oracle.py) — and compared with the
numbers the NumPy oracle put into the fixtures.  Minutes of CPU per config; the outcome is
committed as tests/golden/exact_xcheck.json and asserted by tests/test_golden.py.

    python tests/golden/xcheck_exact.py cfg2_exact cfg3_exact cfg5_exact"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cpu_baseline as CB                                # noqa: E402
from tests import torch_ref as R                                     # noqa: E402
from tests.golden import make_golden as G                            # noqa: E402

F64 = torch.float64


def listener(x, lens, layers):
    h, l = x, lens
    for li, p in enumerate(layers):
        h = torch.cat([CB.dynamic_rnn(h, l, p['fw_kernel'], p['fw_bias'], False),
                       CB.dynamic_rnn(h, l, p['bw_kernel'], p['bw_bias'], True)], 2)
        if li < len(layers) - 1:
            if h.shape[1] % 2:
                h = torch.cat([h, h.new_zeros(h.shape[0], 1, h.shape[2])], 1)
            h = torch.cat([h[:, 0::2], h[:, 1::2]], 2)
            l = (l + 1) // 2
    return h, l


def tt(x):
    if isinstance(x, dict):
        return {k: tt(v) for k, v in x.items()}
    if isinstance(x, list):
        return [tt(v) for v in x]
    return torch.tensor(np.asarray(x), dtype=F64, requires_grad=True)


def check(name):
    fx = dict(np.load(os.path.join(HERE, name + '.npz')))
    names, data, _, loss_name, _, _ = G.exact_setup(name)
    w = G.draw_weights(names)
    b = data.batch(0)
    t0 = time.time()
    tl = tt(G.enc_layers(w, 'Listener', 3))
    x = torch.tensor(b['inputs']['features'], dtype=F64)
    e, el = listener(x, torch.tensor(b['input_seq_length']['features']).long(), tl)
    y, yl = b['targets']['text'], b['target_seq_length']['text']
    grads = {}
    if loss_name == 'CTC':
        W, bo = tt(w['DNNDecoder/text/outlayer/weights']), tt(w['DNNDecoder/text/outlayer/biases'])
        loss = R.ctc_mean(e @ W + bo, el.numpy(), y, yl).mean()
        loss.backward()
        grads['DNNDecoder/text/outlayer/weights'] = W.grad
        grads['DNNDecoder/text/outlayer/biases'] = bo.grad
    else:
        attention = 'vanilla' if name == 'cfg3_exact' else 'location_aware'
        tp = tt(G.speller_view(w, 1, attention))
        lg = R.speller(e, [int(v) for v in el], y, yl, tp, attention, 'softmax')
        loss = R.avg_xent(lg, y, yl, yl)
        loss.backward()
        sg = dict(memory_kernel=tp['memory_kernel'].grad, query_kernel=tp['query_kernel'].grad,
                  attention_v=tp['attention_v'].grad, out_kernel=tp['out_kernel'].grad, out_bias=tp['out_bias'].grad,
                  lstm=[dict(kernel=l['kernel'].grad, bias=l['bias'].grad) for l in tp['lstm']])
        if attention == 'location_aware':
            sg['conv_kernel'], sg['conv_proj'] = tp['conv_kernel'].grad, tp['conv_proj'].grad
        grads.update({k: v for k, v in G.speller_grads_by_name(sg, w, 1, attention).items()})
    grads.update(G.enc_grads_by_name([{k: v.grad for k, v in l.items()} for l in tl], 'Listener', 3))
    res = {'loss_torch': float(loss.detach()), 'loss_oracle': float(fx['losses'][0]),
           'loss_rel': abs(float(loss.detach()) - float(fx['losses'][0])) / float(fx['losses'][0]),
           'seconds': None, 'gnorm_rel_max': 0.0, 'gsample_rel_max': 0.0, 'variables': len(grads)}
    for k, g in grads.items():
        flat = g.detach().numpy().ravel()
        key = k.replace('/', '|')
        gn = float(fx['gnorm:' + key])
        res['gnorm_rel_max'] = max(res['gnorm_rel_max'], abs(np.sqrt((flat ** 2).sum()) - gn) / gn)
        ref = fx['gsample:' + key]
        res['gsample_rel_max'] = max(res['gsample_rel_max'],
                                     float(np.abs(flat[G.sample_index(k, flat.size)] - ref).max() / np.abs(ref).max()))
    res['seconds'] = round(time.time() - t0, 1)
    return res


if __name__ == '__main__':
    path = os.path.join(HERE, 'exact_xcheck.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    for n in sys.argv[1:]:
        out[n] = check(n)
        print(n, out[n], flush=True)
        assert out[n]['loss_rel'] < 1e-9 and out[n]['gnorm_rel_max'] < 1e-7 and out[n]['gsample_rel_max'] < 1e-7
        with open(path, 'w') as fid:
            json.dump(out, fid, indent=1, sort_keys=True)
