"""CPU baseline of the training step (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

SURVEY.md 8(d), "How the reference's CPU path is timed beside it": the reference's own
``non_distributed`` trainer (Python 2 + TensorFlow 1.8, reference trainers/trainer.py:759-767)
cannot run in this image or on the GPU box, so what is timed — and labelled ``kind: "port"`` —
is a restatement of the reference graph at TF's op granularity with PyTorch-CPU float32:

  * ``tf.nn.bidirectional_dynamic_rnn`` (reference components/layer.py:35-47): per direction a
    loop over ALL T_max frames with ONE ``[B, in+H] x [in+H, 4H]`` matmul + the gate elementwise
    ops per frame, ``reverse_sequence`` for the backward direction, zero output / copied-through
    state past each sequence's length;
  * pyramid_stack (components/ops.py:6-60) as strided slices + concat;
  * ``tf.nn.ctc_loss`` (trainers/loss_functions.py:206-210) = torch's CPU CTC kernel;
  * backward = reverse-mode autodiff over that graph (TF: ``tf.gradients``);
  * ``clip_by_value`` + ``AdamOptimizer`` applied variable by variable (trainers/trainer.py:556-569).

A second number replaces the per-frame loop with ``torch.nn.LSTM`` on packed sequences (the fused
CPU kernel): an UPPER baseline, faster than anything TF 1.8's ``dynamic_rnn`` could do.

Only ``bench.py``'s ``cpu_baseline`` leg and ``tests/`` import this file; ``nabu_amd`` never does.
No TF number is claimed anywhere."""
import os
import statistics
import time

import numpy as np
import torch

CONFIGS = {
    # name: (encoder, pyramidal layers, B, T, D, H, C, label range, time reduction, seed)
    'cfg1': ('dblstm', 0, 8, 200, 40, 256, 40, (10, 40), 1, 1234),
    'cfg2': ('listener', 3, 32, 1000, 40, 512, 40, (20, 60), 8, 4234),
}


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def _glorot(gen, *shape):
    fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else shape
    lim = (6.0 / (fan_in + fan_out)) ** 0.5
    return ((torch.rand(*shape, generator=gen) * 2 - 1) * lim).requires_grad_()


def make_params(cfg, seed=99):
    enc, npyr, B, T, D, H, C = CONFIGS[cfg][:7]
    gen = torch.Generator().manual_seed(seed)
    layers = []
    din = D
    n = npyr + 1 if enc == 'listener' else 2
    for l in range(n):
        layers.append({k: _glorot(gen, din + H, 4 * H) if 'kernel' in k else _glorot(gen, 4 * H)
                       for k in ('fw_kernel', 'fw_bias', 'bw_kernel', 'bw_bias')})
        din = 4 * H if (enc == 'listener' and l < npyr) else 2 * H
    out = {'weights': _glorot(gen, 2 * H, C), 'biases': torch.zeros(C, requires_grad=True)}
    return layers, out


def make_batch(cfg):
    from nabu_amd.processing.synthetic import SyntheticData
    enc, npyr, B, T, D, H, C, (lo, hi), red, seed = CONFIGS[cfg]
    b = SyntheticData(B, T, D, min_frames=T, min_labels=lo, max_labels=hi, time_reduction=red, seed=seed).batch(0)
    return (torch.from_numpy(b['inputs']['features']), torch.from_numpy(b['input_seq_length']['features']).long(),
            torch.from_numpy(b['targets']['text']).long(), torch.from_numpy(b['target_seq_length']['text']).long())


def reverse_sequence(x, lens):
    B, T = x.shape[:2]
    t = torch.arange(T)[None, :]
    idx = torch.where(t < lens[:, None], lens[:, None] - 1 - t, t)
    return torch.gather(x, 1, idx[:, :, None].expand(-1, -1, x.shape[2]))


def dynamic_rnn(x, lens, kernel, bias, reverse):
    """one direction at TF op granularity"""
    B, T, _ = x.shape
    H = kernel.shape[1] // 4
    if reverse:
        x = reverse_sequence(x, lens)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = []
    for t in range(T):
        z = torch.cat([x[:, t], h], 1) @ kernel + bias
        i, j, f, o = z.chunk(4, 1)
        c_new = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h_new = torch.tanh(c_new) * torch.sigmoid(o)
        m = (t < lens)[:, None]
        outs.append(torch.where(m, h_new, torch.zeros_like(h_new)))
        c = torch.where(m, c_new, c)
        h = torch.where(m, h_new, h)
    y = torch.stack(outs, 1)
    return reverse_sequence(y, lens) if reverse else y


def fused_lstm_dir(x, lens, kernel, bias, reverse):
    """the same direction through torch.nn.LSTM's fused CPU kernel (upper baseline): TF gate
    order i,j,f,o -> torch i,f,g,o, forget bias folded into the bias"""
    D = x.shape[2]
    H = kernel.shape[1] // 4
    perm = torch.cat([torch.arange(0, H), torch.arange(2 * H, 3 * H), torch.arange(H, 2 * H),
                      torch.arange(3 * H, 4 * H)])
    w = kernel[:, perm]
    b = bias[perm] + torch.cat([torch.zeros(H), torch.ones(H), torch.zeros(2 * H)])
    if reverse:
        x = reverse_sequence(x, lens)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    flat = [w[:D].t().contiguous(), w[D:].t().contiguous(), b, torch.zeros_like(b)]
    y, _, _ = torch._VF.lstm(packed.data, packed.batch_sizes, (x.new_zeros(1, x.shape[0], H),) * 2, flat, True, 1,
                             0.0, True, False)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(
        torch.nn.utils.rnn.PackedSequence(y, packed.batch_sizes, packed.sorted_indices, packed.unsorted_indices),
        batch_first=True, total_length=x.shape[1])
    return reverse_sequence(y, lens) if reverse else y


def train_step(cfg, layers, out, batch, state, fused=False):
    """forward + CTC + backward + per-variable clip + TF-Adam; returns the loss"""
    enc, npyr = CONFIGS[cfg][:2]
    x, lens, labels, label_len = batch
    rnn = fused_lstm_dir if fused else dynamic_rnn
    h, l = x, lens
    for li, p in enumerate(layers):
        h = torch.cat([rnn(h, l, p['fw_kernel'], p['fw_bias'], False),
                       rnn(h, l, p['bw_kernel'], p['bw_bias'], True)], 2)
        if enc == 'listener' and li < npyr:
            if h.shape[1] % 2:
                h = torch.cat([h, h.new_zeros(h.shape[0], 1, h.shape[2])], 1)
            h = torch.cat([h[:, 0::2], h[:, 1::2]], 2)
            l = (l + 1) // 2
    logits = h @ out['weights'] + out['biases']
    C = logits.shape[2]
    lp = torch.log_softmax(logits, 2).transpose(0, 1)
    tg = torch.cat([labels[b, :int(label_len[b])] for b in range(labels.shape[0])])
    loss = torch.nn.functional.ctc_loss(lp, tg, l, label_len, blank=C - 1, reduction='none').mean()
    variables = [v for p in layers for v in p.values()] + list(out.values())
    grads = torch.autograd.grad(loss, variables)
    state['t'] += 1
    t = state['t']
    lr_t = 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
    with torch.no_grad():
        for k, (v, g) in enumerate(zip(variables, grads)):
            g = g.clamp(-1.0, 1.0)
            m, s = state['m'].setdefault(k, torch.zeros_like(v)), state['v'].setdefault(k, torch.zeros_like(v))
            m.mul_(0.9).add_(g, alpha=0.1)
            s.mul_(0.999).addcmul_(g, g, value=0.001)
            v.sub_(lr_t * m / (s.sqrt() + 1e-8))
    return float(loss.detach())


def time_config(cfg, warmup, steps, fused=False, threads=None, budget_s=None):
    """-> dict(seconds=[per step], median, utt_per_s, threads, steps_timed)"""
    threads = threads or physical_cores()
    torch.set_num_threads(threads)
    layers, out = make_params(cfg)
    batch = make_batch(cfg)
    state = {'t': 0, 'm': {}, 'v': {}}
    t_start = time.perf_counter()
    for _ in range(warmup):
        train_step(cfg, layers, out, batch, state, fused)
    secs = []
    for k in range(steps):
        t0 = time.perf_counter()
        loss = train_step(cfg, layers, out, batch, state, fused)
        secs.append(time.perf_counter() - t0)
        if budget_s is not None and k + 1 >= 2 and time.perf_counter() - t_start > budget_s:
            break
    med = statistics.median(secs)
    return {'seconds_per_step': [round(s, 4) for s in secs], 'median_s': round(med, 4),
            'utt_per_s': round(CONFIGS[cfg][2] / med, 3), 'threads': threads, 'warmup': warmup,
            'steps_timed': len(secs), 'final_loss': round(loss, 4)}


if __name__ == '__main__':
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for c, w, s in (('cfg1', 2, 5), ('cfg2', 1, 2)):
        print(c, json.dumps(time_config(c, w, s)), flush=True)
        print(c, 'fused', json.dumps(time_config(c, w, s, fused=True)), flush=True)
