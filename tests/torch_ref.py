"""Independent PyTorch-CPU float64 restatement (autograd) used ONLY to pin the
NumPy oracle (tests/test_oracle.py).  Written from the TF-1.8 op semantics in
SURVEY.md 8(a), not from oracle/nabu_oracle.py, so that a shared mistake is
unlikely."""
import torch


def lstm_dir(x, lens, kernel, bias, reverse):
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    outs = []
    for b in range(B):
        n = int(lens[b])
        xs = x[b, :n]
        if reverse:
            xs = torch.flip(xs, [0])
        h = x.new_zeros(H)
        c = x.new_zeros(H)
        ys = []
        for t in range(n):
            z = torch.cat([xs[t], h]) @ kernel + bias
            i, j, f, o = z.split(H)
            c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            h = torch.tanh(c) * torch.sigmoid(o)
            ys.append(h)
        y = torch.stack(ys) if ys else x.new_zeros(0, H)
        if reverse:
            y = torch.flip(y, [0])
        outs.append(torch.cat([y, x.new_zeros(T - n, H)]))
    return torch.stack(outs)


def blstm(x, lens, p):
    return torch.cat([lstm_dir(x, lens, p['fw_kernel'], p['fw_bias'], False),
                      lstm_dir(x, lens, p['bw_kernel'], p['bw_bias'], True)], 2)


def pyramid(x, lens, n):
    B, T, F = x.shape
    Tp = -(-T // n) * n
    x = torch.cat([x, x.new_zeros(B, Tp - T, F)], 1)
    # reference: gather every n-th frame with offset i and concat on features
    parts = [x[:, i::n] for i in range(n)]
    return torch.cat(parts, 2), [-(-int(l) // n) for l in lens]


def listener(x, lens, layers, n=2):
    h, l = x, list(lens)
    for p in layers[:-1]:
        h = blstm(h, l, p)
        h, l = pyramid(h, l, n)
    return blstm(h, l, layers[-1]), l


def dblstm(x, lens, layers):
    h = x
    for p in layers:
        h = blstm(h, lens, p)
    return h, list(lens)


def ctc_mean(logits, logit_len, labels, label_len):
    B, T, C = logits.shape
    lp = torch.log_softmax(logits, 2).transpose(0, 1)
    tg = torch.cat([torch.as_tensor(labels[b][:int(label_len[b])]) for b in range(B)])
    nll = torch.nn.functional.ctc_loss(
        lp, tg, torch.as_tensor(logit_len), torch.as_tensor(label_len),
        blank=C - 1, reduction='none', zero_infinity=False)
    return nll


def avg_xent(logits, targets, logit_len, target_len):
    B, L, C = logits.shape
    ce = torch.nn.functional.cross_entropy(
        logits.reshape(B * L, C), torch.as_tensor(targets)[:, :L].reshape(-1).long(),
        reduction='none').reshape(B, L)
    mask = torch.arange(L)[None, :] < torch.as_tensor(logit_len)[:, None]
    ce = torch.where(mask, ce, torch.zeros_like(ce))
    return (ce.sum(1) / torch.as_tensor(target_len).to(logits.dtype)).mean()


def speller(enc, enc_len, targets, target_len, p, attention, prob_fn, window=None):
    B, Te, E = enc.shape
    C = p['out_bias'].shape[0]
    U = p['attention_v'].shape[0]
    L = int(max(target_len))
    logits = []
    for b in range(B):
        n = int(enc_len[b])
        values = torch.cat([enc[b, :n], enc.new_zeros(Te - n, E)])
        keys = values @ p['memory_kernel']
        hs = [enc.new_zeros(U) for _ in p['lstm']]
        cs = [enc.new_zeros(U) for _ in p['lstm']]
        ctx = enc.new_zeros(E)
        al = enc.new_zeros(Te)
        if attention == 'windowed':
            al[0] = 1.0
        prev = C - 1
        row = []
        for t in range(L):
            if t >= int(target_len[b]):
                row.append(enc.new_zeros(C))
                continue
            x = torch.cat([torch.nn.functional.one_hot(torch.tensor(prev), C).to(enc.dtype), ctx])
            for k, lp in enumerate(p['lstm']):
                z = torch.cat([x, hs[k]]) @ lp['kernel'] + lp['bias']
                i, j, f, o = z.split(U)
                cs[k] = cs[k] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
                hs[k] = torch.sigmoid(o) * torch.tanh(cs[k])
                x = hs[k]
            q = x @ p['query_kernel']
            s = keys + q[None, :]
            if attention == 'location_aware':
                K = p['conv_kernel'].shape[0]
                pb = (K - 1) // 2
                padded = torch.cat([al.new_zeros(pb), al, al.new_zeros(K - 1 - pb)])
                cf = torch.nn.functional.conv1d(
                    padded[None, None, :], p['conv_kernel'].t()[:, None, :])[0].t()  # [Te,F]
                s = s + cf @ p['conv_proj']
            score = torch.tanh(s) @ p['attention_v']
            valid = torch.arange(Te) < n
            if attention == 'windowed':
                # interval form of the reference's shift/xor window: [m - left - 1, m + right) around the
                # first frame m whose cumulated previous alignment exceeds one half
                over = (torch.cumsum(al.detach(), 0) > 0.5).nonzero()
                m = int(over[0]) if len(over) else Te
                idx = torch.arange(Te)
                valid = valid & (idx >= m - window[0] - 1) & (idx < m + window[1])
            if prob_fn == 'softmax':
                al = torch.softmax(torch.where(valid, score, torch.full_like(score, -float('inf'))), 0)
            else:
                sg = torch.where(valid, torch.sigmoid(score), torch.zeros_like(score))
                al = sg if prob_fn == 'sigmoid' else sg / sg.sum()
            ctx = al @ values
            row.append(torch.cat([x, ctx]) @ p['out_kernel'] + p['out_bias'])
            prev = int(targets[b][t])
        logits.append(torch.stack(row))
    return torch.stack(logits)
