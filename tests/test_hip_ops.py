"""GPU parity tests of the individual HIP kernels, called through the C ABI
(nabu_amd.ops -> libnabu_hip.so), against the NumPy float64 oracle on the same
seeded inputs.  Tolerances are stated per test (fp32 kernels vs float64 oracle).
"""
import numpy as np
import pytest
import torch

from oracle import nabu_oracle as O

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def rel_err(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (200, 136, 40), (37, 40, 1024), (1, 5, 3),
                                   (40, 256, 4100), (300, 40, 48),
                                   # real batches: B*T is rarely a multiple of anything — odd row counts and a
                                   # reduction length with a tail beyond the fast kernel's 16-wide k-tile
                                   (1003, 136, 1592), (512, 2048, 1001), (129, 516, 88)])
def test_gemm_matches_float64(ta, tb, M, N, K):
    from nabu_amd import ops
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = rng.normal(size=(K, M) if ta else (M, K))
    b = rng.normal(size=(N, K) if tb else (K, N))       # asymmetric operands
    c0 = rng.normal(size=(M, N))
    bias = rng.normal(size=N)
    ref = 0.5 * (a.T if ta else a) @ (b.T if tb else b) + 2.0 * c0 + bias
    c = dev(c0)
    ops.gemm(dev(a), dev(b), c, ta, tb, alpha=0.5, beta=2.0, bias=dev(bias))
    # fp32 fma chain over K terms: error ~ 1e-7 * sqrt(K) * |a||b|
    assert rel_err(host(c), ref) < 2e-6 * max(1.0, np.sqrt(K) / 4)


def test_gemm_beta0_ignores_garbage_and_unaligned_ld():
    from nabu_amd import ops
    rng = np.random.default_rng(3)
    a = rng.normal(size=(50, 23)); b = rng.normal(size=(23, 31))       # ld not multiple of 4
    c = torch.full((50, 31), float('nan'), device='cuda')
    ops.gemm(dev(a), dev(b), c)
    assert rel_err(host(c), a @ b) < 2e-6


def test_gemm_segmented_k_shifted_product():
    """dWh = sum_b out[b, :-1]^T dz[b, 1:] without materialising the shift."""
    from nabu_amd import ops
    rng = np.random.default_rng(4)
    B, T, H = 5, 9, 8
    out = rng.normal(size=(B, T, 2 * H)); dz = rng.normal(size=(B, T, 4 * H))
    ref_fw = np.einsum('bth,btg->hg', out[:, :-1, :H], dz[:, 1:])
    ref_bw = np.einsum('bth,btg->hg', out[:, 1:, H:], dz[:, :-1])
    o, d = dev(out), dev(dz)
    c = torch.empty((H, 4 * H), device='cuda')
    ops.gemm(o, d.view(-1)[4 * H:], c, True, False, M=H, N=4 * H, K=B * (T - 1), lda=2 * H, ldb=4 * H,
             ldc=4 * H, kseg=T - 1, a_seg=T * 2 * H, b_seg=T * 4 * H)
    assert rel_err(host(c), ref_fw) < 2e-6
    ops.gemm(o.view(-1)[H + 2 * H:], d, c, True, False, M=H, N=4 * H, K=B * (T - 1), lda=2 * H,
             ldb=4 * H, ldc=4 * H, kseg=T - 1, a_seg=T * 2 * H, b_seg=T * 4 * H)
    assert rel_err(host(c), ref_bw) < 2e-6


def test_gemm_segmented_k_with_a_tail():
    """cfg1's dWh: 8 segments of 199 frames — K = 1592 = 99 k-tiles of 16 + 8; the tail lies inside the
    last segment and is added by the generic kernel"""
    from nabu_amd import ops
    rng = np.random.default_rng(14)
    B, T, H = 8, 200, 64
    out = rng.normal(size=(B, T, 2 * H)); dz = rng.normal(size=(B, T, 4 * H))
    ref_fw = np.einsum('bth,btg->hg', out[:, :-1, :H], dz[:, 1:])
    o, d = dev(out), dev(dz)
    c = torch.full((H, 4 * H), float('nan'), device='cuda')
    ops.gemm(o, d.view(-1)[4 * H:], c, True, False, M=H, N=4 * H, K=B * (T - 1), lda=2 * H, ldb=4 * H,
             ldc=4 * H, kseg=T - 1, a_seg=T * 2 * H, b_seg=T * 4 * H)
    assert rel_err(host(c), ref_fw) < 4e-6
    bias = rng.normal(size=4 * H)
    c2 = dev(np.ones((H, 4 * H)))
    ops.gemm(o, d.view(-1)[4 * H:], c2, True, False, M=H, N=4 * H, K=B * (T - 1), lda=2 * H, ldb=4 * H,
             ldc=4 * H, kseg=T - 1, a_seg=T * 2 * H, b_seg=T * 4 * H, alpha=0.5, beta=3.0, bias=dev(bias))
    assert rel_err(host(c2), 0.5 * ref_fw + 3.0 + bias) < 4e-6


def test_colsum():
    from nabu_amd import ops
    rng = np.random.default_rng(5)
    a = rng.normal(size=(1500, 200))
    out = dev(np.ones(200))
    ops.colsum(dev(a), out, beta=0.5)
    assert rel_err(host(out), a.sum(0) + 0.5) < 1e-5
    out2 = dev(np.ones(200)); out3 = dev(np.ones(200))
    ops.colsum(dev(a), out2); ops.colsum(dev(a), out3)
    assert torch.equal(out2, out3)                      # deterministic reduction


# ------------------------------------------------------------------ BLSTM
def _blstm_params(rng, D, H, scale=0.2):
    return dict(fw_kernel=rng.normal(0, scale, (D + H, 4 * H)), fw_bias=rng.normal(0, scale, 4 * H),
                bw_kernel=rng.normal(0, scale, (D + H, 4 * H)), bw_bias=rng.normal(0, scale, 4 * H))


def _run_blstm(B, T, D, H, lens, mode, seed=0, need_dx=True, precision='default'):
    from nabu_amd import ops
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(B, T, D)).astype(np.float32).astype(np.float64)
    for b in range(B):
        x[b, lens[b]:] = 0
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in _blstm_params(rng, D, H).items()}
    dout = rng.normal(size=(B, T, 2 * H)).astype(np.float32).astype(np.float64)
    ref_out, cache = O.blstm_fwd(x, np.asarray(lens), p)
    ref_dx, ref_g = O.blstm_bwd(dout, cache)

    plan = ops.BlstmPlan(B, T, D, H, int(max(lens)), mode, precision)
    xd, ld = dev(x), dev(np.asarray(lens), torch.int32)
    pd = {k: dev(v) for k, v in p.items()}
    out = torch.full((B, T, 2 * H), float('nan'), device='cuda')
    reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    ops.blstm_fwd(plan, xd, ld, pd['fw_kernel'], pd['fw_bias'], pd['bw_kernel'], pd['bw_bias'], out, reserve)
    got_out = host(out)
    dx = torch.full((B, T, D), float('nan'), device='cuda') if need_dx else None
    g = {k: torch.full(v.shape, float('nan'), device='cuda') for k, v in pd.items()}
    ops.blstm_bwd(plan, xd, ld, pd['fw_kernel'], pd['bw_kernel'], out, dev(dout), reserve, dx,
                  g['fw_kernel'], g['fw_bias'], g['bw_kernel'], g['bw_bias'])
    torch.cuda.synchronize()
    return got_out, ref_out, (host(dx) if need_dx else None), ref_dx, {k: host(v) for k, v in g.items()}, ref_g


@pytest.mark.parametrize('B,T,D,H,lens', [
    (3, 7, 8, 16, [7, 4, 1]),          # ragged, tiny
    (8, 40, 40, 64, None),             # all full length
    (5, 33, 12, 32, [33, 20, 33, 2, 17]),
    (17, 12, 20, 48, None),            # B not a multiple of the block tile, H % 16 == 0
    (2, 9, 4, 20, [9, 5]),             # H not a multiple of 16
    (4, 10, 8, 16, [6, 4, 6, 3]),      # max(len) < T (padded tail never visited)
])
def test_blstm_stepwise_matches_oracle(B, T, D, H, lens):
    from nabu_amd import ops
    lens = lens or [T] * B
    out, rout, dx, rdx, g, rg = _run_blstm(B, T, D, H, lens, ops.LSTM_STEPWISE, seed=B * T)
    assert np.isfinite(out).all()
    assert np.abs(out - rout).max() < 2e-5              # fp32 recurrence vs float64
    for b, n in enumerate(lens):
        assert np.all(out[b, n:] == 0)                   # TF zero-output past len: exact
    assert rel_err(dx, rdx) < 2e-4
    for k in rg:
        assert rel_err(g[k], rg[k]) < 2e-4, k


def test_blstm_T1_and_first_layer_without_dx():
    from nabu_amd import ops
    out, rout, dx, rdx, g, rg = _run_blstm(3, 1, 8, 16, [1, 1, 1], ops.LSTM_STEPWISE, need_dx=False)
    assert np.abs(out - rout).max() < 2e-5
    for k in rg:
        assert np.abs(g[k] - rg[k]).max() < 1e-5, k
    assert np.all(g['fw_kernel'][8:] == 0)               # no recurrent contribution with T=1


# ------------------------------------------------------------------ CTC
def _ctc_case(rng, B, T, C, Lmax, full=False):
    logits = rng.normal(0, 2, (B, T, C)).astype(np.float32).astype(np.float64)
    tl = np.full(B, T) if full else rng.integers(max(2 * Lmax + 1, T // 2), T + 1, B)
    tl[0] = T
    ll = rng.integers(0, Lmax + 1, B)
    ll[-1] = Lmax
    labels = rng.integers(0, C - 1, (B, Lmax))
    if Lmax >= 2:
        labels[0, 1] = labels[0, 0]                      # adjacent repeat
    return logits, tl, labels, ll


@pytest.mark.parametrize('B,T,C,Lmax', [(4, 30, 6, 5), (32, 125, 40, 60), (8, 200, 40, 40), (3, 5, 3, 2),
                                        (2, 300, 40, 140)])
def test_ctc_matches_oracle(B, T, C, Lmax):
    from nabu_amd import ops
    rng = np.random.default_rng(B + T)
    logits, tl, labels, ll = _ctc_case(rng, B, T, C, Lmax)
    rn, rg = O.ctc_loss(logits, tl, labels, ll)
    nll, dl, status = ops.ctc_loss_grad(dev(logits), dev(tl, torch.int32), dev(labels, torch.int32),
                                        dev(ll, torch.int32), 1.0 / B)
    assert int(status.item()) == 0
    assert np.abs(host(nll) - rn).max() / np.abs(rn).max() < 1e-5      # fp32 log-space
    # float32 log-space: alpha, beta ~ -3T, so gamma carries ~1e-7*3T*few relative error
    assert np.abs(host(dl) * B - rg).max() < 1e-6 * T * 4 + 1e-5
    for b in range(B):
        assert np.all(host(dl)[b, tl[b]:] == 0)


def test_ctc_matches_tensorflows_known_answer_vectors():
    """the HIP kernel on TensorFlow's own CTC test vectors (tests/golden/tf_ctc_basic.npz, [TF-1.8 recalled],
    see tests/test_oracle.py): losses to 1e-5, gradients to 2e-6 — blank = last class, repeated labels"""
    import os
    from nabu_amd import ops
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_ctc_basic.npz'))
    nll, dl, status = ops.ctc_loss_grad(dev(np.log(fx['prob'])), dev(fx['logit_len'], torch.int32), dev(fx['labels'], torch.int32),
                                        dev(fx['label_len'], torch.int32), 1.0)
    assert int(status.item()) == 0
    assert np.abs(host(nll) - fx['loss']).max() < 1e-5
    assert np.abs(host(dl) - fx['grad']).max() < 2e-6


def test_ctc_edge_cases():
    from nabu_amd import ops
    C = 5
    # empty label, T=1, and an infeasible utterance (repeat needs a blank)
    logits = np.random.default_rng(1).normal(size=(3, 4, C))
    tl = np.array([4, 1, 2]); ll = np.array([0, 1, 2]); labels = np.array([[0, 0], [2, 0], [1, 1]])
    nll, dl, status = ops.ctc_loss_grad(dev(logits), dev(tl, torch.int32), dev(labels, torch.int32),
                                        dev(ll, torch.int32), 1.0)
    rn, rg = O.ctc_loss(logits[:2], tl[:2], labels[:2], ll[:2])
    assert np.abs(host(nll)[:2] - rn).max() < 1e-5
    assert np.abs(host(dl)[:2] - rg).max() < 1e-5
    assert int(status.item()) == 3 and np.isinf(host(nll)[2])           # TF raises for this one
    assert np.all(host(dl)[2] == 0)


def test_ctc_is_deterministic():
    from nabu_amd import ops
    rng = np.random.default_rng(9)
    logits, tl, labels, ll = _ctc_case(rng, 8, 60, 40, 20)
    args = (dev(logits), dev(tl, torch.int32), dev(labels, torch.int32), dev(ll, torch.int32), 1.0)
    a = ops.ctc_loss_grad(*args); b = ops.ctc_loss_grad(*args)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# ------------------------------------------------------------------ Adam
def test_adam_clip_matches_oracle_over_steps():
    from nabu_amd import ops
    rng = np.random.default_rng(11)
    n = 10007                                            # not a multiple of 4
    th = rng.normal(size=n); m = np.zeros(n); v = np.zeros(n)
    p, md, vd = dev(th), dev(m), dev(v)
    for t in range(1, 6):
        g = rng.normal(0, 1.5, n).astype(np.float32).astype(np.float64)
        th, m, v = O.clip_adam_update(th, g, m, v, t, 1e-3)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_clip_step(p, dev(g), md, vd, lr_t)
    assert np.abs(host(p) - th).max() < 1e-6
    assert np.abs(host(md) - m).max() < 1e-6
    g = dev(rng.normal(0, 3, n)); ops.clip_(g, 1.0)
    assert float(g.abs().max()) <= 1.0


def test_pad_unpad_time():
    from nabu_amd import ops
    x = torch.randn(3, 7, 8, device='cuda')
    y = ops.pad_time(x, 8)
    assert torch.equal(y[:, :7], x) and torch.all(y[:, 7] == 0)
    assert torch.equal(ops.unpad_time(y, 7), x)


# ------------------------------------------------------------------ persistent BLSTM
@pytest.mark.parametrize('B,T,D,H,lens', [
    (8, 12, 8, 64, None),
    (5, 21, 12, 64, [21, 9, 1, 21, 14]),        # ragged, B < shard size
    (19, 16, 20, 128, None),                    # 3 shards, last one partial
    (32, 24, 40, 512, None),                    # the cfg2 geometry: 8 units x 32 workgroups
    (32, 30, 16, 512, 'ragged'),
    (8, 25, 40, 256, 'ragged'),                 # the cfg1 geometry
    (4, 10, 8, 128, [6, 4, 6, 3]),              # max(len) < T
    (64, 12, 16, 512, 'ragged'),                # the cfg5 batch: two launches of 32 rows
    (40, 9, 8, 512, None),                      # chunk of 32 + chunk of 8
    # narrow input (D = 40): the forward kernel projects x itself (lstm_persist.hip, XK); D = 80 keeps the GEMM
    (6, 17, 40, 64, [17, 3, 9, 1, 12, 17]),     # B not a multiple of 4, ragged
    (7, 11, 40, 128, [8, 5, 8, 2, 1, 7, 8]),    # max(len) < T
    (7, 11, 80, 128, [8, 5, 8, 2, 1, 7, 8]),    # D = 80
    (40, 9, 40, 512, 'ragged'),                 # chunk of 32 + chunk of 8
])
def test_blstm_persistent_matches_oracle(B, T, D, H, lens):
    from nabu_amd import ops
    if lens == 'ragged':
        lens = list(np.random.default_rng(B + T).integers(1, T + 1, B))
        lens[0] = T
    lens = lens or [T] * B
    out, rout, dx, rdx, g, rg = _run_blstm(B, T, D, H, lens, ops.LSTM_PERSISTENT, seed=B + H)
    ops.check_persist_status()
    assert np.isfinite(out).all()
    # fp32 dot products of length H with |z| up to ~5 (weights N(0, 0.2)): rounding grows with H
    assert np.abs(out - rout).max() < (2e-5 if H <= 128 else 2e-4)
    for b, n in enumerate(lens):
        assert np.all(out[b, n:] == 0)
    assert rel_err(dx, rdx) < 3e-4
    for k in rg:
        assert rel_err(g[k], rg[k]) < 3e-4, k


def test_blstm_persistent_is_deterministic_and_mode_is_reported():
    import ctypes
    from nabu_amd import ops, _hip
    a = _run_blstm(16, 20, 8, 128, [20] * 16, ops.LSTM_PERSISTENT, seed=1)
    b = _run_blstm(16, 20, 8, 128, [20] * 16, ops.LSTM_PERSISTENT, seed=1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    for k in a[4]:
        assert np.array_equal(a[4][k], b[4][k])
    L = _hip.lib()
    d = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 1000, 40, 512, 1000, ops.LSTM_AUTO, 0)
    assert L.nabu_blstm_uses_persistent(ctypes.byref(d)) == 1
    d.mode = ops.LSTM_STEPWISE
    assert L.nabu_blstm_uses_persistent(ctypes.byref(d)) == 0
    d.mode, d.H = ops.LSTM_AUTO, 48
    assert L.nabu_blstm_uses_persistent(ctypes.byref(d)) == 0       # falls back to stepwise


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('prec,tol', [('bf16x6', 3e-6), ('bf16x3', 2e-5), ('bf16', 1e-2)])
def test_gemm_bf16_split_precisions(ta, tb, prec, tol):
    """nabu_gemm_ex on the bf16 matrix pipe: error against float64, relative to max|C|.  bf16x6 must
    be as accurate as the exact-fp32 MFMA kernel (both are limited by fp32 accumulation)."""
    from nabu_amd import ops
    rng = np.random.default_rng(31 + 2 * ta + tb)
    M, N, K = 392, 260, 1056                     # edge tiles in M and N, K a multiple of 32
    a = rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32) * np.exp(rng.normal(size=(1, 1))).astype(np.float32)
    b = rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    ad, bd = torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda')
    out = {}
    for p in (prec, 'f32'):
        c = torch.zeros(M, N, device='cuda')
        ops.gemm(ad, bd, c, bool(ta), bool(tb), bias=torch.tensor(bias, device='cuda'), precision=p)
        out[p] = np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert out[prec] < tol, out
    if prec == 'bf16x6':
        assert out[prec] < 2.0 * out['f32'] + 1e-7, out
    if prec == 'bf16':
        assert out[prec] > 1e-4                  # really the low-precision path


def test_gemm_bf16x6_split_k_and_segmented_k():
    """the weight-gradient shapes: deterministic split-K and the shifted h^T dz product"""
    from nabu_amd import ops
    rng = np.random.default_rng(5)
    Bn, T, H, G = 8, 64, 64, 256                 # K = Bn*(T-1) = 504 -> not a multiple of 32: exact-fp32 fallback
    x = rng.normal(size=(4096, 128)).astype(np.float32)
    dz = rng.normal(size=(4096, 256)).astype(np.float32)
    ref = x.astype(np.float64).T @ dz.astype(np.float64)
    c = torch.zeros(128, 256, device='cuda')
    ops.gemm(torch.tensor(x, device='cuda'), torch.tensor(dz, device='cuda'), c, True, False, precision='bf16x6')
    assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 3e-6
    c2 = torch.zeros(128, 256, device='cuda')
    ops.gemm(torch.tensor(x, device='cuda'), torch.tensor(dz, device='cuda'), c2, True, False, precision='bf16x6')
    assert torch.equal(c, c2)                    # split-K partials are reduced in a fixed order
    # default precision switch
    assert ops.get_gemm_precision() == 'f32'
    ops.set_gemm_precision('bf16x6')
    try:
        c3 = torch.zeros(128, 256, device='cuda')
        ops.gemm(torch.tensor(x, device='cuda'), torch.tensor(dz, device='cuda'), c3, True, False)
        assert torch.equal(c, c3)
    finally:
        ops.set_gemm_precision('f32')


@pytest.mark.parametrize('M,N,K', [(32, 2048, 1536), (64, 2048, 512), (5, 64, 64), (33, 96, 384), (32, 40, 512)])
def test_gemm_skinny_rows(M, N, K):
    """the decoder-step products (M <= 64 batch rows): skinny kernel + split-K reduce, incl. bias/beta;
    N = 40 is not a multiple of 32 and takes the tile kernel"""
    from nabu_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = 0.5 * (a.astype(np.float64) @ b.astype(np.float64)) + bias + 2.0 * c0
    c = torch.tensor(c0, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c, alpha=0.5, beta=2.0,
             bias=torch.tensor(bias, device='cuda'))
    assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 3e-6
    c2 = torch.tensor(c0, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c2, alpha=0.5, beta=2.0,
             bias=torch.tensor(bias, device='cuda'))
    assert torch.equal(c, c2)


@pytest.mark.parametrize('M,N,K1,K2', [(32, 2048, 1024, 512), (64, 2048, 1024, 512), (32, 512, 512, 0), (32, 1024, 2048, 0),
                                        (5, 64, 64, 128), (33, 96, 192, 64)])
def test_gemm2_two_segment_product_with_the_reduce_inside_the_launch(M, N, K1, K2):
    """nabu_gemm2_f32: [A | A2]·[B ; B2] + beta*C + bias (the Speller cell's concat([context, h])·kernel) in one
    launch with last-arriver split-K reduction: against float64, bitwise repeatable (the order of summation does
    not depend on which workgroup arrives last), and usable again at once (tickets handed back)"""
    from nabu_amd import ops
    rng = np.random.default_rng(M + N + K1 + K2)
    a = rng.normal(size=(M, K1)).astype(np.float32)
    b = rng.normal(size=(K1, N)).astype(np.float32)
    a2 = rng.normal(size=(M, K2)).astype(np.float32) if K2 else None
    b2 = rng.normal(size=(K2, N)).astype(np.float32) if K2 else None
    bias = rng.normal(size=N).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64) + bias + 2.0 * c0
    if K2:
        ref = ref + a2.astype(np.float64) @ b2.astype(np.float64)
    dev = lambda x: None if x is None else torch.tensor(x, device='cuda')
    outs = []
    for _ in range(3):
        c = dev(c0)
        ops.gemm2(dev(a), dev(b), dev(a2), dev(b2), c, beta=2.0, bias=dev(bias))
        outs.append(c)
    assert np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max() < 3e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with pytest.raises(Exception):          # K1 not a multiple of 64
        ops.gemm2(dev(a[:, :40]), dev(b[:40]), None, None, dev(c0))


@pytest.mark.parametrize('M,N,K', [(4000, 260, 40), (2500, 2048, 80), (32000, 128, 8), (2049, 132, 96)])
def test_gemm_small_k_tall_output(M, N, K):
    """the first layer's input projection shape class (K <= 96, M >= 2048): gemm_smallk_kernel — whole K in one load
    phase, 16-byte row stores through LDS — with bias, beta and edge tiles in both dimensions, against float64"""
    from nabu_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    c = torch.empty((M, N), device='cuda').fill_(float('nan'))         # beta = 0 must not read C
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c)
    assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-6
    c = torch.tensor(c0, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c, alpha=0.5, beta=2.0,
             bias=torch.tensor(bias, device='cuda'))
    ref2 = 0.5 * ref + bias + 2.0 * c0
    assert np.abs(c.cpu().numpy() - ref2).max() / np.abs(ref2).max() < 2e-6


@pytest.mark.parametrize('M,N,K', [(256, 256, 128), (16000, 2048, 2048), (2048, 2048, 12800), (130, 200, 64), (1000, 136, 4096)])
def test_gemm_bf16_resident_operands(M, N, K):
    """nabu_cvt_bf16 + nabu_gemm_bf16_nt (configs[4]'s bf16 input GEMMs with the operands converted once): the
    conversion is round-to-nearest-even (bit-identical with torch's), plain and transposed; the product equals
    the float64 product of the ROUNDED operands to fp32 accumulation accuracy, with bias / beta / split-K / edges"""
    from nabu_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = torch.tensor(rng.normal(size=(M, K)).astype(np.float32), device='cuda')
    b = torch.tensor(rng.normal(size=(N, K)).astype(np.float32), device='cuda')
    ab, bb = ops.cvt_bf16(a), ops.cvt_bf16(b)
    assert torch.equal(ab, a.to(torch.bfloat16)) and torch.equal(bb, b.to(torch.bfloat16))
    if M % 2 == 0:
        assert torch.equal(ops.cvt_bf16(a, transpose=True), a.to(torch.bfloat16).t().contiguous())
    ref = ab.double().cpu().numpy() @ bb.double().cpu().numpy().T
    c = torch.empty((M, N), device='cuda').fill_(float('nan'))
    ops.gemm_bf16_nt(ab, bb, c)
    assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 3e-6
    bias = torch.tensor(rng.normal(size=N).astype(np.float32), device='cuda')
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    c = torch.tensor(c0, device='cuda')
    ops.gemm_bf16_nt(ab, bb, c, alpha=0.5, beta=2.0, bias=bias)
    ref2 = 0.5 * ref + bias.cpu().numpy() + 2.0 * c0
    assert np.abs(c.cpu().numpy() - ref2).max() / np.abs(ref2).max() < 3e-6
    with pytest.raises(Exception):
        ops.gemm_bf16_nt(ab[:, :40].contiguous(), bb[:, :40].contiguous(), c)        # K % 64


def test_cross_entropy_losses_match_oracle():
    """average_cross_entropy (loss_functions.py:155-165) and sum_cross_entropy (:142-153) on ragged
    lengths: loss and the gradient the tape receives"""
    from nabu_amd.autodiff import Tape, SeqLen, record
    from nabu_amd.neuralnetworks.trainers import loss_functions
    rng = np.random.default_rng(8)
    B, L, C = 6, 9, 7
    logits = rng.normal(0, 2, (B, L, C))
    tl = np.array([9, 4, 1, 7, 9, 2], np.int32)
    ll = np.array([9, 4, 1, 6, 8, 2], np.int32)              # logit lengths may be shorter than targets
    tg = rng.integers(0, C, (B, L)).astype(np.int32)
    for name, want in (('average_cross_entropy', O.average_cross_entropy(logits, tg, ll, tl)),
                       ('sum_cross_entropy', O.sum_cross_entropy(logits, tg, tl))):
        src = dev(logits)
        lg = dev(logits)
        got = {}

        def capture(g):
            got['g'] = g
            return [None]
        with Tape() as tape:
            record([src], [lg], capture)             # makes the logits an interior node of the tape
            loss = loss_functions.factory(name)({'text': dev(tg, torch.int32)}, {'text': lg},
                                                {'text': SeqLen(ll, 'cuda')}, {'text': SeqLen(tl, 'cuda')})
        tape.backward(loss)
        assert abs(float(loss.item()) - want[0]) / abs(want[0]) < 1e-6, name
        assert rel_err(host(got['g']), want[1]) < 1e-5, name
    with pytest.raises(Exception, match='outside the MI355X hot path'):
        loss_functions.factory('marigin')
    with pytest.raises(Exception, match='unknown loss function'):
        loss_functions.factory('nope')


# ------------------------------------------------------------------ descriptor version 2 (include/nabu_hip.h)
def _layer_call(B, T, D, H, seed=0, **plan_kw):
    """one forward (+ optionally backward) call of a BLSTM layer on seeded operands; returns the tensors"""
    from nabu_amd import ops
    rng = np.random.default_rng(seed)
    x = dev(np.tanh(rng.normal(size=(B, T, D))))                # |x| <= 1: what a previous LSTM layer hands over
    p = {k: dev(v) for k, v in _blstm_params(rng, D, H).items()}
    lens = dev(np.full(B, T), torch.int32)
    plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_AUTO, **plan_kw)
    out = torch.full((B, T, 2 * H), float('nan'), device='cuda')
    reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    ops.blstm_fwd(plan, x, lens, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
    return plan, x, lens, p, out, reserve, dev(rng.normal(size=(B, T, 2 * H)))


def _backward(plan, x, lens, p, out, reserve, dout):
    from nabu_amd import ops
    dx = torch.full(x.shape, float('nan'), device='cuda')
    g = {k: torch.full(v.shape, float('nan'), device='cuda') for k, v in p.items()}
    ops.blstm_bwd(plan, x, lens, p['fw_kernel'], p['bw_kernel'], out, dout, reserve, dx,
                  g['fw_kernel'], g['fw_bias'], g['bw_kernel'], g['bw_bias'])
    torch.cuda.synchronize()
    return dx, g


def test_backward_rejects_a_reserve_of_another_layout():
    """nabu_blstm_fwd records (reserve pointer, layout) and the backward entry points check it (lstm.hip, tag_check):
    a reserve written under another gemm_precision / recurrent_precision / FWD_ONLY flag has its regions elsewhere —
    reading it would hand back numbers, not an error."""
    from nabu_amd import ops, _hip
    # (a shape the packed-operand products take: >= 1024 frames, D >= 256 — below that both plans have ONE layout)
    plan, x, lens, p, out, reserve, dout = _layer_call(8, 256, 256, 64, gemm_precision='f16x3')
    other = ops.BlstmPlan(8, 256, 256, 64, 256, ops.LSTM_AUTO, 'f32')
    assert other.reserve_bytes != plan.reserve_bytes
    big = torch.empty(max(other.reserve_bytes, plan.reserve_bytes), dtype=torch.uint8, device='cuda')
    big[:plan.reserve_bytes].copy_(reserve)                     # same bytes at an address no forward call wrote
    with pytest.raises(_hip.NabuHipError, match='reserve'):
        _backward(plan, x, lens, p, out, big, dout)
    with pytest.raises(_hip.NabuHipError, match='reserve'):     # right address, another layout
        _backward(other, x, lens, p, out, reserve, dout)
    dx, g = _backward(plan, x, lens, p, out, reserve, dout)     # the matching plan still runs after the refusals
    assert torch.isfinite(dx).all() and all(torch.isfinite(v).all() for v in g.values())


@pytest.mark.parametrize('B,T,D,H', [(32, 40, 40, 512), (8, 25, 1024, 256), (16, 80, 1024, 256), (5, 9, 16, 64)])
def test_forward_only_plan_has_the_same_outputs_in_a_smaller_reserve(B, T, D, H):
    """NABU_BLSTM_FWD_ONLY (validation, decoding): no dz / dzT / packed-operand regions in the reserve, bit-identical
    outputs, and the backward entry points refuse the plan."""
    from nabu_amd import _hip
    plan, x, lens, p, out, reserve, dout = _layer_call(B, T, D, H, seed=3, gemm_precision='f16x3')
    fplan, _, _, _, fout, freserve, _ = _layer_call(B, T, D, H, seed=3, gemm_precision='f16x3', fwd_only=True)
    assert torch.equal(out, fout)
    # (the packed dz^T region exists from 1024 frames on: below that the two reserves are the same activations)
    assert fplan.reserve_bytes < plan.reserve_bytes if B * T >= 1024 else fplan.reserve_bytes <= plan.reserve_bytes
    with pytest.raises(_hip.NabuHipError):
        _backward(fplan, x, lens, p, fout, freserve, dout)


def test_input_bound_replaces_the_measuring_pass():
    """x_bound: the caller's guarantee |x| <= bound (an LSTM layer's outputs: 1) sets the f16x3 row scale of x without
    reading x.  A bound is looser than the measured maximum by at most the ratio bound/max|x| — one binade here — so the
    outputs agree to the planes' 22 bits, and the gradients with them."""
    a = _layer_call(16, 30, 256, 128, seed=5, gemm_precision='f16x3')
    b = _layer_call(16, 30, 256, 128, seed=5, gemm_precision='f16x3', x_bound=1.0)
    assert (a[4] - b[4]).abs().max().item() < 2e-6
    da, ga = _backward(*a)
    db, gb = _backward(*b)
    assert rel_err(host(db), host(da)) < 1e-5
    for k in ga:
        assert rel_err(host(gb[k]), host(ga[k])) < 1e-5, k


def test_recurrent_precision_f32_selects_the_exact_kernels_per_call():
    """recurrent_precision = f32 in ONE call's descriptor: that call runs the exact-fp32 persistent kernels, the next
    default call the fp16-plane ones again (no process-wide switch is left behind)."""
    from nabu_amd import ops
    args = (32, 24, 40, 512)
    d0 = _layer_call(*args, seed=7)
    e = _layer_call(*args, seed=7, recurrent_precision='f32')
    d1 = _layer_call(*args, seed=7)
    ops.check_persist_status()
    assert torch.equal(d0[4], d1[4])
    diff = (d0[4] - e[4]).abs().max().item()
    assert 0 < diff < 2e-4                                       # different arithmetic, same function
