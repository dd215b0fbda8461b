"""GPU tests of the packed bf16-plane products (gemm_pk.hip, through the C ABI): the pack kernels are
bit-exact against a NumPy restatement of the split and the layout; the products are compared with float64
and with the exact-fp32 MFMA kernel on the same operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bf16_rne(x):
    """float32 array -> (uint16 bf16 bits, float32 value of the rounded number)"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return r.astype(np.uint16), (r << 16).astype(np.uint32).view(np.float32)


def split_planes(x, planes):
    out = []
    rest = np.ascontiguousarray(x, dtype=np.float32)
    for _ in range(planes):
        bits, val = bf16_rne(rest)
        out.append(bits)
        rest = (rest - val).astype(np.float32)
    return out, rest


def pack_ref(mat, planes, rows_pad, nkb):
    """mat [rows, K] float32 -> uint16 [nkb, planes, rows_pad, 16] in the layout of include/nabu_hip.h"""
    rows, K = mat.shape
    full = np.zeros((rows_pad, nkb * 16), np.float32)
    full[:rows, :K] = mat
    pl, rest = split_planes(full, planes)
    if planes == 3:
        assert not rest.any()                      # the three planes hold the fp32 value exactly
    out = np.zeros((nkb, planes, rows_pad, 16), np.uint16)
    swap = ((np.arange(rows_pad) >> 3) & 1).astype(bool)
    for p in range(planes):
        blk = pl[p].reshape(rows_pad, nkb, 2, 8).copy()
        blk[swap] = blk[swap][:, :, ::-1, :]
        out[:, p] = blk.reshape(rows_pad, nkb, 16).transpose(1, 0, 2)
    return out


@pytest.mark.parametrize('planes', [3, 1])
@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('rows,K', [(256, 64), (300, 1000), (72, 40), (515, 130)])
def test_pack_bit_exact(planes, transposed, rows, K):
    from nabu_amd import ops
    rng = np.random.default_rng(rows + K + planes)
    mat = (rng.normal(size=(rows, K)) * np.exp(3 * rng.normal(size=(rows, 1)))).astype(np.float32)
    mat[0, 0] = 0.0
    mat[1, 1] = np.float32(1.0) + np.float32(2.0) ** -23        # needs all three planes
    src = np.ascontiguousarray(mat.T if transposed else mat)
    pad = 4 - src.shape[1] % 4 if src.shape[1] % 4 else 0       # source rows 16-byte aligned
    srcp = np.zeros((src.shape[0], src.shape[1] + pad), np.float32)
    srcp[:, :src.shape[1]] = src
    sd = torch.tensor(srcp, device='cuda')
    dst = ops.PackedOperand(rows, K, planes, 'cuda')
    dst.buf.fill_(0x5A)
    ops.pk_pack(dst, sd, transposed, R=src.shape[0], C=src.shape[1], ld=srcp.shape[1])
    got = dst.buf.cpu().numpy().view(np.uint16).reshape(dst.nkb, planes, dst.rows_pad, 16)
    ref = pack_ref(mat, planes, dst.rows_pad, dst.nkb)
    assert np.array_equal(got, ref)


def test_pack_shifted_pairs():
    """transposed pack with (period, shift): the h_{t-1}^T dz_t pairing of the recurrent weight gradient"""
    from nabu_amd import ops
    rng = np.random.default_rng(3)
    Bn, T, H = 3, 21, 40
    out = rng.normal(size=(Bn * T, 2 * H)).astype(np.float32)
    od = torch.tensor(out, device='cuda')
    for d, shift in ((0, -1), (1, 1)):
        dst = ops.PackedOperand(H, Bn * T, 3, 'cuda')
        src = od[:, d * H:(d + 1) * H]
        ops.pk_pack(dst, src, True, period=T, shift=shift, R=Bn * T, C=H, ld=2 * H)
        o3 = out.reshape(Bn, T, 2 * H)[:, :, d * H:(d + 1) * H]
        sh = np.zeros_like(o3)
        if shift < 0:
            sh[:, 1:] = o3[:, :-1]
        else:
            sh[:, :-1] = o3[:, 1:]
        ref = pack_ref(np.ascontiguousarray(sh.reshape(Bn * T, H).T), 3, dst.rows_pad, dst.nkb)
        got = dst.buf.cpu().numpy().view(np.uint16).reshape(dst.nkb, 3, dst.rows_pad, 16)
        assert np.array_equal(got, ref)


def _product(a, b, planes, bias=None, beta=0.0, c0=None, alpha=1.0):
    from nabu_amd import ops
    M, K = a.shape
    N = b.shape[0]
    pa, pb = ops.PackedOperand(M, K, planes, 'cuda'), ops.PackedOperand(N, K, planes, 'cuda')
    ops.pk_pack(pa, torch.tensor(a, device='cuda'))
    ops.pk_pack(pb, torch.tensor(b, device='cuda'))
    c = torch.tensor(c0, device='cuda') if c0 is not None else torch.full((M, N), 7.0, device='cuda')
    ops.gemm_pk(pa, pb, c, planes, alpha=alpha, beta=beta, bias=torch.tensor(bias, device='cuda') if bias is not None else None)
    return c.cpu().numpy()


@pytest.mark.parametrize('M,N,K', [(256, 256, 48), (256, 256, 16), (300, 260, 1000), (40, 2048, 2048), (1027, 516, 144),
                                   (512, 256, 4096)])
def test_gemm_pk_x6_matches_float64(M, N, K):
    """asymmetric operands, edge tiles in M and N, K not a multiple of 16; split-K where the policy wants it"""
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(N, K)).astype(np.float32)
    a[:, ::3] *= 1e-3
    bias = rng.normal(size=N).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = 0.5 * a.astype(np.float64) @ b.astype(np.float64).T + 2.0 * c0 + bias
    got = _product(a, b, 3, bias=bias, beta=2.0, c0=c0, alpha=0.5)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 4e-6    # fp32 accumulation over K


def test_gemm_pk_bf16_plain():
    """planes = 1 is the RNE-bf16 product of BASELINE configs[4]: equal to the float64 product of the rounded operands"""
    rng = np.random.default_rng(11)
    M, N, K = 520, 300, 700
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(N, K)).astype(np.float32)
    ar, br = bf16_rne(a)[1].astype(np.float64), bf16_rne(b)[1].astype(np.float64)
    got = _product(a, b, 1)
    ref = ar @ br.T
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6
    full = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.abs(got - full).max() / np.abs(full).max() > 1e-4      # really the low-precision product


def test_gemm_pk_two_destinations_batch_and_plane_reuse():
    """the layer's use: one product filling both directions' buffers (n_split), two products in one launch
    (batch), a 3-plane operand used by the 1-plane product"""
    from nabu_amd import ops
    rng = np.random.default_rng(5)
    M, K, H4 = 700, 336, 512                                   # 21 k-blocks: a multiple of 3 for the 1-plane product
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = rng.normal(size=(2 * H4, K)).astype(np.float32)        # rows: forward cell's 4H columns, then backward's
    b1, b2 = rng.normal(size=H4).astype(np.float32), rng.normal(size=H4).astype(np.float32)
    px, pw = ops.PackedOperand(M, K, 3, 'cuda'), ops.PackedOperand(2 * H4, K, 3, 'cuda')
    ops.pk_pack(px, torch.tensor(x, device='cuda'))
    ops.pk_pack(pw, torch.tensor(w[:H4], device='cuda'), row_off=0, fill_rows=H4)
    ops.pk_pack(pw, torch.tensor(w[H4:], device='cuda'), row_off=H4)
    g1, g2 = torch.zeros(M, H4, device='cuda'), torch.zeros(M, H4, device='cuda')
    ops.gemm_pk(px, pw, g1, 3, bias=torch.tensor(b1, device='cuda'), c2=g2, n_split=H4, bias2=torch.tensor(b2, device='cuda'))
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    tol = 2e-6 * np.abs(ref).max()
    assert np.abs(g1.cpu().numpy() - (ref[:, :H4] + b1)).max() < tol
    assert np.abs(g2.cpu().numpy() - (ref[:, H4:] + b2)).max() < tol
    # plane 0 of the 3-plane operands as a bf16 product
    h1, h2 = torch.zeros(M, H4, device='cuda'), torch.zeros(M, H4, device='cuda')
    ops.gemm_pk(px, pw, h1, 1, c2=h2, n_split=H4, nkb=px.nkb)
    rr = bf16_rne(x)[1].astype(np.float64) @ bf16_rne(w)[1].astype(np.float64).T
    assert np.abs(h1.cpu().numpy() - rr[:, :H4]).max() < tol and np.abs(h2.cpu().numpy() - rr[:, H4:]).max() < tol
    # batch of two products with their own operands
    a2 = rng.normal(size=(2, 300, K)).astype(np.float32)
    pa = [ops.PackedOperand(300, K, 3, 'cuda') for _ in range(2)]
    for i in range(2):
        ops.pk_pack(pa[i], torch.tensor(a2[i], device='cuda'))
    cs = [torch.zeros(300, H4, device='cuda') for _ in range(2)]
    ops.gemm_pk(pa[0], pw, None, 3, M=300, N=H4, a_ptrs=[p.buf.data_ptr() for p in pa],
                b_ptrs=[pw.row_ptr(0), pw.row_ptr(H4)], cs=cs)
    for i in range(2):
        r = a2[i].astype(np.float64) @ w[i * H4:(i + 1) * H4].astype(np.float64).T
        assert np.abs(cs[i].cpu().numpy() - r).max() < tol


@pytest.mark.parametrize('K', [2048, 16000, 32000])
def test_gemm_pk_x6_error_not_above_exact_fp32(K):
    """error against float64 at the real reduction lengths of cfg2, bf16x6 on packed
    operands <= 1.0 x the exact-fp32 MFMA kernel's (both accumulate in fp32; the six-plane product drops
    terms below 2^-26 |a||b| only)."""
    from nabu_amd import ops
    rng = np.random.default_rng(K)
    M, N = 512, 512
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(N, K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    got = _product(a, b, 3)
    c = torch.zeros(M, N, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c, False, True, precision='f32')
    e6 = np.abs(got - ref)
    e32 = np.abs(c.cpu().numpy() - ref)
    print('\nK=%d  bf16x6 max %.3e rms %.3e | fp32 max %.3e rms %.3e' % (K, e6.max(), np.sqrt((e6 ** 2).mean()),
                                                                      e32.max(), np.sqrt((e32 ** 2).mean())))
    assert np.sqrt((e6 ** 2).mean()) <= 1.0 * np.sqrt((e32 ** 2).mean())
    assert e6.max() <= 1.0 * e32.max()


# ---------------------------------------------------------------------------------------------------------------
# f16x3: two scaled fp16 planes, three plane products

def f16_scale_exp(amax):
    """exponent field -> (scale, inverse) as gemm_pk.hip derives them from the row maximum's bit pattern"""
    bits = np.ascontiguousarray(amax, dtype=np.float32).view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int64)
    e = np.where((bits & 0x7FFFFFFF) == 0, 127, e)
    e = np.clip(e, 15, 253)
    return np.exp2(141.0 - e).astype(np.float32), np.exp2(e - 141.0).astype(np.float32)


def pack_ref_f16(mat, amax, rows_pad, nkb):
    """mat [rows, K] float32, amax [rows] -> uint16 [nkb, 2, rows_pad, 16]"""
    rows, K = mat.shape
    full = np.zeros((rows_pad, nkb * 16), np.float32)
    full[:rows, :K] = mat
    am = np.zeros(rows_pad, np.float32)
    am[:rows] = amax
    scale = f16_scale_exp(am)[0]
    xs = (full * scale[:, None]).astype(np.float32)
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)
    out = np.zeros((nkb, 2, rows_pad, 16), np.uint16)
    swap = ((np.arange(rows_pad) >> 3) & 1).astype(bool)
    for p, pl in enumerate((h, l)):
        blk = pl.view(np.uint16).reshape(rows_pad, nkb, 2, 8).copy()
        blk[swap] = blk[swap][:, :, ::-1, :]
        out[:, p] = blk.reshape(rows_pad, nkb, 16).transpose(1, 0, 2)
    return out, h, l, scale


@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('rows,K', [(256, 64), (300, 1000), (72, 40), (515, 130)])
def test_pack_f16_bit_exact_and_row_maxima(transposed, rows, K):
    from nabu_amd import ops
    rng = np.random.default_rng(rows + K)
    mat = (rng.normal(size=(rows, K)) * np.exp(4 * rng.normal(size=(rows, 1))) * np.exp(2 * rng.normal(size=(rows, K)))
           ).astype(np.float32)
    mat[3] = 0.0                                               # an all-zero row
    mat[5, :] = np.float32(2.0) ** -130                        # a row of fp32 subnormals
    src = np.ascontiguousarray(mat.T if transposed else mat)
    pad = 4 - src.shape[1] % 4 if src.shape[1] % 4 else 0
    srcp = np.zeros((src.shape[0], src.shape[1] + pad), np.float32)
    srcp[:, :src.shape[1]] = src
    dst = ops.PackedOperand(rows, K, 2, 'cuda')
    dst.buf.fill_(0x5A)
    ops.pk_pack(dst, torch.tensor(srcp, device='cuda'), transposed, R=src.shape[0], C=src.shape[1], ld=srcp.shape[1])
    amax = dst.amax.cpu().numpy().view(np.float32)
    assert np.array_equal(amax[:rows], np.abs(mat).max(axis=1)) and not amax[rows:].any()
    got = dst.buf.cpu().numpy().view(np.uint16).reshape(dst.nkb, 2, dst.rows_pad, 16)
    ref, h, l, scale = pack_ref_f16(mat, amax[:rows], dst.rows_pad, dst.nkb)
    assert np.array_equal(got, ref)
    # what the two planes hold: the scaled value to 2^-23 relative (11 + 11 bits and the sign of l) or 2^-25 absolute
    # (half the spacing of fp16 subnormals)
    xs = np.zeros((dst.rows_pad, dst.nkb * 16), np.float64)
    xs[:rows, :K] = mat.astype(np.float64) * scale[:rows, None]
    err = np.abs(h.astype(np.float64) + l.astype(np.float64) - xs)
    assert (err <= np.maximum(2.0 ** -23 * np.abs(xs), 2.0 ** -25)).all()
    assert np.abs(xs).max() < 2.0 ** 15


def _product_f16(a, b, bound_a=None, bound_b=None, direct=False, **kw):
    from nabu_amd import ops
    M, K = a.shape
    N = b.shape[0]
    pa, pb = ops.PackedOperand(M, K, 2, 'cuda'), ops.PackedOperand(N, K, 2, 'cuda')
    ops.pk_pack(pa, torch.tensor(a, device='cuda'), bound=bound_a)
    ops.pk_pack(pb, torch.tensor(b, device='cuda'), bound=bound_b)
    c = torch.tensor(kw.pop('c0'), device='cuda') if kw.get('c0') is not None else torch.full((M, N), 7.0, device='cuda')
    kw.pop('c0', None)
    bias = kw.pop('bias', None)
    ops.gemm_pk(pa, pb, c, 2, bias=torch.tensor(bias, device='cuda') if bias is not None else None, direct=direct, **kw)
    return c.cpu().numpy()


@pytest.mark.parametrize('M,N,K', [(256, 256, 48), (256, 256, 16), (300, 260, 1000), (40, 2048, 2048), (1027, 516, 144),
                                   (512, 256, 4096)])
def test_gemm_pk_f16x3_matches_float64(M, N, K):
    """rows of very different magnitude (each has its own scale), edge tiles, K not a multiple of 16, split-K"""
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    a = (rng.normal(size=(M, K)) * np.exp(5 * rng.normal(size=(M, 1)))).astype(np.float32)
    b = (rng.normal(size=(N, K)) * np.exp(5 * rng.normal(size=(N, 1)))).astype(np.float32)
    a[:, ::3] *= 1e-3
    a[7] = 0.0
    bias = rng.normal(size=N).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    prod = a.astype(np.float64) @ b.astype(np.float64).T
    ref = 0.5 * prod + 2.0 * c0 + bias
    got = _product_f16(a, b, bias=bias, beta=2.0, c0=c0, alpha=0.5)
    # per element: relative to the magnitude the row and column scales give the product
    mag = np.sqrt((a.astype(np.float64) ** 2).sum(1))[:, None] * np.sqrt((b.astype(np.float64) ** 2).sum(1))[None, :]
    assert (np.abs(got - ref) <= 4e-6 * (mag + np.abs(2.0 * c0) + np.abs(bias)[None, :]) + 1e-30).all()


def test_gemm_pk_f16x3_two_destinations_and_batch():
    from nabu_amd import ops
    rng = np.random.default_rng(6)
    M, K, H4 = 700, 336, 512
    x = rng.uniform(-1, 1, size=(M, K)).astype(np.float32)          # bounded like LSTM outputs: a-priori bound 1
    w = (0.1 * rng.normal(size=(2 * H4, K))).astype(np.float32)
    b1, b2 = rng.normal(size=H4).astype(np.float32), rng.normal(size=H4).astype(np.float32)
    px, pw = ops.PackedOperand(M, K, 2, 'cuda'), ops.PackedOperand(2 * H4, K, 2, 'cuda')
    ops.pk_pack(px, torch.tensor(x, device='cuda'), bound=1.0)
    ops.pk_pack(pw, torch.tensor(w[:H4], device='cuda'), row_off=0, fill_rows=H4)
    ops.pk_pack(pw, torch.tensor(w[H4:], device='cuda'), row_off=H4)
    g1, g2 = torch.zeros(M, H4, device='cuda'), torch.zeros(M, H4, device='cuda')
    ops.gemm_pk(px, pw, g1, 2, bias=torch.tensor(b1, device='cuda'), c2=g2, n_split=H4, bias2=torch.tensor(b2, device='cuda'))
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    tol = 2e-6 * np.abs(ref).max()
    assert np.abs(g1.cpu().numpy() - (ref[:, :H4] + b1)).max() < tol
    assert np.abs(g2.cpu().numpy() - (ref[:, H4:] + b2)).max() < tol
    a2 = rng.normal(size=(2, 300, K)).astype(np.float32)
    pa = [ops.PackedOperand(300, K, 2, 'cuda') for _ in range(2)]
    for i in range(2):
        ops.pk_pack(pa[i], torch.tensor(a2[i], device='cuda'))
    cs = [torch.zeros(300, H4, device='cuda') for _ in range(2)]
    ops.gemm_pk(pa[0], pw, None, 2, M=300, N=H4, a_ptrs=[p.buf.data_ptr() for p in pa],
                b_ptrs=[pw.row_ptr(0), pw.row_ptr(H4)], cs=cs, a_amax=[p.amax.data_ptr() for p in pa],
                b_amax=[pw.amax.data_ptr(), pw.amax.data_ptr() + 4 * H4])
    for i in range(2):
        r = a2[i].astype(np.float64) @ w[i * H4:(i + 1) * H4].astype(np.float64).T
        assert np.abs(cs[i].cpu().numpy() - r).max() < tol


@pytest.mark.parametrize('sigma', [0.0, 3.0])
@pytest.mark.parametrize('K', [2048, 16000, 32000])
def test_gemm_pk_f16x3_error_not_above_exact_fp32(K, sigma):
    """the bar of bf16x6 for the three-product arithmetic: error against float64 <= 1.0 x the exact-fp32 MFMA
    kernel's at the reduction lengths of cfg2 on normal data.  On heavy-tailed data (elements spread over
    e^(+-3 sigma) inside every row: a few terms dominate every sum, small elements live in the fp16 planes'
    subnormal range) the two planes' 2^-23 representation error shows: <= 1.25 x."""
    from nabu_amd import ops
    rng = np.random.default_rng(K)
    M, N = 512, 512
    a = (rng.normal(size=(M, K)) * np.exp(sigma * rng.normal(size=(M, K)))).astype(np.float32)
    b = (rng.normal(size=(N, K)) * np.exp(sigma * rng.normal(size=(N, K)))).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    got = _product_f16(a, b)
    c = torch.zeros(M, N, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c, False, True, precision='f32')
    e3 = np.abs(got - ref)
    e32 = np.abs(c.cpu().numpy() - ref)
    print('\nK=%d sigma=%g  f16x3 max %.3e rms %.3e | fp32 max %.3e rms %.3e' % (
        K, sigma, e3.max(), np.sqrt((e3 ** 2).mean()), e32.max(), np.sqrt((e32 ** 2).mean())))
    bar = 1.0 if sigma == 0.0 else 1.25
    assert np.sqrt((e3 ** 2).mean()) <= bar * np.sqrt((e32 ** 2).mean())
    assert e3.max() <= (1.0 if sigma == 0.0 else 1.5) * e32.max()


@pytest.mark.parametrize('M,N,K', [(2048, 4096, 16000), (2048, 4096, 4000), (552, 2048, 32000), (512, 2048, 8000),
                                   (16000, 4096, 2048), (4000, 4096, 2048), (8000, 2048, 8192)])
def test_gemm_pk_f16x3_direct_chain_at_the_layer_shapes(M, N, K):
    """the BLSTM layer asks for direct = 2: the three plane products chained directly into the accumulators where that
    rounds less often than the exact-fp32 kernel would.  At the shapes of cfg2's weight-gradient, forward and
    input-gradient products the rule picks the direct chain and the error against float64 stays <= 1.0 x the
    exact-fp32 MFMA kernel's; at 512 x 512 x 2048 (the fp32 kernel splits K there, this one barely) it picks the
    promoted sums"""
    from nabu_amd import ops
    rng = np.random.default_rng(M + K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    b = rng.normal(size=(N, K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    got = _product_f16(a, b, direct=1)
    assert np.array_equal(got, _product_f16(a, b, direct=2))          # the rule takes the direct chain here
    c = torch.zeros(M, N, device='cuda')
    ops.gemm(torch.tensor(a, device='cuda'), torch.tensor(b, device='cuda'), c, False, True, precision='f32')
    e3, e32 = got - ref, c.cpu().numpy() - ref
    r = np.sqrt((e3 ** 2).mean()) / np.sqrt((e32 ** 2).mean())
    print('\n%d x %d x %d  direct f16x3 / exact fp32: rms %.3f max %.3f' % (M, N, K, r, np.abs(e3).max() / np.abs(e32).max()))
    assert r <= 1.0 and np.abs(e3).max() <= 1.0 * np.abs(e32).max()


def test_gemm_pk_f16x3_direct_rule_keeps_promoted_sums_where_fp32_splits_finer():
    rng = np.random.default_rng(8)
    a = rng.normal(size=(512, 2048)).astype(np.float32)
    b = rng.normal(size=(512, 2048)).astype(np.float32)
    auto, promoted, direct = (_product_f16(a, b, direct=m) for m in (2, 0, 1))
    assert np.array_equal(auto, promoted) and not np.array_equal(auto, direct)


def test_gemm_pk_f16x3_non_finite_inputs_stay_visible():
    rng = np.random.default_rng(2)
    a = rng.normal(size=(256, 64)).astype(np.float32)
    b = rng.normal(size=(256, 64)).astype(np.float32)
    a[10, 3] = np.inf
    a[20, 5] = np.nan
    got = _product_f16(a, b)
    assert not np.isfinite(got[10]).any() and np.isnan(got[20]).all()
    ok = np.ones(256, bool)
    ok[[10, 20]] = False
    ref = a[ok].astype(np.float64) @ b.astype(np.float64).T
    assert np.abs(got[ok] - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('precision,tol', [('bf16x6', 2e-4), ('f16x3', 2e-4), ('bf16', 5e-2)])
@pytest.mark.parametrize('B,T,D,H,lens', [
    (32, 64, 256, 64, None),                                       # every product on the packed path
    (17, 130, 260, 128, [130, 7, 99, 130, 1, 64, 65, 129, 30, 130, 2, 77, 128, 13, 100, 55, 130]),   # ragged, edge tiles,
                                                                   # B*T and D not multiples of 16
    (16, 160, 40, 64, None),                                       # narrow input: only the recurrent gradient is packed
])   # (B*T >= 2048: below that the layer computes f16x3 requests as bf16x6)
def test_blstm_layer_on_packed_products_matches_oracle(precision, tol, B, T, D, H, lens):
    """nabu_blstm_fwd/_bwd with gemm_precision = bf16x6 / f16x3 (fp32-equivalent, same bounds as the exact-fp32 layer
    tests) and bf16 (operand rounding ~2^-9) against the float64 oracle; zero rows past each length stay exact"""
    from nabu_amd import ops
    from tests.test_hip_ops import _run_blstm, rel_err
    lens = lens or [T] * B
    out, rout, dx, rdx, g, rg = _run_blstm(B, T, D, H, lens, ops.LSTM_AUTO, seed=B + T, precision=precision)
    assert np.isfinite(out).all()
    assert np.abs(out - rout).max() < (2e-5 if precision in ('bf16x6', 'f16x3') else 6e-2)
    for b, n in enumerate(lens):
        assert np.all(out[b, n:] == 0)
    assert rel_err(dx, rdx) < tol
    for k in rg:
        assert rel_err(g[k], rg[k]) < tol, k


def test_blstm_layer_f16x3_is_its_own_arithmetic_above_2048_frames():
    """a layer asked for f16x3 runs the three-product kernel from 2048 frames on (its results differ from bf16x6's in
    the last bits) and bf16x6 below (bit-identical results)"""
    from nabu_amd import ops
    from tests.test_hip_ops import _run_blstm
    for B, T, same in ((32, 64, False), (16, 64, True)):
        a = _run_blstm(B, T, 256, 64, [T] * B, ops.LSTM_AUTO, seed=5, precision='f16x3')
        b = _run_blstm(B, T, 256, 64, [T] * B, ops.LSTM_AUTO, seed=5, precision='bf16x6')
        assert np.array_equal(a[0], b[0]) == same and np.array_equal(a[2], b[2]) == same, (B, T)


@pytest.mark.parametrize('precision', ['f32', 'bf16x6', 'f16x3', 'bf16'])
@pytest.mark.parametrize('B,T,D,H', [(32, 64, 256, 64), (16, 160, 40, 64)])
def test_blstm_backward_in_two_calls_equals_one(precision, B, T, D, H):
    """nabu_blstm_bwd_data + nabu_blstm_bwd_weights (the weight-gradient products deferred behind the last recurrence of
    a backward pass) == nabu_blstm_bwd, bit for bit, on every arithmetic of the products"""
    from nabu_amd import ops
    rng = np.random.default_rng(B + T)
    lens = rng.integers(T // 2, T + 1, B).astype(np.int32)
    lens[0] = T
    x = torch.tensor(rng.normal(size=(B, T, D)).astype(np.float32), device='cuda')
    for b in range(B):
        x[b, lens[b]:] = 0
    ld = torch.tensor(lens, device='cuda')
    p = [torch.tensor(rng.normal(0, 0.2, s).astype(np.float32), device='cuda') for s in [(D + H, 4 * H), (4 * H,), (D + H, 4 * H), (4 * H,)]]
    dout = torch.tensor(rng.normal(size=(B, T, 2 * H)).astype(np.float32), device='cuda')
    plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_AUTO, precision)
    res = []
    for split in (False, True):
        out = torch.zeros(B, T, 2 * H, device='cuda')
        reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
        ops.blstm_fwd(plan, x, ld, p[0], p[1], p[2], p[3], out, reserve)
        g = [torch.full_like(q, float('nan')) for q in p]
        dx = torch.full_like(x, float('nan'))
        if split:
            ops.blstm_bwd_data(plan, x, ld, p[0], p[2], out, dout, reserve, dx, g[1], g[3])
            junk = torch.randn(1 << 20, device='cuda')          # other work in between
            junk.mul_(2.0)
            ops.blstm_bwd_weights(plan, x, ld, out, reserve, g[0], g[2])
        else:
            ops.blstm_bwd(plan, x, ld, p[0], p[2], out, dout, reserve, dx, g[0], g[1], g[2], g[3])
        torch.cuda.synchronize()
        res.append([dx] + g)
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and torch.equal(a, b)
