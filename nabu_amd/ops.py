"""Typed Python wrappers over the C ABI (one function per entry point of
include/nabu_hip.h).  Tensors are torch CUDA tensors used as device-memory
handles; all arithmetic happens inside libnabu_hip.so."""
import ctypes

import torch

from . import _hip
from ._hip import ptr, stream, check, Workspace

LSTM_AUTO, LSTM_STEPWISE, LSTM_PERSISTENT = 0, 1, 2


def _f32(t, name):
    if t.dtype != torch.float32:
        raise _hip.NabuHipError('%s must be float32, got %s' % (name, t.dtype))
    return t


def gemm(a, b, c, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, bias=None,
         M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
         kseg=0, a_seg=0, b_seg=0, precision='default'):
    """c = alpha*op(a)@op(b) + beta*c + bias on 2-D row-major tensors (or raw
    views when M/N/K/ld* are given explicitly).  precision: 'default' | 'f32' | 'bf16' |
    'bf16x3' | 'bf16x6' (include/nabu_hip.h, nabu_gemm_ex)."""
    L = _hip.lib()
    if M is None:
        M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
        N = b.shape[0] if trans_b else b.shape[1]
        lda, ldb, ldc = a.stride(0), b.stride(0), c.stride(0)
    ws_bytes = L.nabu_gemm_ws_bytes(M, N, K)
    ws = Workspace.get(ws_bytes, c.device, 'gemm') if ws_bytes else None
    check(L.nabu_gemm_ex(_hip.GEMM_PRECISIONS[precision], int(trans_a), int(trans_b), M, N, K, alpha,
                         a.data_ptr(), lda, b.data_ptr(), ldb, beta, c.data_ptr(), ldc,
                         bias.data_ptr() if bias is not None else None, kseg, a_seg, b_seg,
                         ptr(ws), ws_bytes, stream()), 'nabu_gemm_ex')
    return c


def gemm2(a, b, a2, b2, c, beta=0.0, bias=None):
    """c = [a | a2] @ [b ; b2] + beta*c + bias for M <= 64 rows in one launch (nabu_gemm2_f32); a2/b2 may be None"""
    L = _hip.lib()
    M, K1 = a.shape
    N = b.shape[1]
    K2 = a2.shape[1] if a2 is not None else 0
    ws_bytes = L.nabu_gemm2_ws_bytes(M, N, K1, K2)
    ws = Workspace.get(ws_bytes, c.device, 'gemm2')
    check(L.nabu_gemm2_f32(M, N, K1, ptr(a), a.stride(0), ptr(b), b.stride(0), K2, ptr(a2),
                           a2.stride(0) if a2 is not None else 0, ptr(b2), b2.stride(0) if b2 is not None else 0,
                           beta, ptr(c), c.stride(0), ptr(bias), ptr(ws), ws_bytes, stream()), 'nabu_gemm2_f32')
    return c


def cvt_bf16(src, transpose=False):
    """bf16 copy (RNE) of a 2-D fp32 tensor, optionally transposed; returned as a torch.bfloat16 tensor"""
    R, C = src.shape
    out = torch.empty((C, R) if transpose else (R, C), dtype=torch.bfloat16, device=src.device)
    check(_hip.lib().nabu_cvt_bf16(R, C, ptr(src), src.stride(0), ptr(out), out.stride(0), int(transpose), stream()),
          'nabu_cvt_bf16')
    return out


def gemm_bf16_nt(a_bf16, b_bf16, c, alpha=1.0, beta=0.0, bias=None):
    """c (fp32) = alpha * a_bf16 @ b_bf16.T + beta*c + bias for bf16 operands with k contiguous"""
    L = _hip.lib()
    M, K = a_bf16.shape
    N = b_bf16.shape[0]
    ws_bytes = L.nabu_gemm_bf16_nt_ws_bytes(M, N, K)
    ws = Workspace.get(ws_bytes, c.device, 'gemm') if ws_bytes else None
    check(L.nabu_gemm_bf16_nt(M, N, K, alpha, ptr(a_bf16), a_bf16.stride(0), ptr(b_bf16), b_bf16.stride(0), beta,
                              ptr(c), c.stride(0), ptr(bias), ptr(ws), ws_bytes, stream()), 'nabu_gemm_bf16_nt')
    return c


class PackedOperand(object):
    """An fp32 matrix converted to the packed plane layout of gemm_pk.hip (include/nabu_hip.h, nabu_pk_pack):
    `rows` = the operand's M (or N) index, `K` = the reduction length.  planes = 3 / 1: bf16 planes; planes = 2:
    the scaled fp16 planes of f16x3, with `amax` = the bit patterns of the packed rows' largest magnitudes
    (pk_pack measures them unless told a bound)."""

    def __init__(self, rows, K, planes, device):
        L = _hip.lib()
        self.rows, self.K, self.planes = rows, K, planes
        self.rows_pad = L.nabu_pk_rows_pad(rows)
        self.nkb = L.nabu_pk_kblocks(K, planes)
        self.buf = torch.empty(L.nabu_pk_bytes(rows, K, planes), dtype=torch.uint8, device=device)
        self.amax = torch.zeros(self.rows_pad, dtype=torch.int32, device=device) if planes == 2 else None

    def kb_ptr(self, kb):
        return self.buf.data_ptr() + kb * self.planes * self.rows_pad * 32

    def row_ptr(self, row):
        return self.buf.data_ptr() + row * 32


def pk_amax(src, rows=None, cols=None, R=None, C=None, ld=None):
    """atomic maxima (bit patterns of |x|) of the rows / columns of the 2-D fp32 tensor `src` into the int32 device
    arrays `rows` [>= R] / `cols` [>= C], which the caller zeroed or seeded (nabu_pk_amax)"""
    if R is None:
        R, C = src.shape
        ld = src.stride(0)
    check(_hip.lib().nabu_pk_amax(src.data_ptr(), ld, R, C, ptr(rows), ptr(cols), stream()), 'nabu_pk_amax')


def pk_pack(dst, src, transposed=False, row_off=0, kb_off=0, fill_rows=None, fill_kb=None, period=0, shift=0,
            R=None, C=None, ld=None, bound=None, measure=True):
    """write the 2-D fp32 tensor `src` into the packed operand `dst` (rows / k-blocks beyond the source are zero).
    f16x3 operands (dst.planes == 2): the row maxima of the packed rows are measured first (`measure`; they
    accumulate when several sources share packed rows along k) or set to the a-priori `bound`."""
    if R is None:
        R, C = src.shape
        ld = src.stride(0)
    rows, kred = (C, R) if transposed else (R, C)
    if fill_rows is None:
        fill_rows = dst.rows_pad - row_off if row_off + rows >= dst.rows else rows
    if fill_kb is None:
        fill_kb = dst.nkb - kb_off if kb_off + (kred + 15) // 16 >= (dst.K + 15) // 16 else (kred + 15) // 16
    if dst.planes == 2:
        L = _hip.lib()
        am = dst.amax[row_off:]
        if bound is not None:
            check(L.nabu_pk_amax_fill(ptr(am), rows, float(bound), stream()), 'nabu_pk_amax_fill')
        elif measure:
            if kb_off == 0:        # a fresh measurement of these rows (the maxima accumulate only along k: kb_off > 0)
                am[:rows].zero_()
            check(L.nabu_pk_amax(src.data_ptr(), ld, R, C, None if transposed else ptr(am), ptr(am) if transposed else None,
                                 stream()), 'nabu_pk_amax')
        check(L.nabu_pk_pack_f16(int(transposed), src.data_ptr(), ld, R, C, ptr(dst.buf), dst.rows_pad, row_off, kb_off,
                                 fill_rows, fill_kb, period, shift, ptr(dst.amax), stream()), 'nabu_pk_pack_f16')
        return dst
    check(_hip.lib().nabu_pk_pack(dst.planes, int(transposed), src.data_ptr(), ld, R, C, ptr(dst.buf), dst.rows_pad,
                                  row_off, kb_off, fill_rows, fill_kb, period, shift, stream()), 'nabu_pk_pack')
    return dst


def gemm_pk(a, b, c, planes=None, alpha=1.0, beta=0.0, bias=None, c2=None, n_split=0, bias2=None, M=None, N=None,
            nkb=None, a_ptrs=None, b_ptrs=None, cs=None, c2s=None, a_amax=None, b_amax=None, direct=False):
    """c[M,N] = alpha * a·b^T + beta*c + bias over packed operands (nabu_gemm_pk); batched form through
    a_ptrs / b_ptrs / cs (lists of raw pointers / tensors, <= 2 entries)"""
    L = _hip.lib()
    planes = planes or min(a.planes, b.planes)
    d = _hip.PkGemmDesc()
    d.size = ctypes.sizeof(_hip.PkGemmDesc)
    d.planes = planes
    d.M = a.rows if M is None else M
    d.N = b.rows if N is None else N
    d.nkb = L.nabu_pk_kblocks(a.K, planes) if nkb is None else nkb
    cs = cs or [c]
    d.nbatch = len(cs)
    for i in range(d.nbatch):
        d.A[i] = a_ptrs[i] if a_ptrs else a.buf.data_ptr()
        d.B[i] = b_ptrs[i] if b_ptrs else b.buf.data_ptr()
        d.C[i] = ptr(cs[i])
        d.C2[i] = ptr(c2s[i]) if c2s else ptr(c2)
    d.a_rows_pad, d.b_rows_pad, d.a_planes, d.b_planes = a.rows_pad, b.rows_pad, a.planes, b.planes
    d.ldc, d.n_split = cs[0].stride(0), n_split
    d.bias, d.bias2 = ptr(bias), ptr(bias2)
    d.alpha, d.beta = alpha, beta
    d.direct = int(direct)            # 0 promoted accumulation, 1 direct chain, 2 by rounding count (nabu_hip.h)
    if planes == 2:
        for i in range(d.nbatch):
            d.a_amax[i] = (a_amax[i] if a_amax else a.amax.data_ptr())
            d.b_amax[i] = (b_amax[i] if b_amax else b.amax.data_ptr())
    ws_bytes = L.nabu_gemm_pk_ws_bytes(ctypes.byref(d))
    ws = Workspace.get(ws_bytes, cs[0].device, 'gemm') if ws_bytes else None
    check(L.nabu_gemm_pk(ctypes.byref(d), ptr(ws), ws_bytes, stream()), 'nabu_gemm_pk')
    return c


def set_gemm_precision(precision):
    """process default of every GEMM that does not name a precision ('f32' | 'bf16' | 'bf16x3' | 'bf16x6' | 'f16x3')"""
    check(_hip.lib().nabu_gemm_set_default_precision(_hip.GEMM_PRECISIONS[precision]), 'nabu_gemm_set_default_precision')


def get_gemm_precision():
    code = _hip.lib().nabu_gemm_get_default_precision()
    return [k for k, v in _hip.GEMM_PRECISIONS.items() if v == code][0]


def colsum(a, out, beta=0.0):
    """out[n] = beta*out[n] + sum_m a[m,n] for a 2-D row-major tensor."""
    L = _hip.lib()
    M, N = a.shape
    ws_bytes = L.nabu_colsum_ws_bytes(M, N)
    ws = Workspace.get(ws_bytes, a.device, 'gemm')
    check(L.nabu_colsum_f32(M, N, ptr(a), a.stride(0), beta, ptr(out), ptr(ws), ws_bytes, stream()),
          'nabu_colsum_f32')
    return out


class BlstmPlan(object):
    """Shape descriptor + buffers of one BLSTM layer call."""

    def __init__(self, B, T, D, H, max_len, mode=LSTM_AUTO, gemm_precision='default', x_bound=0.0, fwd_only=False,
                 recurrent_precision='default', out_stack=0):
        """x_bound > 0: |x| <= x_bound is guaranteed (the previous layer's LSTM outputs) — the f16x3 packs of x skip their
        measuring pass; fwd_only: no backward pass follows (validation): the reserve holds the activations only;
        recurrent_precision 'f32': the exact-fp32 recurrent kernels (include/nabu_hip.h, nabu_blstm_desc)"""
        import os
        # (NABU_DESC_V1=1: the 32-byte ABI-version-1 descriptor — A/B runs against a library built before round 5)
        self.desc = _hip.BlstmDesc(32 if os.environ.get('NABU_DESC_V1') == '1' else ctypes.sizeof(_hip.BlstmDesc), B, T, D, H, int(max_len), mode,
                                   _hip.GEMM_PRECISIONS[gemm_precision], float(x_bound),
                                   _hip.BLSTM_FWD_ONLY if fwd_only else 0, _hip.REC_PRECISIONS[recurrent_precision],
                                   int(out_stack), None, None, None, None, None)
        L = _hip.lib()
        # packed companions (ABI version 3): sizes of x_pk_rows, x_pk_cols, hT_pk, out_pk_rows, out_pk_cols for this layer
        # (0 = the layer would not use that companion); the buffers themselves are attached by the caller (set_companions)
        pkb = (ctypes.c_size_t * 5)()
        self.pk_bytes = [0] * 5
        if self.desc.size == ctypes.sizeof(_hip.BlstmDesc) and L.nabu_blstm_pk_bytes(ctypes.byref(self.desc), pkb) == 0:
            self.pk_bytes = [int(v) for v in pkb]
        self._keep = []
        self.reserve_bytes = L.nabu_blstm_reserve_bytes(ctypes.byref(self.desc))
        self.ws_bytes = L.nabu_blstm_ws_bytes(ctypes.byref(self.desc))
        if self.reserve_bytes == 0:
            raise _hip.NabuHipError('blstm: unsupported shape B=%d T=%d D=%d H=%d: %s' % (
                B, T, D, H, L.nabu_last_error().decode()))


def blstm_set_companions(plan, x_pk=None, out_pk=None, hT_pk=None):
    """attach packed companions to a plan (include/nabu_hip.h, nabu_blstm_desc ABI version 3): x_pk / out_pk = (rows, cols)
    uint8 tensors of plan.pk_bytes[0:2] / [3:5] bytes, hT_pk one tensor of plan.pk_bytes[2] bytes, each zero-filled once
    by its owner.  The plan keeps them alive."""
    d = plan.desc
    if x_pk is not None:
        assert x_pk[0].numel() == plan.pk_bytes[0] and x_pk[1].numel() == plan.pk_bytes[1]
        d.x_pk_rows, d.x_pk_cols = x_pk[0].data_ptr(), x_pk[1].data_ptr()
    if out_pk is not None:
        assert out_pk[0].numel() == plan.pk_bytes[3] and out_pk[1].numel() == plan.pk_bytes[4]
        d.out_pk_rows, d.out_pk_cols = out_pk[0].data_ptr(), out_pk[1].data_ptr()
    if hT_pk is not None:
        assert hT_pk.numel() == plan.pk_bytes[2]
        d.hT_pk = hT_pk.data_ptr()
    plan._keep += [t for t in (x_pk or ()) + (out_pk or ()) + ((hT_pk,) if hT_pk is not None else ())]


def blstm_emits_packed(plan):
    """which companions attached to `plan` nabu_blstm_fwd's recurrent kernel writes itself: bit 0 rows, 1 transposed, 2 h^T"""
    return int(_hip.lib().nabu_blstm_emits_packed(ctypes.byref(plan.desc)))


def blstm_drop_companions(plan, out_pk=False, hT_pk=False):
    d = plan.desc
    if out_pk:
        d.out_pk_rows, d.out_pk_cols = None, None
    if hT_pk:
        d.hT_pk = None


def blstm_fwd(plan, x, lens_dev, k_fw, b_fw, k_bw, b_bw, out, reserve):
    L = _hip.lib()
    ws = Workspace.get(plan.ws_bytes, x.device, 'blstm')
    if BEFORE_RECURRENT[0] is not None:
        BEFORE_RECURRENT[0]()
    if PROFILER is not None:
        PROFILER.arm('fwd', plan)
    check(L.nabu_blstm_fwd(ctypes.byref(plan.desc), ptr(_f32(x, 'x')), ptr(lens_dev), ptr(k_fw), ptr(b_fw),
                           ptr(k_bw), ptr(b_bw), ptr(out), ptr(reserve), ptr(ws), plan.ws_bytes,
                           stream()), 'nabu_blstm_fwd')
    if PROFILER is not None:
        PROFILER.disarm()
    return out


def blstm_bwd(plan, x, lens_dev, k_fw, k_bw, out, d_out, reserve, d_x, dk_fw, db_fw, dk_bw, db_bw):
    L = _hip.lib()
    ws = Workspace.get(plan.ws_bytes, x.device, 'blstm')
    if BEFORE_RECURRENT[0] is not None:
        BEFORE_RECURRENT[0]()
    if PROFILER is not None:
        PROFILER.arm('bwd', plan)
    check(L.nabu_blstm_bwd(ctypes.byref(plan.desc), ptr(x), ptr(lens_dev), ptr(k_fw), ptr(k_bw), ptr(out),
                           ptr(d_out), ptr(reserve), ptr(d_x), ptr(dk_fw), ptr(db_fw), ptr(dk_bw),
                           ptr(db_bw), ptr(ws), plan.ws_bytes, stream()), 'nabu_blstm_bwd')
    if PROFILER is not None:
        PROFILER.disarm()
    return d_x


def blstm_bwd_data(plan, x, lens_dev, k_fw, k_bw, out, d_out, reserve, d_x, db_fw, db_bw):
    """first half of blstm_bwd (nabu_blstm_bwd_data): recurrence backward, bias gradients, d_x; dz stays in `reserve`"""
    L = _hip.lib()
    ws = Workspace.get(plan.ws_bytes, x.device, 'blstm')
    if BEFORE_RECURRENT[0] is not None:
        BEFORE_RECURRENT[0]()
    if PROFILER is not None:
        PROFILER.arm('bwd', plan)
    check(L.nabu_blstm_bwd_data(ctypes.byref(plan.desc), ptr(x), ptr(lens_dev), ptr(k_fw), ptr(k_bw), ptr(out),
                                ptr(d_out), ptr(reserve), ptr(d_x), ptr(db_fw), ptr(db_bw), ptr(ws), plan.ws_bytes,
                                stream()), 'nabu_blstm_bwd_data')
    if PROFILER is not None:
        PROFILER.disarm()
    return d_x


def blstm_bwd_weights(plan, x, lens_dev, out, reserve, dk_fw, dk_bw):
    """second half (nabu_blstm_bwd_weights): the kernel gradients from the dz blstm_bwd_data left in `reserve`"""
    ws = Workspace.get(plan.ws_bytes, x.device, 'blstm')
    check(_hip.lib().nabu_blstm_bwd_weights(ctypes.byref(plan.desc), ptr(x), ptr(lens_dev), ptr(out), ptr(reserve),
                                            ptr(dk_fw), ptr(dk_bw), ptr(ws), plan.ws_bytes, stream()),
          'nabu_blstm_bwd_weights')


def pad_time(x, Tp):
    B, T, F = x.shape
    y = torch.empty((B, Tp, F), dtype=x.dtype, device=x.device)
    check(_hip.lib().nabu_pad_time_f32(B, T, Tp, F, ptr(x), ptr(y), stream()), 'nabu_pad_time_f32')
    return y


def unpad_time(y, T):
    B, Tp, F = y.shape
    x = torch.empty((B, T, F), dtype=y.dtype, device=y.device)
    check(_hip.lib().nabu_unpad_time_f32(B, T, Tp, F, ptr(y), ptr(x), stream()), 'nabu_unpad_time_f32')
    return x


def ctc_loss_grad(logits, logit_len_dev, labels_dev, label_len_dev, grad_scale):
    """Returns (nll [B], dlogits [B,T,C], status [1] int32) — all on device."""
    L = _hip.lib()
    B, T, C = logits.shape
    Lmax = labels_dev.shape[1]
    nll = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    status = torch.empty(1, dtype=torch.int32, device=logits.device)
    ws_bytes = L.nabu_ctc_ws_bytes(B, T, Lmax)
    ws = Workspace.get(ws_bytes, logits.device, 'ctc')
    check(L.nabu_ctc_loss_grad(B, T, C, Lmax, ptr(_f32(logits, 'logits')), ptr(logit_len_dev),
                               ptr(labels_dev), ptr(label_len_dev), grad_scale, ptr(nll), ptr(dlogits),
                               ptr(status), ptr(ws), ws_bytes, stream()), 'nabu_ctc_loss_grad')
    return nll, dlogits, status


def adam_clip_step(param, grad, m, v, lr_t, b1=0.9, b2=0.999, eps=1e-8, clip=1.0, grad_scale=1.0):
    check(_hip.lib().nabu_adam_clip_step(param.numel(), ptr(param), ptr(grad), ptr(m), ptr(v), lr_t, b1, b2,
                                         eps, clip, grad_scale, stream()), 'nabu_adam_clip_step')


def clip_(g, clip=1.0):
    check(_hip.lib().nabu_clip_f32(g.numel(), ptr(g), clip, stream()), 'nabu_clip_f32')
    return g


def dropout(x, keep_prob, seed, offset):
    y = torch.empty_like(x)
    check(_hip.lib().nabu_dropout_f32(x.numel(), ptr(x), ptr(y), keep_prob, seed, offset, stream()),
          'nabu_dropout_f32')
    return y


def sample_ids(logits, prob, seed, offset, teacher_ids):
    """scheduled-sampling choice of the next decoder input (ScheduledEmbeddingTrainingHelper)"""
    B, C = logits.shape
    out = torch.empty_like(teacher_ids)
    check(_hip.lib().nabu_sample_ids(B, C, ptr(logits), prob, seed, offset, ptr(teacher_ids), ptr(out), stream()),
          'nabu_sample_ids')
    return out


def gaussian_noise(x, stddev, seed, offset):
    y = torch.empty_like(x)
    check(_hip.lib().nabu_gaussian_noise_f32(x.numel(), ptr(x), ptr(y), stddev, seed, offset, stream()),
          'nabu_gaussian_noise_f32')
    return y


def sum_(x, scale=1.0):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_hip.lib().nabu_sum_f32(x.numel(), ptr(x), scale, ptr(out), stream()), 'nabu_sum_f32')
    return out


def axpy_(y, x, a=1.0):
    check(_hip.lib().nabu_axpy_f32(x.numel(), a, ptr(x), ptr(y), stream()), 'nabu_axpy_f32')
    return y


def xent_loss_grad(logits, targets_dev, logit_len_dev, target_len_dev, grad_scale):
    """Returns (loss [B], dlogits [B,L,C])."""
    B, L, C = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(_hip.lib().nabu_xent_loss_grad(B, L, C, targets_dev.shape[1], ptr(_f32(logits, 'logits')),
                                         ptr(targets_dev), ptr(logit_len_dev), ptr(target_len_dev),
                                         grad_scale, ptr(loss), ptr(dlogits), stream()),
          'nabu_xent_loss_grad')
    return loss, dlogits


class RecurrentProfiler(object):
    """Times the recurrent kernel(s) of every BLSTM call with HIP events recorded by
    the library on the launch stream (nabu_blstm_set_profile_events).  Used by
    bench.py for the live roofline figure; off by default."""

    def __init__(self):
        self.hip = ctypes.CDLL('libamdhip64.so')
        self.records = []          # (kind, B, T_steps, H, ev_begin, ev_end)
        self.enabled = False

    def _event(self):
        ev = ctypes.c_void_p()
        err = self.hip.hipEventCreate(ctypes.byref(ev))
        if err:
            raise _hip.NabuHipError('hipEventCreate failed: %d' % err)
        return ev

    def arm(self, kind, plan):
        if not self.enabled:
            return
        b, e = self._event(), self._event()
        _hip.lib().nabu_blstm_set_profile_events(b, e)
        d = plan.desc
        self.records.append((kind, d.B, d.max_len if d.max_len > 0 else d.T, d.H, b, e))

    def disarm(self):
        if self.enabled:
            _hip.lib().nabu_blstm_set_profile_events(None, None)

    def collect(self):
        """[(kind, B, steps, H, milliseconds)] — call after a device synchronize."""
        out = []
        for kind, B, steps, H, b, e in self.records:
            ms = ctypes.c_float()
            err = self.hip.hipEventElapsedTime(ctypes.byref(ms), b, e)
            if err:
                raise _hip.NabuHipError('hipEventElapsedTime failed: %d' % err)
            out.append((kind, B, steps, H, ms.value))
            self.hip.hipEventDestroy(b)
            self.hip.hipEventDestroy(e)
        self.records = []
        return out


PROFILER = None


def enable_profiler():
    global PROFILER
    if PROFILER is None:
        PROFILER = RecurrentProfiler()
    PROFILER.enabled = True
    return PROFILER


_phase_hook = [None]       # keeps the ctypes trampoline alive


def set_phase_hook(fn):
    """fn() is called by nabu_blstm_bwd between its recurrent kernel(s) and its dense products
    (include/nabu_hip.h nabu_blstm_set_phase_hook); None switches the hook off"""
    if fn is None:
        _phase_hook[0] = None
        check(_hip.lib().nabu_blstm_set_phase_hook(None, None), 'nabu_blstm_set_phase_hook')
        return
    def trampoline(user):
        # ctypes prints and swallows an exception raised inside a C callback: keep it and let the
        # caller re-raise it once the C call has returned (take_phase_hook_error)
        try:
            fn()
        except BaseException as exc:                     # noqa: B902 — re-raised by the caller
            if _phase_hook_error[0] is None:
                _phase_hook_error[0] = exc
    _phase_hook[0] = _hip.PHASE_HOOK_T(trampoline)
    check(_hip.lib().nabu_blstm_set_phase_hook(ctypes.cast(_phase_hook[0], ctypes.c_void_p), None),
          'nabu_blstm_set_phase_hook')


_phase_hook_error = [None]


def take_phase_hook_error():
    """the first exception a phase hook raised since the last call (None if none); clears it"""
    exc, _phase_hook_error[0] = _phase_hook_error[0], None
    return exc


# called right before a recurrent launch is enqueued (forward and backward): the data-parallel
# trainer joins its communication stream here (the persistent kernels must run alone)
BEFORE_RECURRENT = [None]


def set_persist_timeout_ms(ms):
    """bound of the in-kernel waits of the persistent recurrent kernels (0 = the 200 ms default)"""
    check(_hip.lib().nabu_persist_set_timeout_us(int(ms * 1000)), 'nabu_persist_set_timeout_us')


def check_persist_status(device=None):
    """Raise if a persistent recurrent kernel gave up (bounded-spin timeout); the
    status word is the first int32 of the 'blstm' / 'speller' workspace.  Synchronises."""
    for (dev, tag), buf in list(Workspace._bufs.items()):
        if tag not in ('blstm', 'speller'):     # 'speller': the persistent decoder kernel (speller_persist.hip)
            continue
        code = int(buf[:4].view(torch.int32).item())
        if code:
            buf[:4].zero_()
            raise _hip.NabuHipError(
                'persistent %s kernel timed out waiting for a peer workgroup (code %d: block %d, %s); '
                'results of this step are invalid' % ('LSTM' if tag == 'blstm' else 'decoder', code, code // 4,
                                                      {1: 'forward pass', 2: 'backward pass',
                                                       3: 'start-up handshake'}.get(code % 4, '?')))


# ---------------------------------------------------------------- speller step kernels
def lstm_cell_fwd(step, seq_len_dev, z, bias, emb_rows, ids, c_prev, h_prev, acts, c_new, h_new):
    B, U = c_prev.shape
    check(_hip.lib().nabu_lstm_cell_fwd(B, U, step, ptr(seq_len_dev), ptr(z), ptr(bias), ptr(emb_rows),
                                        ptr(ids), ptr(c_prev), ptr(h_prev), ptr(acts), ptr(c_new), ptr(h_new),
                                        stream()), 'nabu_lstm_cell_fwd')


def lstm_cell_bwd(step, seq_len_dev, acts, c_new, c_prev, dh, dh2, dc_in, dz, dc_out):
    B, U = c_prev.shape
    check(_hip.lib().nabu_lstm_cell_bwd(B, U, step, ptr(seq_len_dev), ptr(acts), ptr(c_new), ptr(c_prev),
                                        ptr(dh), ptr(dh2), ptr(dc_in), ptr(dz), ptr(dc_out), stream()),
          'nabu_lstm_cell_bwd')


PROB_FNS = {'softmax': 0, 'sigmoid': 1, 'normalized_sigmoid': 2}


def attn_desc(B, Te, E, U, kind, K=0, F=0, prob_fn=0):
    return _hip.AttnDesc(ctypes.sizeof(_hip.AttnDesc), B, Te, E, U, kind, K, F, prob_fn)


def attn_fwd(desc, step, dec_len, enc_len, keys, values, q, v, conv_kernel, conv_proj, align_prev,
             ctx_prev, align, ctx, znorm=None):
    L = _hip.lib()
    nbytes = L.nabu_attn_fwd_ws_bytes(ctypes.byref(desc))
    ws = Workspace.get(nbytes, keys.device, 'attn_fwd') if nbytes else None
    check(L.nabu_attn_fwd(ctypes.byref(desc), step, ptr(dec_len), ptr(enc_len), ptr(keys), ptr(values),
                          ptr(q), ptr(v), ptr(conv_kernel), ptr(conv_proj), ptr(align_prev),
                          ptr(ctx_prev), ptr(align), ptr(ctx), ptr(znorm), ptr(ws), nbytes, stream()), 'nabu_attn_fwd')


def attn_bwd(desc, step, dec_len, enc_len, keys, values, q, v, conv_kernel, conv_proj, align_prev, align, ctx,
             dctx, dalign_in, dq, dkeys, dv_part, dcp_part, dck_part, dalign_out, znorm=None):
    """dv_part / dcp_part have B * attn_bwd_slices(desc) rows; ctx = this step's context"""
    L = _hip.lib()
    nbytes = L.nabu_attn_bwd_ws_bytes(ctypes.byref(desc))
    ws = Workspace.get(nbytes, keys.device, 'attn_bwd')
    check(L.nabu_attn_bwd(ctypes.byref(desc), step, ptr(dec_len), ptr(enc_len), ptr(keys), ptr(values),
                          ptr(q), ptr(v), ptr(conv_kernel), ptr(conv_proj), ptr(align_prev),
                          ptr(align), ptr(ctx), ptr(dctx), ptr(dalign_in), ptr(dq), ptr(dkeys), ptr(dv_part),
                          ptr(dcp_part), ptr(dck_part), ptr(dalign_out), ptr(znorm), ptr(ws), nbytes, stream()),
          'nabu_attn_bwd')


def attn_bwd_slices(desc):
    return _hip.lib().nabu_attn_bwd_slices(ctypes.byref(desc))


def mask_time_(x, len_dev):
    B, L, F = x.shape
    check(_hip.lib().nabu_mask_time_f32(B, L, F, ptr(x), ptr(len_dev), stream()), 'nabu_mask_time_f32')
    return x


def swap01(x):
    """[L,B,F] -> [B,L,F] (new tensor)."""
    L, B, F = x.shape
    y = torch.empty((B, L, F), dtype=x.dtype, device=x.device)
    check(_hip.lib().nabu_swap01_f32(L, B, F, ptr(x), ptr(y), stream()), 'nabu_swap01_f32')
    return y


def scatter_rows(ids, dz, dK):
    """dK[c,:] = sum_{i: ids[i]==c} dz[i,:]; ids [N] int32, dz [N,W], dK [C,W]."""
    N, W = dz.shape
    check(_hip.lib().nabu_scatter_rows_f32(dK.shape[0], N, W, ptr(ids), ptr(dz), ptr(dK), stream()),
          'nabu_scatter_rows_f32')
    return dK


def relu(x):
    y = torch.empty_like(x)
    check(_hip.lib().nabu_relu_f32(x.numel(), ptr(x), ptr(y), stream()), 'nabu_relu_f32')
    return y


def relu_bwd(y, dy):
    dx = torch.empty_like(y)
    check(_hip.lib().nabu_relu_bwd_f32(y.numel(), ptr(y), ptr(dy), ptr(dx), stream()), 'nabu_relu_bwd_f32')
    return dx


def layer_norm_fwd(x, gamma, beta, eps=1e-12):
    """x [B,...,F]: moments over everything but the batch axis (tf.contrib.layers.layer_norm)"""
    B, F = x.shape[0], x.shape[-1]
    N = x.numel() // B
    y = torch.empty_like(x)
    mean = torch.empty(B, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, dtype=torch.float32, device=x.device)
    check(_hip.lib().nabu_layer_norm_fwd(B, N, F, ptr(x), ptr(gamma), ptr(beta), eps, ptr(y), ptr(mean), ptr(rstd),
                                         stream()), 'nabu_layer_norm_fwd')
    return y, mean, rstd


def layer_norm_bwd(x, gamma, dy, mean, rstd):
    B, F = x.shape[0], x.shape[-1]
    N = x.numel() // B
    dx = torch.empty_like(x)
    dgp = torch.empty((B, F), dtype=torch.float32, device=x.device)
    dbp = torch.empty((B, F), dtype=torch.float32, device=x.device)
    check(_hip.lib().nabu_layer_norm_bwd(B, N, F, ptr(x), ptr(gamma), ptr(dy), ptr(mean), ptr(rstd), ptr(dx),
                                         ptr(dgp), ptr(dbp), stream()), 'nabu_layer_norm_bwd')
    return dx, dgp, dbp


def ceil_div_i32(x, d):
    out = torch.empty_like(x)
    check(_hip.lib().nabu_ceil_div_i32(x.numel(), ptr(x), int(d), ptr(out), stream()), 'nabu_ceil_div_i32')
    return out


# --------------------------------------------------------------------------
# inference decoders (decode.hip)
def ctc_beam_search(logits, logit_len, beam_width=100, merge_repeated=True):
    """tf.nn.ctc_beam_search_decoder(top_paths=1) on batch-major logits [B,T,C]:
    returns (ids [B,T] int32 padded with -1, lengths [B] int32, log-probabilities [B])"""
    B, T, C = logits.shape
    logits = logits.contiguous()
    lib = _hip.lib()
    nbytes = lib.nabu_ctc_beam_ws_bytes(B, T, C, int(beam_width))
    ws = _hip.Workspace.get(nbytes, logits.device, 'ctc_beam')
    ids = torch.empty((B, T), dtype=torch.int32, device=logits.device)
    lens = torch.empty((B,), dtype=torch.int32, device=logits.device)
    lp = torch.empty((B,), dtype=torch.float32, device=logits.device)
    check(lib.nabu_ctc_beam_search(B, T, C, int(beam_width), int(bool(merge_repeated)), ptr(logits),
                                   ptr(logit_len), ptr(ids), ptr(lens), ptr(lp), ptr(ws), nbytes, stream()),
          'nabu_ctc_beam_search')
    return ids, lens, lp


def edit_distance(hyp, hyp_len, truth, truth_len):
    """Levenshtein distance per row: hyp [B,Lh], truth [B,Lt] int32 with their lengths -> [B] int32"""
    B = hyp.shape[0]
    hyp, truth = hyp.contiguous(), truth.contiguous()
    dist = torch.empty((B,), dtype=torch.int32, device=hyp.device)
    check(_hip.lib().nabu_edit_distance(B, ptr(hyp) if hyp.numel() else None, hyp.shape[1], ptr(hyp_len),
                                        ptr(truth) if truth.numel() else None, truth.shape[1],
                                        ptr(truth_len), ptr(dist), stream()), 'nabu_edit_distance')
    return dist


def beam_prune(logits, logprobs, lengths, finished, seen, temperature=1.0, length_penalty=0.0):
    """one expand+prune step of the attention beam search; logits [B,W,C]; the [B,W] state tensors
    are updated in place.  Returns (pred_ids, parent, stay, all_seen)"""
    B, W, C = logits.shape
    dev = logits.device
    pred = torch.empty((B, W), dtype=torch.int32, device=dev)
    parent, stay = torch.empty_like(pred), torch.empty_like(pred)
    all_seen = torch.empty((B,), dtype=torch.int32, device=dev)
    scratch = torch.empty((B, W * C + W), dtype=torch.float32, device=dev)
    check(_hip.lib().nabu_beam_prune(B, W, C, ptr(logits.contiguous()), float(temperature), float(length_penalty),
                                     ptr(logprobs), ptr(lengths), ptr(finished), ptr(seen), ptr(pred),
                                     ptr(parent), ptr(stay), ptr(all_seen), ptr(scratch), stream()),
          'nabu_beam_prune')
    return pred, parent, stay, all_seen


def beam_gather(fresh, old, parent, stay):
    B, W, F = fresh.shape
    dst = torch.empty_like(fresh)
    check(_hip.lib().nabu_beam_gather(B, W, F, ptr(fresh.contiguous()), ptr(old.contiguous()), ptr(parent),
                                      ptr(stay), ptr(dst), stream()), 'nabu_beam_gather')
    return dst


def persist_clocks(device=None):
    """{'fwd': GHz, 'bwd': GHz} — the shader clock the chip sustained under the LAST fp16-plane recurrent launch of each
    pass on this device's 'blstm' workspace (lstm_persist_dev.h, clock_stamp: shader-cycle and 100 MHz wall-clock ticks
    left by block 0 at the start and the end of the launch).  None for a pass that has not run.  Synchronises."""
    out = {'fwd': None, 'bwd': None}
    for (dev, tag), buf in list(Workspace._bufs.items()):
        if tag != 'blstm' or (device is not None and dev != str(device)):
            continue
        w = buf[4 * 280:4 * 288].view(torch.int32).cpu().tolist()
        for i, name in enumerate(('fwd', 'bwd')):
            c0, w0, c1, w1 = w[4 * i:4 * i + 4]
            dc, dw = (c1 - c0) & 0xFFFFFFFF, (w1 - w0) & 0xFFFFFFFF
            if dw > 0 and dc > 0:
                out[name] = round(dc / dw * 0.1, 4)
    return out
