"""ctypes binding of libnabu_hip.so (the C ABI declared in include/nabu_hip.h).

The product path has NO fallback: if the HIP library is missing or a tensor is
not on the GPU, the call raises.  PyTorch is used only to own device memory and
to name the HIP stream the kernels are enqueued on."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NABU_HIP_LIB') or os.path.join(_HERE, 'libnabu_hip.so')   # (the override: A/B builds in experiments)

_c = ctypes
_vp, _i, _f, _sz, _ll = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_longlong


class BlstmDesc(_c.Structure):
    _fields_ = [('size', _c.c_uint32), ('B', _c.c_int32), ('T', _c.c_int32), ('D', _c.c_int32),
                ('H', _c.c_int32), ('max_len', _c.c_int32), ('mode', _c.c_int32),
                ('gemm_precision', _c.c_int32), ('x_bound', _c.c_float), ('flags', _c.c_int32),
                ('recurrent_precision', _c.c_int32),
                # ABI version 3: packed companions (include/nabu_hip.h)
                ('out_stack', _c.c_int32), ('out_pk_rows', _c.c_void_p), ('out_pk_cols', _c.c_void_p),
                ('hT_pk', _c.c_void_p), ('x_pk_rows', _c.c_void_p), ('x_pk_cols', _c.c_void_p)]


BLSTM_FWD_ONLY = 1
REC_PRECISIONS = {'default': 0, 'f32': 1}


class PkGemmDesc(_c.Structure):
    _fields_ = [('size', _c.c_uint32), ('planes', _c.c_int32), ('M', _c.c_int32), ('N', _c.c_int32),
                ('nkb', _c.c_int32), ('nbatch', _c.c_int32), ('A', _c.c_void_p * 2), ('B', _c.c_void_p * 2),
                ('a_rows_pad', _c.c_int32), ('b_rows_pad', _c.c_int32), ('a_planes', _c.c_int32),
                ('b_planes', _c.c_int32), ('C', _c.c_void_p * 2), ('C2', _c.c_void_p * 2),
                ('ldc', _c.c_int32), ('n_split', _c.c_int32), ('bias', _c.c_void_p), ('bias2', _c.c_void_p),
                ('alpha', _c.c_float), ('beta', _c.c_float), ('a_amax', _c.c_void_p * 2), ('b_amax', _c.c_void_p * 2), ('direct', _c.c_int32)]


# name -> (restype, argtypes); must list every symbol of include/nabu_hip.h
SIGNATURES = {
    'nabu_version': (_i, []),
    'nabu_last_error': (_c.c_char_p, []),
    'nabu_gemm_ws_bytes': (_sz, [_i, _i, _i]),
    'nabu_gemm_f32': (_i, [_i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _ll, _ll,
                           _vp, _sz, _vp]),
    'nabu_gemm_ex': (_i, [_i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _ll, _ll,
                          _vp, _sz, _vp]),
    'nabu_gemm2_ws_bytes': (_sz, [_i, _i, _i, _i]),
    'nabu_gemm2_f32': (_i, [_i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _i, _vp, _vp, _sz, _vp]),
    'nabu_cvt_bf16': (_i, [_sz, _i, _vp, _i, _vp, _i, _i, _vp]),
    'nabu_gemm_bf16_nt_ws_bytes': (_sz, [_i, _i, _i]),
    'nabu_gemm_bf16_nt': (_i, [_i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _vp, _vp, _sz, _vp]),
    'nabu_pk_rows_pad': (_i, [_i]),
    'nabu_pk_kblocks': (_i, [_i, _i]),
    'nabu_pk_bytes': (_sz, [_i, _i, _i]),
    'nabu_pk_pack': (_i, [_i, _i, _vp, _ll, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'nabu_pk_amax': (_i, [_vp, _ll, _i, _i, _vp, _vp, _vp]),
    'nabu_pk_amax_fill': (_i, [_vp, _i, _f, _vp]),
    'nabu_pk_pack_f16': (_i, [_i, _vp, _ll, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'nabu_gemm_pk_ws_bytes': (_sz, [_c.POINTER(PkGemmDesc)]),
    'nabu_gemm_pk': (_i, [_c.POINTER(PkGemmDesc), _vp, _sz, _vp]),
    'nabu_gemm_set_default_precision': (_i, [_i]),
    'nabu_gemm_get_default_precision': (_i, []),
    'nabu_colsum_ws_bytes': (_sz, [_i, _i]),
    'nabu_colsum_f32': (_i, [_i, _i, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    'nabu_blstm_reserve_bytes': (_sz, [_c.POINTER(BlstmDesc)]),
    'nabu_blstm_ws_bytes': (_sz, [_c.POINTER(BlstmDesc)]),
    'nabu_blstm_fwd': (_i, [_c.POINTER(BlstmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_blstm_bwd': (_i, [_c.POINTER(BlstmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _sz, _vp]),
    'nabu_blstm_bwd_data': (_i, [_c.POINTER(BlstmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_blstm_bwd_weights': (_i, [_c.POINTER(BlstmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_blstm_uses_persistent': (_i, [_c.POINTER(BlstmDesc)]),
    'nabu_blstm_pk_bytes': (_i, [_c.POINTER(BlstmDesc), _c.POINTER(_c.c_size_t)]),
    'nabu_blstm_emits_packed': (_i, [_c.POINTER(BlstmDesc)]),
    'nabu_blstm_set_profile_events': (_i, [_vp, _vp]),
    'nabu_persist_set_timeout_us': (_i, [_ll]),
    'nabu_blstm_set_phase_hook': (_i, [_vp, _vp]),
    'nabu_pad_time_f32': (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    'nabu_unpad_time_f32': (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    'nabu_ctc_ws_bytes': (_sz, [_i, _i, _i]),
    'nabu_ctc_loss_grad': (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_xent_loss_grad': (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    'nabu_lstm_cell_fwd': (_i, [_i, _i, _i] + [_vp] * 10 + [_vp]),
    'nabu_lstm_cell_bwd': (_i, [_i, _i, _i] + [_vp] * 9 + [_vp]),
    'nabu_attn_fwd': (_i, [_vp, _i] + [_vp] * 14 + [_sz, _vp]),
    'nabu_attn_fwd_ws_bytes': (_sz, [_vp]),
    'nabu_attn_bwd': (_i, [_vp, _i] + [_vp] * 21 + [_sz, _vp]),
    'nabu_attn_bwd_slices': (_i, [_vp]),
    'nabu_attn_bwd_ws_bytes': (_sz, [_vp]),
    'nabu_sample_ids': (_i, [_i, _i, _vp, _f, _c.c_ulonglong, _c.c_ulonglong, _vp, _vp, _vp]),
    'nabu_speller_decoder_inputs': (_i, [_vp, _vp, _vp, _vp]),
    'nabu_speller_reserve_bytes': (_sz, [_vp]),
    'nabu_speller_ws_bytes': (_sz, [_vp]),
    'nabu_speller_uses_persistent': (_i, [_vp, _i]),
    'nabu_speller_fwd': (_i, [_vp] * 9 + [_sz, _vp]),
    'nabu_speller_bwd': (_i, [_vp] * 11 + [_sz, _vp]),
    'nabu_mask_time_f32': (_i, [_i, _i, _i, _vp, _vp, _vp]),
    'nabu_swap01_f32': (_i, [_i, _i, _i, _vp, _vp, _vp]),
    'nabu_scatter_rows_f32': (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    'nabu_adam_clip_step': (_i, [_sz, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp]),
    'nabu_clip_f32': (_i, [_sz, _vp, _f, _vp]),
    'nabu_dropout_f32': (_i, [_sz, _vp, _vp, _f, _c.c_ulonglong, _c.c_ulonglong, _vp]),
    'nabu_gaussian_noise_f32': (_i, [_sz, _vp, _vp, _f, _c.c_ulonglong, _c.c_ulonglong, _vp]),
    'nabu_sum_f32': (_i, [_sz, _vp, _f, _vp, _vp]),
    'nabu_axpy_f32': (_i, [_sz, _f, _vp, _vp, _vp]),
    'nabu_ceil_div_i32': (_i, [_i, _vp, _i, _vp, _vp]),
    'nabu_ctc_beam_ws_bytes': (_sz, [_i, _i, _i, _i]),
    'nabu_ctc_beam_search': (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_edit_distance': (_i, [_i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'nabu_speller_beam_ws_bytes': (_sz, [_vp]),
    'nabu_speller_beam_search': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nabu_beam_prune': (_i, [_i, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nabu_beam_gather': (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nabu_relu_f32': (_i, [_sz, _vp, _vp, _vp]),
    'nabu_relu_bwd_f32': (_i, [_sz, _vp, _vp, _vp, _vp]),
    'nabu_layer_norm_fwd': (_i, [_i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    'nabu_crc32c_host': (_c.c_uint32, [_c.c_char_p, _sz, _c.c_uint32]),
    'nabu_layer_norm_bwd': (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None
PHASE_HOOK_T = _c.CFUNCTYPE(None, _c.c_void_p)


class NabuHipError(RuntimeError):
    pass


def lib():
    """Load libnabu_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NabuHipError(
                'libnabu_hip.so is missing (%s): run `python -m nabu_amd.build` or '
                '__graft_entry__.build(); there is no CPU fallback' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.nabu_version() != ABI_VERSION:
            raise NabuHipError('libnabu_hip.so reports ABI version %d, this binding is written against %d: rebuild '
                               '(python -m nabu_amd.build)' % (handle.nabu_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().nabu_last_error().decode(errors='replace')
        raise NabuHipError('%s failed (%d): %s' % (what, code, msg))


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NabuHipError('tensor is not on the GPU (device=%s): the HIP path has no CPU fallback'
                           % t.device)
    if not t.is_contiguous():
        raise NabuHipError('tensor must be contiguous')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class Workspace(object):
    """Grow-only scratch buffers, one per (device, tag)."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes, device, tag='ws'):
        key = (str(device), tag)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            # zero-filled: the first word of the 'blstm' workspace is the persistent
            # kernels' status word (0 = ok)
            buf = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


SPELLER_MAX_LAYERS = 4
GEMM_DEFAULT, GEMM_F32, GEMM_BF16, GEMM_BF16X3, GEMM_BF16X6 = 0, 1, 2, 3, 4
# NABU_ABI_VERSION of include/nabu_hip.h this binding was written against (lib() refuses another library)
ABI_VERSION = 3

GEMM_PRECISIONS = {'default': 0, 'f32': 1, 'bf16': 2, 'bf16x3': 3, 'bf16x6': 4, 'f16x3': 5}


class SpellerDesc(_c.Structure):
    _fields_ = [('size', _c.c_uint32)] + [(n, _c.c_int32) for n in
                                          ('B', 'Te', 'E', 'U', 'C', 'L', 'num_layers', 'kind', 'K', 'F', 'prob_fn')] + \
               [('keep_prob', _c.c_float), ('seed', _c.c_ulonglong), ('seed_offset', _c.c_ulonglong),
                ('sample_prob', _c.c_float), ('sample_seed', _c.c_ulonglong), ('sample_offset', _c.c_ulonglong)]


class BeamDesc(_c.Structure):
    _fields_ = [('size', _c.c_uint32)] + [(n, _c.c_int32) for n in
                                          ('B', 'Te', 'E', 'U', 'C', 'num_layers', 'kind', 'K', 'F', 'prob_fn',
                                           'beam_width', 'max_steps')] + \
               [('length_penalty', _c.c_float), ('temperature', _c.c_float)]


class SpellerPtrs(_c.Structure):
    """nabu_speller_params / nabu_speller_grads (same layout, const or not)"""
    _fields_ = [(n, _c.c_void_p) for n in ('memory_kernel', 'query_kernel', 'attention_v', 'conv_kernel',
                                           'conv_proj', 'out_kernel', 'out_bias')] + \
               [('lstm_kernel', _c.c_void_p * SPELLER_MAX_LAYERS), ('lstm_bias', _c.c_void_p * SPELLER_MAX_LAYERS)]


class AttnDesc(_c.Structure):
    _fields_ = [('size', _c.c_uint32), ('B', _c.c_int32), ('Te', _c.c_int32), ('E', _c.c_int32),
                ('U', _c.c_int32), ('kind', _c.c_int32), ('K', _c.c_int32), ('F', _c.c_int32),
                ('prob_fn', _c.c_int32)]
