"""ctypes binding of libmasr_hip.so (C ABI: include/masr_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this
module raises, and so does everything that imports it.
"""
import ctypes as C
import os

from .build import LIB_PATH

_lib = None


class MasrConfig(C.Structure):
    _fields_ = [('model_kind', C.c_int32), ('d_model', C.c_int32), ('heads', C.c_int32), ('d_ff', C.c_int32),
                ('num_blocks', C.c_int32), ('cnn_kernel', C.c_int32), ('n_mels', C.c_int32),
                ('vocab_size', C.c_int32), ('causal', C.c_int32), ('max_pos', C.c_int32), ('device_id', C.c_int32),
                ('reserved', C.c_int32 * 5)]


class MasrError(RuntimeError):
    pass


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/masr_hip.h one to one
SIGNATURES = {
    'masr_last_error': [],
    'masr_version': [],
    'masr_create': [C.POINTER(MasrConfig), C.POINTER(_P)],
    'masr_destroy': [_P],
    'masr_load_tensor': [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I],
    'masr_finalize': [_P, _P],
    'masr_fbank_batch': [_P, _P, _I, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    'masr_encode_full': [_P, _P, _P, _I, _I, _I, _P, _P],
    'masr_ctc_probs': [_P, _P, _I, _P, _P, _P, _P],
    'masr_ctc_greedy_frames': [_P, _P, _I, _P, _P, _P],
    'masr_ctc_collapse': [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    'masr_argmax_rows': [_P, _P, _I, _I, _P, _P, _P],
    'masr_ctc_topk': [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    'masr_ctc_topk_blank': [_P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    'masr_beam_create': [_I, _I, C.POINTER(_P)],
    'masr_beam_destroy': [_P],
    'masr_beam_reset': [_P],
    'masr_beam_advance': [_P, _P, _P, _P, _I, _I],
    'masr_beam_advance_lm': [_P, _P, _P, _P, _P, _I, _I],
    'masr_beam_result': [_P, _P, _I, C.POINTER(_I), C.POINTER(_F)],
    'masr_beam_search_batch': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    'masr_beam_search_gpu': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    'masr_gbeam_open': [_P, _I, _I, _I, C.POINTER(_I)],
    'masr_gbeam_advance': [_P, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P],
    'masr_gbeam_advance_lm': [_P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P],
    'masr_gbeam_reset': [_P, _I],
    'masr_gbeam_close': [_P, _I],
    'masr_gbeam_set_lm': [_P, _I, _P, _F, _F],
    'masr_beam_set_lm': [_P, _P, _F, _F],
    'masr_beam_search_batch_lm': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _F, _P, _P, _I, _P, _P],
    'masr_beam_search_gpu_lm': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _F, _F, _P, _P, _I, _P, _P, _P],
    'masr_lm_load_arpa': [C.c_char_p, C.POINTER(C.c_char_p), _I, C.POINTER(_P)],
    'masr_lm_destroy': [_P],
    'masr_lm_last_error': [],
    'masr_lm_info': [_P, C.POINTER(_I), C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(C.c_int64)],
    'masr_lm_cond_log_prob': [_P, C.POINTER(_I), _I, C.POINTER(_F)],
    'masr_lm_sentence_log_prob': [_P, C.POINTER(_I), _I, C.POINTER(_F)],
    'masr_lm_word_id': [_P, C.c_char_p, C.POINTER(_I)],
    'masr_lm_dict_size': [_P, C.POINTER(_I)],
    'masr_vad_create': [_I, C.POINTER(_P)],
    'masr_vad_destroy': [_P],
    'masr_vad_last_error': [],
    'masr_vad_load_tensor': [_P, _I, C.c_char_p, _P, C.c_int64],
    'masr_vad_finalize': [_P, _I],
    'masr_vad_forward': [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P],
    'masr_mean_square': [_P, _P, _I, _P, _I, _I, _P, _P],
    'masr_mfcc_batch': [_P, _P, _I, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P],
    'masr_linear_batch': [_P, _P, _I, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    'masr_transcribe_batch': [_P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P],
    'masr_transcribe_rows': [_P, _P, _I, _P, _I, _I, _I, _F, _P, _I, _P, _P],
    'masr_pool_create': [_P, _I, _I, _I, _F, _I, C.POINTER(_P)],
    'masr_pool_destroy': [_P],
    'masr_pool_open': [_P, C.POINTER(_I)],
    'masr_pool_close': [_P, _I],
    'masr_pool_reset': [_P, _I],
    'masr_pool_step': [_P, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_I), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                       C.POINTER(_I), C.POINTER(_P), _P],
    'masr_pool_profile': [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), _I],
    'masr_encoder_frames': [_P, _I, C.POINTER(_I)],
    'masr_engine_info': [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    'masr_stream_open': [_P, _I, C.POINTER(_I)],
    'masr_stream_reset': [_P, _I],
    'masr_stream_close': [_P, _I],
    'masr_stream_offset': [_P, _I, C.POINTER(_I)],
    'masr_stream_cache_len': [_P, _I, C.POINTER(_I)],
    'masr_stream_room': [_P, _I, C.POINTER(_I)],
    'masr_resample_f32': [_P, C.c_int64, C.c_double, _P, _P, C.c_int64, _I, _P, C.c_int64],
    'masr_stream_set_history': [_P, _I, _I],
    'masr_encode_chunk': [_P, C.POINTER(_I), _I, _P, _I, _P, _P, _P, _P],
    'masr_stream_export_cache': [_P, _I, _P, _P, _P],
    'masr_op_layernorm': [_P, _P, _P, _P, _P, _I, _F, _P],
    'masr_op_gemm': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    'masr_side_stream': [_P, _I, C.POINTER(_P)],
    'masr_select_lane': [_P, _I],
    'masr_stage_rows': [_P, C.c_int64, _P, _P, _I, _I, _I],
    'masr_debug_set': [_P, _I, _I],
    'masr_profile_select': [_P, _I],
    'masr_profile_read': [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), _I],
}
_RESTYPE = {'masr_last_error': C.c_char_p, 'masr_destroy': None, 'masr_beam_destroy': None, 'masr_lm_destroy': None,
            'masr_lm_last_error': C.c_char_p, 'masr_vad_last_error': C.c_char_p, 'masr_vad_destroy': None, 'masr_pool_destroy': None}


# int gain_fn(const float* mean_square, int32 n, float target_db, float* gain_out, void* user)
GAIN_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_float), C.c_int32, C.c_float, C.POINTER(C.c_float), C.c_void_p)


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (same soname): import it FIRST so that this library binds
    # to the HIP runtime torch already loaded -- two runtimes in one process cannot both see the GPU
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MasrError(f'{LIB_PATH} not found: build it with `python -m masr_amd.build` '
                        f'(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    h = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = h
    return h


def check(rc):
    if rc != 0:
        raise MasrError(lib().masr_last_error().decode('utf-8', 'replace'))
