"""Python handle on the HIP engine (libmasr_hip.so).  torch is used only for device memory,
streams and host<->device copies; all arithmetic of the hot path runs in the HIP kernels."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import MasrConfig, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def positional_table(max_len, d):
    """Same torch ops as the reference (conformer/embedding.py:31-37) -> bit-identical table."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def subsampled_len(t):
    return ((t - 1) // 2 - 1) // 2


def reference_gains_scalar(mean_square, target_db, max_gain_db=300.0):
    """float32 mean squares [B] -> linear gains [B] with the reference's own scalar expressions, verbatim in numpy on this host:
    ``rms_db = 10 * np.log10(mean_square)`` (audio.py:524-529), ``gain = target_db - rms_db`` and the max_gain_db check
    (:300-304), ``samples *= 10. ** (gain / 20.)`` (:256-264).  A zero mean square (digital silence) counts as 1, like the
    reference's ``rms_db`` -- the gain is then just target_dB and nothing raises; a NaN mean square (NaN samples in the input) is
    NOT special-cased there either: it propagates into the gain, and the transcript of that utterance is whatever NaN features
    decode to, as in the reference."""
    out = np.empty(len(mean_square), np.float32)
    for i, ms in enumerate(np.asarray(mean_square, np.float32)):
        if ms == 0:
            ms = 1
        rms_db = 10 * np.log10(ms)
        gain = target_db - rms_db
        if gain > max_gain_db:
            raise ValueError(f"无法将段规范化到{target_db}dB，音频增益{gain}增益已经超过max_gain_db ({max_gain_db}dB)")
        out[i] = 10. ** (min(max_gain_db, gain) / 20.)
    return out


def reference_gains(mean_square, target_db, max_gain_db=300.0):
    """``reference_gains_scalar`` for a whole batch.  The two transcendental steps -- ``np.log10(mean_square)`` and
    ``10. ** (gain / 20.)`` -- are evaluated element by element on numpy float32 SCALARS, exactly as the reference evaluates them:
    numpy's float32 ARRAY loops of log10 / power are SIMD routines (SVML on AVX-512 hosts) whose last bit differs from the scalar
    libm path on some arguments -- measured on the GPU box's EPYC 9575F: 1 of 32 gains, 273 int16 samples of that utterance off by
    one LSB.  The products, differences and quotients in between are IEEE-exact float32 operations (identical in any loop) and
    run as array operations.  128 streams: ~0.15 ms per pool step."""
    ms = np.asarray(mean_square, np.float32)
    if len(ms) <= 2 or not isinstance(target_db, (int, float)) or np.float32(target_db) != target_db:
        return reference_gains_scalar(ms, target_db, max_gain_db)     # one utterance (predict): 3 us; or a target that float32
                                                                      # does not hold: the long way
    silent = ms == 0
    lg = np.array([np.log10(x) for x in np.where(silent, np.float32(1), ms).astype(np.float32)], np.float32)
    gain = np.float32(target_db) - np.float32(10) * lg
    if (gain > max_gain_db).any():
        g = gain[gain > max_gain_db][0]
        raise ValueError(f"无法将段规范化到{target_db}dB，音频增益{g}增益已经超过max_gain_db ({max_gain_db}dB)")
    out = np.array([10. ** (x / 20.) for x in gain], np.float32)
    if silent.any():
        # digital silence: the scalar code sets ``ms = 1`` -- a Python int, so its logarithm, the gain and the power are evaluated in
        # float64 and rounded once
        out[silent] = np.float32(10. ** (min(max_gain_db, target_db - 10 * np.log10(1)) / 20.))
    return out


def _validate_encoder_conf(use_model, enc, state_dict):
    """The kernels implement exactly the module variants the shipped YAMLs select; anything else must fail here instead of
    loading cleanly into the wrong arithmetic (e.g. ``cnn_module_norm: batch_norm`` has LayerNorm-shaped ``norm.weight``)."""
    def want(key, allowed, default):
        v = enc.get(key, default)
        if v not in allowed:
            raise _lib.MasrError(f'{use_model}: encoder_conf.{key}={v!r} is not implemented (supported: {allowed})')
    if use_model in ('conformer', 'efficient_conformer'):
        # batch_norm (conformer/convolution.py:60-67): Conformer only, full-context forward only (eval-mode statistics folded)
        want('cnn_module_norm', ('layer_norm', 'batch_norm') if use_model == 'conformer' else ('layer_norm',), 'layer_norm')
        want('activation_type', ('swish',), 'swish')
        want('normalize_before', (True,), True)
        want('use_cnn_module', (True,), True)
        want('macaron_style', (True,), True)
        want('input_layer', ('conv2d',), 'conv2d')
        want('pos_enc_layer_type', ('rel_pos',), 'rel_pos')
        has_bn = state_dict is not None and any(k.endswith('conv_module.norm.running_mean') for k in state_dict)
        is_bn = enc.get('cnn_module_norm', 'layer_norm') == 'batch_norm'
        if has_bn != is_bn and state_dict is not None:
            raise _lib.MasrError(f'{use_model}: encoder_conf.cnn_module_norm={enc.get("cnn_module_norm", "layer_norm")!r} but the checkpoint '
                                 f'{"holds" if has_bn else "has no"} BatchNorm statistics (conv_module.norm.running_mean)')
    elif use_model == 'squeezeformer':
        want('cnn_norm_type', ('batch_norm',), 'batch_norm')
        want('activation_type', ('swish',), 'swish')
        want('normalize_before', (False,), False)
        want('pos_enc_layer_type', ('rel_pos',), 'rel_pos')
        want('adaptive_scale', (True,), True)
        want('dw_stride', (False,), False)


class HipEngine:
    """One engine per GPU rank.  ``state_dict`` uses the reference key names
    (``encoder.*`` / ``ctc.*``; extra keys are ignored); ``state_dict=None`` gives a weight-less
    engine that can run the model-independent kernels (fbank, argmax, CTC collapse)."""

    def __init__(self, state_dict, encoder_conf=None, vocab_size=None, streaming=True, n_mels=80, device=None,
                 max_pos=5000, use_model='conformer'):
        if not torch.cuda.is_available():
            raise _lib.MasrError('no HIP device visible to torch: the MI355X engine has no CPU fallback')
        if device is None:          # the calling rank's GPU (one process per GPU: torch.cuda.set_device(LOCAL_RANK) comes first)
            device = torch.cuda.current_device()
        device = int(device)
        self.lib = _lib.lib()
        enc = dict(encoder_conf or {})
        if vocab_size is None:
            ctc_key = 'decoder.ctc_lo.weight' if use_model == 'deepspeech2' else 'ctc.ctc_lo.weight'
            vocab_size = int(state_dict[ctc_key].shape[0]) if state_dict is not None else 1
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        _validate_encoder_conf(use_model, enc, state_dict)
        if use_model == 'conformer':
            cfg = MasrConfig(model_kind=0, d_model=int(enc.get('output_size', 256)),
                             heads=int(enc.get('attention_heads', 4)), d_ff=int(enc.get('linear_units', 2048)),
                             num_blocks=int(enc.get('num_blocks', 12)),
                             cnn_kernel=int(enc.get('cnn_module_kernel', 15)), n_mels=n_mels,
                             vocab_size=int(vocab_size), causal=1 if streaming else 0, max_pos=max_pos,
                             device_id=device)
            cfg.reserved[0] = 1 if enc.get('cnn_module_norm', 'layer_norm') == 'batch_norm' else 0
        elif use_model == 'squeezeformer':
            # configs/squeezeformer.yml: encoder_dim, feed_forward_expansion_factor, reduce_idx / recover_idx
            dim = int(enc.get('encoder_dim', 256))
            cfg = MasrConfig(model_kind=1, d_model=dim, heads=int(enc.get('attention_heads', 4)),
                             d_ff=dim * int(enc.get('feed_forward_expansion_factor', 8)),
                             num_blocks=int(enc.get('num_blocks', 12)),
                             cnn_kernel=int(enc.get('cnn_module_kernel', 31)), n_mels=n_mels,
                             vocab_size=int(vocab_size), causal=1 if streaming else 0, max_pos=max_pos,
                             device_id=device)
            red, rec = enc.get('reduce_idx', 5), enc.get('recover_idx', 11)
            cfg.reserved[0] = -1 if red is None else int(red)
            cfg.reserved[1] = -1 if rec is None else int(rec)
        elif use_model == 'efficient_conformer':
            # configs/efficient_conformer.yml: the nested efficient_conf is swallowed by **kwargs in the reference
            # (efficient_conformer/encoder.py:54); its constructor defaults equal the shipped YAML values
            eff = dict(enc.get('efficient_conf', {}) or {})
            stride_idx = eff.get('stride_layer_idx', [3])
            stride_idx = stride_idx if isinstance(stride_idx, (list, tuple)) else [stride_idx]
            groups = list(eff.get('group_layer_idx', [0, 1, 2, 3]))
            stride = eff.get('stride', [2])
            stride = list(stride) if isinstance(stride, (list, tuple)) else [stride]
            if len(stride_idx) != 1 or stride != [2] or groups != list(range(len(groups))):
                raise _lib.MasrError('efficient_conformer: only one stride-2 layer and leading grouped layers are supported')
            cfg = MasrConfig(model_kind=2, d_model=int(enc.get('output_size', 256)),
                             heads=int(enc.get('attention_heads', 4)), d_ff=int(enc.get('linear_units', 2048)),
                             num_blocks=int(enc.get('num_blocks', 12)),
                             cnn_kernel=int(enc.get('cnn_module_kernel', 15)), n_mels=n_mels,
                             vocab_size=int(vocab_size), causal=1 if streaming else 0, max_pos=max_pos,
                             device_id=device)
            cfg.reserved[0] = int(stride_idx[0])
            cfg.reserved[1] = len(groups)
            cfg.reserved[2] = int(eff.get('group_size', 3))
        elif use_model == 'deepspeech2':
            # configs/deepspeech2.yml encoder_conf: rnn_size, num_rnn_layers; streaming <=> uni-directional LSTMs
            # (deepspeech2/encoder.py:14-19: rnn_direction = 'forward' if streaming else 'bidirect')
            cfg = MasrConfig(model_kind=3, d_model=int(enc.get('rnn_size', 1024)), heads=0, d_ff=0,
                             num_blocks=int(enc.get('num_rnn_layers', 5)), cnn_kernel=0, n_mels=n_mels,
                             vocab_size=int(vocab_size), causal=1 if streaming else 0, max_pos=max_pos,
                             device_id=device)
        else:
            raise _lib.MasrError(f'use_model={use_model}: conformer, squeezeformer, efficient_conformer, deepspeech2 '
                                 f'are implemented')
        self.use_model = use_model
        self.cfg = cfg
        self.d_model, self.vocab_size, self.n_mels = cfg.d_model, cfg.vocab_size, n_mels
        # width of the encoder output rows (DeepSpeech2: rnn_size x directions)
        self.enc_dim = cfg.d_model * (1 if streaming else 2) if use_model == 'deepspeech2' else cfg.d_model
        self.num_blocks, self.heads, self.cnn_kernel = cfg.num_blocks, cfg.heads, cfg.cnn_kernel
        h = C.c_void_p()
        check(self.lib.masr_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.has_weights = state_dict is not None
        if state_dict is None:      # weight-less engine: fbank / argmax / collapse kernels only
            return
        for name, t in state_dict.items():
            if not (name.startswith('encoder.') or name.startswith('ctc.') or name.startswith('decoder.ctc_lo.')):
                continue
            self._load(name, t)
        if use_model != 'deepspeech2':
            self._load('__pos_table__', positional_table(max_pos, cfg.d_model))
        check(self.lib.masr_finalize(self.h, _stream()))

    def _load(self, name, t):
        a = np.ascontiguousarray(t.detach().cpu().float().numpy())
        shape = (C.c_int64 * a.ndim)(*a.shape)
        check(self.lib.masr_load_tensor(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def close(self):
        if getattr(self, 'h', None):
            for pool in list(self.__dict__.get('_pools', ())):        # serving pools hold streams of this engine
                pool.shutdown()
            torch.cuda.synchronize()
            self.lib.masr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- features -------------------------------------------------------------------------------
    def mean_square(self, samples, n_samples, out=None):
        """float32 ``np.mean(samples ** 2)`` per utterance in numpy's summation order (audio.py:524) -> [B] f32 (device).  The
        kernels use a scratch of the calling stream's own (not the feature launch's): preparing the next pass on a side
        stream never touches what the current pass reads."""
        B, n_max = samples.shape
        fmt = {torch.int16: 0, torch.float32: 1}[samples.dtype]
        ms = out if out is not None else torch.empty(B, dtype=torch.float32, device=self.device)
        check(self.lib.masr_mean_square(self.h, _ptr(samples), fmt, _ptr(n_samples), B, n_max, _ptr(ms), _stream()))
        return ms

    def side_stream(self, kind):
        """a side stream of THIS DEVICE owned by libmasr_hip.so (masr_side_stream; kind 0 / 1: prefix searches of consecutive
        passes, 2: per-pass preparation, 3: copies, 4: encoder passes of lane 1), as a torch stream.  One set per device, created
        with the device's first engine at default priority -- hardware queues of their own, whatever streams the process creates
        before or after (include/masr_hip.h); borrowed, never destroyed."""
        cache = self.__dict__.setdefault('_side', {})
        st = cache.get(kind)
        if st is None:
            ptr = C.c_void_p()
            check(self.lib.masr_side_stream(self.h, int(kind), C.byref(ptr)))
            st = cache[kind] = torch.cuda.ExternalStream(ptr.value, device=self.device)
        return st

    def select_lane(self, lane):
        """masr_select_lane: the calls that follow use workspace set ``lane`` (0 / 1) of this engine, so two offline passes can be
        in flight on two streams; launches sharing a lane must be ordered by their stream (include/masr_hip.h)."""
        check(self.lib.masr_select_lane(self.h, int(lane)))
        self.lane = int(lane)

    def to_host(self, t):
        """small device tensor -> numpy through a pinned buffer, waiting for THIS stream only.  (``tensor.cpu()`` copies to
        pageable memory, which the HIP runtime serialises against every stream of the device: a prefix search running on a side
        stream then stalls the next pass's features and encoder for its whole duration -- 32 ms per predict_batch call of
        BASELINE configs[2], tools/beam_batch_profile.py.)"""
        pin = self.__dict__.setdefault('_pins', {})
        buf = pin.get(t.dtype)
        if buf is None or buf.numel() < t.numel():
            buf = torch.empty(max(int(t.numel()), 256), dtype=t.dtype, pin_memory=True)
            pin[t.dtype] = buf
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return view.numpy().copy()

    def to_device(self, a):
        """small host array -> device through pinned memory (asynchronous, no device-wide serialisation).  A ring of eight pinned
        slots per dtype; every slot carries the event of its last copy (recorded on the stream the copy was issued on), which is
        waited for before the slot is rewritten -- a copy queued on a side stream behind a long prefix search keeps its source
        however many uploads follow."""
        t = torch.from_numpy(np.ascontiguousarray(a).reshape(-1))
        slot = self.__dict__.setdefault('_up_ring', {}).setdefault(t.dtype, {'bufs': [None] * 8, 'events': [None] * 8, 'turn': 0})
        k = slot['turn']
        slot['turn'] = (k + 1) & 7
        if slot['events'][k] is not None:
            slot['events'][k].synchronize()
        buf = slot['bufs'][k]
        if buf is None or buf.numel() < t.numel():
            buf = slot['bufs'][k] = torch.empty(max(t.numel(), 64), dtype=t.dtype, pin_memory=True)
        view = buf[:t.numel()]
        view.copy_(t)
        out = view.view(np.shape(a)).to(self.device, non_blocking=True)
        ev = slot['events'][k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot['events'][k] = ev
        return out

    def host_gains(self, samples, n_samples, target_db, max_gain_db=300.0):
        """The reference's normalisation gain evaluated where the reference evaluates it: the mean square comes from the
        device (bit-identical to numpy's), the scalar float32 expressions of ``rms_db`` / ``normalize`` / ``gain_db``
        (audio.py:256-264,287-304,519-529) run on THIS host's numpy, whose float32 log10 / power are not correctly rounded
        and differ between machines.  Returns linear gains [B] f32 (device); raises like ``normalize`` beyond max_gain_db."""
        return self.to_device(reference_gains(self.to_host(self.mean_square(samples, n_samples)), target_db, max_gain_db))

    def _db_mode(self, use_db_normalization, gain_in, B, return_gain):
        """-> (mode for the C ABI, gain tensor): 0 off / 1 device gains (returned in the tensor when asked) / 2 supplied gains"""
        if use_db_normalization and gain_in is not None:
            return 2, gain_in.to(device=self.device, dtype=torch.float32).contiguous().clone()
        gain = torch.ones(B, dtype=torch.float32, device=self.device) if return_gain else None
        return (1 if use_db_normalization else 0), gain

    def fbank_batch(self, samples, n_samples, use_db_normalization=True, target_db=-20.0, return_norm=False,
                    return_gain=False, gain_in=None):
        """samples int16 PCM or float32 [B, n_max] (device), n_samples int32 [B] (device)
        -> feats [B,T,80], frames [B] (+ normalised int16 samples, + linear gains).  ``gain_in`` [B]: linear gains supplied
        by the caller (``host_gains``) instead of the device's own evaluation."""
        B, n_max = samples.shape
        fmt = {torch.int16: 0, torch.float32: 1}[samples.dtype]
        T = 1 + (n_max - 400) // 160 if n_max >= 400 else 0
        feats = torch.empty(B, T, 80, dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        norm = torch.empty(B, n_max, dtype=torch.int16, device=self.device) if return_norm else None
        mode, gain = self._db_mode(use_db_normalization, gain_in, B, return_gain)
        check(self.lib.masr_fbank_batch(self.h, _ptr(samples), fmt, _ptr(n_samples), B, n_max,
                                        mode, float(target_db), _ptr(feats), _ptr(frames),
                                        _ptr(norm), _ptr(gain), _stream()))
        res = [feats, frames]
        if return_norm:
            res.append(norm)
        if return_gain:
            res.append(gain)
        return tuple(res)

    def mfcc_batch(self, samples, n_samples, n_mfcc=40, use_db_normalization=True, target_db=-20.0, return_gain=False,
                   gain_in=None):
        """kaldi.mfcc(num_mel_bins=80, num_ceps=n_mfcc) for a padded batch -> feats [B,T,n_mfcc], frames [B] (+ gains)."""
        B, n_max = samples.shape
        fmt = {torch.int16: 0, torch.float32: 1}[samples.dtype]
        T = 1 + (n_max - 400) // 160 if n_max >= 400 else 0
        feats = torch.empty(B, T, int(n_mfcc), dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        mode, gain = self._db_mode(use_db_normalization, gain_in, B, return_gain)
        check(self.lib.masr_mfcc_batch(self.h, _ptr(samples), fmt, _ptr(n_samples), B, n_max, mode,
                                       float(target_db), int(n_mfcc), _ptr(feats), _ptr(frames), _ptr(gain), _stream()))
        return (feats, frames, gain) if return_gain else (feats, frames)

    def linear_batch(self, samples, n_samples, use_db_normalization=True, target_db=-20.0, return_gain=False, gain_in=None):
        """linear log power spectrogram (20 ms / 10 ms, 161 bins) for a padded batch -> feats [B,T,161], frames [B]."""
        B, n_max = samples.shape
        fmt = {torch.int16: 0, torch.float32: 1}[samples.dtype]
        T = (n_max - 320) // 160 + 1 if n_max >= 320 else 0
        feats = torch.empty(B, T, 161, dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        mode, gain = self._db_mode(use_db_normalization, gain_in, B, return_gain)
        check(self.lib.masr_linear_batch(self.h, _ptr(samples), fmt, _ptr(n_samples), B, n_max,
                                         mode, float(target_db), _ptr(feats), _ptr(frames),
                                         _ptr(gain), _stream()))
        return (feats, frames, gain) if return_gain else (feats, frames)

    def features_batch(self, method, samples, n_samples, use_db_normalization=True, target_db=-20.0, n_mfcc=40,
                       return_gain=False, gain_in=None):
        """dispatch on the reference's ``feature_method`` (audio_featurizer.py:51-69)"""
        if method == 'fbank':
            return self.fbank_batch(samples, n_samples, use_db_normalization, target_db, return_gain=return_gain,
                                    gain_in=gain_in)
        if method == 'mfcc':
            return self.mfcc_batch(samples, n_samples, n_mfcc, use_db_normalization, target_db, return_gain=return_gain,
                                   gain_in=gain_in)
        if method == 'linear':
            return self.linear_batch(samples, n_samples, use_db_normalization, target_db, return_gain=return_gain,
                                     gain_in=gain_in)
        raise Exception('没有{}预处理方法'.format(method))

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_full(self, feats, lens, decoding_chunk_size=-1):
        """feats f32 [B,T,80] (device, zero padded), lens int32 [B] (device) -> enc [B,T',d]."""
        B, T, _ = feats.shape
        Tp = self.out_frames(T)
        enc = torch.empty(B, Tp, self.enc_dim, dtype=torch.float32, device=self.device)
        check(self.lib.masr_encode_full(self.h, _ptr(feats), _ptr(lens), B, T, int(decoding_chunk_size), _ptr(enc),
                                        _stream()))
        return enc

    def out_frames(self, T):
        """encoder output frames for T feature frames (the Efficient Conformer halves the rate once more)."""
        Tp = subsampled_len(T)
        return (Tp + 1) // 2 if getattr(self, 'use_model', 'conformer') == 'efficient_conformer' else Tp

    def enc_frames(self, feat_frames):
        """valid encoder frames per utterance for its feature frames (tensor or array; mirrors launch_frame_counts in
        elementwise.hip: Conv2dSubsampling4, once more halved with ceil by the Efficient Conformer's stride layer)."""
        n4 = ((feat_frames - 1) // 2 - 1) // 2
        n4 = n4.clamp(min=0) if torch.is_tensor(n4) else np.maximum(n4, 0)
        return (n4 + 1) // 2 if self.use_model == 'efficient_conformer' else n4

    def ctc_probs(self, enc, want_argmax=False):
        M = enc.numel() // self.enc_dim
        probs = torch.empty(*enc.shape[:-1], self.vocab_size, dtype=torch.float32, device=self.device)
        idx = torch.empty(M, dtype=torch.int32, device=self.device) if want_argmax else None
        mp = torch.empty(M, dtype=torch.float32, device=self.device) if want_argmax else None
        check(self.lib.masr_ctc_probs(self.h, _ptr(enc), M, _ptr(probs), _ptr(idx), _ptr(mp), _stream()))
        return (probs, idx, mp) if want_argmax else probs

    def ctc_greedy_frames(self, enc):
        M = enc.numel() // self.enc_dim
        idx = torch.empty(enc.shape[:-1], dtype=torch.int32, device=self.device)
        mp = torch.empty(enc.shape[:-1], dtype=torch.float32, device=self.device)
        check(self.lib.masr_ctc_greedy_frames(self.h, _ptr(enc), M, _ptr(idx), _ptr(mp), _stream()))
        return idx, mp

    def ctc_collapse(self, idx, maxp, n_frames=None, blank=0):
        B, Tp = idx.shape
        tokens = torch.empty(B, Tp, dtype=torch.int32, device=self.device)
        ntok = torch.empty(B, dtype=torch.int32, device=self.device)
        score = torch.empty(B, dtype=torch.float32, device=self.device)
        check(self.lib.masr_ctc_collapse(self.h, _ptr(idx), _ptr(maxp), _ptr(n_frames), B, Tp, blank, _ptr(tokens),
                                         _ptr(ntok), _ptr(score), _stream()))
        return tokens, ntok, score

    def argmax_rows(self, probs):
        M, V = probs.shape
        idx = torch.empty(M, dtype=torch.int32, device=self.device)
        mp = torch.empty(M, dtype=torch.float32, device=self.device)
        check(self.lib.masr_argmax_rows(self.h, _ptr(probs), M, V, _ptr(idx), _ptr(mp), _stream()))
        return idx, mp

    def transcribe_batch(self, pcm, n_samples, use_db_normalization=True, target_db=-20.0, decode_all_frames=False,
                         out=None):
        """Whole offline hot path in one call, no host sync: PCM -> token ids."""
        B, n_max = pcm.shape
        Tp = self.out_frames(1 + (n_max - 400) // 160)
        if out is None:
            out = (torch.empty(B, Tp, dtype=torch.int32, device=self.device),
                   torch.empty(B, dtype=torch.int32, device=self.device),
                   torch.empty(B, dtype=torch.float32, device=self.device))
        tokens, ntok, score = out
        check(self.lib.masr_transcribe_batch(self.h, _ptr(pcm), _ptr(n_samples), B, n_max,
                                             1 if use_db_normalization else 0, float(target_db),
                                             1 if decode_all_frames else 0, _ptr(tokens), _ptr(ntok), _ptr(score),
                                             _stream()))
        return tokens, ntok, score

    def transcribe_rows(self, samples, n_samples, use_db_normalization=True, target_db=-20.0, gain_in=None,
                        decode_all_frames=False, out=None):
        """The same pass with the facade's inputs (int16 PCM or float32 samples, optionally the caller's gains: the bit-exact
        route of ``host_gains``) and ONE packed int32 row per utterance out: rows [B, T' + 2] = tokens | count | score bits."""
        B, n_max = samples.shape
        fmt = {torch.int16: 0, torch.float32: 1}[samples.dtype]
        Tp = self.out_frames(1 + (n_max - 400) // 160)
        rows = out if out is not None else torch.empty(B, Tp + 2, dtype=torch.int32, device=self.device)
        mode = 0 if not use_db_normalization else (2 if gain_in is not None else 1)
        check(self.lib.masr_transcribe_rows(self.h, _ptr(samples), fmt, _ptr(n_samples), B, n_max, mode, float(target_db),
                                            _ptr(gain_in), 1 if decode_all_frames else 0, _ptr(rows), _stream()))
        return rows

    # ---- streaming ----------------------------------------------------------------------------------
    def stream_open(self, max_frames_out=0):
        sid = C.c_int32()
        check(self.lib.masr_stream_open(self.h, int(max_frames_out), C.byref(sid)))
        return sid.value

    def stream_reset(self, sid):
        check(self.lib.masr_stream_reset(self.h, sid))

    def stream_close(self, sid):
        check(self.lib.masr_stream_close(self.h, sid))

    def stream_set_history(self, sid, required_cache_size):
        """forward_chunk's required_cache_size for this stream: < 0 keep all cached keys, >= 0 at most that many (input-rate
        frames; Conformer, Squeezeformer and Efficient-Conformer)"""
        check(self.lib.masr_stream_set_history(self.h, sid, int(required_cache_size)))

    def stream_offset(self, sid):
        off = C.c_int32()
        check(self.lib.masr_stream_offset(self.h, sid, C.byref(off)))
        return off.value

    def encode_chunk(self, stream_ids, feats, want_probs=True, want_argmax=False):
        """feats f32 [n, Tc, 80] (device) -> probs [n, Tc', V] (+ argmax/maxprob [n, Tc'])."""
        n, Tc, _ = feats.shape
        Tq = self.out_frames(Tc)
        ids = (C.c_int32 * n)(*stream_ids)
        probs = torch.empty(n, Tq, self.vocab_size, dtype=torch.float32, device=self.device) if want_probs else None
        idx = torch.empty(n, Tq, dtype=torch.int32, device=self.device) if want_argmax else None
        mp = torch.empty(n, Tq, dtype=torch.float32, device=self.device) if want_argmax else None
        check(self.lib.masr_encode_chunk(self.h, ids, n, _ptr(feats), Tc, _ptr(probs), _ptr(idx), _ptr(mp), _stream()))
        return probs, idx, mp

    def stream_export_cache(self, sid):
        if self.use_model == 'deepspeech2':       # (h, c), each [num_rnn_layers, 1, 1, rnn_size] like the reference state
            h = torch.zeros(self.num_blocks, 1, 1, self.d_model, dtype=torch.float32, device=self.device)
            c = torch.zeros_like(h)
            check(self.lib.masr_stream_export_cache(self.h, sid, _ptr(h), _ptr(c), _stream()))
            return h, c
        n = C.c_int32(0)
        check(self.lib.masr_stream_cache_len(self.h, sid, C.byref(n)))       # att_cache.size(2) of the reference after the last step
        t = n.value
        dk = self.d_model // self.heads
        att = torch.zeros(self.num_blocks, self.heads, t, 2 * dk, dtype=torch.float32, device=self.device)
        cnn = torch.zeros(self.num_blocks, 1, self.d_model, self.cnn_kernel - 1, dtype=torch.float32, device=self.device)
        check(self.lib.masr_stream_export_cache(self.h, sid, _ptr(att), _ptr(cnn), _stream()))
        return att, cnn

    # ---- single ops / profiling -----------------------------------------------------------------------
    def op_layernorm(self, x, w, b, eps=1e-5):
        y = torch.empty_like(x)
        check(self.lib.masr_op_layernorm(self.h, _ptr(x), _ptr(w), _ptr(b), _ptr(y), x.numel() // x.shape[-1],
                                         float(eps), _stream()))
        return y

    def op_gemm(self, a, w, bias=None, res=None, act=0, alpha=1.0):
        M, K = a.shape
        N = w.shape[0]
        c = torch.empty(M, N, dtype=torch.float32, device=self.device)
        check(self.lib.masr_op_gemm(self.h, _ptr(a), _ptr(w), _ptr(bias), _ptr(res), _ptr(c), M, N, K, int(act),
                                    float(alpha), _stream()))
        return c

    def profile_select(self, kind):
        check(self.lib.masr_profile_select(self.h, int(kind)))

    def profile_read(self, reset=True):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        check(self.lib.masr_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(fl), 1 if reset else 0))
        return ms.value, n.value, fl.value
