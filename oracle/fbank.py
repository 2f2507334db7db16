"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the feature front-end.

Restates, in numpy:

* ``AudioSegment`` sample handling of the reference
  (``masr/data_utils/audio.py``): int->float32 (``:532-546``), RMS-dB
  normalisation (``normalize`` ``:287-304``, ``gain_db`` ``:256-264``,
  ``rms_db`` ``:519-529``) and float32->int16 conversion with clip + C
  truncation (``_convert_samples_from_float32`` ``:549-574``).
* ``torchaudio.compliance.kaldi.fbank`` as called by
  ``masr/data_utils/featurizer/audio_featurizer.py:120-138``
  (``num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0,
  sample_frequency=16000``; everything else torchaudio defaults:
  povey window, remove_dc_offset, preemphasis 0.97, round_to_power_of_two,
  snip_edges, use_power, use_log_fbank, low_freq 20, high_freq 0 -> Nyquist).

torchaudio (third-party; the reference pins no version, docs/install.md:7
recommends torchaudio==2.0.2) is NOT vendored in the reference and NOT
installed in the build container, and the reference ships no fbank vectors:
**parity unpinned** at the reference level.  ``tests/test_oracle_golden.py``
(``test_fbank_cross_check_transformers``) cross-checks this restatement against ``transformers.audio_utils`` (an
independent numpy Kaldi mimic).
* ``torchaudio.compliance.kaldi.mfcc`` as called by ``audio_featurizer.py:98-117`` (``num_mel_bins=n_mels=80,
  num_ceps=n_mfcc``; torchaudio defaults ``cepstral_lifter=22, use_energy=False``): the fbank above times torchaudio's
  orthonormal DCT-II matrix, times the lifter.  Same status: **parity unpinned**, cross-checked against ``scipy.fft.dct``.
* ``AudioFeaturizer._compute_linear`` (``audio_featurizer.py:73-95``, in-tree numpy): restated AND pinned -- the fixture
  ``tests/golden/features.npz`` is the output of the reference's own function (``oracle/make_golden.py --only-features``).
"""
import math

import numpy as np

EPS_F32 = np.float32(1.1920928955078125e-07)  # torch.finfo(torch.float32).eps


# --------------------------------------------------------------------------
# AudioSegment logic
# --------------------------------------------------------------------------
def pcm16_to_float32(pcm: np.ndarray) -> np.ndarray:
    """audio.py:532-546 -- int16 -> float32 / 2**15."""
    assert pcm.dtype == np.int16
    out = pcm.astype('float32')
    out *= (1. / 2 ** 15)
    return out


def rms_db(samples: np.ndarray) -> float:
    """audio.py:519-529."""
    mean_square = np.mean(samples ** 2)
    if mean_square == 0:
        mean_square = 1
    return 10 * np.log10(mean_square)


def db_normalize(samples: np.ndarray, target_db=-20, max_gain_db=300.0) -> np.ndarray:
    """audio.py:287-304 + :256-264 (in place in the reference; returns a copy here)."""
    gain = target_db - rms_db(samples)
    if gain > max_gain_db:
        raise ValueError("gain exceeds max_gain_db")
    out = samples.copy()
    out *= 10. ** (min(max_gain_db, gain) / 20.)
    return out


def float32_to_int16(samples: np.ndarray) -> np.ndarray:
    """audio.py:549-574 -- *32768, clip to [-32768, 32767], astype (truncation)."""
    out = samples.copy()
    out *= (2 ** 15 / 1.)
    out[out > 32767] = 32767
    out[out < -32768] = -32768
    return out.astype(np.int16)


# --------------------------------------------------------------------------
# Kaldi fbank
# --------------------------------------------------------------------------
def povey_window(n=400, dtype=np.float64):
    """torch.hann_window(n, periodic=False).pow(0.85)."""
    k = np.arange(n, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * math.pi * k / (n - 1))
    return (hann ** 0.85).astype(dtype)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins=80, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0,
              dtype=np.float32):
    """torchaudio ``get_mel_banks`` (no VTLN).  Returns [num_bins, padded//2 + 1]
    (last column zero, as fbank() pads it).  Computed in ``dtype`` like torch does
    (float32 tensors; scalars in python float64)."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=dtype)[:, None]
    left = (dtype(mel_low) + b * dtype(delta)).astype(dtype)
    center = (dtype(mel_low) + (b + dtype(1.0)) * dtype(delta)).astype(dtype)
    right = (dtype(mel_low) + (b + dtype(2.0)) * dtype(delta)).astype(dtype)
    freqs = (dtype(fft_bin_width) * np.arange(num_fft_bins, dtype=dtype)).astype(dtype)
    mel = (dtype(1127.0) * np.log(dtype(1.0) + freqs / dtype(700.0))).astype(dtype)[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = np.maximum(dtype(0.0), np.minimum(up, down)).astype(dtype)
    return np.concatenate([bins, np.zeros((num_bins, 1), dtype)], axis=1)


def num_frames(n_samples, win=400, hop=160):
    """snip_edges=True framing."""
    if n_samples < win:
        return 0
    return 1 + (n_samples - win) // hop


def kaldi_fbank(waveform: np.ndarray, num_mel_bins=80, dtype=np.float32) -> np.ndarray:
    """waveform: 1-D array of int16-valued samples (any numeric dtype).
    Returns [m, num_mel_bins] in ``dtype`` (float32 mimics torch; float64 is the
    high-precision statement of the same algorithm)."""
    win, hop, padded = 400, 160, 512
    x = np.asarray(waveform).astype(dtype)
    m = num_frames(x.shape[0], win, hop)
    if m == 0:
        return np.zeros((0, num_mel_bins), dtype)
    idx = np.arange(win)[None, :] + hop * np.arange(m)[:, None]
    fr = x[idx]                                            # _get_strided
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=dtype)  # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)  # replicate pad left
    fr = fr - dtype(0.97) * prev                           # pre-emphasis
    fr = fr * povey_window(win, dtype)[None, :]
    fr = np.concatenate([fr, np.zeros((m, padded - win), dtype)], axis=1)
    spec = np.fft.rfft(fr.astype(np.float64) if dtype == np.float64 else fr, axis=1)
    if dtype == np.float32:
        spec = spec.astype(np.complex64)
    power = (np.abs(spec) ** 2).astype(dtype)
    banks = mel_banks(num_mel_bins, padded, dtype=np.float32).astype(dtype)
    mel = power @ banks.T
    eps = dtype(EPS_F32)
    return np.log(np.maximum(mel, eps)).astype(dtype)


def featurize_pcm16(pcm: np.ndarray, use_db_normalization=True, target_db=-20, dtype=np.float32):
    """AudioFeaturizer.featurize (audio_featurizer.py:37-69) for 16 kHz int16 PCM:
    /32768 -> dB-normalise -> int16 -> fbank.  Returns (feats[T,80], int16 samples)."""
    s = pcm16_to_float32(pcm)
    if use_db_normalization:
        s = db_normalize(s, target_db)
    i16 = float32_to_int16(s)
    return kaldi_fbank(i16, 80, dtype), i16


# --------------------------------------------------------------------------
# kaldi.mfcc (torchaudio/compliance/kaldi.py: mfcc, _get_dct_matrix, _get_lifter_coeffs; functional.create_dct)
# --------------------------------------------------------------------------
def dct_matrix(num_ceps=40, num_mel_bins=80):
    """[num_mel_bins, num_ceps] float32: create_dct(n_mels, n_mels, 'ortho') with column 0 := sqrt(1/n_mels), first
    num_ceps columns.  float32 arithmetic in torch's order (python scalars multiply float32 tensors)."""
    n = np.arange(num_mel_bins, dtype=np.float32)
    k = np.arange(num_mel_bins, dtype=np.float32)[:, None]
    dct = np.cos((np.float32(math.pi / float(num_mel_bins)) * (n + np.float32(0.5))) * k).astype(np.float32)   # [k, n]
    dct[0] *= np.float32(1.0 / math.sqrt(2.0))
    dct *= np.float32(math.sqrt(2.0 / float(num_mel_bins)))
    dct = dct.T.copy()                                          # [n_mels, n_mfcc] (right-multiply form)
    dct[:, 0] = np.float32(math.sqrt(1 / float(num_mel_bins)))
    return dct[:, :num_ceps]


def lifter_coeffs(num_ceps=40, cepstral_lifter=22.0):
    i = np.arange(num_ceps).astype(np.float32)
    return (np.float32(1.0) + np.float32(0.5 * cepstral_lifter) * np.sin(np.float32(math.pi) * i / np.float32(cepstral_lifter))
            ).astype(np.float32)


def kaldi_mfcc(waveform, num_mel_bins=80, num_ceps=40, dtype=np.float32):
    """waveform: int16-valued samples -> [m, num_ceps]"""
    fb = kaldi_fbank(waveform, num_mel_bins, dtype)
    out = fb @ dct_matrix(num_ceps, num_mel_bins).astype(dtype)
    out = out * lifter_coeffs(num_ceps).astype(dtype)[None, :]
    return out.astype(dtype)


# --------------------------------------------------------------------------
# linear log power spectrogram (audio_featurizer.py:73-95)
# --------------------------------------------------------------------------
def linear_spectrogram(samples, sample_rate=16000, frame_shift=10.0, frame_length=20.0, eps=1e-14):
    """samples: float32 in [-1, 1] (AudioSegment.samples after normalize) -> float64 [T, 161]"""
    stride = int(0.001 * sample_rate * frame_shift)
    window = int(0.001 * sample_rate * frame_length)
    n = len(samples)
    if n < window:
        return np.zeros((0, window // 2 + 1))
    t = (n - window) // stride + 1                                            # (:76-79: the tail that fills no frame is cut)
    frames = np.stack([samples[i * stride:i * stride + window] for i in range(t)], axis=1)      # [window, T]
    w = np.hanning(window)[:, None]
    spec = np.abs(np.fft.rfft(frames * w, axis=0)) ** 2
    scale = np.sum(w ** 2) * sample_rate
    spec[1:-1, :] *= 2.0 / scale
    spec[(0, -1), :] /= scale
    return np.log(spec + eps).T


def featurize_samples(samples_f32, method='fbank', use_db_normalization=True, target_db=-20, n_mfcc=40):
    """AudioFeaturizer.featurize (audio_featurizer.py:36-69) on float32 samples for any feature_method -> float32 [T, D]"""
    s = np.asarray(samples_f32, np.float32)
    if use_db_normalization:
        s = db_normalize(s, target_db)
    if method == 'linear':
        return linear_spectrogram(s).astype(np.float32)
    i16 = float32_to_int16(s)
    if method == 'mfcc':
        return kaldi_mfcc(i16, 80, n_mfcc)
    return kaldi_fbank(i16, 80)
