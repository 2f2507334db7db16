"""TEST INFRASTRUCTURE ONLY -- import shims for running the UNMODIFIED reference
package (``/root/reference/masr``) on CPU in the build container.

The reference is pinned (implicitly) to numpy<2 / older torch and imports a few
packages that are absent here; none of the shims changes any arithmetic of the
hot path except ``torchaudio.compliance.kaldi.fbank`` which is replaced by the
numpy restatement in ``oracle/fbank.py`` (torchaudio is not installed).

``/root/reference`` only exists in the build container; this module is used by
``oracle/make_golden.py`` (fixture generation) and by CPU tests that are
skipped when the reference is absent.
"""
import importlib.machinery
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get('MASR_REFERENCE_ROOT', '/root/reference')


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'masr'))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)     # importlib.util.find_spec(name) must keep working on a stub
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Make ``import masr`` (the reference) work.  Idempotent."""
    if getattr(install, '_done', False):
        return
    import numpy as np
    import torch
    try:  # must be imported BEFORE the torchaudio stub exists (find_spec probing)
        import transformers  # noqa: F401
    except Exception:
        pass

    # typeguard.typechecked -> identity (conformer/encoder.py:5, convolution.py:5, loss/ctc.py:3)
    _stub('typeguard', typechecked=lambda f=None, **kw: f if f is not None else (lambda g: g))
    # absent I/O packages that the hot path never calls with ndarray / PCM input
    for name in ('av', 'resampy', 'soundfile', 'visualdl', 'Levenshtein', 'pydub', 'yeaudio'):
        if name not in sys.modules:
            _stub(name)
    sys.modules['pydub'].AudioSegment = object
    _stub('zhconv', convert=lambda s, *_a, **_k: s)
    _stub('termcolor', colored=lambda s, *_a, **_k: s)
    sys.modules['visualdl'].LogWriter = object
    # torch.nn.modules.conv.Union (squeezeformer/conv2d.py:2)
    import torch.nn.modules.conv as _conv
    if not hasattr(_conv, 'Union'):
        _conv.Union = typing.Union
    # np.sctypes (audio.py:542,567) was removed in numpy 2
    if not hasattr(np, 'sctypes'):
        np.sctypes = {'float': [np.float16, np.float32, np.float64],
                      'int': [np.int8, np.int16, np.int32, np.int64],
                      'uint': [np.uint8, np.uint16, np.uint32, np.uint64]}
    # torchaudio.compliance.kaldi.{fbank,mfcc} -> numpy restatement
    from oracle import fbank as _fb

    def _fbank(waveform, num_mel_bins=23, frame_length=25.0, frame_shift=10.0, dither=0.0,
               sample_frequency=16000.0, **_kw):
        assert frame_length == 25 and frame_shift == 10 and dither == 0.0 and sample_frequency == 16000
        x = waveform.detach().cpu().numpy()[0]
        return torch.from_numpy(_fb.kaldi_fbank(x, num_mel_bins, np.float32))

    def _mfcc(waveform, num_mel_bins=23, num_ceps=13, frame_length=25.0, frame_shift=10.0, dither=0.0,
              sample_frequency=16000.0, **_k):
        assert frame_length == 25 and frame_shift == 10 and dither == 0.0 and sample_frequency == 16000
        x = waveform.numpy()[0]
        return torch.from_numpy(_fb.kaldi_mfcc(x, num_mel_bins, num_ceps, np.float32))

    ta = _stub('torchaudio')
    comp = _stub('torchaudio.compliance')
    kaldi = _stub('torchaudio.compliance.kaldi', fbank=_fbank, mfcc=_mfcc)
    ta.compliance = comp
    comp.kaldi = kaldi

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
