"""CPU: masr_amd's VADPredictor (segmentation logic of the reference's Silero wrapper, network pluggable) against the
REAL reference class (masr/infer_utils/vad_predictor.py, imported unmodified through oracle/shims with a stub ``onnxruntime``
whose InferenceSession replays scripted speech probabilities).  Skipped where /root/reference is absent."""
import sys
import types

import numpy as np
import pytest

from oracle import shims

pytestmark = pytest.mark.skipif(not shims.reference_available(), reason='reference checkout not present')


class ScriptedSession:
    """onnxruntime.InferenceSession stand-in: the i-th call returns probs[i]; (h, c) count the calls so that state handling
    is visible; checks the input contract of the Silero graph"""

    def __init__(self, probs):
        self.probs, self.i = list(probs), 0

    def run(self, _names, inputs):
        assert set(inputs) == {'input', 'h', 'c', 'sr'} and inputs['input'].ndim == 2
        assert inputs['h'].shape == (2, inputs['input'].shape[0], 64) and inputs['sr'].dtype == np.int64
        p = self.probs[self.i % len(self.probs)]
        self.i += 1
        return np.array([[p]], np.float32), inputs['h'] + 1, inputs['c'] + 1


@pytest.fixture(scope='module')
def RefVAD():
    shims.install()
    ort = types.ModuleType('onnxruntime')
    ort.InferenceSession = lambda path: ScriptedSession([0.0])
    sys.modules['onnxruntime'] = ort
    from masr.infer_utils.vad_predictor import VADPredictor
    return VADPredictor


def _pair(RefVAD, probs, **kw):
    from masr_amd.infer_utils.vad_predictor import VADPredictor
    ref = RefVAD(path='unused.onnx', **kw)
    ref.session = ScriptedSession(probs)
    return ref, VADPredictor(session=ScriptedSession(probs), **kw)


def _scripts():
    rng = np.random.default_rng(0)
    yield [0.0] * 40
    yield [0.9] * 40
    yield [0.1] * 5 + [0.9] * 20 + [0.1] * 3 + [0.9] * 10 + [0.1] * 30          # a gap shorter than min_silence is bridged
    yield [0.9] * 3 + [0.1] * 30 + [0.9] * 30                                    # a blip below min_speech is dropped
    yield [0.9] * 20 + [0.4] * 30 + [0.9] * 20                                   # between the two thresholds: stays open
    yield [0.1] * 10 + [0.9] * 20 + [0.2] * 6 + [0.9] * 20 + [0.2] * 2 + [0.9] * 20 + [0.0] * 9    # neighbours share padding
    for _ in range(40):                                                          # random walks through the hysteresis band
        n = int(rng.integers(20, 400))
        x = np.clip(np.cumsum(rng.normal(0, 0.18, n)) % 1.3 - 0.15, 0, 1)
        yield x.tolist()
    for _ in range(20):
        yield rng.choice([0.05, 0.3, 0.45, 0.5, 0.95], size=int(rng.integers(5, 300))).tolist()


@pytest.mark.parametrize('kw', [{}, {'threshold': 0.35, 'min_speech_duration_ms': 100, 'min_silence_duration_ms': 300,
                                     'speech_pad_ms': 200},
                                {'window_size_samples': 1024, 'speech_pad_ms': 0, 'min_silence_duration_ms': 0}])
def test_get_speech_timestamps_equals_reference(RefVAD, kw):
    w = kw.get('window_size_samples', 512)
    for k, probs in enumerate(_scripts()):
        n = len(probs) * w - (k * 37) % w                      # the last window is usually partial (zero padded)
        audio = np.zeros(max(n, 1), np.float32)
        ref, mine = _pair(RefVAD, probs, **kw)
        want = ref.get_speech_timestamps(audio, 16000)
        got = mine.get_speech_timestamps(audio, 16000)
        assert got == want, (k, kw, got, want)
        assert mine.session.i == ref.session.i                 # one network call per window, state carried between them
        assert np.array_equal(mine._h, ref._h)


def test_stream_vad_equals_reference(RefVAD):
    rng = np.random.default_rng(3)
    for probs in list(_scripts())[:30]:
        for secs in (False, True):
            ref, mine = _pair(RefVAD, probs)
            for _ in probs:
                x = rng.normal(0, 0.1, 512).astype(np.float32)
                assert mine.stream_vad(x, 16000, return_seconds=secs) == ref.stream_vad(x, 16000, return_seconds=secs)
            assert (mine.triggered, mine.temp_end, mine.current_sample) == (ref.triggered, ref.temp_end, ref.current_sample)
    ref, mine = _pair(RefVAD, [0.9])
    assert mine.stream_vad(np.zeros(100, np.float32), 16000) is None is ref.stream_vad(np.zeros(100, np.float32), 16000)


def test_input_validation_equals_reference(RefVAD):
    ref, mine = _pair(RefVAD, [0.7])
    for x, sr in ((np.zeros(512, np.float32), 16000), (np.zeros((1, 256), np.float32), 8000), (np.zeros(100, np.float32), 16000),
                  (np.zeros(512, np.float32), 44100), (np.zeros((1, 1, 512), np.float32), 16000)):
        try:
            want = ref(x, sr)
        except ValueError:
            with pytest.raises(ValueError):
                mine(x, sr)
            continue
        except Exception:
            with pytest.raises(Exception):      # (the reference's own message formatting fails for > 2 dimensions)
                mine(x, sr)
            continue
        assert np.array_equal(mine(x, sr), want)


def test_constructor_without_network_fails_loudly():
    from masr_amd.infer_utils.vad_predictor import VADPredictor
    sys.modules.pop('onnxruntime', None)
    with pytest.raises(Exception):
        VADPredictor()
