"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/silero_testwav.npz from the REFERENCE's own artefacts:

* the 16 kHz AND the 8 kHz model of /root/reference/masr/infer_utils/silero_vad.onnx (weights under the names of oracle/silero.py; they are what
  the GPU tests load into the product network -- /root/reference does not exist on the GPU box),
* the speech probability of every 512-sample window of /root/reference/dataset/test.wav and the final LSTM state, from the
  operator-by-operator evaluation of that file (oracle/onnx_run.py) driven by the UNMODIFIED reference class
  masr.infer_utils.vad_predictor.VADPredictor (imported through oracle/shims.py, with a stand-in ``onnxruntime`` module whose
  InferenceSession is that evaluation -- onnxruntime itself is absent from the image: parity unpinned vs onnxruntime),
* what the reference class makes of them: ``get_speech_timestamps`` at the default and at a second parameter set, and the
  ``stream_vad`` events window by window.

    python -m oracle.make_silero_golden"""
import os
import sys
import types
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from masr_amd.utils import onnx_lite                     # noqa: E402
from oracle import shims, silero                          # noqa: E402


class Recording:
    """session wrapper that records every probability the reference class obtains"""

    def __init__(self, inner):
        self.inner, self.probs = inner, []
        self.intra_op_num_threads = self.inter_op_num_threads = 1

    def run(self, names, feeds):
        out = self.inner.run(names, feeds)
        self.probs.append(float(np.asarray(out[0]).reshape(-1)[0]))
        return out


def reference_vad(graph, **kw):
    shims.install()
    ort = types.ModuleType('onnxruntime')
    ort.InferenceSession = lambda path: Recording(silero.OracleSession(None, graph=graph))
    sys.modules['onnxruntime'] = ort
    from masr.infer_utils.vad_predictor import VADPredictor
    return VADPredictor(**kw)


def test_wav():
    with wave.open(os.path.join(REF, 'dataset', 'test.wav'), 'rb') as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    return (pcm.astype(np.float32) / 32768.0).astype(np.float32)


def main():
    graph = onnx_lite.load(os.path.join(REF, 'masr', 'infer_utils', 'silero_vad.onnx'))
    out = {'w16.' + k: np.asarray(v, np.float32) for k, v in silero.weights_from_graph(graph, 16000).items()}
    audio = test_wav()
    # make the recording less trivial for the state machine: test.wav, 0.6 s of silence, test.wav at a tenth of the level
    long_audio = np.concatenate([audio, np.zeros(9600, np.float32), 0.1 * audio]).astype(np.float32)
    out['audio_tail_silence'] = np.array([9600], np.int64)
    vad = reference_vad(graph)
    stamps = vad.get_speech_timestamps(long_audio, 16000)
    out['probs'] = np.array(vad.session.probs, np.float32)
    out['h'], out['c'] = vad._h, vad._c
    out['stamps'] = np.array([[s['start'], s['end']] for s in stamps], np.int64)
    print(len(vad.session.probs), 'windows;', stamps)
    kw = dict(threshold=0.35, min_speech_duration_ms=100, min_silence_duration_ms=300, speech_pad_ms=200)
    vad2 = reference_vad(graph, **kw)
    stamps2 = vad2.get_speech_timestamps(long_audio, 16000)
    out['stamps_alt'] = np.array([[s['start'], s['end']] for s in stamps2], np.int64)
    print(stamps2)
    vad3 = reference_vad(graph)
    events = []
    for i in range(0, len(long_audio) - 511, 512):
        ev = vad3.stream_vad(long_audio[i:i + 512], 16000)
        if ev:
            events.append((i // 512, 0 if 'start' in ev else 1, list(ev.values())[0]))
    out['stream_events'] = np.array(events, np.int64).reshape(-1, 3)
    print(events)
    # a 1536-sample window (three LSTM steps per call)
    vad4 = reference_vad(graph, window_size_samples=1536)
    vad4.get_speech_timestamps(long_audio[:64000], 16000)
    out['probs_1536'] = np.array(vad4.session.probs, np.float32)
    # the 8 kHz branch of the same file (vad_predictor.py:51,94-95: sampling_rate 8000, window 256 / 512 / 768): its weights, and the
    # reference class on every second sample of the same recording (a plain decimation -- the fixture pins arithmetic, not audio
    # quality): probabilities, segments, final state
    out.update({'w8.' + k: np.asarray(v, np.float32) for k, v in silero.weights_from_graph(graph, 8000).items()})
    audio8 = np.ascontiguousarray(long_audio[::2])
    vad8 = reference_vad(graph, window_size_samples=256)
    stamps8 = vad8.get_speech_timestamps(audio8, 8000)
    out['probs_8k'] = np.array(vad8.session.probs, np.float32)
    out['h_8k'], out['c_8k'] = vad8._h, vad8._c
    out['stamps_8k'] = np.array([[s['start'], s['end']] for s in stamps8], np.int64).reshape(-1, 2)
    print('8 kHz:', len(vad8.session.probs), 'windows;', stamps8)
    vad9 = reference_vad(graph, window_size_samples=768)
    vad9.get_speech_timestamps(audio8[:32000], 8000)
    out['probs_8k_768'] = np.array(vad9.session.probs, np.float32)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'silero_testwav.npz'), **out)
    print('written', os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'silero_testwav.npz')), 'bytes')


if __name__ == '__main__':
    main()
