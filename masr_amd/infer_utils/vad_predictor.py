"""Voice-activity segmentation for ``MASRPredictor.predict_long``.

The reference cuts long audio with Silero VAD, a third-party ONNX model run through onnxruntime
(masr/infer_utils/vad_predictor.py:13-175, predict.py:195-234).  Neither the model nor onnxruntime belongs to the MI355X hot
path, so ``predict_long`` takes ANY object with the reference's interface

    get_speech_timestamps(audio: np.float32[N], sampling_rate: int) -> [{'start': int, 'end': int}, ...]   (sample indices)

-- the reference's own ``VADPredictor`` can be passed unchanged where onnxruntime is installed.  ``EnergyVAD`` below is the
built-in stand-in: short-time energy against an adaptive noise floor with hysteresis, minimum speech / silence durations and
padding (the same knobs as the reference class, :19-35).  It is NOT Silero and will not cut at the same samples.
"""
import numpy as np


class EnergyVAD(object):
    def __init__(self, threshold_db=12.0, min_speech_duration_ms=250, min_silence_duration_ms=100, window_size_samples=512,
                 speech_pad_ms=30, max_speech_duration_s=30.0):
        self.threshold_db = float(threshold_db)          # frame is speech when this far above the noise floor
        self.min_speech_duration_ms = min_speech_duration_ms
        self.min_silence_duration_ms = min_silence_duration_ms
        self.window_size_samples = int(window_size_samples)
        self.speech_pad_ms = speech_pad_ms
        self.max_speech_duration_s = float(max_speech_duration_s)

    def frame_levels_db(self, audio):
        w = self.window_size_samples
        n = (len(audio) + w - 1) // w
        x = np.zeros(n * w, np.float32)
        x[:len(audio)] = audio
        e = np.mean(x.reshape(n, w).astype(np.float64) ** 2, axis=1)
        return 10.0 * np.log10(e + 1e-12)

    def get_speech_timestamps(self, audio, sampling_rate):
        audio = np.asarray(audio, np.float32)
        total = len(audio)
        if total == 0:
            return []
        w = self.window_size_samples
        level = self.frame_levels_db(audio)
        floor = np.percentile(level, 10)                      # noise floor of this recording
        on = level >= floor + self.threshold_db
        off = level < floor + self.threshold_db - 3.0         # hysteresis
        min_speech = sampling_rate * self.min_speech_duration_ms / 1000
        min_silence = sampling_rate * self.min_silence_duration_ms / 1000
        pad = int(sampling_rate * self.speech_pad_ms / 1000)
        max_len = int(self.max_speech_duration_s * sampling_rate)
        # pass 1: raw regions with hysteresis and minimum-silence bridging
        regions, start, quiet_from = [], None, None
        for i in range(len(level)):
            pos = i * w
            if start is None:
                if on[i]:
                    start, quiet_from = pos, None
                continue
            if off[i]:
                if quiet_from is None:
                    quiet_from = pos
                if pos + w - quiet_from >= min_silence:
                    regions.append([start, quiet_from])
                    start, quiet_from = None, None
            elif on[i]:
                quiet_from = None
        if start is not None:
            regions.append([start, total if quiet_from is None else quiet_from])
        regions = [r for r in regions if r[1] - r[0] >= min_speech]
        # pass 2: padding (never across the midpoint of the gap to the neighbour), then the length cap
        out = []
        for k, (s, e) in enumerate(regions):
            lo = 0 if k == 0 else (regions[k - 1][1] + s) // 2
            hi = total if k == len(regions) - 1 else (e + regions[k + 1][0]) // 2
            s, e = max(lo, s - pad), min(hi, e + pad)
            # the encoder's positional table is finite: cap the segment length -- cut an over-long region into equal pieces
            # (never a full-length piece plus a remainder too short to recognise)
            pieces = max(1, -(-(e - s) // max_len))
            edges = [s + (e - s) * k // pieces for k in range(pieces + 1)]
            out.extend({'start': int(a), 'end': int(b)} for a, b in zip(edges, edges[1:]))
        return out
