"""Voice-activity segmentation for ``MASRPredictor.predict_long``.

The reference cuts long audio with Silero VAD: a small ONNX network (``silero_vad.onnx``, shipped next to its vad_predictor.py and
run through onnxruntime) that maps a 512-sample window + LSTM state to a speech probability, followed by a hysteresis /
minimum-duration / padding state machine in plain Python (masr/infer_utils/vad_predictor.py:13-175, predict.py:195-234).  Here:

* ``VADPredictor`` -- the reference class's interface and its segmentation logic (``get_speech_timestamps`` :106-175,
  ``stream_vad`` :177-216, ``reset_states`` :73-81, the chunk validation :53-71), pinned against the reference class on scripted
  probability sequences and on ``dataset/test.wav`` (tests/test_vad_cpu.py, tests/test_silero.py).  The network behind it is a
  ``session`` with onnxruntime's ``run(None, {'input', 'h', 'c', 'sr'}) -> (prob, h, c)`` signature; the default is
  ``infer_utils.silero_vad.SileroVAD``: the network on the GPU (csrc/silero.hip) with the weights read out of the user's copy of
  the ONNX file (no onnxruntime involved).  When the session can score a whole recording at once (``speech_probs``: two launches
  instead of one session.run per window) ``get_speech_timestamps`` uses that -- same probabilities, same state afterwards.
  Without a model file the class refuses to construct; it never guesses probabilities.
* ``EnergyVAD`` -- a stand-in for callers WITHOUT the Silero file (explicit opt-in: ``predict_long(vad_predictor=EnergyVAD())``):
  short-time energy against an adaptive noise floor with hysteresis, minimum speech / silence durations and padding (the same knobs
  as the reference class, :19-35).  It is NOT Silero and will not cut at the same samples.

``predict_long`` takes ANY object with ``get_speech_timestamps(audio: np.float32[N], sampling_rate) -> [{'start', 'end'}]``.
"""
import os

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


class VADPredictor(object):
    """masr/infer_utils/vad_predictor.py:11-216 with a pluggable network ``session`` (see the module docstring)."""

    def __init__(self, path=None, threshold: float = 0.5, min_speech_duration_ms: int = 250, min_silence_duration_ms: int = 100,
                 window_size_samples: int = 512, speech_pad_ms: int = 30, session=None):
        if session is None:
            # vad_predictor.py:33-36 builds an onnxruntime session on silero_vad.onnx; here the same file feeds the GPU network
            from masr_amd.infer_utils.silero_vad import SileroVAD
            session = SileroVAD(path)
        self.session = session
        self.threshold = threshold
        self.min_speech_duration_ms = min_speech_duration_ms
        self.min_silence_duration_ms = min_silence_duration_ms
        self.window_size_samples = window_size_samples
        self.speech_pad_ms = speech_pad_ms
        self.sample_rates = [8000, 16000]
        self.reset_states()

    # ---- network call (:53-104) -------------------------------------------------------------------------------------------
    def reset_states(self, batch_size=1):
        self._h = np.zeros((2, batch_size, 64), np.float32)
        self._c = np.zeros((2, batch_size, 64), np.float32)
        self._last_sr = 0
        self._last_batch_size = 0
        self.triggered = False
        self.temp_end = 0
        self.current_sample = 0

    def _validate_input(self, x, sr):
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[np.newaxis, :]
        if x.ndim > 2:
            raise ValueError(f'Too many dimensions for input audio chunk {x.ndim}')
        if sr != 16000 and sr % 16000 == 0:
            x = x[::sr // 16000]           # (the reference strides the FIRST axis here, :61-64 -- kept as it is)
            sr = 16000
        if sr not in self.sample_rates:
            raise ValueError(f'Supported sampling rates: {self.sample_rates} (or multiply of 16000)')
        if sr / x.shape[1] > 31.25:
            raise ValueError('Input audio chunk is too short')
        return x, sr

    def __call__(self, x, sr):
        x, sr = self._validate_input(x, sr)
        batch = x.shape[0]
        if not self._last_batch_size or (self._last_sr and self._last_sr != sr) or self._last_batch_size != batch:
            self.reset_states(batch)
        out, self._h, self._c = self.session.run(None, {'input': x, 'h': self._h, 'c': self._c,
                                                        'sr': np.array(sr, dtype=np.int64)})
        self._last_sr, self._last_batch_size = sr, batch
        return out

    # ---- offline segmentation (:106-175) ------------------------------------------------------------------------------------
    def speech_probabilities(self, audio, sampling_rate):
        """the per-window loop of get_speech_timestamps (:122-129).  A session that scores whole recordings (SileroVAD) does it
        in one call; the state bookkeeping of ``__call__`` is kept identical (same self._h / self._c afterwards)."""
        w = self.window_size_samples
        batched = getattr(self.session, 'speech_probs', None)
        if batched is not None and len(audio) > 0:
            head = np.asarray(audio[:w])
            if len(head) < w:
                head = np.pad(head, (0, w - len(head)))
            _, sr = self._validate_input(head, sampling_rate)
            if sr == sampling_rate:                          # (a multiple of 16 kHz goes through the per-window path as written)
                if not self._last_batch_size or (self._last_sr and self._last_sr != sr) or self._last_batch_size != 1:
                    self.reset_states(1)
                probs, self._h, self._c = batched(audio, sr, w, self._h, self._c)
                self._last_sr, self._last_batch_size = sr, 1
                return [float(p) for p in probs]
        probs = []
        for start in range(0, len(audio), w):
            chunk = audio[start:start + w]
            if len(chunk) < w:
                chunk = np.pad(chunk, (0, int(w - len(chunk))))
            probs.append(self(chunk, sampling_rate).item())
        return probs

    def get_speech_timestamps(self, audio, sampling_rate):
        """audio np.float32 [N] -> [{'start': sample, 'end': sample}]: a window opens a segment at >= threshold; the segment
        closes at the first window below (threshold - 0.15) that is followed by min_silence of such windows without another
        >= threshold window in between; segments not longer than min_speech are dropped; then padding, shared gaps halved."""
        self.reset_states()
        total = len(audio)
        w = self.window_size_samples
        min_speech = sampling_rate * self.min_speech_duration_ms / 1000
        min_silence = sampling_rate * self.min_silence_duration_ms / 1000
        pad = sampling_rate * self.speech_pad_ms / 1000
        low = self.threshold - 0.15
        segments, start, quiet_at = [], None, 0
        for i, p in enumerate(self.speech_probabilities(audio, sampling_rate)):
            pos = w * i
            if p >= self.threshold:
                quiet_at = 0
                if start is None:
                    start = pos
                continue
            if start is None or p >= low:
                continue
            if not quiet_at:
                quiet_at = pos
            if pos - quiet_at >= min_silence:
                if quiet_at - start > min_speech:
                    segments.append({'start': start, 'end': quiet_at})
                start, quiet_at = None, 0
        if start is not None and total - start > min_speech:
            segments.append({'start': start, 'end': total})
        for k, seg in enumerate(segments):
            if k == 0:
                seg['start'] = int(max(0, seg['start'] - pad))
            if k == len(segments) - 1:
                seg['end'] = int(min(total, seg['end'] + pad))
                continue
            nxt = segments[k + 1]
            gap = nxt['start'] - seg['end']
            if gap < 2 * pad:
                seg['end'] += int(gap // 2)
                nxt['start'] = int(max(0, nxt['start'] - gap // 2))
            else:
                seg['end'] = int(min(total, seg['end'] + pad))
                nxt['start'] = int(max(0, nxt['start'] - pad))
        return segments

    # ---- streaming (:177-216) ---------------------------------------------------------------------------------------------------
    def stream_vad(self, x, sampling_rate, return_seconds=False):
        """one window at a time -> {'start': t} when speech begins, {'end': t} when it has ended, else None"""
        if len(x) < self.window_size_samples:
            return None
        x = np.asarray(x)
        min_silence = sampling_rate * self.min_silence_duration_ms / 1000
        pad = sampling_rate * self.speech_pad_ms / 1000
        self.current_sample += x.shape[-1]
        p = self(x, sampling_rate).item()
        stamp = lambda t: round(t / sampling_rate, 1) if return_seconds else int(t)
        if p >= self.threshold:
            self.temp_end = 0
            if not self.triggered:
                self.triggered = True
                return {'start': stamp(self.current_sample - pad)}
            return None
        if p < self.threshold - 0.15 and self.triggered:
            if not self.temp_end:
                self.temp_end = self.current_sample
            if self.current_sample - self.temp_end >= min_silence:
                end = self.temp_end + pad
                self.temp_end, self.triggered = 0, False
                return {'end': stamp(end)}
        return None
