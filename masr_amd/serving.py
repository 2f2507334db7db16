"""Multi-stream session manager: many concurrent ``predict_stream`` sessions on ONE engine (BASELINE
configs[4]).  The reference server holds one MASRPredictor -- i.e. one stream -- per websocket and decodes them one after the
other (infer_server.py:42-46,103-156).  Here every session keeps the reference's per-stream state (carried-over samples,
cached feature frames, greedy decoder history; predict.py:237-343) but the device work of all sessions that have audio
pending is done together: ONE ragged fbank launch for the new samples of all sessions, and the 67-frame windows advance in
lock-step through ``masr_encode_chunk`` (n streams per call).  Every session receives exactly the partial results it would
get from its own ``MASRPredictor.predict_stream`` (greedy decoding; Conformer-family models and streaming DeepSpeech2).
"""
import numpy as np
import torch

from masr_amd.data_utils.audio import AudioSegment
from masr_amd.decoders.ctc_greedy_decoder import greedy_decoder_chunk


class _Session:
    __slots__ = ('sid', 'remained', 'cached_feat', 'last_prob', 'last_idx', 'result')

    def __init__(self, sid):
        self.sid = sid
        self.remained = None          # float32 samples not yet turned into frames (re-normalised on every call, like the reference)
        self.cached_feat = None       # [T, 80] frames not yet consumed by a window
        self.last_prob, self.last_idx = None, None
        self.result = None


class StreamPool:
    """``pool = StreamPool(predictor)`` on a streaming MASRPredictor (conformer / squeezeformer / efficient_conformer /
    deepspeech2, ``decoder: ctc_greedy``).  ``open()`` -> handle; ``feed(handle, pcm_bytes, is_end)`` queues audio; ``step()`` processes
    everything queued since the last step and returns ``{handle: {'text', 'score'} or None}`` for the sessions that were fed;
    ``close(handle)`` releases the stream."""

    def __init__(self, predictor, max_frames_out=0):
        cfg = predictor.configs
        if not cfg.streaming or not ('former' in cfg.use_model or cfg.use_model == 'deepspeech2'):
            raise Exception('StreamPool needs a streaming model (Conformer family or uni-directional DeepSpeech2)')
        if cfg.decoder != 'ctc_greedy':
            raise Exception('StreamPool decodes with ctc_greedy')
        if cfg.preprocess_conf.get('feature_method', 'fbank') != 'fbank':
            raise Exception('StreamPool batches the fbank front-end (feature_method: fbank)')
        self.predictor = predictor
        self.engine = predictor.predictor.engine
        self.vocab = predictor._text_featurizer.vocab_list
        pc = cfg.preprocess_conf
        self.use_db, self.target_db = bool(pc.use_dB_normalization), float(pc.target_dB)
        self.max_frames_out = max_frames_out
        self.sessions = {}
        self._fed = {}

    # ---- session life cycle ---------------------------------------------------------------------------------------------
    def open(self):
        sid = self.engine.stream_open(self.max_frames_out)
        self.sessions[sid] = _Session(sid)
        return sid

    def close(self, handle):
        self.engine.stream_close(handle)
        self.sessions.pop(handle)
        self._fed.pop(handle, None)

    def feed(self, handle, audio_data, is_end=False, channels=1, samp_width=2, sample_rate=16000):
        """queue raw PCM bytes (or a float / int numpy array) for a session; processed by the next ``step()``"""
        if isinstance(audio_data, np.ndarray):
            seg = AudioSegment.from_ndarray(audio_data, sample_rate)
        else:
            seg = AudioSegment.from_pcm_bytes(audio_data, channels=channels, samp_width=samp_width, sample_rate=sample_rate)
        if seg.sample_rate != 16000:
            seg.resample(16000)
        s = self.sessions[handle]
        s.remained = seg.samples if s.remained is None else np.concatenate([s.remained, seg.samples])
        self._fed[handle] = bool(is_end) or self._fed.get(handle, False)

    # ---- one batched step -------------------------------------------------------------------------------------------------
    def step(self):
        fed, self._fed = self._fed, {}
        if not fed:
            return {}
        eng = self.engine
        sess = [self.sessions[h] for h in fed]
        # 1. features of all pending samples in one ragged fbank launch (predict.py:274-281 per session)
        lens = np.array([len(s.remained) for s in sess], np.int32)
        buf = np.zeros((len(sess), max(int(lens.max()), 400)), np.float32)
        for i, s in enumerate(sess):
            buf[i, :lens[i]] = s.remained
        feats, frames, gain = eng.fbank_batch(torch.from_numpy(buf).to(eng.device), torch.from_numpy(lens).to(eng.device),
                                              self.use_db, self.target_db, return_gain=True)
        feats, frames, gain = feats.cpu().numpy(), frames.cpu().numpy(), gain.cpu().numpy()
        for i, s in enumerate(sess):
            if self.use_db and lens[i] > 0:
                s.remained = s.remained * np.float32(gain[i])          # normalised in place, like AudioSegment.normalize
            nf = int(frames[i]) if lens[i] >= 400 else 0
            new = feats[i, :nf]
            s.cached_feat = new if s.cached_feat is None else np.concatenate([s.cached_feat, new], axis=0)
            s.remained = s.remained[160 * nf:]
        # 2. windows of every session (predict.py:283-306), advanced in lock-step
        win, stride, ctx = 67, 64, 7
        plans = []
        for s in sess:
            nfr = s.cached_feat.shape[0]
            is_end = fed[s.sid]
            s.result = None
            if (nfr < win and not is_end) or nfr < ctx:
                plans.append([])
                continue
            left = ctx if is_end else win
            plans.append([(cur, min(cur + win, nfr)) for cur in range(0, nfr - left + 1, stride)])
        for k in range(max((len(p) for p in plans), default=0)):
            groups = {}
            for s, p in zip(sess, plans):
                if k < len(p):
                    groups.setdefault(p[k][1] - p[k][0], []).append((s, p[k]))
            for length, items in groups.items():          # full windows together; a short last window on its own
                x = np.stack([s.cached_feat[a:b] for s, (a, b) in items])
                probs, _, _ = eng.encode_chunk([s.sid for s, _ in items], torch.from_numpy(x).to(eng.device))
                probs = probs.cpu().numpy()
                for j, (s, _) in enumerate(items):
                    score, text, s.last_prob, s.last_idx = greedy_decoder_chunk(
                        probs_seq=probs[j], vocabulary=self.vocab, last_max_index_list=s.last_idx,
                        last_max_prob_list=s.last_prob)
                    s.result = {'text': text, 'score': score}
        out = {}
        for s, p in zip(sess, plans):
            if p:
                s.cached_feat = s.cached_feat[p[-1][1] - 3:]            # keep the 3 overlap frames (predict.py:329)
            out[s.sid] = s.result
        return out

    def reset(self, handle):
        """start a new utterance on an open session (MASRPredictor.reset_stream)"""
        self.engine.stream_reset(handle)
        self.sessions[handle] = _Session(handle)
        self._fed.pop(handle, None)
