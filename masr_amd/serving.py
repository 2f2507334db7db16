"""Multi-stream session manager: many concurrent ``predict_stream`` sessions on ONE engine (BASELINE
configs[4]).  The reference server holds one MASRPredictor -- i.e. one stream -- per websocket and decodes them one after the
other (infer_server.py:42-46,103-156).  Here every session keeps the reference's per-stream state (carried-over samples,
cached feature frames, greedy decoder history; predict.py:237-343) but the device work of all sessions that have audio
pending is done together: ONE ragged fbank launch for the new samples of all sessions, and the 67-frame windows advance in
lock-step through ``masr_encode_chunk`` (n streams per call), only the per-frame (argmax, max prob) pairs leave the CTC head
(never the [n, 16, V] probabilities), they are appended to a device-resident history [session, frame], and ONE
``masr_ctc_collapse`` launch per step turns the histories of all sessions that advanced into tokens + scores (the
reference re-decodes its python lists per session, ctc_greedy_decoder.py:52-89).  Every session receives exactly the
partial results it would get from its own ``MASRPredictor.predict_stream`` (greedy decoding; Conformer-family models and
streaming DeepSpeech2).
"""
import numpy as np
import torch

from masr_amd.data_utils.audio import AudioSegment


class _Session:
    __slots__ = ('sid', 'remained', 'cached_feat', 'row', 'frames', 'result')

    def __init__(self, sid, row):
        self.sid = sid
        self.remained = None          # float32 samples not yet turned into frames (re-normalised on every call, like the reference)
        self.cached_feat = None       # [T, 80] frames not yet consumed by a window
        self.row = row                # row of the pool's device-resident (argmax, max prob) history
        self.frames = 0               # encoder frames decoded so far
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
        self._free_rows = []
        self._hist_idx = torch.zeros(0, 0, dtype=torch.int32, device=self.engine.device)      # [rows, frames]
        self._hist_mp = torch.zeros(0, 0, dtype=torch.float32, device=self.engine.device)

    # ---- session life cycle ---------------------------------------------------------------------------------------------
    def _grow(self, rows, frames):
        """make the history tensors at least [rows, frames] (geometric growth, contents kept)"""
        r0, f0 = self._hist_idx.shape
        if rows <= r0 and frames <= f0:
            return
        r1 = r0 if rows <= r0 else max(rows, 2 * r0, 16)
        f1 = f0 if frames <= f0 else max(frames, 2 * f0, 256)
        for name, dt in (('_hist_idx', torch.int32), ('_hist_mp', torch.float32)):
            new = torch.zeros(r1, f1, dtype=dt, device=self.engine.device)
            new[:r0, :f0] = getattr(self, name)
            setattr(self, name, new)

    def open(self):
        sid = self.engine.stream_open(self.max_frames_out)
        used = {s.row for s in self.sessions.values()}
        row = self._free_rows.pop() if self._free_rows else len(used)
        self._grow(row + 1, 256)
        self.sessions[sid] = _Session(sid, row)
        return sid

    def close(self, handle):
        self.engine.stream_close(handle)
        self._free_rows.append(self.sessions.pop(handle).row)
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
                _, idx, mp = eng.encode_chunk([s.sid for s, _ in items], torch.from_numpy(x).to(eng.device), want_probs=False,
                                              want_argmax=True)
                tq = idx.shape[1]
                self._grow(0, max(s.frames for s, _ in items) + tq)
                rows = torch.tensor([s.row for s, _ in items], device=eng.device)[:, None]
                cols = torch.tensor([s.frames for s, _ in items], device=eng.device)[:, None] + \
                    torch.arange(tq, device=eng.device)[None, :]
                self._hist_idx[rows, cols] = idx                     # append this window's frames to the sessions' histories
                self._hist_mp[rows, cols] = mp
                for s, _ in items:
                    s.frames += tq
        # one collapse launch for every session that advanced: full-history best path + score (greedy_decoder_chunk semantics)
        adv = [s for s, p in zip(sess, plans) if p]
        if adv:
            rows = torch.tensor([s.row for s in adv], device=eng.device)
            tmax = max(s.frames for s in adv)
            nfr = torch.tensor([s.frames for s in adv], dtype=torch.int32, device=eng.device)
            tok, ntok, score = eng.ctc_collapse(self._hist_idx[rows, :tmax].contiguous(), self._hist_mp[rows, :tmax].contiguous(), nfr)
            tok, ntok, score = tok.cpu().numpy(), ntok.cpu().numpy(), score.cpu().numpy()
            for j, s in enumerate(adv):
                text = ''.join(self.vocab[t] for t in tok[j, :ntok[j]]).replace('<space>', ' ')
                # the score counts every non-blank frame (repeats included); with none the reference returns 0
                s.result = {'text': text, 'score': float(np.float32(score[j])) * 100.0 if ntok[j] > 0 else 0}
        out = {}
        for s, p in zip(sess, plans):
            if p:
                s.cached_feat = s.cached_feat[p[-1][1] - 3:]            # keep the 3 overlap frames (predict.py:329)
            out[s.sid] = s.result
        return out

    def reset(self, handle):
        """start a new utterance on an open session (MASRPredictor.reset_stream)"""
        self.engine.stream_reset(handle)
        self.sessions[handle] = _Session(handle, self.sessions[handle].row)
        self._fed.pop(handle, None)
