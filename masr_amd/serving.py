"""Multi-stream session manager: any number of concurrent ``predict_stream`` sessions on ONE engine (SURVEY.md 8(f) rank 1,
BASELINE configs[4]); ``MASRPredictor.predict_stream`` itself is a pool with one session.

The reference server holds one MASRPredictor -- i.e. one stream -- per websocket and decodes them one after the other
(infer_server.py:42-46,103-156).  Here every session keeps the state the reference keeps per stream (samples not yet framed,
feature frames not yet consumed, decoder history; predict.py:237-343) but the device work of all sessions that have audio
pending is done together:

  * ONE ragged feature launch for the new samples of all sessions (dB normalisation gains evaluated like the reference); the
    feature frames never leave the device: they are appended to a device-resident pool [session, frame, F] and the 67-frame
    decoding windows are gathered from it;
  * the windows advance in lock-step through ``masr_encode_chunk`` (n streams per call);
  * ``decoder: ctc_greedy`` -- only the per-frame (argmax, max prob) pairs leave the CTC head (never the [n, 16, V]
    probabilities); they are appended to a device-resident history [session, frame] and ONE ``masr_ctc_collapse`` launch per
    step turns the histories of all sessions that advanced into tokens + scores (the reference re-decodes its python lists
    per session, ctc_greedy_decoder.py:52-89);
  * ``decoder: ctc_beam_search`` -- the probabilities stay on the device and feed each session's own device-resident prefix
    beam search (``BeamSearchDecoder.decode_chunk``, beam_search_decoder.py:75-91).

Every session receives exactly the partial results its own reference ``predict_stream`` loop would produce.
"""
import numpy as np
import torch

from masr_amd.data_utils.audio import AudioSegment

# chunked decoding geometry of the reference facade (predict.py:283-290): 16 encoder frames per chunk, subsampling 4, context 7
DECODING_CHUNK, SUBSAMPLING, CONTEXT = 16, 4, 7
WINDOW = (DECODING_CHUNK - 1) * SUBSAMPLING + CONTEXT      # 67 feature frames per window
STRIDE = SUBSAMPLING * DECODING_CHUNK                      # 64
OVERLAP = CONTEXT - SUBSAMPLING                            # 3 frames carried over


class _Session:
    __slots__ = ('sid', 'remained', 'f0', 'nf', 'row', 'frames', 'result', 'decoder', 'tokens')

    def __init__(self, sid, row, decoder=None):
        self.sid = sid
        self.remained = None          # float32 samples not yet turned into frames (re-normalised on every call, like the reference)
        self.f0 = 0                   # feature frames not yet consumed by a window (the reference's cached_feat): frames
        self.nf = 0                   # f0 .. f0 + nf of this session's row of the device-resident feature pool
        self.row = row                # row of the pool's device-resident feature frames and (argmax, max prob) history
        self.frames = 0               # encoder frames decoded so far
        self.result = None
        self.decoder = decoder        # ctc_beam_search: this session's own search state
        self.tokens = []              # token ids of the last partial result


class StreamPool:
    """``pool = StreamPool(predictor)`` on a streaming MASRPredictor (conformer / squeezeformer / efficient_conformer /
    uni-directional deepspeech2; ``ctc_greedy`` or ``ctc_beam_search``).  ``open()`` -> handle; ``feed(handle, pcm_bytes,
    is_end)`` queues audio; ``step()`` processes everything queued since the last step and returns ``{handle: {'text',
    'score'} or None}`` for the sessions that were fed; ``close(handle)`` releases the stream."""

    def __init__(self, predictor, max_frames_out=0):
        cfg = predictor.configs
        if not cfg.streaming or not ('former' in cfg.use_model or cfg.use_model == 'deepspeech2'):
            raise Exception(f"不支持改该模型流式识别，当前模型：{cfg.use_model}，参数streaming为：{cfg.streaming}")
        if cfg.decoder not in ('ctc_greedy', 'ctc_beam_search'):
            raise Exception(f'unknown decoder {cfg.decoder}')
        self.predictor = predictor
        self.engine = predictor.predictor.engine
        self.vocab = predictor._text_featurizer.vocab_list
        pc = cfg.preprocess_conf
        self.method = pc.get('feature_method', 'fbank')
        self.n_mfcc = int(pc.get('n_mfcc', 40))
        self.sample_rate = int(pc.get('sample_rate', 16000))
        self.use_db, self.target_db = bool(pc.use_dB_normalization), pc.target_dB
        self.min_samples = 320 if self.method == 'linear' else 400
        self.beam = cfg.decoder == 'ctc_beam_search'
        self.max_frames_out = max_frames_out
        self.sessions = {}
        self._fed = {}
        self._free_rows = []
        self._hist_idx = torch.zeros(0, 0, dtype=torch.int32, device=self.engine.device)      # [rows, frames]
        self._hist_mp = torch.zeros(0, 0, dtype=torch.float32, device=self.engine.device)
        self.feat_dim = {'linear': 161, 'mfcc': self.n_mfcc}.get(self.method, 80)
        self._feat_cap = 512                                                                   # frames per session row
        self._feat = torch.zeros(0, self._feat_cap, self.feat_dim, dtype=torch.float32, device=self.engine.device)

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
        if max(r1, rows) > self._feat.shape[0]:
            new = torch.zeros(max(r1, rows), self._feat_cap, self.feat_dim, dtype=torch.float32, device=self.engine.device)
            new[:self._feat.shape[0]] = self._feat
            self._feat = new

    def _dev_index(self, idx):
        """host int64 index array -> device (one small H2D copy)"""
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(self.engine.device)

    def _new_decoder(self):
        return self.predictor.beam_search_decoder.fork() if self.beam else None

    def open(self):
        sid = self.engine.stream_open(self.max_frames_out)
        used = {s.row for s in self.sessions.values()}
        row = self._free_rows.pop() if self._free_rows else len(used)
        self._grow(row + 1, 256)
        self.sessions[sid] = _Session(sid, row, self._new_decoder())
        return sid

    def close(self, handle):
        self.engine.stream_close(handle)
        s = self.sessions.pop(handle)
        if s.decoder is not None:
            s.decoder.close()
        self._free_rows.append(s.row)
        self._fed.pop(handle, None)

    def reset(self, handle):
        """start a new utterance on an open session (MASRPredictor.reset_stream, predict.py:346-353)"""
        self.engine.stream_reset(handle)
        old = self.sessions[handle]
        if old.decoder is not None:
            old.decoder.reset_decoder()
        self.sessions[handle] = _Session(handle, old.row, old.decoder)
        self._fed.pop(handle, None)

    def feed(self, handle, audio_data, is_end=False, channels=1, samp_width=2, sample_rate=16000):
        """queue raw PCM bytes (or a float / int numpy array) for a session (predict.py:260-272); processed by the next
        ``step()``"""
        if isinstance(audio_data, np.ndarray):
            seg = AudioSegment.from_ndarray(audio_data, sample_rate)
        elif isinstance(audio_data, (bytes, bytearray, memoryview)):
            seg = AudioSegment.from_pcm_bytes(bytes(audio_data), channels=channels, samp_width=samp_width,
                                              sample_rate=sample_rate)
        else:
            raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')
        if seg.sample_rate != self.sample_rate:
            seg.resample(self.sample_rate)
        s = self.sessions[handle]
        s.remained = seg.samples if s.remained is None else np.concatenate([s.remained, seg.samples])
        self._fed[handle] = bool(is_end) or self._fed.get(handle, False)

    def last_tokens(self, handle):
        """token ids behind the session's last partial result (what a multi-GPU front-end gathers instead of text)"""
        return self.sessions[handle].tokens

    # ---- one batched step -------------------------------------------------------------------------------------------------
    def _featurize(self, sess):
        """features of all pending samples in one ragged launch (predict.py:274-281 per session), appended on the device to
        the sessions' rows of the feature pool; the carried-over samples are re-normalised in place on every call, exactly
        like the reference (audio.py:304).  Nothing but the gains' mean squares travels back to the host."""
        eng = self.engine
        lens = np.array([len(s.remained) for s in sess], np.int32)
        buf = np.zeros((len(sess), max(int(lens.max()), self.min_samples)), np.float32)
        for i, s in enumerate(sess):
            buf[i, :lens[i]] = s.remained
        xs, ns = torch.from_numpy(buf).to(eng.device), torch.from_numpy(lens).to(eng.device)
        gain = eng.host_gains(xs, ns, self.target_db) if self.use_db else None
        feats, _ = eng.features_batch(self.method, xs, ns, self.use_db, self.target_db, n_mfcc=self.n_mfcc, gain_in=gain)
        gain = gain.cpu().numpy() if gain is not None else None
        # frames per session: the front-end's own count (1 + (n - window) // 160), known without asking the device
        new = np.where(lens >= self.min_samples, (lens.astype(np.int64) - self.min_samples) // 160 + 1, 0)
        tmax = feats.shape[1]
        need = max(s.nf + int(new[i]) for i, s in enumerate(sess))
        if need > self._feat_cap:                                      # a whole utterance fed in one call: wider rows
            cap = max(need, 2 * self._feat_cap)
            wider = torch.zeros(self._feat.shape[0], cap, self.feat_dim, dtype=torch.float32, device=eng.device)
            wider[:, :self._feat_cap] = self._feat
            self._feat, self._feat_cap = wider, cap
        src, dst = [], []
        for i, s in enumerate(sess):
            if self.use_db and lens[i] > 0:
                s.remained = s.remained * np.float32(gain[i])          # normalised in place, like AudioSegment.normalize
            nf = int(new[i])
            if s.f0 + s.nf + nf > self._feat_cap:                      # make room: the live frames move to the front of the row
                self._feat[s.row, :s.nf] = self._feat[s.row, s.f0:s.f0 + s.nf].clone()
                s.f0 = 0
            if nf:
                src.append(i * tmax + np.arange(nf))
                dst.append(s.row * self._feat_cap + s.f0 + s.nf + np.arange(nf))
                s.nf += nf
            s.remained = s.remained[160 * nf:]
        if src:
            flat = self._feat.view(-1, self.feat_dim)
            flat[self._dev_index(np.concatenate(dst))] = feats.view(-1, self.feat_dim)[self._dev_index(np.concatenate(src))]

    def step(self):
        fed, self._fed = self._fed, {}
        if not fed:
            return {}
        eng = self.engine
        sess = [self.sessions[h] for h in fed]
        self._featurize(sess)
        # windows of every session (predict.py:283-306), advanced in lock-step
        plans = []
        for s in sess:
            nfr = s.nf
            is_end = fed[s.sid]
            s.result = None
            if (nfr < WINDOW and not is_end) or nfr < CONTEXT:
                plans.append([])
                continue
            left = CONTEXT if is_end else WINDOW
            plans.append([(cur, min(cur + WINDOW, nfr)) for cur in range(0, nfr - left + 1, STRIDE)])
        for k in range(max((len(p) for p in plans), default=0)):
            groups = {}
            for s, p in zip(sess, plans):
                if k < len(p):
                    groups.setdefault(p[k][1] - p[k][0], []).append((s, p[k]))
            for length, items in groups.items():          # full windows together; a short last window on its own
                # the windows are gathered from the device-resident feature pool: [items, length] flat frame indices
                base = np.array([s.row * self._feat_cap + s.f0 + a for s, (a, _) in items], np.int64)
                x = self._feat.view(-1, self.feat_dim)[self._dev_index(base[:, None] + np.arange(length)[None, :])]
                sids = [s.sid for s, _ in items]
                if self.beam:
                    probs, _, _ = eng.encode_chunk(sids, x, want_probs=True)
                    for i, (s, _) in enumerate(items):
                        score, text = s.decoder.decode_chunk(probs=probs[i:i + 1], logits_lens=[probs.shape[1]])
                        s.frames += probs.shape[1]
                        s.result = {'text': text, 'score': score}
                        s.tokens = list(s.decoder.last_tokens)
                    continue
                _, idx, mp = eng.encode_chunk(sids, x, want_probs=False, want_argmax=True)
                tq = idx.shape[1]
                self._grow(0, max(s.frames for s, _ in items) + tq)
                rows = torch.tensor([s.row for s, _ in items], device=eng.device)[:, None]
                cols = torch.tensor([s.frames for s, _ in items], device=eng.device)[:, None] + \
                    torch.arange(tq, device=eng.device)[None, :]
                self._hist_idx[rows, cols] = idx                     # append this window's frames to the sessions' histories
                self._hist_mp[rows, cols] = mp
                for s, _ in items:
                    s.frames += tq
        # greedy: one collapse launch for every session that advanced -- full-history best path + score
        # (greedy_decoder_chunk semantics, ctc_greedy_decoder.py:52-89)
        adv = [s for s, p in zip(sess, plans) if p]
        if adv and not self.beam:
            rows = torch.tensor([s.row for s in adv], device=eng.device)
            tmax = max(s.frames for s in adv)
            nfr = torch.tensor([s.frames for s in adv], dtype=torch.int32, device=eng.device)
            tok, ntok, score = eng.ctc_collapse(self._hist_idx[rows, :tmax].contiguous(), self._hist_mp[rows, :tmax].contiguous(), nfr)
            tok, ntok, score = tok.cpu().numpy(), ntok.cpu().numpy(), score.cpu().numpy()
            for j, s in enumerate(adv):
                s.tokens = tok[j, :ntok[j]].tolist()
                text = ''.join(self.vocab[t] for t in s.tokens).replace('<space>', ' ')
                # the score counts every non-blank frame (repeats included); with none the reference returns 0
                s.result = {'text': text, 'score': float(np.float32(score[j])) * 100.0 if ntok[j] > 0 else 0}
        out = {}
        for s, p in zip(sess, plans):
            if p:
                used = p[-1][1] - OVERLAP                                     # keep the overlap frames (predict.py:329)
                s.f0 += used
                s.nf -= used
            out[s.sid] = s.result
        return out
