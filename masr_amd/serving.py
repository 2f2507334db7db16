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
import ctypes as C
import os
import weakref

import numpy as np
import torch

from masr_amd import _lib
from masr_amd._lib import MasrError
from masr_amd.data_utils.audio import AudioSegment
from masr_amd.engine import reference_gains

_PCM_SCALE = np.float32(1.0 / 32768.0)

# chunked decoding geometry of the reference facade (predict.py:283-290): 16 encoder frames per chunk, subsampling 4, context 7
DECODING_CHUNK, SUBSAMPLING, CONTEXT = 16, 4, 7
WINDOW = (DECODING_CHUNK - 1) * SUBSAMPLING + CONTEXT      # 67 feature frames per window
STRIDE = SUBSAMPLING * DECODING_CHUNK                      # 64
OVERLAP = CONTEXT - SUBSAMPLING                            # 3 frames carried over


class _HostStage:
    """Pinned host memory for the per-call uploads of a pool (pending samples, lengths, index arrays): a bump allocator whose
    copies to the device are asynchronous on the current stream, so the host never waits for the device between the launches
    of a step.  ``begin()`` opens a call: it waits for the previous call's copies (long done) before the area is reused."""

    def __init__(self, device, nbytes=1 << 20):
        self.device = device
        self._retired = []
        self._event = None
        self._alloc(nbytes)

    def _alloc(self, nbytes):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.host = self.buf.numpy()
        self.used = 0

    def begin(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        elif self.used:                                      # a call that raised before end(): its copies may still be in flight
            torch.cuda.current_stream().synchronize()
        self._retired.clear()
        self.used = 0

    def take(self, shape, dtype):
        """-> (numpy view [shape] of dtype inside the pinned area, upload() -> device tensor of the same shape)"""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        at = (self.used + 63) & ~63
        if at + nbytes > self.host.size:                    # copies in flight keep the old area alive until the next begin()
            self._retired.append(self.buf)
            self._alloc(max(2 * self.host.size, 2 * nbytes))
            at = 0
        self.used = at + nbytes
        view = self.host[at:at + nbytes].view(dtype).reshape(shape)
        src = self.buf[at:at + nbytes].view({'float32': torch.float32, 'int32': torch.int32,
                                             'int64': torch.int64}[dtype.name]).view(*shape)
        return view, lambda: src.to(self.device, non_blocking=True)

    def put(self, array, dtype=np.int64):
        view, upload = self.take(np.shape(array), dtype)
        view[...] = array
        return upload()

    def end(self):
        self._event = torch.cuda.Event()
        self._event.record()


class _Session:
    __slots__ = ('sid', 'remained', 'fresh', 'f0', 'nf', 'row', 'frames', 'result', 'decoder', 'tokens')

    def __init__(self, sid, row, decoder=None):
        self.sid = sid
        self.remained = None          # float32 samples not yet turned into frames (re-normalised on every call, like the reference)
        self.fresh = []               # audio fed since the last step: int16 PCM views (scaled when staged) / float32 arrays
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
        self.errors = {}              # handle -> message of a session the last step left out (full stream)
        self._fed = {}
        self._free_rows = []
        self._hist_idx = torch.zeros(0, 0, dtype=torch.int32, device=self.engine.device)      # [rows, frames]
        self._hist_mp = torch.zeros(0, 0, dtype=torch.float32, device=self.engine.device)
        self.feat_dim = {'linear': 161, 'mfcc': self.n_mfcc}.get(self.method, 80)
        self._feat_cap = 512                                                                   # frames per session row
        self._feat = torch.zeros(0, self._feat_cap, self.feat_dim, dtype=torch.float32, device=self.engine.device)
        self._stage = _HostStage(self.engine.device)
        # greedy sessions: the whole step is ONE C-ABI call (masr_pool_step, csrc/pool.hip); this class keeps the handles, hands
        # over the fed buffers and builds the text.  Beam-search sessions (a device-resident prefix search per session) and
        # MASR_POOL_PY=1 (A/B) keep the python framing below.
        self._c = None
        self._feeds = []
        self._packed = None
        if not self.beam and not os.environ.get('MASR_POOL_PY'):
            self._lib = _lib.lib()
            h = C.c_void_p()
            method = {'fbank': 0, 'mfcc': 1, 'linear': 2}[self.method]
            _lib.check(self._lib.masr_pool_create(self.engine.h, method, self.n_mfcc, 1 if self.use_db else 0, float(self.target_db),
                                                  int(max_frames_out), C.byref(h)))
            self._c = h
            self._gain_error = None
            me = weakref.ref(self)                             # (no reference cycle through the callback: the pool dies with

            def gain_cb(ms_ptr, n, target_db, out_ptr, user):  #  its last reference, not at some later garbage collection)
                pool = me()
                return pool._gains(ms_ptr, n, target_db, out_ptr, user) if pool is not None else 1
            self._gain_cb = _lib.GAIN_FN(gain_cb)              # (kept alive with the pool)
            self.engine.__dict__.setdefault('_pools', weakref.WeakSet()).add(self)     # HipEngine.close() shuts its pools down first

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
        """host int64 index array -> device (staged through pinned memory, asynchronous)"""
        return self._stage.put(idx, np.int64)

    def _new_decoder(self):
        return self.predictor.beam_search_decoder.fork() if self.beam else None

    def shutdown(self):
        """destroy the C pool (closes its streams on the engine, synchronises the engine's device).  Idempotent; called by
        ``HipEngine.close()`` for every pool still alive on it, so a pool never touches a destroyed engine."""
        c, self._c = getattr(self, '_c', None), None
        self._closed = True
        if c is not None and getattr(self.engine, 'h', None):
            with torch.cuda.device(self.engine.device):
                self._lib.masr_pool_destroy(c)
        self.sessions = {}
        self._feeds = []

    def __del__(self):                                         # fallback only: owners call shutdown()
        try:
            self.shutdown()
        except Exception:                                      # noqa: BLE001  (interpreter shutdown)
            pass

    def _gains(self, ms_ptr, n, target_db, out_ptr, _user):
        """masr_gain_fn: the reference's scalar numpy expressions (audio.py:287-304,519-529) on the device's mean squares"""
        try:
            ms = np.ctypeslib.as_array(ms_ptr, (n,))
            np.ctypeslib.as_array(out_ptr, (n,))[:] = reference_gains(ms, self.target_db)
            return 0
        except Exception as exc:                               # noqa: BLE001  (re-raised by step() once the C call has returned)
            self._gain_error = exc
            return 1

    def _alive(self):
        if getattr(self, '_closed', False):
            raise MasrError('this StreamPool is shut down (its engine was closed): open a new pool on a live engine')

    def open(self):
        self._alive()
        if self._c is not None:
            h = C.c_int32()
            _lib.check(self._lib.masr_pool_open(self._c, C.byref(h)))
            self.sessions[h.value] = _Session(h.value, 0, None)
            return h.value
        sid = self.engine.stream_open(self.max_frames_out)
        used = {s.row for s in self.sessions.values()}
        row = self._free_rows.pop() if self._free_rows else len(used)
        self._grow(row + 1, 256)
        self.sessions[sid] = _Session(sid, row, self._new_decoder())
        return sid

    def close(self, handle):
        self._alive()
        if self._c is not None:
            _lib.check(self._lib.masr_pool_close(self._c, int(handle)))
            self.sessions.pop(handle)
            self.errors.pop(handle, None)
            self._feeds = [f for f in self._feeds if f[0] != handle]
            return
        self.engine.stream_close(handle)
        s = self.sessions.pop(handle)
        if s.decoder is not None:
            s.decoder.close()
        self._free_rows.append(s.row)
        self._fed.pop(handle, None)

    def reset(self, handle):
        """start a new utterance on an open session (MASRPredictor.reset_stream, predict.py:346-353)"""
        self._alive()
        if self._c is not None:
            _lib.check(self._lib.masr_pool_reset(self._c, int(handle)))
            self.sessions[handle] = _Session(handle, 0, None)
            self.errors.pop(handle, None)
            self._feeds = [f for f in self._feeds if f[0] != handle]
            return
        self.engine.stream_reset(handle)
        old = self.sessions[handle]
        if old.decoder is not None:
            old.decoder.reset_decoder()
        self.sessions[handle] = _Session(handle, old.row, old.decoder)
        self._fed.pop(handle, None)

    def feed(self, handle, audio_data, is_end=False, channels=1, samp_width=2, sample_rate=16000):
        """queue raw PCM bytes (or a float / int numpy array) for a session (predict.py:260-272); processed by the next
        ``step()``"""
        self._alive()
        s = self.sessions[handle]
        wire = isinstance(audio_data, (bytes, bytearray, memoryview)) and samp_width == 2 and channels == 1 and \
            sample_rate == self.sample_rate
        if wire and self._c is not None:
            # (the buffer is handed to masr_pool_step as it is: the array keeps the caller's bytes alive until the step has run)
            self._feeds.append((handle, np.frombuffer(bytes(audio_data) if isinstance(audio_data, bytearray) else audio_data, '<i2'),
                                0, bool(is_end)))
            return
        if wire:
            # the common wire format: kept as int16 (a view of the caller's bytes); x / 2^15 -- the float32 value
            # from_pcm_bytes produces -- happens when the step stages the samples for the device
            s.fresh.append(np.frombuffer(bytes(audio_data) if isinstance(audio_data, bytearray) else audio_data, '<i2'))
        else:
            if isinstance(audio_data, np.ndarray):
                seg = AudioSegment.from_ndarray(audio_data, sample_rate)
            elif isinstance(audio_data, (bytes, bytearray, memoryview)):
                seg = AudioSegment.from_pcm_bytes(bytes(audio_data), channels=channels, samp_width=samp_width,
                                                  sample_rate=sample_rate)
            else:
                raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')
            if seg.sample_rate != self.sample_rate:
                seg.resample(self.sample_rate)
            if self._c is not None:
                self._feeds.append((handle, np.ascontiguousarray(seg.samples, np.float32), 1, bool(is_end)))
                return
            s.fresh.append(seg.samples)
        self._fed[handle] = bool(is_end) or self._fed.get(handle, False)

    def last_tokens(self, handle):
        """token ids behind the session's last partial result (what a multi-GPU front-end gathers instead of text)"""
        return self.sessions[handle].tokens

    # ---- one batched step -------------------------------------------------------------------------------------------------
    def _featurize(self, sess):
        """features of all pending samples in one ragged launch (predict.py:274-281 per session), appended on the device to
        the sessions' rows of the feature pool; the carried-over samples are re-normalised in place on every call, exactly
        like the reference (audio.py:304).  The samples are assembled in pinned memory and copied asynchronously; nothing
        but the gains' mean squares travels back to the host."""
        eng, st = self.engine, self._stage
        n = len(sess)
        carried = [0 if s.remained is None else len(s.remained) for s in sess]
        lens = np.array([c + sum(len(a) for a in s.fresh) for c, s in zip(carried, sess)], np.int32)
        buf, upload_buf = st.take((n, max(int(lens.max()), self.min_samples)), np.float32)
        for i, s in enumerate(sess):
            row, at = buf[i], carried[i]
            if at:
                row[:at] = s.remained
            for a in s.fresh:
                if a.dtype == np.float32:
                    row[at:at + len(a)] = a
                else:                                                  # int16 PCM: x / 2^15 in float32 (buf_to_float)
                    np.multiply(a, _PCM_SCALE, out=row[at:at + len(a)])
                at += len(a)
            row[at:] = 0
        xs, ns = upload_buf(), st.put(lens, np.int32)
        gain_h = gain = None
        if self.use_db:                                                # the reference's scalar expressions on this host
            gain_h = reference_gains(eng.mean_square(xs, ns).cpu().numpy(), self.target_db)
            gain = st.put(gain_h, np.float32)
        feats, _ = eng.features_batch(self.method, xs, ns, self.use_db, self.target_db, n_mfcc=self.n_mfcc, gain_in=gain)
        # frames per session: the front-end's own count (1 + (n - window) // 160), known without asking the device
        new = np.where(lens >= self.min_samples, (lens.astype(np.int64) - self.min_samples) // 160 + 1, 0)
        tmax = feats.shape[1]
        need = max(s.nf + int(new[i]) for i, s in enumerate(sess))
        if need > self._feat_cap:                                      # a whole utterance fed in one call: wider rows
            cap = max(need, 2 * self._feat_cap)
            wider = torch.zeros(self._feat.shape[0], cap, self.feat_dim, dtype=torch.float32, device=eng.device)
            wider[:, :self._feat_cap] = self._feat
            self._feat, self._feat_cap = wider, cap
        first = np.zeros(n, np.int64)                                  # flat pool index of each session's first new frame
        for i, s in enumerate(sess):
            nf, L = int(new[i]), int(lens[i])
            s.fresh = []
            rest = buf[i, 160 * nf:L]                                  # normalised in place, like AudioSegment.normalize
            s.remained = rest * np.float32(gain_h[i]) if self.use_db and L > 0 else rest.copy()
            if s.f0 + s.nf + nf > self._feat_cap:                      # make room: the live frames move to the front of the row
                self._feat[s.row, :s.nf] = self._feat[s.row, s.f0:s.f0 + s.nf].clone()
                s.f0 = 0
            first[i] = s.row * self._feat_cap + s.f0 + s.nf
            s.nf += nf
        total = int(new.sum())
        if total:
            offs = np.arange(total) - np.repeat(np.cumsum(new) - new, new)
            moves = self._dev_index(np.stack([np.repeat(np.arange(n, dtype=np.int64) * tmax, new) + offs,
                                              np.repeat(first, new) + offs]))
            self._feat.view(-1, self.feat_dim)[moves[1]] = feats.view(-1, self.feat_dim)[moves[0]]

    @property
    def last_packed(self):
        """(device rows [n, tmax + 2] = tokens | count | score bits, tmax, handles) of the sessions that advanced in the last
        step -- what a multi-GPU front-end all-gathers (parallel.ShardedStreamPool) -- or None"""
        pk = self._packed
        if pk is not None and not torch.is_tensor(pk[0]):
            ptr, na, width, sids, host = pk
            try:            # zero-copy view of the pool's device rows (valid until the next step)
                view = type('_DevRows', (), {'__cuda_array_interface__': {'shape': (na, width), 'typestr': '<i4', 'data': (ptr, False),
                                                                         'version': 2, 'strides': None}})()
                rows = torch.as_tensor(view, device=self.engine.device)
            except Exception:                                  # noqa: BLE001
                rows = torch.from_numpy(host.copy()).to(self.engine.device)
            pk = self._packed = (rows, width - 2, sids)
        return pk

    @last_packed.setter
    def last_packed(self, value):
        self._packed = value

    def _step_c(self):
        """one masr_pool_step call over everything fed since the last step"""
        feeds, self._feeds = self._feeds, []
        self._packed = None
        if not feeds:
            return {}
        n = len(feeds)
        handles = np.fromiter((f[0] for f in feeds), np.int32, n)
        ptrs = np.fromiter((f[1].__array_interface__['data'][0] for f in feeds), np.uint64, n)
        counts = np.fromiter((f[1].shape[0] for f in feeds), np.int64, n)
        fmts = np.fromiter((f[2] for f in feeds), np.int32, n)
        ends = np.fromiter((f[3] for f in feeds), np.int32, n)
        ns, width = C.c_int32(), C.c_int32()
        h_out, st_out, rows_h, rows_d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._gain_error = None
        rc = self._lib.masr_pool_step(self._c, n, handles.ctypes.data, ptrs.ctypes.data, counts.ctypes.data, fmts.ctypes.data,
                                      ends.ctypes.data, C.cast(self._gain_cb, C.c_void_p), None, C.byref(ns), C.byref(h_out),
                                      C.byref(st_out), C.byref(rows_h), C.byref(width), C.byref(rows_d),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if self._gain_error is not None:
            raise self._gain_error
        _lib.check(rc)
        m, w = ns.value, width.value
        hs = np.ctypeslib.as_array(C.cast(h_out, C.POINTER(C.c_int32)), (m,))
        st = np.ctypeslib.as_array(C.cast(st_out, C.POINTER(C.c_int32)), (m,))
        na = int((st == 1).sum())
        rows = np.ctypeslib.as_array(C.cast(rows_h, C.POINTER(C.c_int32)), (na, w)) if na else None
        out, j, adv = {}, 0, []
        vocab = self.vocab
        for h, ok in zip(hs.tolist(), st.tolist()):
            s = self.sessions[h]
            if ok < 0:
                # left out of the step: its stream is full (max_frames_out / the positional table).  The reference asserts there
                # (embedding.py:48-50); here the OTHER sessions keep their results and this one carries the error until it is reset
                s.result = None
                out[h] = None
                self.errors[h] = 'stream exceeds max_frames_out / max_pos: reset_stream() or close it'
                continue
            if not ok:
                s.result = None
                out[h] = None
                continue
            r = rows[j]
            j += 1
            nt = int(r[w - 2])
            s.tokens = r[:nt].tolist()
            text = ''.join([vocab[t] for t in s.tokens]).replace('<space>', ' ')
            # the score counts every non-blank frame (repeats included); with none the reference returns 0
            s.result = out[h] = {'text': text, 'score': float(r[w - 1:w].view(np.float32)[0]) * 100.0 if nt > 0 else 0}
            adv.append(h)
        if na:
            self._packed = (rows_d.value, na, w, adv, rows)
        return out

    def step(self):
        self._alive()
        if self._c is not None:
            return self._step_c()
        self.last_packed = None
        fed, self._fed = self._fed, {}
        if not fed:
            return {}
        eng = self.engine
        sess = [self.sessions[h] for h in fed]
        self._stage.begin()
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
                at = self._dev_index(np.array([s.row * self._hist_idx.shape[1] + s.frames for s, _ in items],
                                              np.int64)[:, None] + np.arange(tq)[None, :])
                self._hist_idx.view(-1)[at] = idx                    # append this window's frames to the sessions' histories
                self._hist_mp.view(-1)[at] = mp
                for s, _ in items:
                    s.frames += tq
        # greedy: one collapse launch for every session that advanced -- full-history best path + score
        # (greedy_decoder_chunk semantics, ctc_greedy_decoder.py:52-89)
        adv = [s for s, p in zip(sess, plans) if p]
        if adv and not self.beam:
            rows = self._dev_index([s.row for s in adv])
            tmax = max(s.frames for s in adv)
            nfr = self._stage.put([s.frames for s in adv], np.int32)
            tok, ntok, score = eng.ctc_collapse(self._hist_idx[rows, :tmax].contiguous(), self._hist_mp[rows, :tmax].contiguous(), nfr)
            # one copy back: [tokens | count | score bits]
            packed_dev = torch.cat([tok, ntok[:, None], score.view(torch.int32)[:, None]], 1)
            self.last_packed = (packed_dev, tmax, [s.sid for s in adv])       # device copy: what a multi-GPU gather ships
            packed = packed_dev.cpu().numpy()
            tok, ntok, score = packed[:, :tmax], packed[:, tmax], packed[:, tmax + 1].copy().view(np.float32)
            for j, s in enumerate(adv):
                s.tokens = tok[j, :ntok[j]].tolist()
                text = ''.join(self.vocab[t] for t in s.tokens).replace('<space>', ' ')
                # the score counts every non-blank frame (repeats included); with none the reference returns 0
                s.result = {'text': text, 'score': float(np.float32(score[j])) * 100.0 if ntok[j] > 0 else 0}
        self._stage.end()
        out = {}
        for s, p in zip(sess, plans):
            if p:
                used = p[-1][1] - OVERLAP                                     # keep the overlap frames (predict.py:329)
                s.f0 += used
                s.nf -= used
            out[s.sid] = s.result
        return out
