"""BeamSearchDecoder with the reference's constructor / method contract
(masr/decoders/beam_search_decoder.py:9-96).  The reference hands the search to the third-party SWIG
module ``paddlespeech_ctcdecoders`` (+ a KenLM language model); here the per-frame vocabulary pruning runs
on the GPU (masr_ctc_topk) and so does the CTC prefix beam search of whole utterances
(masr_beam_search_gpu_lm: one workgroup per utterance) and of streams (masr_gbeam_*: the live prefixes stay on the device
between decode_chunk calls); sizes beyond the kernel's limits use the host-thread search inside libmasr_hip.so
(masr_beam_*).

External scorer (alpha, beta): ``language_model_path`` is read as an ARPA n-gram model (decoders/lm_scorer.py; KenLM's binary
.klm format is not parsed -> an error that says so).  Character-based models (every LM word one character, the reference's
Mandarin models) are applied inside the search on the GPU table or the host table: a prefix extended by a character adds
alpha * ln P_LM + beta.  Word-based models (an LM word longer than one character, the English configurations): the scorer owns
the spelling dictionary, a word is scored when its space arrives and the unfinished last word at the end of the utterance;
that search runs on host threads.  With a scorer bound the decoder's min_cutoff / full_beam pruning applies (ln p(blank) of
every frame comes out of the pruning kernel).  The returned score is the decoder's approx_ctc.

A missing language model file: the reference downloads its default model and otherwise ASSERTS
(beam_search_decoder.py:19-28) -- there is no network here, so the assertion is what a caller sees.  The scorer-free search
(alpha = beta = 0, no pruning rule) is still available, but only when asked for: ``language_model_path=None``."""
import ctypes as C
import logging
import os

import numpy as np
import torch

from masr_amd import _lib, runtime
from masr_amd._lib import check

logger = logging.getLogger(__name__)


class BeamSearchDecoder:
    def __init__(self, alpha, beta, beam_size, cutoff_prob, cutoff_top_n, vocab_list, num_processes=10, blank_id=0,
                 language_model_path='lm/zh_giga.no_cna_cmn.prune01244.klm'):
        self.alpha, self.beta = alpha, beta
        self.beam_size = int(beam_size)
        self.cutoff_prob = float(cutoff_prob)
        self.cutoff_top_n = int(cutoff_top_n)
        self.vocab_list = vocab_list
        self.num_processes = int(num_processes)
        self.blank_id = int(blank_id)
        self.use_gpu_search = True       # False: prefix search on host threads (masr_beam_search_batch)
        self._gstream = None             # device-resident streaming search (masr_gbeam_*), opened on first use
        self._geng = None                # ... and the (per-device) auxiliary engine that owns it
        self._gout = None
        self.last_tokens = []            # token ids of the last decode_chunk result
        self._lib = _lib.lib()
        self._ext_scorer = None
        self.prune_min_cutoff = True     # the decoder's min_cutoff / full_beam rule (only ever applies with a scorer bound)
        if language_model_path is None:
            # explicit scorer-free mode (not a reference configuration: its decoder always owns a Scorer)
            logger.info('masr_amd BeamSearchDecoder: language_model_path=None, searching with the acoustic CTC scores only')
            self.alpha = self.beta = 0
        else:
            import os
            # beam_search_decoder.py:28 (the download of the default model that precedes it needs a network)
            assert os.path.exists(language_model_path), f'语言模型不存在：{language_model_path}'
            from masr_amd.decoders.lm_scorer import LanguageModel
            self._ext_scorer = LanguageModel(language_model_path, vocab_list)
            logger.info(f'language model: model path = {language_model_path}, is_character_based = '
                        f'{self._ext_scorer.is_character_based}, max_order = {self._ext_scorer.get_max_order()}, '
                        f'dict_size = {self._ext_scorer.get_dict_size()}')
        h = C.c_void_p()
        if self._lib.masr_beam_create(self.beam_size, self.blank_id, C.byref(h)) != 0:
            raise _lib.MasrError('masr_beam_create failed')
        self._stream = h
        self._bind_host_lm()

    def _lm_args(self):
        """(lm handle or NULL, alpha, beta) for the *_lm entry points"""
        if self._ext_scorer is None:
            return C.c_void_p(0), C.c_float(0.0), C.c_float(0.0)
        return self._ext_scorer.h, C.c_float(float(self.alpha)), C.c_float(float(self.beta))

    def _bind_host_lm(self):
        if self._ext_scorer is not None:
            self._lib.masr_beam_set_lm(self._stream, *self._lm_args())

    def fork(self):
        """a decoder with the same configuration (and the same language model tables) but its own streaming search state --
        one per concurrent ``predict_stream`` session (serving.StreamPool)"""
        other = object.__new__(BeamSearchDecoder)
        other.__dict__.update(self.__dict__)
        other._gstream, other._geng, other._gout, other.last_tokens = None, None, None, []
        h = C.c_void_p()
        if self._lib.masr_beam_create(self.beam_size, self.blank_id, C.byref(h)) != 0:
            raise _lib.MasrError('masr_beam_create failed')
        other._stream = h
        other._bind_host_lm()
        return other

    def close(self):
        """release the streaming search state (host trie + device-resident beam)"""
        if getattr(self, '_gstream', None) is not None:
            check(self._lib.masr_gbeam_close(self._geng.h, self._gstream))
            self._gstream = self._geng = None
        if getattr(self, '_stream', None):
            self._lib.masr_beam_destroy(self._stream)
            self._stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- GPU: per-frame candidate pruning ----------------------------------------------------------
    def _candidates(self, probs, to_host=True):
        """probs np/torch [M, V] -> idx [M,K] int32, logp [M,K] f32, count [M] int32, blank_lp [M] f32 or None, K (host arrays,
        or device tensors).  blank_lp = ln p(blank) per frame, produced when a scorer is bound (the pruning rule's input)."""
        # device-resident probabilities are searched on the GPU they live on (one worker thread per GPU, server.WorkerRouter)
        eng = runtime.aux_engine(probs.device if torch.is_tensor(probs) and probs.is_cuda else None)
        p = torch.as_tensor(np.asarray(probs) if not torch.is_tensor(probs) else probs, dtype=torch.float32)
        p = p.to(eng.device).contiguous()
        M, V = p.shape
        K = min(self.cutoff_top_n, V)
        idx = torch.zeros(M, K, dtype=torch.int32, device=eng.device)
        logp = torch.zeros(M, K, dtype=torch.float32, device=eng.device)
        cnt = torch.zeros(M, dtype=torch.int32, device=eng.device)
        blp = None
        if self._ext_scorer is not None and self.prune_min_cutoff:
            blp = torch.zeros(max(M, 1), dtype=torch.float32, device=eng.device)
        if M:
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if blp is None:
                check(self._lib.masr_ctc_topk(eng.h, C.c_void_p(p.data_ptr()), M, V, K, C.c_float(self.cutoff_prob),
                                              C.c_void_p(idx.data_ptr()), C.c_void_p(logp.data_ptr()),
                                              C.c_void_p(cnt.data_ptr()), stream))
            else:
                check(self._lib.masr_ctc_topk_blank(eng.h, C.c_void_p(p.data_ptr()), M, V, K, C.c_float(self.cutoff_prob),
                                                    self.blank_id, C.c_void_p(idx.data_ptr()), C.c_void_p(logp.data_ptr()),
                                                    C.c_void_p(cnt.data_ptr()), C.c_void_p(blp.data_ptr()), stream))
        if not to_host:
            return idx, logp, cnt, blp, K
        return idx.cpu().numpy(), logp.cpu().numpy(), cnt.cpu().numpy(), None if blp is None else blp.cpu().numpy(), K

    @staticmethod
    def _ptr(x):
        """device tensor / host array / None -> void* for the C ABI"""
        if x is None:
            return C.c_void_p(0)
        return C.c_void_p(x.data_ptr()) if torch.is_tensor(x) else x.ctypes.data_as(C.c_void_p)

    def gpu_search_supported(self, T, V):
        """limits of masr_beam_search_gpu (include/masr_hip.h): LDS-resident entry table and 32-bit trie keys; word-based
        scorers (spelling dictionary) are searched on host threads."""
        K = min(self.cutoff_top_n, V)
        lm = self._ext_scorer is not None
        if lm and not self._ext_scorer.is_character_based:
            return False
        # beam_gpu_lds_bytes (beam_gpu.hip): extension keys + survivor list, 29 (+ 15 with a scorer) words of live-prefix state,
        # fixed tables
        return (K <= 64 and 2 <= self.beam_size <= 512 and
                self.beam_size * K * 6 + (176 if lm else 116) * self.beam_size + 24752 <= 160 * 1024)

    def _text(self, toks):
        return ''.join(self.vocab_list[t] for t in toks).replace('<space>', ' ')

    # ---- reference API -----------------------------------------------------------------------------
    def decode_beam_search_offline(self, probs_split):
        """one utterance: probs [T, V] -> (score, text)  (beam_search_decoder.py:45-56)."""
        score, text = self._batch([np.asarray(probs_split)])[0]
        return score, text

    def decode_batch_beam_search_offline(self, probs_split):
        """list of [T_i, V] -> list of texts (beam_search_decoder.py:59-73), num_processes host threads."""
        return [t for _, t in self._batch([np.asarray(p) for p in probs_split])]

    def _pinned_out(self, n):
        """a pinned host buffer of >= n int32 for the results of one deferred search, from a small free list"""
        pool = self.__dict__.setdefault('_pin_free', [])
        for i, b in enumerate(pool):
            if b.numel() >= n:
                return pool.pop(i)
        return torch.empty(max(n, 1 << 15), dtype=torch.int32, pin_memory=True)

    def _batch_collect(self, pending, want_tokens=False):
        """results of a ``_batch(..., defer=True)`` launch.  The host waits for the EVENT behind this pass's search, then copies
        the packed rows (tokens | length | score bits, int32) back on the library's copy stream and waits for that copy only.
        Round 6 measured the three simpler ways and why they stall: through the current (main) stream the first of three passes
        came back at 35 instead of 15 ms (the encoders of the later passes are queued there); an asynchronous copy enqueued right
        behind the search kernel is a DMA-engine command that waits 6 ms and holds up the next pass's upload on the same engine
        (the host sat 17 ms in the next pass's preparation); a copy enqueued on the search's stream at collection time queues
        behind the search of pass k + 2, which shares that stream."""
        _, B, max_len, packed, _keep, done = pending
        host = self._pinned_out(B * (max_len + 2))
        done.synchronize()                           # the search (and the packing behind it) of THIS pass
        with torch.cuda.stream(runtime.aux_engine(packed.device).side_stream(3)):          # the library's copy stream: idle
            host[:B * (max_len + 2)].copy_(packed.view(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        ev.synchronize()
        r = host[:B * (max_len + 2)].view(B, max_len + 2).numpy()
        toks, lens, scores = r[:, :max_len], r[:, max_len].copy(), r[:, max_len + 1].copy().view(np.float32)
        res_tok = [toks[i, :lens[i]].tolist() for i in range(B)]
        self._pin_free.append(host)
        if want_tokens:
            return [(res_tok[i], float(scores[i])) for i in range(B)]
        return [(float(scores[i]), self._text(res_tok[i])) for i in range(B)]

    def _batch(self, probs_list, defer=False, want_tokens=False, frames=None):
        """``defer=True`` (device-resident probabilities, GPU search): launch pruning + search on the current torch stream and
        return a handle for ``_batch_collect`` without synchronising -- lets the caller run the search of one sub-batch on a
        side stream while the encoder works on the next one (the search occupies one workgroup per utterance).
        ``frames`` given: ``probs_list`` is ONE padded [B, Ts, V] array / device tensor (what the CTC head of a device pass
        leaves behind) with ``frames[i]`` valid rows per utterance -- searched in place, no restacking (round 6: the 32 slice
        copies + the fill of a 134 MB buffer were 3 ms on the critical path of every BASELINE configs[2] pass)."""
        if frames is not None:
            stacked = probs_list
            B, Ts, V = stacked.shape
            frames = np.asarray(frames, np.int32)
            assert frames.shape == (B,) and (B == 0 or int(frames.max()) <= Ts)
            probs_list = stacked
        else:
            B = len(probs_list)
            frames = np.array([p.shape[0] for p in probs_list], np.int32)
            Ts = int(frames.max()) if B else 0
            V = probs_list[0].shape[1]
            if torch.is_tensor(probs_list[0]):          # device-resident probabilities: no host round trip
                stacked = torch.zeros(B, Ts, V, dtype=torch.float32, device=probs_list[0].device)
            else:
                stacked = np.zeros((B, Ts, V), np.float32)
            for i, p in enumerate(probs_list):
                stacked[i, :p.shape[0]] = p
        max_len = max(Ts, 1)
        if B and Ts and self.use_gpu_search and self.gpu_search_supported(Ts, V):
            # whole search on the device: candidates never leave HBM, one workgroup per utterance
            eng = runtime.aux_engine(stacked.device if torch.is_tensor(stacked) else None)
            idx, logp, cnt, blp, K = self._candidates(stacked.reshape(B * Ts, V), to_host=False)
            fr = eng.to_device(frames)
            toks = torch.zeros(B, max_len, dtype=torch.int32, device=eng.device)
            lens = torch.zeros(B, dtype=torch.int32, device=eng.device)
            scores = torch.zeros(B, dtype=torch.float32, device=eng.device)
            check(self._lib.masr_beam_search_gpu_lm(eng.h, C.c_void_p(idx.data_ptr()), C.c_void_p(logp.data_ptr()),
                                                    C.c_void_p(cnt.data_ptr()), C.c_void_p(fr.data_ptr()), B, Ts, K,
                                                    self.beam_size, self.blank_id, *self._lm_args(), self._ptr(blp),
                                                    C.c_void_p(toks.data_ptr()), max_len,
                                                    C.c_void_p(lens.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            if defer:                      # nothing has been synchronised: _batch_collect() fetches the result later
                packed = torch.cat([toks, lens[:, None], scores.view(torch.int32)[:, None]], 1)       # one contiguous copy back
                done = torch.cuda.Event()
                done.record()
                return ('gpu', B, max_len, packed, (stacked, idx, logp, cnt, blp, fr, probs_list, toks, lens, scores), done)
            toks, lens, scores = toks.cpu().numpy(), lens.cpu().numpy(), scores.cpu().numpy()
            if want_tokens:          # (token ids, score): what a multi-GPU caller gathers instead of text
                return [(toks[i, :lens[i]].tolist(), float(scores[i])) for i in range(B)]
            return [(float(scores[i]), self._text(toks[i, :lens[i]])) for i in range(B)]
        if defer:
            raise Exception('deferred batch search needs the GPU search (device-resident probabilities within its limits)')
        if self._ext_scorer is not None and not self._ext_scorer.is_character_based and not getattr(self, '_said_host', False):
            self._said_host = True
            logger.info('masr_amd BeamSearchDecoder: word-based language model -> prefix search on host threads')
        elif self.use_gpu_search and B and Ts and not getattr(self, '_said_host', False):
            self._said_host = True
            logger.warning(f'masr_amd BeamSearchDecoder: beam_size={self.beam_size} x cutoff_top_n={self.cutoff_top_n} is beyond the '
                           f'GPU search (LDS), using {self.num_processes} host threads')
        idx, logp, cnt, blp, K = self._candidates(stacked.reshape(B * Ts, V))
        toks = np.zeros((B, max_len), np.int32)
        lens = np.zeros(B, np.int32)
        scores = np.zeros(B, np.float32)
        rc = self._lib.masr_beam_search_batch_lm(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                                 cnt.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p), B, Ts, K,
                                                 self.beam_size, self.blank_id, self.num_processes, *self._lm_args(),
                                                 self._ptr(blp), toks.ctypes.data_as(C.c_void_p), max_len,
                                                 lens.ctypes.data_as(C.c_void_p),
                                                 scores.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise _lib.MasrError('masr_beam_search_batch failed')
        if want_tokens:
            return [(toks[i, :lens[i]].tolist(), float(scores[i])) for i in range(B)]
        return [(float(scores[i]), self._text(toks[i, :lens[i]])) for i in range(B)]

    # ---- one search for the utterances of several device passes ---------------------------------------------------------------
    def prune_padded(self, probs, frames):
        """vocabulary pruning of ONE device pass (padded probabilities [B, Ts, V] on the device, ``frames[i]`` valid rows) on the
        current stream; the probabilities are not needed afterwards.  -> a part for ``search_pruned``."""
        B, Ts, V = probs.shape
        idx, logp, cnt, blp, K = self._candidates(probs.reshape(B * Ts, V), to_host=False)
        return {'idx': idx.view(B, Ts, K), 'logp': logp.view(B, Ts, K), 'cnt': cnt.view(B, Ts), 'blp': None if blp is None else blp.view(B, Ts),
                'frames': np.asarray(frames, np.int32), 'B': B, 'Ts': Ts, 'K': K}

    def search_pruned(self, parts):
        """ONE prefix-search launch over the utterances of every part (one workgroup per utterance): the candidates of the passes
        are gathered into common [B_total, T_max, K] arrays (a few MB) and searched together, on the current stream.  Returns a
        handle for ``_batch_collect``; results come back in the order of the parts.  (Round 6: a search that runs BESIDE an
        encoder pass holds 32 compute units for milliseconds, and every row-block launch of that encoder pass -- sized for one
        round of 256 -- then takes two rounds; searched together behind the encoders of the group, all utterances advance at the
        kernel's stand-alone rate and the call is encoders + the longest utterance's search.)"""
        eng = runtime.aux_engine(parts[0]['idx'].device)
        Bt, Tm, K = sum(p['B'] for p in parts), max(p['Ts'] for p in parts), parts[0]['K']
        if len(parts) == 1:
            p = parts[0]
            idx, logp, cnt, blp = p['idx'], p['logp'], p['cnt'], p['blp']
        else:
            idx = torch.zeros(Bt, Tm, K, dtype=torch.int32, device=eng.device)
            logp = torch.zeros(Bt, Tm, K, dtype=torch.float32, device=eng.device)
            cnt = torch.zeros(Bt, Tm, dtype=torch.int32, device=eng.device)
            blp = None if parts[0]['blp'] is None else torch.zeros(Bt, Tm, dtype=torch.float32, device=eng.device)
            lo = 0
            for p in parts:
                idx[lo:lo + p['B'], :p['Ts']] = p['idx']
                logp[lo:lo + p['B'], :p['Ts']] = p['logp']
                cnt[lo:lo + p['B'], :p['Ts']] = p['cnt']
                if blp is not None:
                    blp[lo:lo + p['B'], :p['Ts']] = p['blp']
                lo += p['B']
        frames = np.concatenate([p['frames'] for p in parts]).astype(np.int32)
        fr = eng.to_device(frames)
        max_len = max(Tm, 1)
        toks = torch.zeros(Bt, max_len, dtype=torch.int32, device=eng.device)
        lens = torch.zeros(Bt, dtype=torch.int32, device=eng.device)
        scores = torch.zeros(Bt, dtype=torch.float32, device=eng.device)
        check(self._lib.masr_beam_search_gpu_lm(eng.h, C.c_void_p(idx.data_ptr()), C.c_void_p(logp.data_ptr()),
                                                C.c_void_p(cnt.data_ptr()), C.c_void_p(fr.data_ptr()), Bt, Tm, K,
                                                self.beam_size, self.blank_id, *self._lm_args(), self._ptr(blp),
                                                C.c_void_p(toks.data_ptr()), max_len,
                                                C.c_void_p(lens.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        packed = torch.cat([toks, lens[:, None], scores.view(torch.int32)[:, None]], 1)
        done = torch.cuda.Event()
        done.record()
        return ('gpu', Bt, max_len, packed, (idx, logp, cnt, blp, fr, toks, lens, scores, parts), done)

    # ---- host-thread search of a device pass, deferred ---------------------------------------------------------------------------
    def search_host_deferred(self, probs, frames, want_tokens=False):
        """The host-thread prefix search (word-based scorers, sizes beyond the GPU kernel) of ONE device pass without stalling the
        device: the vocabulary pruning is queued on the current stream right behind the pass's CTC head, and a function is
        returned that -- when called -- waits for the pruning, brings the candidates to the host on the library's copy stream
        (5 MB for 32 x 494 frames x 40; NOT queued behind the encoder, see ``_batch_collect``) and runs the search on
        ``num_processes`` host threads.  The caller launches the encoders of the next passes first and collects afterwards: host
        searches run under device passes instead of between them."""
        part = self.prune_padded(probs, frames)
        done = torch.cuda.Event()
        done.record()
        if self._ext_scorer is not None and not self._ext_scorer.is_character_based and not getattr(self, '_said_host', False):
            self._said_host = True
            logger.info('masr_amd BeamSearchDecoder: word-based language model -> prefix search on host threads')

        def collect():
            B, Ts, K = part['B'], part['Ts'], part['K']
            dev = [part['idx'].view(-1), part['logp'].view(-1).view(torch.int32), part['cnt'].view(-1)]
            if part['blp'] is not None:
                dev.append(part['blp'].view(-1).view(torch.int32))
            total = sum(t.numel() for t in dev)
            host = self._pinned_out(total)
            done.synchronize()
            with torch.cuda.stream(runtime.aux_engine(part['idx'].device).side_stream(3)):
                at, views = 0, []
                for t in dev:
                    host[at:at + t.numel()].copy_(t, non_blocking=True)
                    views.append(host[at:at + t.numel()].numpy())
                    at += t.numel()
                ev = torch.cuda.Event()
                ev.record()
            ev.synchronize()
            idx, logp, cnt = views[0], views[1].view(np.float32), views[2]
            blp = views[3].view(np.float32) if part['blp'] is not None else None
            max_len = max(Ts, 1)
            toks = np.zeros((B, max_len), np.int32)
            lens = np.zeros(B, np.int32)
            scores = np.zeros(B, np.float32)
            fr = np.ascontiguousarray(part['frames'], np.int32)
            rc = self._lib.masr_beam_search_batch_lm(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                                     cnt.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p), B, Ts, K,
                                                     self.beam_size, self.blank_id, self.num_processes, *self._lm_args(),
                                                     self._ptr(blp), toks.ctypes.data_as(C.c_void_p), max_len,
                                                     lens.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p))
            self._pin_free.append(host)
            if rc != 0:
                raise _lib.MasrError('masr_beam_search_batch failed')
            if want_tokens:
                return [(toks[i, :lens[i]].tolist(), float(scores[i])) for i in range(B)]
            return [(float(scores[i]), self._text(toks[i, :lens[i]])) for i in range(B)]
        return collect

    def decode_chunk(self, probs, logits_lens):
        """streaming: feed a chunk probs [1, T, V]; returns (score, text) of the best prefix so far
        (beam_search_decoder.py:75-91)."""
        n_valid = int(np.asarray(logits_lens).reshape(-1)[0])
        p = probs[0][:n_valid] if torch.is_tensor(probs) else np.asarray(probs)[0][:n_valid]     # device tensors stay there
        if self.use_gpu_search and self.gpu_search_supported(1, p.shape[1]):
            return self._decode_chunk_gpu(p)
        idx, logp, cnt, blp, K = self._candidates(p)
        if p.shape[0]:
            self._lib.masr_beam_advance_lm(self._stream, idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                           cnt.ctypes.data_as(C.c_void_p), self._ptr(blp), p.shape[0], K)
        toks = np.zeros(4096, np.int32)
        n, sc = C.c_int32(), C.c_float()
        self._lib.masr_beam_result(self._stream, toks.ctypes.data_as(C.c_void_p), 4096, C.byref(n), C.byref(sc))
        self.last_tokens = toks[:n.value].tolist()
        return float(sc.value), self._text(toks[:n.value])

    def _decode_chunk_gpu(self, p):
        """device-resident search state (masr_gbeam_*): only the best prefix travels back per chunk"""
        eng = self._geng if self._gstream is not None else runtime.aux_engine()
        if self._gstream is None:
            self._geng = eng             # the stream's state lives in THIS device's engine until close()
            h = C.c_int32()
            check(self._lib.masr_gbeam_open(eng.h, self.beam_size, self.blank_id, 5000, C.byref(h)))
            self._gstream = h.value
            if self._ext_scorer is not None:
                check(self._lib.masr_gbeam_set_lm(eng.h, self._gstream, *self._lm_args()))
        T = p.shape[0]
        idx, logp, cnt, blp, K = self._candidates(p, to_host=False)
        max_len = 5000
        if self._gout is None:
            self._gout = (torch.zeros(1, max_len, dtype=torch.int32, device=eng.device),
                          torch.zeros(1, dtype=torch.int32, device=eng.device),
                          torch.zeros(1, dtype=torch.float32, device=eng.device))
        toks, lens, scores = self._gout
        check(self._lib.masr_gbeam_advance_lm(eng.h, self._gstream, C.c_void_p(idx.data_ptr()), C.c_void_p(logp.data_ptr()),
                                              C.c_void_p(cnt.data_ptr()), self._ptr(blp), T, K, C.c_void_p(toks.data_ptr()),
                                              max_len, C.c_void_p(lens.data_ptr()), C.c_void_p(scores.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        n = int(lens.item())
        self.last_tokens = toks[0, :n].cpu().numpy().tolist()
        return float(scores.item()), self._text(self.last_tokens)

    def reset_decoder(self):
        """beam_search_decoder.py:93-96."""
        self._lib.masr_beam_reset(self._stream)
        if self._gstream is not None:
            check(self._lib.masr_gbeam_reset(self._geng.h, self._gstream))
