"""MASRPredictor -- the reference facade (masr/predict.py) on the MI355X engine.

Same constructor and ``predict`` / ``predict_long`` / ``predict_stream`` / ``reset_stream`` contract
(predict.py:20-26,167-171,195-199,237-244,346); everything between the audio and the text runs in libmasr_hip.so.
How the calls map onto the engine:

  * ``predict`` is ``predict_batch`` with a batch of one: padded PCM -> features -> encoder -> CTC decode in one device
    pass, no numpy round trips between the reference's stages;
  * ``predict_stream`` is a ``serving.StreamPool`` with a single session (the pool is the one implementation of the
    reference's stream framing; a server opens more sessions on the same pool);
  * ``predict_batch`` / ``evaluate`` shard their utterances over the ranks of an initialised ``torch.distributed`` process
    group (one process per GPU, ``parallel.py``) and all-gather the hypotheses.
"""
import ctypes as C
import json
import logging
import os
from io import BufferedReader

import numpy as np
import torch
import yaml

from masr_amd import SUPPORT_MODEL, parallel
from masr_amd._lib import check
from masr_amd.data_utils.audio import AudioSegment
from masr_amd.data_utils.featurizer.audio_featurizer import AudioFeaturizer
from masr_amd.data_utils.featurizer.text_featurizer import TextFeaturizer
from masr_amd.decoders.ctc_greedy_decoder import greedy_decoder
from masr_amd.infer_utils.inference_predictor import InferencePredictor
from masr_amd.utils.utils import dict_to_object

logger = logging.getLogger(__name__)


def balanced_cuts(sorted_lengths, budget):
    """[lo, hi) index ranges over an ASCENDING list of utterance lengths such that every range holds as many utterances as fit
    count x (its longest utterance) <= budget, cut from the long end (at least one utterance per range); ranges in ascending
    order.  A range is one device pass: all passes then cost about the same padded work -- one full round of workgroups each --
    instead of a pass of long utterances costing twice a pass of short ones."""
    cuts, hi = [], len(sorted_lengths)
    while hi > 0:
        k = max(1, min(hi, int(float(budget) // max(float(sorted_lengths[hi - 1]), 1.0))))
        cuts.append((hi - k, hi))
        hi -= k
    cuts.reverse()
    return cuts


class MASRPredictor:
    def __init__(self, configs=None, model_tag='conformer_streaming_fbank_aishell',
                 model_path='models/conformer_streaming_fbank/inference.pt', use_pun=False,
                 pun_model_dir='models/pun_models/', use_gpu=True, state_dict=None):
        if not configs:
            raise Exception('no network here: pass `configs` (YAML path or dict) and `model_path` explicitly')
        if isinstance(configs, str):
            with open(configs, 'r', encoding='utf-8') as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.configs = dict_to_object(configs)
        assert self.configs.use_model in SUPPORT_MODEL, f'没有该模型：{self.configs.use_model}'
        if use_pun:
            raise Exception('punctuation (PaddleNLP) is outside the hot path and not provided')
        self.running = False
        self.use_gpu = use_gpu
        self._text_featurizer = TextFeaturizer(vocab_filepath=self.configs.dataset_conf.dataset_vocab)
        self._audio_featurizer = AudioFeaturizer(**self.configs.preprocess_conf)
        self.beam_search_decoder = None
        self.vad_predictor = None
        self._pool = None             # predict_stream: a StreamPool with one session, opened on first use
        self._session = None
        self._stage = {}              # pinned host staging buffers of predict_batch, per sample type
        if self.configs.decoder == 'ctc_beam_search':
            # predict.py:96-109; the search is masr_amd's own (vocabulary pruning, prefix search and LM scoring on the GPU)
            from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
            self.beam_search_decoder = BeamSearchDecoder(vocab_list=self._text_featurizer.vocab_list,
                                                         **self.configs.ctc_beam_search_decoder_conf)
        if state_dict is None and not os.path.exists(model_path):
            raise Exception("模型文件不存在，请检查{}是否存在！".format(model_path))
        self.predictor = InferencePredictor(configs=self.configs, use_model=self.configs.use_model,
                                            streaming=self.configs.streaming, model_path=model_path,
                                            use_gpu=self.use_gpu, state_dict=state_dict)
        self._audio_featurizer.bind(self.predictor.engine)
        # warm-up like the reference (predict.py:89-92)
        warmup_audio = np.random.uniform(low=-2.0, high=2.0, size=(134240,))
        self.predict(audio_data=warmup_audio, is_itn=False)
        self.reset_stream()

    # ---- helpers ----------------------------------------------------------------------------------------------------------
    def decode(self, output_data, use_pun, is_itn):
        """predict.py:118-144: probabilities [T', V] of one utterance -> (score, text)."""
        if self.configs.decoder == 'ctc_beam_search':
            score, text = self.beam_search_decoder.decode_beam_search_offline(probs_split=output_data)
        else:
            score, text = greedy_decoder(probs_seq=output_data, vocabulary=self._text_featurizer.vocab_list)
        self._no_itn(is_itn)
        return score, text

    @staticmethod
    def _no_itn(is_itn):
        if is_itn:
            raise Exception('inverse text normalisation (WeTextProcessing) is outside the hot path')

    @staticmethod
    def _load_audio(audio_data, sample_rate=16000):
        """predict.py:146-164: path / binary file object / ndarray / wav bytes -> AudioSegment."""
        loaders = ((str, AudioSegment.from_file), (BufferedReader, AudioSegment.from_file),
                   (np.ndarray, lambda a: AudioSegment.from_ndarray(a, sample_rate)), (bytes, AudioSegment.from_bytes))
        for kind, load in loaders:
            if isinstance(audio_data, kind):
                return load(audio_data)
        raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')

    def _text(self, ids):
        vocab = self._text_featurizer.vocab_list
        return ''.join(vocab[j] for j in ids).replace('<space>', ' ')

    # ---- offline ------------------------------------------------------------------------------------------------------------
    def predict(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000):
        """predict.py:167-192: one utterance -> {'text', 'score'} (a batch of one on the device path)."""
        self._no_itn(is_itn)
        return self._predict_local([self._load_audio(audio_data, sample_rate)])[0]

    def init_vad(self, vad_predictor=None):
        """predict.py:110-115: ``VADPredictor()`` -- the Silero network (on the GPU, weights from the reference's
        ``silero_vad.onnx``: infer_utils/silero_vad.py says where the file is looked for; without it this raises).
        ``vad_predictor``: any object with the reference VADPredictor's ``get_speech_timestamps`` instead, e.g. the built-in
        ``EnergyVAD`` stand-in for callers without the Silero file."""
        if vad_predictor is not None:
            self.vad_predictor = vad_predictor
        elif getattr(self, 'vad_predictor', None) is None:
            from masr_amd.infer_utils.vad_predictor import VADPredictor
            self.vad_predictor = VADPredictor()

    def predict_long(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000, vad_predictor=None, batch_size=32):
        """predict.py:195-234: long audio -> VAD segments -> text.  The reference recognises the segments one after the
        other (``self.predict`` per segment, :220); here they go through ``predict_batch`` in length-sorted batches of
        ``batch_size`` (one device pass per batch); texts are joined in time order exactly like the reference (:222-227,233).
        A padded batch follows the reference's own batch > 1 semantics (its pad mask keeps one padding-contaminated key per
        shorter utterance, trainer.py:632); ``batch_size=1`` is the reference's per-segment ``predict`` to the bit."""
        if use_pun:
            raise Exception('punctuation (PaddleNLP) is outside the hot path and not provided')
        self._no_itn(is_itn)
        self.init_vad(vad_predictor)
        audio_segment = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        if audio_segment.sample_rate != self.configs.preprocess_conf.sample_rate:
            audio_segment.resample(self.configs.preprocess_conf.sample_rate)
        samples = audio_segment.samples
        stamps = self.vad_predictor.get_speech_timestamps(samples, audio_segment.sample_rate)
        pieces = [samples[t['start']: t['end']] for t in stamps]
        order = sorted(range(len(pieces)), key=lambda i: len(pieces[i]))
        results = [None] * len(pieces)
        for lo in range(0, len(order), batch_size):
            idx = order[lo:lo + batch_size]
            for i, res in zip(idx, self._predict_local([AudioSegment.from_ndarray(pieces[i], audio_segment.sample_rate)
                                                        for i in idx])):
                results[i] = res
        texts, scores = '', []
        for res in results:
            if res['text'] != '':
                texts = texts + '，' + res['text']
            scores.append(res['score'])
            logger.info(f'长语音识别片段结果：{res["text"]}')
        if texts[:1] == '，':
            texts = texts[1:]
        return {'text': texts, 'score': round(sum(scores) / len(scores), 2) if scores else 0}

    def _prep_stream(self):
        """side stream of the per-pass preparation (upload, mean squares and their read-back, gains): the host waits for THIS
        stream only, so preparing pass k + 1 never waits for the encoder of pass k on the main stream"""
        if getattr(self, '_prep', None) is None:
            self._prep = self.predictor.engine.side_stream(2)
        return self._prep

    def _stage_batch(self, segs, n):
        """padded batch through a pinned staging buffer -> device (int16 when every utterance still is the PCM it was
        loaded from: half the bytes over PCIe, x / 2^15 happens in the kernel), on the CURRENT stream.  Staging buffers take
        turns; each carries the event of its last upload, so filling one never waits for the device unless that very
        buffer's previous copy (two passes ago) is still in flight."""
        return self._stage_upload(self._stage_fill(segs, n))

    def _stage_fill(self, segs, n):
        """host half of ``_stage_batch``: the rows into the pinned buffer -> what ``_stage_upload`` needs"""
        as_pcm = all(s._pcm16 is not None for s in segs)
        dt = torch.int16 if as_pcm else torch.float32
        n_max = int(n.max())
        need = len(segs) * n_max
        ring = self._stage.setdefault(dt, {'bufs': [None] * 2, 'events': [None] * 2, 'turn': 0, 'lens': [None] * 2})      # (a buffer's upload is complete long before the pass that staged it ends)
        k = ring['turn']
        ring['turn'] = k ^ 1
        if ring['events'][k] is not None:
            ring['events'][k].synchronize()
        if ring['bufs'][k] is None or ring['bufs'][k].numel() < need:
            ring['bufs'][k] = torch.zeros(need + need // 4, dtype=dt, pin_memory=True)
        if ring['lens'][k] is None or ring['lens'][k].numel() < len(segs):
            ring['lens'][k] = torch.zeros(max(len(segs), 64), dtype=torch.int32, pin_memory=True)
        stage = ring['bufs'][k][:need].view(len(segs), n_max)
        # masr_stage_rows (csrc/stage.cpp): the rows of the pass into the pinned buffer, zero-padded, on four library threads --
        # 20 MB per pass of 32 x 20 s.  (Python threads over numpy row assignments: 0.45 ms per pass, one thread 0.66 ms; 2 / 8 / 12
        # python threads 17.0 / 17.4 / 18.5 ms per configs[2] greedy call against 16.9 with four; 16 row groups uploaded block by
        # block under the fill of the next block: 18.6 ms.)
        rows = [np.ascontiguousarray(s._pcm16 if as_pcm else s._samples, np.int16 if as_pcm else np.float32) for s in segs]
        ptrs = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        n32 = np.ascontiguousarray(n, np.int32)
        assert all(r.shape[0] >= int(m) for r, m in zip(rows, n32))
        check(self.predictor.engine.lib.masr_stage_rows(C.c_void_p(stage.data_ptr()), n_max * stage.element_size(), ptrs,
                                                       n32.ctypes.data_as(C.c_void_p), len(rows), stage.element_size(), 4))
        lens = ring['lens'][k][:len(segs)]
        lens.numpy()[:] = n32
        return ring, k, stage, lens

    def _stage_upload(self, filled):
        """device half of ``_stage_batch``: queues the two copies on the current stream"""
        eng = self.predictor.engine
        ring, k, stage, lens = filled
        xs = stage.to(eng.device, non_blocking=True)
        ns = lens.to(eng.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring['events'][k] = ev
        return xs, ns

    def _prepare_begin(self, live, n, use_db):
        """first half of a pass's preparation, nothing waited for: staging + upload + (bit-exact normalisation route) the mean
        squares on their way back to a pinned slot, all on the preparation stream.  With two lanes it is issued for pass k + 1
        BEFORE the encoder of pass k is launched: issued beside a running encoder pass, the upload and the mean squares start
        4 - 6 ms late (round 6; whichever side stream, 4 or 8 hardware queues)."""
        eng = self.predictor.engine
        with torch.cuda.stream(self._prep_stream()):
            xs, ns = self._stage_upload(self._stage_fill(live, n))
            ms_host = None
            if use_db:
                ring = self.__dict__.setdefault('_ms_ring', {'bufs': [None] * 4, 'turn': 0})
                k = ring['turn']
                ring['turn'] = (k + 1) % 4
                if ring['bufs'][k] is None or ring['bufs'][k].numel() < len(live):
                    ring['bufs'][k] = torch.empty(max(len(live), 64), dtype=torch.float32, pin_memory=True)
                ms_host = ring['bufs'][k][:len(live)]
                ms_host.copy_(eng.mean_square(xs, ns), non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        return xs, ns, ms_host, ready

    def _prepare(self, live, n, use_db, target_db):
        """both halves at once (a pass prepared right before its encoder)"""
        return self._prepare_finish(self._prepare_begin(live, n, use_db), target_db)

    def _prepare_finish(self, began, target_db):
        """second half: the reference's gains -- the scalar float32 expressions of audio.py:287-304,519-529 in this host's numpy on
        the mean squares the device computed -- uploaded on the CURRENT stream (the pass's encoder stream), which then waits for
        the upload of the samples; the host has waited for the preparation stream's event only."""
        from masr_amd.engine import reference_gains
        eng = self.predictor.engine
        xs, ns, ms_host, ready = began
        main = torch.cuda.current_stream(eng.device)
        gain = None
        if ms_host is not None:
            ready.synchronize()
            gain = eng.to_device(reference_gains(ms_host.numpy().copy(), target_db))
        main.wait_event(ready)
        for t in (xs, ns):
            t.record_stream(main)
        return xs, ns, gain

    def _rows_slot(self, B, width):
        """pinned host buffer for the packed result rows of one pass (a ring of 4: at most two passes are in flight)"""
        ring = self.__dict__.setdefault('_rows_ring', {'bufs': [None] * 4, 'turn': 0})
        k = ring['turn']
        ring['turn'] = (k + 1) % 4
        if ring['bufs'][k] is None or ring['bufs'][k].numel() < B * width:
            ring['bufs'][k] = torch.empty(max(B * width, 1 << 14), dtype=torch.int32, pin_memory=True)
        return ring['bufs'][k][:B * width].view(B, width)

    def _begin_pass(self, segs):
        """host side of a device pass + the asynchronous half of its preparation (``_prepare_begin``); ``_predict_local`` takes
        the result as ``began``.  Utterances too short for one feature frame are set aside here."""
        pc = self.configs.preprocess_conf
        rate = int(pc.get('sample_rate', 16000))
        min_samples = 320 if pc.get('feature_method', 'fbank') == 'linear' else 400
        for s in segs:
            if s.sample_rate != rate:
                s.resample(rate)
        # 7 feature frames is the least Conv2dSubsampling4 accepts (subsampling.py:65-112)
        ok = [i for i, s in enumerate(segs) if s.num_samples >= min_samples + 6 * 160]
        live = [segs[i] for i in ok]
        n = np.array([s.num_samples for s in live], np.int32)
        return {'count': len(segs), 'ok': ok, 'n': n, 'min_samples': min_samples,
                'prep': self._prepare_begin(live, n, pc.use_dB_normalization) if ok else None}

    def _predict_local(self, segs, decode_all_frames=False, as_tokens=False, defer=False, hold_search=False, began=None):
        """AudioSegments -> [{'text','score'}] on THIS rank's engine.  Utterances too short for one feature frame decode to
        the empty transcript (the reference's encoder cannot take them either); a digitally silent utterance (mean square 0) is
        normalised with gain = target_dB like the reference (audio.py:519-529), only a gain above max_gain_db raises (:300-303).
        ``defer=True``: nothing is waited for -- a zero-argument function that returns the results is handed back, so the caller
        prepares and launches the next device pass underneath this one (greedy: the packed hypothesis rows come back in one
        copy behind an event; beam search on the GPU: the prefix search runs on a side stream)."""
        eng = self.predictor.engine
        pc = self.configs.preprocess_conf
        method = pc.get('feature_method', 'fbank')
        if began is None:
            began = self._begin_pass(segs)
        ok, n, min_samples = began['ok'], began['n'], began['min_samples']
        out = [None] * began['count']
        if len(ok) != began['count']:
            for i in set(range(began['count'])) - set(ok):
                out[i] = ([], 0) if as_tokens else {'text': '', 'score': 0}
        if not ok:
            return (lambda: out) if defer else out
        live = ok                # (only its length is used below)
        xs, ns, gain = self._prepare_finish(began['prep'], pc.target_dB)
        greedy = self.configs.decoder != 'ctc_beam_search'
        if greedy and method == 'fbank':
            # the whole pass is ONE C-ABI call and ONE copy back: rows [B, T' + 2] = tokens | count | score bits
            rows = eng.transcribe_rows(xs, ns, pc.use_dB_normalization, pc.target_dB, gain_in=gain,
                                       decode_all_frames=decode_all_frames)
            host = self._rows_slot(*rows.shape)
            host.copy_(rows, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()

            def fetch():
                ev.synchronize()
                r = host.numpy()
                tp = r.shape[1] - 2
                ntok, score = r[:, tp], r[:, tp + 1].copy().view(np.float32)
                for j, i in enumerate(ok):
                    sc = float(score[j]) * 100.0 if ntok[j] > 0 or score[j] > 0 else 0
                    ids = r[j, :ntok[j]]
                    out[i] = (ids.tolist(), sc) if as_tokens else {'text': self._text(ids), 'score': sc}
                return out
            return fetch if defer else fetch()
        feats, frames = eng.features_batch(method, xs, ns, pc.use_dB_normalization, pc.target_dB, n_mfcc=pc.get('n_mfcc', 40),
                                           gain_in=gain)
        if decode_all_frames:        # the padded frames are decoded too: the engine must compute them (masr_debug_set key 38)
            check(eng.lib.masr_debug_set(eng.h, 38, 0))
        try:
            enc = eng.encode_full(feats, frames, -1)
        finally:
            if decode_all_frames:
                check(eng.lib.masr_debug_set(eng.h, 38, 7))
        nenc = None if decode_all_frames else eng.enc_frames(frames).to(torch.int32)
        if not greedy:
            # probabilities stay on the GPU: vocabulary pruning, prefix search and LM scoring run there
            probs = eng.ctc_probs(enc)
            n_host = [probs.shape[1]] * len(live) if nenc is None else eng.enc_frames((n.astype(np.int64) - min_samples) // 160 + 1).tolist()
            seqs = probs                       # searched in place: padded [B, T', V] + the valid frames per utterance
            dec = self.beam_search_decoder

            def fill(res):
                for i, r in zip(ok, res):
                    out[i] = r if as_tokens else {'text': r[1], 'score': r[0]}
                return out
            if defer and dec.use_gpu_search and dec.gpu_search_supported(probs.shape[1], probs.shape[2]):
                # The prefix search of a pass (one workgroup per utterance, frames in sequence) runs on a side stream; side streams
                # take turns, so the searches of consecutive passes run next to each other, not one behind the other.  Two of
                # them: with four side streams next to the main and the preparation stream the streams alias onto hardware
                # queues (round 4: 50.1 vs 46.6 ms per configs[2] call).  Round 5 measured the alternative of HOLDING the
                # searches of a group of passes until the group's encoders are enqueued (MASR_BEAM_GROUP=n, ``hold_search``):
                # slower (flat posteriors 58.2 vs 45.7 ms, sharpened head 38.2 vs 29.1 ms per call, gpurun_out r05c) -- the
                # call is bound by the longest utterance's ~30 us per frame x 494 frames, and starting that search late costs
                # more than sharing CUs with the next encoder pass.  Default: launch at once.
                main = torch.cuda.current_stream(eng.device)          # (explicit device: a server runs one worker thread per GPU)
                if getattr(self, '_sides', None) is None:
                    # the library's two search streams of this device (masr_side_stream: queues of their own, highest priority)
                    self._sides, self._side_turn = [eng.side_stream(k) for k in range(max(1, min(2, int(os.environ.get('MASR_BEAM_SIDES', '2')))))], 0

                def launch_search():
                    side = self._sides[self._side_turn]
                    self._side_turn = (self._side_turn + 1) % len(self._sides)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        pending = dec._batch(seqs, defer=True, frames=n_host)
                    return lambda: fill(dec._batch_collect(pending, want_tokens=as_tokens))
                if hold_search:
                    # the pass's candidates are pruned here, on the main stream right behind its CTC head (the 134 MB of
                    # probabilities are then dead); the search itself is launched ONCE for the passes of a group (_run_sorted)
                    return ('held', dec.prune_padded(seqs, n_host), fill)
                return launch_search()
            if defer and torch.is_tensor(seqs) and seqs.is_cuda and os.environ.get('MASR_HOST_SEARCH_DEFER', '1') == '1':
                # host-thread search (word-based scorer, sizes beyond the GPU kernel): pruning queued now, candidates fetched and
                # searched when the results are collected -- under the encoders of the passes launched in between
                handle = dec.search_host_deferred(seqs, n_host, want_tokens=as_tokens)
                return lambda: fill(handle())
            res = fill(dec._batch(seqs, want_tokens=as_tokens, frames=n_host))
            return (lambda: res) if defer else res
        idx, mp = eng.ctc_greedy_frames(enc)
        tok, ntok, score = eng.ctc_collapse(idx, mp, nenc)
        tok, ntok, score = eng.to_host(tok), eng.to_host(ntok), eng.to_host(score)
        for j, i in enumerate(ok):
            sc = float(score[j]) * 100.0 if ntok[j] > 0 or score[j] > 0 else 0
            ids = tok[j, :ntok[j]]
            out[i] = (ids.tolist(), sc) if as_tokens else {'text': self._text(ids), 'score': sc}
        return (lambda: out) if defer else out

    def _length_hint(self, audio_data, sample_rate):
        """number of samples an utterance will have at the model's rate, WITHOUT decoding it where that is possible (wav
        path / wav bytes: the header; ndarray: its length) -- what sorting into passes and sharding over ranks need"""
        rate = int(self.configs.preprocess_conf.get('sample_rate', 16000))
        try:
            if isinstance(audio_data, np.ndarray):
                return int(audio_data.shape[0] * rate // max(int(sample_rate), 1))
            if isinstance(audio_data, (str, bytes)):
                import io
                import wave
                with wave.open(audio_data if isinstance(audio_data, str) else io.BytesIO(audio_data), 'rb') as w:
                    return int(w.getnframes() * rate // max(w.getframerate(), 1))
        except Exception:
            pass
        return self._load_audio(audio_data, sample_rate).num_samples        # file objects and anything odd: decode

    def pass_size(self):
        """utterances of ~10 s per device pass that fill the chip for this model family: 32 are 248 row blocks of 32 for 256 CUs;
        the Efficient-Conformer runs at half the frame rate behind its stride layer, where it takes 64"""
        return 64 if self.configs.use_model == 'efficient_conformer' else 32

    def predict_batch(self, audio_list, sample_rate=16000, decode_all_frames=False, batch_size=0, distributed=None,
                      lengths=None, pass_padded=None):
        """Batched offline path (an addition; the reference's only batched consumer is MASRTrainer.evaluate,
        trainer.py:592-651): a list of utterances -> [{'text','score'}] in input order.

        ``decode_all_frames=True`` reproduces the reference's batch evaluation quirk of decoding padded frames
        (trainer.py:340).  ``batch_size`` > 0 cuts the (length-sorted) work into device passes of that many utterances
        (``batch_size='auto'``: ``pass_size()`` -- 32, or 64 for the Efficient-Conformer; a LIST gives the pass sizes explicitly,
        counted from the longest utterance down, the last size repeating).  ``batch_size='balanced'``: passes of
        EQUAL PADDED SIZE instead of equal count -- a pass takes utterances (longest first) while count x its longest
        utterance stays within ``pass_padded`` (in the unit of the lengths; default: ``pass_size()`` utterances of 10 s, the row
        blocks that fill the chip once), so a pass of long utterances holds few of them, a pass of short ones many, every pass is
        one full round of workgroups and nobody is padded to the longest utterance of the whole list.  The
        audio of a pass is decoded when the pass is formed and dropped when its results are in, at most two passes are in
        flight (features + encoder of pass k under the prefix search of pass k - 1), so host memory and HBM hold two passes
        whatever the list's length.  ``lengths`` (samples or seconds, any common unit): known durations, e.g. a manifest's,
        used for sorting and sharding instead of reading every file's header.
        With an initialised ``torch.distributed`` group (``distributed=None``: automatically when world > 1) the utterances
        are dealt out length-balanced over the ranks -- every rank must call with the same list -- each rank decodes ONLY
        its shard on its own GPU and ONE all-gather of the token ids returns all hypotheses to every rank."""
        hints = list(lengths) if lengths is not None else [self._length_hint(a, sample_rate) for a in audio_list]
        if batch_size == 'auto':
            batch_size = self.pass_size()
        if batch_size == 'balanced':
            rate = int(self.configs.preprocess_conf.get('sample_rate', 16000))
            if pass_padded is None:
                if lengths is not None:
                    raise ValueError("batch_size='balanced' with caller-supplied lengths needs pass_padded in the same unit "
                                     "(evaluate() passes manifest durations and derives it: pass_size() x 10 seconds)")
                pass_padded = self.pass_size() * 10 * rate
            batch_size = ('balanced', float(pass_padded))
        rank, world = parallel.world_info()
        if distributed is None:
            distributed = world > 1
        if not distributed or not parallel.collectives_on():
            return self._run_sorted(audio_list, sample_rate, hints, list(range(len(audio_list))), decode_all_frames, batch_size,
                                    as_tokens=False)
        shards = parallel.length_balanced_shards(hints, world)
        mine = shards[rank]
        local = self._run_sorted(audio_list, sample_rate, hints, mine, decode_all_frames, batch_size, as_tokens=True)
        tmax = max([len(t) for t, _ in local], default=0)
        tok = torch.full((len(mine), max(tmax, 1)), -1, dtype=torch.int32)
        nt = torch.zeros(len(mine), dtype=torch.int32)
        sc = torch.zeros(len(mine), dtype=torch.float32)
        for j, (t, s) in enumerate(local):
            tok[j, :len(t)] = torch.tensor(t, dtype=torch.int32)
            nt[j], sc[j] = len(t), s
        tok, nt, sc = parallel.gather_sharded_results(tok, nt, sc, shards, len(audio_list))
        return [{'text': self._text(tok[i, :nt[i]]), 'score': float(sc[i])} for i in range(len(audio_list))]

    def predict_batch_deferred(self, audio_list, sample_rate=16000):
        """ONE device pass over ``audio_list`` launched, nothing waited for: returns a function that -- when called -- waits for the
        pass and returns [{'text', 'score'}] in input order.  What a serving loop calls for the batch it has formed BEFORE it
        collects the previous one: staging, upload and launches of batch k + 1 then run under the device work of batch k, and
        consecutive calls alternate over the engine's two lanes (not with the prefix search on the GPU, whose launches already
        run beside the next pass).  The reference serves one request at a time (infer_server.py:48-71)."""
        eng = self.predictor.engine
        dec = self.beam_search_decoder if self.configs.decoder == 'ctc_beam_search' else None
        gpu_search = dec is not None and getattr(dec, 'use_gpu_search', False) and \
            dec.gpu_search_supported(1, len(self._text_featurizer.vocab_list))
        lanes = max(1, min(2, int(os.environ.get('MASR_LANES', '1' if gpu_search else '2'))))
        turn = self.__dict__.get('_deferred_turn', 0)
        self._deferred_turn = turn + 1
        lane = turn % lanes
        main = torch.cuda.current_stream(eng.device)
        stream = main
        if lane:
            stream = eng.side_stream(4)
            if not self.__dict__.get('_lane1_ordered'):
                stream.wait_stream(main)             # (once: whatever the caller had queued before the first pass on lane 1)
                self._lane1_ordered = True
        began = self._begin_pass([self._load_audio(a, sample_rate) for a in audio_list])
        if lane:
            eng.select_lane(lane)
        try:
            with torch.cuda.stream(stream):
                return self._predict_local(None, defer=True, began=began)
        finally:
            if lane:
                eng.select_lane(0)

    def _run_sorted(self, audio_list, sample_rate, hints, which, decode_all_frames, batch_size, as_tokens):
        """decode ``audio_list[i] for i in which`` in length-sorted device passes (cut shortest first, ties in input order -- the
        batches ``evaluate`` forms from a duration-sorted manifest); results in the order of ``which``.  Pipeline depth 2:
        pass k is launched (its prefix search on a side stream), then pass k - 1 is collected and its audio dropped."""
        order = sorted(which, key=lambda i: hints[i]) if batch_size else list(which)      # (a list / tuple / number: sorted)
        if isinstance(batch_size, tuple):             # ('balanced', budget): count x longest <= budget per pass
            cuts = balanced_cuts([hints[i] for i in order], batch_size[1])
        elif isinstance(batch_size, list):            # explicit pass sizes, counted from the LONGEST utterance down (the last size repeats)
            if not batch_size:
                raise ValueError('batch_size: an empty list of pass sizes')
            cuts, hi, k = [], len(order), 0
            while hi > 0:
                step = max(1, int(batch_size[min(k, len(batch_size) - 1)]))
                cuts.append((max(0, hi - step), hi))
                hi -= step
                k += 1
            cuts.reverse()
        else:
            step = batch_size if batch_size else max(len(order), 1)
            cuts = [(lo, min(lo + step, len(order))) for lo in range(0, len(order), step)]
        got, pending = {}, []

        def collect(item):
            idx, fetch = item
            for i, r in zip(idx, fetch()):
                got[i] = r

        # same passes either way; with the GPU prefix search the LONGEST pass goes first: its search (one workgroup per utterance,
        # the longest utterance is the critical path of the call) then starts right after the first encoder pass and the
        # shorter passes' encoders and searches run underneath it
        depth = 2
        dec = self.beam_search_decoder if self.configs.decoder == 'ctc_beam_search' else None
        # (a word-based scorer, or beam x top-n beyond the kernel's LDS: the search runs on host threads whatever use_gpu_search says)
        gpu_search = dec is not None and getattr(dec, 'use_gpu_search', False) and \
            dec.gpu_search_supported(1, len(self._text_featurizer.vocab_list))
        if dec is not None and not gpu_search and os.environ.get('MASR_HOST_SEARCH_DEFER', '1') == '1':
            # host-thread search: pruning is queued behind each pass, the search itself runs when the pass is collected -- two
            # passes behind the launches, so the device always has a pass queued; the pass of the shortest utterances (the
            # quickest search) comes last: nothing runs under the last search
            cuts.reverse()
            depth = 3
        if gpu_search:
            cuts.reverse()
            # a prefix search is a long serial kernel on a few CUs (one workgroup per utterance, frames in sequence): the encoders
            # of up to three further passes are launched underneath it before its results are waited for
            depth = 4
        # MASR_BEAM_GROUP=n (A/B, off by default): the prefix searches of a GROUP of passes as ONE launch behind the group's last
        # encoder pass (one workgroup per utterance; a group closes at n utterances or with the list), on a library side stream,
        # so that nothing runs beside an encoder pass inside a group.  Round 6, BASELINE configs[2] (sharpened head / flat): 33.0 /
        # 55.7 ms per call against 30.2 / 48.5 ms with each pass's search launched behind its own encoder -- 64 searches side by
        # side advance at 19 us per frame where 32 do at 16 (tools/beam_profile.py, BEAM_PROFILE_B), and the first pass's search
        # no longer hides under the second encoder.  Default: per-pass launches.
        group_utts = int(os.environ.get('MASR_BEAM_GROUP', '0')) if gpu_search else 0
        held = []

        def release_group():
            eng = self.predictor.engine
            dec = self.beam_search_decoder
            main = torch.cuda.current_stream(eng.device)
            if getattr(self, '_sides', None) is None:
                self._sides, self._side_turn = [eng.side_stream(k) for k in range(2)], 0
            side = self._sides[self._side_turn]
            self._side_turn = (self._side_turn + 1) % len(self._sides)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                handle = dec.search_pruned([part for _, part, _ in held])
            group = list(held)
            held.clear()

            def fetch():
                res = dec._batch_collect(handle, want_tokens=as_tokens)
                lo = 0
                for idx_h, part, fill in group:
                    for i, r in zip(idx_h, fill(res[lo:lo + part['B']])):
                        got[i] = r
                    lo += part['B']
                return []
            pending.append(([], fetch))
        # Two LANES (masr_select_lane): consecutive passes run on two streams and two workspace sets of the one engine, so the
        # kernels of one pass fill the CUs the other leaves idle -- a fused Squeezeformer stage holds one 32-row block per CU,
        # the 429 / 179 valid row blocks of BASELINE configs[2]'s two passes are 2 + 1 rounds of 256 CUs one after the other
        # and 2.4 side by side (encoders alone: 17.0 -> 14.8 ms; the greedy call 18.5 -> 17.2 ms).  The preparation of pass
        # k + 1 (upload, mean squares) is queued BEFORE the encoder of pass k.  NOT with the prefix search on the GPU: a search
        # is a chain of dependent memory round trips and slows down with everything that runs beside it -- its pass's encoder
        # must finish EARLY, not share the chip (sharpened head 23.5 -> 24.7 ms, flat 28.0 -> 30.8 ms per call with two lanes;
        # passes of 16 or uneven passes on two lanes: 23.4 - 31 / 33 - 38 ms).  MASR_LANES=1 | 2 overrides (A/B).
        eng = self.predictor.engine
        lanes = int(os.environ.get('MASR_LANES', '1' if gpu_search else '2'))
        lanes = max(1, min(2, lanes)) if len(cuts) > 1 and not group_utts else 1
        main = torch.cuda.current_stream(eng.device)
        lane_streams = [main] + ([eng.side_stream(4)] if lanes > 1 else [])
        for st in lane_streams[1:]:
            st.wait_stream(main)             # (once, before the first pass: whatever the caller queued comes first)
        began = {}

        def begin(k):
            lo, hi = cuts[k]
            segs = [self._load_audio(audio_list[i], sample_rate) for i in order[lo:hi]]
            began[k] = self._begin_pass(segs)
        for k, (lo, hi) in enumerate(cuts):
            idx = order[lo:hi]
            # (MASR_PREP_AHEAD=0, A/B: the next pass prepared AFTER this pass's encoder is launched -- its upload and mean squares
            #  then wait 6 ms for CUs and the second lane starts late: 18.6 against 17.0 ms per configs[2] greedy call)
            ahead = lanes > 1 and k + 1 < len(cuts) and os.environ.get('MASR_PREP_AHEAD', '1') == '1'
            if k not in began:
                begin(k)
            if ahead:
                # (the staging fills of the two passes started TOGETHER on eight threads, uploads behind them: measured slower --
                #  first encoder kernel at 1.4 instead of 1.15 ms, 17.5 vs 17.0 ms per call)
                begin(k + 1)
            lane = k % lanes
            if lane:
                eng.select_lane(lane)
            try:
                with torch.cuda.stream(lane_streams[lane]):
                    res = self._predict_local(None, decode_all_frames, as_tokens, defer=True, hold_search=group_utts > 0,
                                              began=began.pop(k))
            finally:
                if lane:
                    eng.select_lane(0)
            if isinstance(res, tuple) and res[0] == 'held':
                held.append((idx, res[1], res[2]))
                if sum(part['B'] for _, part, _ in held) >= group_utts:
                    release_group()
                    while len(pending) > 1:           # (the previous group's results: its search ran beside this group's encoders)
                        collect(pending.pop(0))
                continue
            pending.append((idx, res))
            if len(pending) >= depth:
                collect(pending.pop(0))
        if held:
            release_group()
        if pending:
            for item in pending:
                collect(item)
            for side in (getattr(self, '_sides', None) or []) + lane_streams[1:]:
                main.wait_stream(side)
        return [got[i] for i in which]

    def evaluate(self, manifest, batch_size='auto', display_result=False, decode_all_frames=False, pass_padded=None):
        """Batched offline evaluation on the engine (the reference's batch > 1 consumer: MASRTrainer.evaluate,
        trainer.py:592-651): ``manifest`` is the reference's txt manifest (one JSON object per line with ``audio_filepath``
        and ``text``, data_utils/reader.py:32-40,55), utterances are sorted by duration, padded per batch and decoded with
        the configured decoder; returns (loss, error_rate) like the reference with loss = -1 (no CTC loss on this path) and
        error_rate = mean CER or WER over utterances (configs.metrics_type).  ``decode_all_frames=True`` reproduces the
        reference's decoding of the padded frames (trainer.py:340-344).  Under an initialised process group every rank reads
        the manifest, decodes its length-balanced shard and all ranks return the same error rate."""
        from masr_amd.utils.metrics import cer, wer
        items = []
        with open(manifest, 'r', encoding='utf-8') as f:
            for line in f:
                line = line.strip()
                if line:
                    d = json.loads(line)
                    items.append((d['audio_filepath'], d['text'], float(d.get('duration', 0.0))))
        items.sort(key=lambda it: it[2])
        metric = wer if self.configs.metrics_type == 'wer' else cer
        # manifest durations sort and shard the work: a rank opens only the files of its own shard, one pass at a time
        known = [it[2] for it in items] if all(it[2] > 0 for it in items) else None
        # batch_size='balanced': the pass budget in the unit of the lengths -- manifest durations are seconds
        if batch_size == 'balanced' and known is not None and pass_padded is None:
            pass_padded = self.pass_size() * 10.0
        results = self.predict_batch([it[0] for it in items], decode_all_frames=decode_all_frames, batch_size=batch_size,
                                     lengths=known, pass_padded=pass_padded)
        errors = []
        for (path, label, _), res in zip(items, results):
            err = metric(res['text'], label)
            errors.append(err)
            if display_result:
                logger.info(f'预测结果为：{res["text"]}')
                logger.info(f'实际标签为：{label}')
                logger.info(f'这条数据的{self.configs.metrics_type}：{round(err, 6)}，'
                            f'当前{self.configs.metrics_type}：{round(sum(errors) / len(errors), 6)}')
        return -1, (float(sum(errors) / len(errors)) if errors else -1)

    # ---- streaming ------------------------------------------------------------------------------------------------------------
    def predict_stream(self, audio_data, is_end=False, use_pun=False, is_itn=False, channels=1, samp_width=2,
                       sample_rate=16000):
        """predict.py:237-343: feed raw PCM bytes / ndarray chunks, get the transcript so far ({'text','score'}), or None
        while fewer frames than one decoding window have arrived.  One session of a ``StreamPool``."""
        if not self.configs.streaming:
            raise Exception(f"不支持改该模型流式识别，当前模型：{self.configs.use_model}，参数streaming为：{self.configs.streaming}")
        self._no_itn(is_itn)
        if self._pool is None:
            from masr_amd.serving import StreamPool
            self._pool = StreamPool(self)
            self._session = self._pool.open()
        self._pool.feed(self._session, audio_data, is_end, channels=channels, samp_width=samp_width, sample_rate=sample_rate)
        return self._pool.step()[self._session]

    def reset_stream(self):
        """predict.py:346-353: forget the stream (caches, pending audio and frames, decoder history)."""
        self.predictor.reset_stream()
        if self._pool is not None:
            self._pool.reset(self._session)
