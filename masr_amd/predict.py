"""MASRPredictor -- the reference facade (masr/predict.py) on the MI355X engine.

Same constructor and ``predict`` / ``predict_stream`` / ``reset_stream`` contract
(predict.py:20-26,167-171,237-244,346); the three hot components are the HIP-backed mirrors:
AudioFeaturizer (fbank), InferencePredictor (Conformer encoder + CTC head) and the greedy
decoders.  ``predict_batch`` is an addition: the whole PCM -> text path for a padded batch in one
device call (no host round trip of probabilities).
"""
import logging
import os
from io import BufferedReader

import numpy as np
import torch
import yaml

from masr_amd import SUPPORT_MODEL
from masr_amd.data_utils.audio import AudioSegment
from masr_amd.data_utils.featurizer.audio_featurizer import AudioFeaturizer
from masr_amd.data_utils.featurizer.text_featurizer import TextFeaturizer
from masr_amd.decoders.ctc_greedy_decoder import greedy_decoder, greedy_decoder_chunk, greedy_decoder_chunk_frames
from masr_amd.infer_utils.inference_predictor import InferencePredictor
from masr_amd.utils.utils import dict_to_object

logger = logging.getLogger(__name__)


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
        # streaming decode state (predict.py:69-73)
        self.remained_wav = None
        self.cached_feat = None
        self.greedy_last_max_prob_list = None
        self.greedy_last_max_index_list = None
        self.beam_search_decoder = None
        self.vad_predictor = None
        if self.configs.decoder == 'ctc_beam_search':
            # predict.py:96-109; here the search is masr_amd's own (GPU pruning + host prefix search, LM-free)
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

    def decode(self, output_data, use_pun, is_itn):
        """predict.py:118-144."""
        if self.configs.decoder == 'ctc_beam_search':
            score, text = self.beam_search_decoder.decode_beam_search_offline(probs_split=output_data)
        else:
            score, text = greedy_decoder(probs_seq=output_data, vocabulary=self._text_featurizer.vocab_list)
        if is_itn:
            raise Exception('inverse text normalisation (WeTextProcessing) is outside the hot path')
        return score, text

    @staticmethod
    def _load_audio(audio_data, sample_rate=16000):
        """predict.py:146-164."""
        if isinstance(audio_data, str):
            return AudioSegment.from_file(audio_data)
        elif isinstance(audio_data, BufferedReader):
            return AudioSegment.from_file(audio_data)
        elif isinstance(audio_data, np.ndarray):
            return AudioSegment.from_ndarray(audio_data, sample_rate)
        elif isinstance(audio_data, bytes):
            return AudioSegment.from_bytes(audio_data)
        raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')

    def predict(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000):
        """predict.py:167-192: one utterance -> {'text', 'score'}.  With ``decoder: ctc_greedy`` the utterance takes the batched
        device path as a batch of one (features, encoder, fused CTC head and best-path collapse without the numpy round trips
        between the reference's stages: features -> host -> device, probabilities [T', V] -> host -> device)."""
        if self.configs.decoder != 'ctc_beam_search' and not is_itn:
            return self.predict_batch([audio_data], sample_rate=sample_rate)[0]
        audio_segment = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        audio_feature = self._audio_featurizer.featurize(audio_segment)
        input_data = np.array(audio_feature).astype(np.float32)[np.newaxis, :]
        audio_len = np.array([input_data.shape[1]]).astype(np.int64)
        output_data = self.predictor.predict(input_data, audio_len)[0]
        score, text = self.decode(output_data=output_data, use_pun=use_pun, is_itn=is_itn)
        return {'text': text, 'score': score}

    def init_vad(self, vad_predictor=None):
        """predict.py:110-115.  ``vad_predictor``: any object with the reference VADPredictor's ``get_speech_timestamps``;
        default: the built-in energy VAD (Silero / onnxruntime are outside this path, see infer_utils/vad_predictor.py)."""
        if vad_predictor is not None:
            self.vad_predictor = vad_predictor
        elif getattr(self, 'vad_predictor', None) is None:
            from masr_amd.infer_utils.vad_predictor import EnergyVAD
            self.vad_predictor = EnergyVAD()

    def predict_long(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000, vad_predictor=None, batch_size=32):
        """predict.py:195-234: long audio -> VAD segments -> text.  The reference recognises the segments one after the
        other (``self.predict`` per segment, :220); here they go through ``predict_batch`` in length-sorted batches of
        ``batch_size`` (one device call per batch); texts are joined in time order exactly like the reference (:222-227,233).
        A padded batch follows the reference's own batch > 1 semantics (its pad mask keeps one padding-contaminated key per
        shorter utterance, trainer.py:632); ``batch_size=1`` is the reference's per-segment ``predict`` to the bit."""
        if use_pun:
            raise Exception('punctuation (PaddleNLP) is outside the hot path and not provided')
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
            if len(idx) == 1:
                results[idx[0]] = self.predict(audio_data=pieces[idx[0]], sample_rate=audio_segment.sample_rate)
                continue
            for i, res in zip(idx, self.predict_batch([pieces[i] for i in idx], sample_rate=audio_segment.sample_rate)):
                results[i] = res
        texts, scores = '', []
        for res in results:
            if is_itn:
                raise Exception('inverse text normalisation (WeTextProcessing) is outside the hot path')
            if res['text'] != '':
                texts = texts + '，' + res['text']
            scores.append(res['score'])
            logger.info(f'长语音识别片段结果：{res["text"]}')
        if texts[:1] == '，':
            texts = texts[1:]
        return {'text': texts, 'score': round(sum(scores) / len(scores), 2) if scores else 0}

    def predict_batch(self, audio_list, sample_rate=16000, decode_all_frames=False):
        """Batched offline path on the device: padded int16/float PCM -> fbank -> encoder -> greedy
        in ONE library call.  Returns [{'text','score'}].  ``decode_all_frames=True`` reproduces
        the reference's batch evaluation quirk of decoding padded frames (trainer.py:340)."""
        eng = self.predictor.engine
        segs = [self._load_audio(a, sample_rate) for a in audio_list]
        for s in segs:
            if s.sample_rate != 16000:
                s.resample(16000)
        n = np.array([s.num_samples for s in segs], np.int32)
        # padded batch in a pinned staging buffer (int16 when every utterance still is the PCM it was loaded from)
        as_pcm = all(s._pcm16 is not None for s in segs)
        dt = torch.int16 if as_pcm else torch.float32
        need = len(segs) * int(n.max())
        pool = self.__dict__.setdefault('_stage', {})
        if dt not in pool or pool[dt].numel() < need:
            pool[dt] = torch.zeros(need + need // 4, dtype=dt, pin_memory=True)          # reused across calls
        stage = pool[dt][:need].view(len(segs), int(n.max()))
        buf = stage.numpy()
        buf[:] = 0
        for i, s in enumerate(segs):
            buf[i, :n[i]] = s._pcm16 if as_pcm else s._samples
        xs = stage.to(eng.device, non_blocking=True)
        ns = torch.from_numpy(n).to(eng.device)
        torch.cuda.current_stream().synchronize()          # the staging buffer is reused by the next call
        pc = self.configs.preprocess_conf
        feats, frames = eng.features_batch(pc.get('feature_method', 'fbank'), xs, ns, pc.use_dB_normalization, pc.target_dB,
                                           n_mfcc=pc.get('n_mfcc', 40))
        enc = eng.encode_full(feats, frames, -1)
        nenc = None if decode_all_frames else (((frames - 1) // 2 - 1) // 2).clamp(min=0).to(torch.int32)
        if self.configs.decoder == 'ctc_beam_search':
            # probabilities stay on the GPU for the vocabulary pruning; the prefix search runs on host threads
            probs = eng.ctc_probs(enc)
            n_host = [probs.shape[1]] * len(segs) if nenc is None else nenc.cpu().tolist()
            res = self.beam_search_decoder._batch([probs[i, :n_host[i]] for i in range(len(segs))])
            return [{'text': t, 'score': sc} for sc, t in res]
        idx, mp = eng.ctc_greedy_frames(enc)
        tok, ntok, score = eng.ctc_collapse(idx, mp, nenc)
        tok, ntok, score = tok.cpu().numpy(), ntok.cpu().numpy(), score.cpu().numpy()
        vocab = self._text_featurizer.vocab_list
        out = []
        for i in range(len(segs)):
            text = ''.join(vocab[j] for j in tok[i, :ntok[i]]).replace('<space>', ' ')
            out.append({'text': text, 'score': float(score[i]) * 100.0 if ntok[i] > 0 or score[i] > 0 else 0})
        return out

    def evaluate(self, manifest, batch_size=32, display_result=False, decode_all_frames=False):
        """Batched offline evaluation on the engine (the reference's batch > 1 consumer: MASRTrainer.evaluate,
        trainer.py:592-651): ``manifest`` is the reference's txt manifest (one JSON object per line with ``audio_filepath``
        and ``text``, data_utils/reader.py:32-40,55), utterances are sorted by duration, padded per batch and decoded with
        the configured decoder; returns (loss, error_rate) like the reference with loss = -1 (no CTC loss on this path) and
        error_rate = mean CER or WER over utterances (configs.metrics_type).  ``decode_all_frames=True`` reproduces the
        reference's decoding of the padded frames (trainer.py:340-344)."""
        import json
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
        errors = []
        for lo in range(0, len(items), batch_size):
            chunk = items[lo:lo + batch_size]
            results = self.predict_batch([it[0] for it in chunk], decode_all_frames=decode_all_frames)
            for (path, label, _), res in zip(chunk, results):
                err = metric(res['text'], label)
                errors.append(err)
                if display_result:
                    logger.info(f'预测结果为：{res["text"]}')
                    logger.info(f'实际标签为：{label}')
                    logger.info(f'这条数据的{self.configs.metrics_type}：{round(err, 6)}，'
                                f'当前{self.configs.metrics_type}：{round(sum(errors) / len(errors), 6)}')
        return -1, (float(sum(errors) / len(errors)) if errors else -1)

    def predict_stream(self, audio_data, is_end=False, use_pun=False, is_itn=False, channels=1, samp_width=2,
                       sample_rate=16000):
        """predict.py:237-343: feed raw PCM bytes / ndarray chunks, get the transcript so far."""
        if not self.configs.streaming:
            raise Exception(f"不支持改该模型流式识别，当前模型：{self.configs.use_model}，参数streaming为：{self.configs.streaming}")
        if isinstance(audio_data, np.ndarray):
            audio_data = AudioSegment.from_ndarray(audio_data, sample_rate)
        elif isinstance(audio_data, bytes):
            audio_data = AudioSegment.from_pcm_bytes(audio_data, channels=channels, samp_width=samp_width,
                                                     sample_rate=sample_rate)
        else:
            raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')
        if self.remained_wav is None:
            self.remained_wav = audio_data
        else:
            self.remained_wav = AudioSegment(np.concatenate([self.remained_wav.samples, audio_data.samples]),
                                             audio_data.sample_rate)
        # featurize everything not yet turned into frames; NB this re-normalises the carried-over
        # samples in place on every call, exactly like the reference (predict.py:274, audio.py:304)
        x_chunk = self._audio_featurizer.featurize(self.remained_wav)
        x_chunk = np.array(x_chunk).astype(np.float32)[np.newaxis, :]
        if self.cached_feat is None:
            self.cached_feat = x_chunk
        else:
            self.cached_feat = np.concatenate([self.cached_feat, x_chunk], axis=1)
        self.remained_wav._samples = self.remained_wav.samples[160 * x_chunk.shape[1]:]

        decoding_chunk_size, context, subsampling = 16, 7, 4
        cached_feature_num = context - subsampling                           # 3 frames of overlap
        decoding_window = (decoding_chunk_size - 1) * subsampling + context  # 67
        stride = subsampling * decoding_chunk_size                           # 64
        num_frames = self.cached_feat.shape[1]
        if num_frames < decoding_window and not is_end:
            return None
        if num_frames < context:
            return None
        left_frames = context if is_end else decoding_window
        score, text, end = None, None, None
        for cur in range(0, num_frames - left_frames + 1, stride):
            end = min(cur + decoding_window, num_frames)
            x_chunk = self.cached_feat[:, cur:end, :]
            if self.configs.decoder != 'ctc_beam_search':
                # greedy: only the per-frame (argmax, max prob) pairs leave the device (fused CTC head)
                ids, mps = self.predictor.predict_chunk_frames(x_chunk)
                score, text, self.greedy_last_max_prob_list, self.greedy_last_max_index_list = \
                    greedy_decoder_chunk_frames(ids, mps, vocabulary=self._text_featurizer.vocab_list,
                                                last_max_index_list=self.greedy_last_max_index_list,
                                                last_max_prob_list=self.greedy_last_max_prob_list)
                continue
            if self.configs.use_model == 'deepspeech2':
                output_chunk_probs, output_lens = self.predictor.predict_chunk_deepspeech(x_chunk=x_chunk)
            elif 'former' in self.configs.use_model:
                num_decoding_left_chunks = -1
                required_cache_size = decoding_chunk_size * num_decoding_left_chunks
                output_chunk_probs = self.predictor.predict_chunk_conformer(x_chunk=x_chunk,
                                                                            required_cache_size=required_cache_size)
                output_lens = np.array([output_chunk_probs.shape[1]])
            else:
                raise Exception(f'当前模型不支持该方法，当前模型为：{self.configs.use_model}')
            if self.configs.decoder == 'ctc_beam_search':
                score, text = self.beam_search_decoder.decode_chunk(probs=output_chunk_probs, logits_lens=output_lens)
            else:
                score, text, self.greedy_last_max_prob_list, self.greedy_last_max_index_list = \
                    greedy_decoder_chunk(probs_seq=output_chunk_probs[0], vocabulary=self._text_featurizer.vocab_list,
                                         last_max_index_list=self.greedy_last_max_index_list,
                                         last_max_prob_list=self.greedy_last_max_prob_list)
        self.cached_feat = self.cached_feat[:, end - cached_feature_num:, :]
        if is_itn:
            raise Exception('inverse text normalisation (WeTextProcessing) is outside the hot path')
        return {'text': text, 'score': score}

    def reset_stream(self):
        """predict.py:346-353."""
        self.predictor.reset_stream()
        self.remained_wav = None
        self.cached_feat = None
        self.greedy_last_max_prob_list = None
        self.greedy_last_max_index_list = None
        if self.configs.decoder == 'ctc_beam_search' and getattr(self, 'beam_search_decoder', None) is not None:
            self.beam_search_decoder.reset_decoder()
