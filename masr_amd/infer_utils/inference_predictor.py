"""InferencePredictor -- drop-in for masr/infer_utils/inference_predictor.py on the HIP engine.

Same constructor, same four methods, same numpy-in / numpy-out contract and error behaviour
(:10-15, :52-102).  Instead of ``torch.jit.load(...).to(device)`` the exported artefact is only
read for its weights: ``inference.pt`` (TorchScript of the whole model, trainer.py:684-689) or
``model.pt`` (state_dict, trainer.py:308) -> ``state_dict`` -> libmasr_hip.so.
"""
import os

import numpy as np
import torch

from masr_amd.engine import HipEngine


def load_state_dict(model_path):
    """inference.pt (TorchScript), model.pt (plain state_dict) or a packed file (utils/packed.py) -> {name: tensor}."""
    from masr_amd.utils import packed
    if packed.is_packed(model_path):
        return packed.load_packed(model_path)[0]
    try:
        m = torch.jit.load(model_path, map_location='cpu')
        return {k: v.detach().cpu() for k, v in m.state_dict().items()}
    except RuntimeError:
        sd = torch.load(model_path, map_location='cpu', weights_only=True)
        if not isinstance(sd, dict):
            raise
        return sd


class InferencePredictor:
    def __init__(self, configs, use_model, streaming=True, model_path='models/conformer_streaming_fbank/inference.pt',
                 use_gpu=True, state_dict=None):
        self.configs = configs
        self.use_gpu = use_gpu
        self.use_model = use_model
        self.streaming = streaming
        if state_dict is None:
            if not os.path.exists(model_path):
                raise Exception(f"模型文件不存在，请检查{model_path}是否存在！")
            state_dict = load_state_dict(model_path)
        if not use_gpu:
            raise Exception('masr_amd is the MI355X path: use_gpu=False is not available (no CPU fallback)')
        assert (torch.cuda.is_available()), 'GPU不可用'
        if use_model not in ('conformer', 'squeezeformer', 'efficient_conformer', 'deepspeech2'):
            raise Exception(f'masr_amd implements conformer / squeezeformer / efficient_conformer / deepspeech2; '
                            f'got use_model={use_model}')
        self.device = torch.device('cuda', torch.cuda.current_device())
        enc_conf = dict(configs.get('encoder_conf', {})) if configs is not None else {}
        # input feature size of the model = AudioFeaturizer.feature_dim (audio_featurizer.py:141-154)
        pc = configs.get('preprocess_conf', {}) if configs is not None else {}
        n_mels = {'linear': 161, 'mfcc': int(pc.get('n_mfcc', 40))}.get(pc.get('feature_method', 'fbank'), int(pc.get('n_mels', 80)))
        self.engine = HipEngine(state_dict, encoder_conf=enc_conf, streaming=streaming, n_mels=n_mels,
                                use_model=use_model)
        self._sid = None
        self._history = -1

    # offline (inference_predictor.py:52-64): probs = softmax(ctc_lo(encoder(speech)))
    def predict(self, speech, speech_lengths):
        audio_data = torch.as_tensor(np.asarray(speech), dtype=torch.float32).to(self.device).contiguous()
        audio_len = torch.as_tensor(np.asarray(speech_lengths)).to(torch.int32).to(self.device).contiguous()
        enc = self.engine.encode_full(audio_data, audio_len, -1)
        probs = self.engine.ctc_probs(enc)
        if self.use_model == 'deepspeech2':      # pad_packed_sequence trims to the longest sequence (encoder.py:42)
            longest = int(((np.asarray(speech_lengths).astype(np.int64) - 1) // 2 - 1).max() // 2)
            probs = probs[:, :longest]
        return probs.cpu().numpy()

    def predict_chunk_deepspeech(self, x_chunk):
        if not (self.use_model == 'deepspeech2' and self.streaming):
            raise Exception(f'当前模型不支持该方法，当前模型为：{self.use_model}，参数streaming为：{self.streaming}')
        # inference_predictor.py:66-78: the (h, c) state of every LSTM layer lives in the engine's stream
        if self._sid is None:
            self._sid = self.engine.stream_open(0)
        x = torch.as_tensor(np.asarray(x_chunk), dtype=torch.float32).to(self.device).contiguous()
        probs, _, _ = self.engine.encode_chunk([self._sid], x)
        return probs.cpu().numpy(), np.array([probs.shape[1]], dtype=np.int64)

    # streaming (inference_predictor.py:80-94): att_cache / cnn_cache / offset live in the engine
    def predict_chunk_conformer(self, x_chunk, required_cache_size):
        if not ('former' in self.use_model and self.streaming):
            raise Exception(f'当前模型不支持该方法，当前模型为：{self.use_model}，参数streaming为：{self.streaming}')
        if self._sid is None:
            self._sid = self.engine.stream_open(0)
        required_cache_size = int(np.asarray(required_cache_size).reshape(-1)[0])
        if required_cache_size != self._history:          # (>= 0: bounded attention history, conformer/encoder.py:397-410)
            self.engine.stream_set_history(self._sid, required_cache_size)
            self._history = required_cache_size
        x = torch.as_tensor(np.asarray(x_chunk), dtype=torch.float32).to(self.device).contiguous()
        probs, _, _ = self.engine.encode_chunk([self._sid], x)
        return probs.cpu().numpy()

    def predict_chunk_frames(self, x_chunk):
        """Greedy fast path of the two chunk methods above (an addition): the per-frame (argmax id, max probability) pairs of
        the chunk -- all that greedy_decoder_chunk reads from the probabilities (ctc_greedy_decoder.py:73-77) -- computed by
        the fused CTC head, so the [T, V] probabilities are neither materialised nor copied to the host."""
        if not self.streaming:
            raise Exception(f'当前模型不支持该方法，当前模型为：{self.use_model}，参数streaming为：{self.streaming}')
        if self._sid is None:
            self._sid = self.engine.stream_open(0)
        x = torch.as_tensor(np.asarray(x_chunk), dtype=torch.float32).to(self.device).contiguous()
        _, idx, mp = self.engine.encode_chunk([self._sid], x, want_probs=False, want_argmax=True)
        return idx[0].cpu().numpy(), mp[0].cpu().numpy()

    @property
    def offset(self):
        return 0 if self._sid is None else self.engine.stream_offset(self._sid)

    def reset_stream(self):
        if self._sid is not None:
            self.engine.stream_reset(self._sid)
