"""AudioFeaturizer with the reference's constructor / ``featurize`` contract
(masr/data_utils/featurizer/audio_featurizer.py:21-69,141-154), computed by the batched HIP front-end
(masr_fbank_batch / masr_mfcc_batch / masr_linear_batch): dB normalisation, int16 truncation and the Kaldi fbank,
Kaldi MFCC or linear log-power spectrogram all run on the GPU.
"""
import numpy as np
import torch

from masr_amd import runtime


class AudioFeaturizer(object):
    def __init__(self, feature_method='fbank', n_mels=80, n_mfcc=40, sample_rate=16000, use_dB_normalization=True,
                 target_dB=-20, train=False):
        if feature_method not in ('fbank', 'mfcc', 'linear'):
            raise Exception('没有{}预处理方法'.format(feature_method))
        if sample_rate != 16000 or (feature_method != 'linear' and n_mels != 80):
            raise Exception('the MI355X front-end kernels are built for 16 kHz and 80 mel bins')
        if feature_method == 'mfcc' and not 0 < n_mfcc <= n_mels:
            raise Exception('n_mfcc must be in [1, n_mels]')
        self._feature_method = feature_method
        self._target_sample_rate = sample_rate
        self._n_mels = n_mels
        self._n_mfcc = n_mfcc
        self._use_dB_normalization = use_dB_normalization
        self._target_dB = target_dB
        self._train = train
        self._engine = None

    def bind(self, engine):
        """Use a model engine's stream / device instead of the shared weight-less one."""
        self._engine = engine

    def _eng(self):
        return self._engine if self._engine is not None else runtime.aux_engine()

    def featurize(self, audio_segment):
        """AudioSegment -> np.float32 [T, 80].  Like the reference, this mutates the segment
        (resample, dB gain applied in place)."""
        if audio_segment.sample_rate != self._target_sample_rate:
            audio_segment.resample(self._target_sample_rate)
        eng = self._eng()
        x = audio_segment._samples
        n = x.shape[0]
        if n < (320 if self._feature_method == 'linear' else 400):
            if self._use_dB_normalization and n > 0:
                self._normalize_only(eng, audio_segment)
            return np.zeros((0, self.feature_dim), np.float32)
        xs = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))[None].to(eng.device)
        ns = torch.tensor([n], dtype=torch.int32, device=eng.device)
        # the gain is evaluated with the reference's own numpy expressions on this host (engine.reference_gains; raises beyond
        # max_gain_db like AudioSegment.normalize), the samples are scaled by it on the device
        gain = eng.host_gains(xs, ns, self._target_dB) if self._use_dB_normalization else None
        feats, frames = eng.features_batch(self._feature_method, xs, ns, self._use_dB_normalization, self._target_dB,
                                           n_mfcc=self._n_mfcc, gain_in=gain)
        if self._use_dB_normalization:
            audio_segment.gain_linear(float(gain[0]))          # the reference normalises in place (audio.py:304)
        return feats[0].cpu().numpy()

    def _normalize_only(self, eng, seg):
        xs = torch.from_numpy(np.ascontiguousarray(seg._samples, dtype=np.float32))[None].to(eng.device)
        ns = torch.tensor([xs.shape[1]], dtype=torch.int32, device=eng.device)
        seg.gain_linear(float(eng.host_gains(xs, ns, self._target_dB)[0]))

    @property
    def feature_method(self):
        return self._feature_method

    @property
    def feature_dim(self):
        """audio_featurizer.py:141-154"""
        return {'linear': 161, 'mfcc': self._n_mfcc, 'fbank': self._n_mels}[self._feature_method]
