"""AudioSegment -- the subset of the reference container (masr/data_utils/audio.py) that the
inference hot path touches: construction from ndarray / PCM bytes / wav file, float32 samples
scaled to [-1, 1], in-place gain, int16 export.  RMS measurement and the actual normalisation
arithmetic of the hot path run on the GPU (masr_fbank_batch); ``gain_linear`` lets the featurizer
mirror the reference's in-place mutation of ``samples`` (audio.py:256-264, relied upon by
predict_stream, predict.py:274).
"""
import io
import wave

import numpy as np


class AudioSegment(object):
    def __init__(self, samples, sample_rate):
        """Samples are converted to float32, ints scaled to [-1, 1]; multi-channel -> mean
        (audio.py:24-32,532-546)."""
        samples = np.asarray(samples)
        self._sample_rate = sample_rate
        # mono int16 input: the batched device front-end takes the PCM as it is (x / 2^15 happens in the kernel, the same
        # float32 value) -- half the bytes over PCIe -- and the float32 copy is made only when somebody asks for it
        # (``_samples``).  Dropped as soon as the float samples are modified.
        self._pcm16 = samples if samples.dtype == np.int16 and samples.ndim == 1 else None
        self._float = None if self._pcm16 is not None else self._to_float(samples)

    @staticmethod
    def _to_float(samples):
        f = samples.astype('float32')
        if samples.dtype in (np.int8, np.int16, np.int32, np.int64):
            f *= (1. / 2 ** (np.iinfo(samples.dtype).bits - 1))
        elif samples.dtype not in (np.float16, np.float32, np.float64):
            raise TypeError("Unsupported sample type: %s." % samples.dtype)
        if f.ndim >= 2:
            f = np.mean(f, 1)
        return f

    @property
    def _samples(self):
        if self._float is None:
            self._float = self._to_float(self._pcm16)
        return self._float

    @_samples.setter
    def _samples(self, value):
        self._float = value

    # ---- constructors (audio.py:56-152) ---------------------------------------------------------
    @classmethod
    def from_file(cls, file):
        """wav path or binary file object (stdlib ``wave``; the reference uses soundfile)."""
        with wave.open(file, 'rb') as w:
            return cls._from_wave(w)

    @classmethod
    def from_bytes(cls, data):
        with wave.open(io.BytesIO(data), 'rb') as w:
            return cls._from_wave(w)

    @classmethod
    def _from_wave(cls, w):
        width, ch = w.getsampwidth(), w.getnchannels()
        if width not in (2, 4):
            raise ValueError(f'unsupported sample width {width}')
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype='<i%d' % width)
        if ch > 1:
            raw = raw.reshape(-1, ch)
        return cls(raw, w.getframerate())

    @classmethod
    def from_pcm_bytes(cls, data, channels=1, samp_width=2, sample_rate=16000):
        """audio.py:122-139 + buf_to_float (data_utils/utils.py:382-411)."""
        scale = 1.0 / float(1 << ((8 * samp_width) - 1))
        samples = scale * np.frombuffer(data, '<i%d' % samp_width).astype(np.float32)
        if channels > 1:
            samples = samples.reshape(-1, channels)
        return cls(samples, sample_rate)

    @classmethod
    def from_ndarray(cls, data, sample_rate=16000):
        return cls(data, sample_rate)

    # ---- accessors ------------------------------------------------------------------------------
    @property
    def samples(self):
        return self._samples.copy()

    @property
    def sample_rate(self):
        return self._sample_rate

    @property
    def num_samples(self):
        return (self._pcm16 if self._float is None else self._float).shape[0]

    @property
    def duration(self):
        return self.num_samples / float(self._sample_rate)

    def gain_linear(self, factor):
        """In-place ``samples *= factor`` in float32 (what gain_db does with 10**(gain/20))."""
        x = self._samples
        x *= np.float32(factor)
        self._pcm16 = None

    def resample(self, target_sample_rate, filter='kaiser_best'):
        """audio.py:306-317: ``resampy.resample(samples, sample_rate, target, filter=filter)`` in place.  resampy is absent from
        the image; its published band-limited sinc interpolation and filter parameters are restated in ``data_utils/resample.py``
        (same output length ``int(n * target / rate)``, same tap order and accumulation; **parity unpinned** -- nothing to
        compare with here)."""
        if target_sample_rate == self._sample_rate:
            return
        from .resample import resample_native
        self._samples = resample_native(self._samples, self._sample_rate, target_sample_rate, filter)
        self._pcm16 = None
        self._sample_rate = target_sample_rate
