"""masr_amd -- MI355X (gfx950) native inference hot path behind MASR's Python surface.

PCM -> Kaldi fbank -> Conformer encoder -> CTC greedy, as hand-written HIP kernels in
``libmasr_hip.so`` (C ABI: ``include/masr_hip.h``).  There is no CPU fallback: anything that
computes imports ``masr_amd._lib`` which raises if the library is missing.
"""
__version__ = '0.1.0'
SUPPORT_MODEL = ['squeezeformer', 'efficient_conformer', 'conformer', 'deepspeech2']
