"""masr_amd -- MI355X (gfx950) native inference hot path behind MASR's Python surface.

PCM -> Kaldi fbank -> Conformer encoder -> CTC greedy, as hand-written HIP kernels in
``libmasr_hip.so`` (C ABI: ``include/masr_hip.h``).  There is no CPU fallback: anything that
computes imports ``masr_amd._lib`` which raises if the library is missing.
"""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; two streams that share a queue run one
# behind the other.  predict_batch keeps the prefix searches of consecutive passes on two side streams next to the main stream,
# and a process that has created a few other streams before (copy streams, a second engine) would find them aliased: BASELINE
# configs[2] 14 300 -> 10 400 audio-s/s.  The variable is read when the HIP runtime starts, so this only takes effect when
# masr_amd is imported before the first device call; a value set by the user wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
SUPPORT_MODEL = ['squeezeformer', 'efficient_conformer', 'conformer', 'deepspeech2']
