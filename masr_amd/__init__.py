"""masr_amd -- MI355X (gfx950) native inference hot path behind MASR's Python surface.

PCM -> Kaldi fbank -> Conformer encoder -> CTC greedy, as hand-written HIP kernels in
``libmasr_hip.so`` (C ABI: ``include/masr_hip.h``).  There is no CPU fallback: anything that
computes imports ``masr_amd._lib`` which raises if the library is missing.
"""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin in creation order; two streams that share a
# queue run one behind the other.  predict_batch keeps the prefix searches of consecutive passes on two side streams next to the
# main stream, and a process that has created other streams before (copy / preparation streams, pools, a second engine) finds
# them aliased with the main stream: BASELINE configs[2] 45.9 -> 63.4 ms per call inside the long bench process with 8 queues,
# 47.7 ms with 16 (round 5, same box; rounds 3-4 ran with 8).  The variable is read when the HIP runtime starts, so this only
# takes effect when masr_amd is imported before the first device call; a value set by the user wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

__version__ = '0.1.0'
SUPPORT_MODEL = ['squeezeformer', 'efficient_conformer', 'conformer', 'deepspeech2']
