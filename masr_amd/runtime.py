"""Process-wide weight-less HIP engine for the model-independent kernels (fbank, argmax, CTC
collapse).  Created lazily on first use; raises (no CPU fallback) when there is no GPU."""
import torch

_aux = None


def aux_engine():
    global _aux
    if _aux is None:
        from .engine import HipEngine
        _aux = HipEngine(None, device=torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _aux
