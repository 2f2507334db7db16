"""Weight-less HIP engines for the model-independent kernels (fbank, argmax, CTC collapse, vocabulary pruning, prefix beam
search): ONE PER DEVICE, created lazily on first use from a thread whose current torch device is that GPU; raises (no CPU
fallback) when there is no GPU.

Per device because the C-ABI entry points select their engine's device (``hipSetDevice``) and take the caller's current stream:
a worker thread of GPU k (server.WorkerRouter: one thread per engine in one process) must never be handed the engine -- and with
it the workspaces, pinned read-back buffers and beam-search state -- of the device some other thread touched first."""
import threading

import torch

_aux = {}
_lock = threading.Lock()


def aux_engine(device=None):
    """the auxiliary engine of ``device`` (default: the calling thread's current torch device)"""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    device = int(getattr(device, 'index', device) or 0)
    eng = _aux.get(device)
    if eng is None:
        with _lock:
            eng = _aux.get(device)
            if eng is None:
                from .engine import HipEngine
                eng = _aux[device] = HipEngine(None, device=device)
    return eng
