"""Timing experiment: the chunk step of n lock-step streams with every layer running on layer 0's weights (masr_debug_set key
19: the 11.5 MB of a layer stay hot in the L2s / Infinity Cache) against the real step, whose 138 MB of weights cycle through
once per step.  The difference is what prefetching the next launch's weights could buy at most.  The transcripts of the
experiment are meaningless.  usage: python tools/studies/chunk_hot_weights.py [n_streams ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
for ns in [int(a) for a in sys.argv[1:]] or [16, 128]:
    feats = torch.randn(ns, 998, 80, device='cuda') * 3 + 13
    sids = [e.stream_open(300) for _ in range(ns)]

    def run():
        lat = []
        for sid in sids:
            e.stream_reset(sid)
        for cur in range(0, 998 - 67 + 1, 64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, idx, _ = e.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        return lat

    for hot in (0, 1, 0, 1):
        e.lib.masr_debug_set(e.h, 19, hot)
        run()
        lat = []
        for _ in range(4):
            lat += run()
        print(f'streams={ns:4d}  layer-0 weights everywhere={hot}: chunk step p50 {np.percentile(lat, 50) * 1e3:.3f} ms  '
              f'p95 {np.percentile(lat, 95) * 1e3:.3f} ms')
    e.lib.masr_debug_set(e.h, 19, 0)
    for sid in sids:
        e.stream_close(sid)
