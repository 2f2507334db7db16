"""A/B for the streaming chunk step: masr_debug_set key KEY (argv[1]) at the values argv[2:], chunk-call latency of 16 and 128
lock-step streams and identity of the frame argmax (alternating on one box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

KEY = int(sys.argv[1])
VALS = [int(v) for v in sys.argv[2:]]
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
for ns in (16, 128):
    feats = torch.randn(ns, 998, 80, device='cuda') * 3 + 13
    sids = [e.stream_open(300) for _ in range(ns)]

    def run():
        lat, out = [], []
        for sid in sids:
            e.stream_reset(sid)
        for cur in range(0, 998 - 67 + 1, 64):
            t0 = time.perf_counter()
            _, idx, mp = e.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
            out.append((idx.cpu(), mp.cpu()))
            lat.append(time.perf_counter() - t0)
        return lat, out

    ref = None
    for rnd in range(3):
        for v in VALS:
            e.lib.masr_debug_set(e.h, KEY, v)
            run()
            lat = []
            for _ in range(4):
                l, out = run()
                lat += l
            if ref is None:
                ref = out
            same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ref, out))
            print(f'streams={ns:4d} key {KEY}={v}: chunk call p50 {np.percentile(lat, 50) * 1e3:.3f} ms  p95 {np.percentile(lat, 95) * 1e3:.3f} ms'
                  f'  identical to the first run: {same}')
    for sid in sids:
        e.stream_close(sid)
