"""Chunk-call latency (p50 / p95) of 16 and 128 lock-step streams for whatever libmasr_hip.so is installed, and a checksum of the
frame decisions -- for A/B runs of two builds on one box (REPS=3 bash tools/lib_ab.sh with KT=tools/chunk_lat.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ''
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
line = f'{label:8s}'
# optional A/B inside one process: MASR_AB='30:0,30:1' runs the sizes once per setting of debug key 30
settings = [tuple(int(v) for v in kv.split(':')) for kv in os.environ.get('MASR_AB', '').split(',') if kv] or [None]
for ns, ab in [(n, a) for a in settings for n in (16, 128)]:
    if ab is not None:
        e.lib.masr_debug_set(e.h, ab[0], ab[1])
        line += f' | key {ab[0]} = {ab[1]}'
    feats = torch.randn(ns, 998, 80, device='cuda', generator=torch.Generator('cuda').manual_seed(3)) * 3 + 13
    sids = [e.stream_open(300) for _ in range(ns)]

    def run():
        lat, chk = [], 0
        for sid in sids:
            e.stream_reset(sid)
        for cur in range(0, 998 - 67 + 1, 64):
            t0 = time.perf_counter()
            _, idx, mp = e.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
            chk += int(idx.cpu().sum())
            lat.append(time.perf_counter() - t0)
        return lat, chk

    run()
    lat = []
    for _ in range(5):
        l, chk = run()
        lat += l
    line += f' | {ns} streams: p50 {np.percentile(lat, 50) * 1e3:.3f} ms p95 {np.percentile(lat, 95) * 1e3:.3f} ms (sum of frame ids {chk})'
    for sid in sids:
        e.stream_close(sid)
print(line)
