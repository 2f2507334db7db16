"""Chunk-call latency of n lock-step streams against the two size thresholds of the streaming kernels: the row-block count
below which the K-split projection kernel (rowgemm_small.hip) is used (masr_debug_set key 12) and below which the fused FFN
splits d_ff across workgroups (key 13).  usage: python tools/studies/chunk_step_ab.py n_streams [n_streams ...]"""
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
            t0 = time.perf_counter()
            _, idx, _ = e.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
            idx.cpu()
            lat.append(time.perf_counter() - t0)
        return lat

    for small, split in ((64, 64), (128, 64), (64, 192), (128, 192), (256, 256), (128, 128), (64, 64)):
        e.lib.masr_debug_set(e.h, 12, small)
        e.lib.masr_debug_set(e.h, 13, split)
        run()
        lat = []
        for _ in range(4):
            lat += run()
        print(f'streams={ns:4d} (M = {16 * ns:5d} rows)  small-M kernel below {small:3d} row blocks, FFN split below {split:3d}: '
              f'chunk call p50 {np.percentile(lat, 50) * 1e3:.3f} ms  p95 {np.percentile(lat, 95) * 1e3:.3f} ms')
    for sid in sids:
        e.stream_close(sid)
