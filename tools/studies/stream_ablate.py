"""Diagnostic: chunk-call latency of n lock-step streams (argv[1], default 16) with the small-M projection kernel
(rowgemm_small.hip) and the few-query attention kernel switched on and off (masr_debug_set keys 6 / 7), plus the per-launch time of the gemm-class / FFN launches
(HIP events, incl. gaps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16
feats = torch.randn(ns, 998, 80, device='cuda') * 3 + 13
sids = [e.stream_open(300) for _ in range(ns)]


def run(kind=0):
    lat = []
    for sid in sids:
        e.stream_reset(sid)
    e.profile_select(kind); e.profile_read()
    for cur in range(0, 998 - 67 + 1, 64):
        t0 = time.perf_counter()
        _, idx, _ = e.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
        idx.cpu()
        lat.append(time.perf_counter() - t0)
    ms, n, fl = e.profile_read()
    return lat, ms * 1e3 / max(n, 1), n


for small_gemm in (1, 0):
    for fewq in (1, 0):
        e.lib.masr_debug_set(e.h, 6, small_gemm)
        e.lib.masr_debug_set(e.h, 7, fewq)
        run()
        lat = []
        for _ in range(6):
            lat += run()[0]
        _, g_us, g_n = run(1)
        _, f_us, f_n = run(2)
        print(f'streams={ns} rowgemm_small={small_gemm} attention_fewq={fewq}: chunk call p50 {np.percentile(lat, 50) * 1e3:.3f} ms  p95 '
              f'{np.percentile(lat, 95) * 1e3:.3f} ms | gemm-class {g_us:.2f} us x {g_n}, ffn {f_us:.2f} us x {f_n}')
