"""Where the host time of one StreamPool call goes: cProfile over feed + step of N lock-step streams (0.5 s chunks).
usage: python tools/stream_host_profile.py [n_streams] [top]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from masr_amd.serving import StreamPool  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pred = bench.facade('conformer', 'ctc_greedy', 0)
pool = StreamPool(pred, max_frames_out=320)
ids = [pool.open() for _ in range(n)]
chunk, n_chunks = 8000, 20
pcm = synthetic.synthetic_pcm(n, chunk * n_chunks, seed=4321)
chunks = [[pcm[j, c * chunk:(c + 1) * chunk].tobytes() for j in range(n)] for c in range(n_chunks)]


def utterance(lat=None):
    for g in ids:
        pool.reset(g)
    for c in range(n_chunks):
        t0 = time.perf_counter()
        for j, g in enumerate(ids):
            pool.feed(g, chunks[c][j], is_end=(c == n_chunks - 1))
        t1 = time.perf_counter()
        pool.step()
        if lat is not None:
            lat.append((t1 - t0, time.perf_counter() - t1))


utterance()
lat = []
for _ in range(3):
    utterance(lat)
a = np.array(lat) * 1e3
print(f'{n} streams: feed p50 {np.percentile(a[:, 0], 50):.3f} ms, step p50 {np.percentile(a[:, 1], 50):.3f} ms, '
      f'call p50 {np.percentile(a.sum(1), 50):.3f} ms')
if getattr(pool, '_c', None) is not None:
    import ctypes as C
    ph, st = (C.c_double * 6)(), C.c_int64()
    pool._lib.masr_pool_profile(pool._c, ph, C.byref(st), 1)
    for _ in range(3):
        utterance()
    pool._lib.masr_pool_profile(pool._c, ph, C.byref(st), 1)
    names = ('assemble samples', 'upload + mean squares + wait', 'gains (python evaluator)', 'features + frame bookkeeping',
             'windows: chunk steps enqueued', 'collapse + copy back + wait')
    print('masr_pool_step host time per call: ' + ' | '.join(f'{nm} {v / st.value:.3f} ms' for nm, v in zip(names, ph)))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    utterance()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(top)
