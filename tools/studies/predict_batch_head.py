"""Study (round 6): where the HOST spends the head of a configs[2] predict_batch call (the GPU time line shows the first encoder
kernel 1.8 ms after the call starts).  Wall time of the wrapped steps, per call, with the device idle at the start of every call.
usage: python tools/studies/predict_batch_head.py [greedy|beam]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from masr_amd.predict import MASRPredictor     # noqa: E402
from masr_amd.utils import synthetic           # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'greedy'
rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm_h[i, :lens[i]] for i in range(64)]
pred = bench.facade('squeezeformer', 'ctc_greedy' if kind == 'greedy' else 'ctc_beam_search', 0, streaming=False)
eng = pred.predictor.engine
acc, marks = defaultdict(float), []
t_call = [0.0]


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            acc[label] += t1 - t0
            marks.append((label, (t0 - t_call[0]) * 1e3, (t1 - t_call[0]) * 1e3))
    setattr(obj, name, timed)


wrap(pred, '_load_audio')
wrap(pred, '_stage_batch')
wrap(pred, '_prepare')
wrap(eng, 'host_gains')
wrap(eng, 'transcribe_rows')
wrap(pred, '_predict_local')
for _ in range(3):
    pred.predict_batch(audio, batch_size=32)
torch.cuda.synchronize()
acc.clear()
N = 10
tot = 0.0
for it in range(N):
    marks.clear()
    t_call[0] = time.perf_counter()
    pred.predict_batch(audio, batch_size=32)
    tot += time.perf_counter() - t_call[0]
    torch.cuda.synchronize()
print(f'{kind}: {tot / N * 1e3:.3f} ms per call; host wall time inside (per call, nested steps counted in their parents too):')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:20s} {v / N * 1e3:8.3f} ms')
print('last call, steps in order (start -> end, ms from the start of the call; _load_audio summed per pass):')
la = None
for label, a, b in marks:
    if label == '_load_audio':
        la = (la[0], b, la[2] + b - a) if la else (a, b, b - a)
        continue
    if la:
        print(f'  _load_audio x pass     {la[0]:7.3f} -> {la[1]:7.3f}  ({la[2]:.3f} ms inside)')
        la = None
    print(f'  {label:22s} {a:7.3f} -> {b:7.3f}')
