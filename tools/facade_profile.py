"""Where a MASRPredictor.predict_batch(32 x 10 s) / predict(test.wav) call spends its host time (GPU box).
usage: python tools/facade_profile.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

pred = bench.facade('conformer', 'ctc_greedy', 0)
eng = pred.predictor.engine
audio = list(synthetic.synthetic_pcm(32, 160000, seed=1234))
wav = np.load(os.path.join(bench.ROOT, 'tests', 'golden', 'testwav.npz'))['pcm']


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name, batch in (('b32', audio), ('b1', [wav])):
    segs = [pred._load_audio(a, 16000) for a in batch]
    n = np.array([s.num_samples for s in segs], np.int32)
    print(name, 'predict_batch            %.3f ms' % timed(lambda: pred.predict_batch(batch)))
    print(name, '  _length_hint + _load   %.3f ms' % timed(lambda: [pred._length_hint(a, 16000) for a in batch] and [pred._load_audio(a, 16000) for a in batch]))
    print(name, '  _stage_batch           %.3f ms' % timed(lambda: pred._stage_batch(segs, n)))
    print(name, '  _prepare (stage+gains) %.3f ms' % timed(lambda: pred._prepare(segs, n, True, -20.0)))
    xs, ns, gain = pred._prepare(segs, n, True, -20.0)
    print(name, '  transcribe_rows (GPU)  %.3f ms' % timed(lambda: eng.transcribe_rows(xs, ns, True, -20.0, gain_in=gain)))
    print(name, '  _predict_local         %.3f ms' % timed(lambda: pred._predict_local(segs)))
    rows = eng.transcribe_rows(xs, ns, True, -20.0, gain_in=gain).cpu().numpy()
    tp = rows.shape[1] - 2
    print(name, '  text of the rows       %.3f ms' % timed(lambda: [pred._text(r[:r[tp]]) for r in rows]))
