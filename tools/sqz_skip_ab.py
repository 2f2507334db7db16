"""BASELINE configs[2]'s 64 utterances (2-20 s) through the Squeezeformer with ctc_greedy: ms per predict_batch call by pass policy
('balanced' = passes of equal padded size, 32 = fixed passes, 64 = ONE pass padded to the longest utterance) with the all-padding
row blocks skipped (masr_debug_set key 38 = 1, default) and computed (0); transcripts must agree across all of them.
usage: python tools/sqz_skip_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm_h[i, :lens[i]] for i in range(64)]
pred = bench.facade('squeezeformer', 'ctc_greedy', 0, streaming=False)
eng = pred.predictor.engine
ref = None
ONLY = os.environ.get('SQZ_AB_ONLY')          # e.g. '1:64' -> one configuration (for a rocprofv3 trace)
for skip in ((int(ONLY.split(":")[0]),) if ONLY else (7, 0)):
    eng.lib.masr_debug_set(eng.h, 38, skip)
    for mode in ((int(ONLY.split(':')[1]),) if ONLY else ('balanced', 32, 64)):
        for _ in range(2):
            res = pred.predict_batch(audio, batch_size=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            res = pred.predict_batch(audio, batch_size=mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        texts = [r['text'] for r in res]
        if ref is None:
            ref = texts
        same = sum(a == b for a, b in zip(ref, texts))
        print(f'skip padded blocks {skip}, passes {mode}: {1e3 * dt:.2f} ms per call, {same}/64 transcripts equal to the first run')
