"""Study (round 6): the fused-stage threshold of the Squeezeformer's half-rate layers (masr_debug_set key 36: row blocks from
which a layer takes the two fused stage kernels; below it the d_ff-split separate launches) when the two passes of BASELINE
configs[2] run side by side on the engine's two lanes -- unused CUs of a small grid are then filled by the other pass, so the
fused kernels' lower CU-time may win where their latency lost (one lane: fused from 96 row blocks 21.2 vs 18.9 ms)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from masr_amd.utils import synthetic           # noqa: E402

rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm_h[i, :lens[i]] for i in range(64)]
pred = bench.facade('squeezeformer', 'ctc_greedy', 0, streaming=False)
eng = pred.predictor.engine
ref = None
for lanes in ('2', '1'):
    os.environ['MASR_LANES'] = lanes
    for blocks in (192, 128, 96, 64, 0, 192, 128):
        eng.lib.masr_debug_set(eng.h, 36, blocks)
        for _ in range(3):
            res = pred.predict_batch(audio, batch_size=32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            res = pred.predict_batch(audio, batch_size=32)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        if ref is None:
            ref = res
        same = sum(a == b for a, b in zip(ref, res))
        print(f'lanes {lanes}, fused stages from {blocks} row blocks: {dt * 1e3:.3f} ms per call, {same}/64 results equal to the first setting', flush=True)
eng.lib.masr_debug_set(eng.h, 36, 128)
