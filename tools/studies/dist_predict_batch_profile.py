"""Study (round 6): host profile of MASRPredictor.predict_batch on the RCCL path with ONE rank (MASR_FORCE_DIST=1) against the plain
path, configs[2] sharpened head: where the sharded call spends its extra time.  usage: MASR_FORCE_DIST=1 python tools/studies/dist_predict_batch_profile.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from masr_amd import parallel                  # noqa: E402
from masr_amd.decoders.lm_scorer import write_synthetic_arpa   # noqa: E402
from masr_amd.utils import synthetic           # noqa: E402

rank, world, local = parallel.init_from_env()
torch.cuda.set_device(local)
rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm_h[i, :lens[i]] for i in range(64)]
d = tempfile.mkdtemp()
conf = {'alpha': 2.2, 'beta': 4.3, 'beam_size': 300, 'cutoff_prob': 0.99, 'cutoff_top_n': 40, 'num_processes': 10,
        'language_model_path': write_synthetic_arpa(os.path.join(d, 'lm.arpa'), synthetic.synthetic_vocab(bench.VOCAB), seed=5)}
pred = bench.facade('squeezeformer', 'ctc_beam_search', local, streaming=False, beam_conf=conf, head_gain=bench.SHARP_HEAD_GAIN)
for _ in range(3):
    pred.predict_batch(audio, batch_size=32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    pred.predict_batch(audio, batch_size=32)
torch.cuda.synchronize()
print(f'collectives on: {parallel.collectives_on()}; {(time.perf_counter() - t0) * 100:.3f} ms per call')
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    pred.predict_batch(audio, batch_size=32)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
