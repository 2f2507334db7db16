"""Study (round 6): which of {one lane, two lanes} x {GPU search, host search} disagree on BASELINE configs[2]'s transcripts"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from masr_amd.decoders.lm_scorer import write_synthetic_arpa   # noqa: E402
from masr_amd.utils import synthetic           # noqa: E402

rng = np.random.default_rng(1234)
lens = rng.integers(32000, 320001, 64)
pcm = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm[i, :lens[i]].copy() for i in range(64)]
d = tempfile.mkdtemp()
conf = {'alpha': 2.2, 'beta': 4.3, 'beam_size': 300, 'cutoff_prob': 0.99, 'cutoff_top_n': 40, 'num_processes': 10,
        'language_model_path': write_synthetic_arpa(os.path.join(d, 'lm.arpa'), synthetic.synthetic_vocab(bench.VOCAB), order=3, seed=5)}
pred = bench.facade('squeezeformer', 'ctc_beam_search', 0, streaming=False, beam_conf=conf)
dec = pred.beam_search_decoder
res = {}
for rep in range(2):
    for lanes in ('1', '2'):
        for gpu in (True, False):
            os.environ['MASR_LANES'] = lanes
            dec.use_gpu_search = gpu
            res[(rep, lanes, gpu)] = pred.predict_batch(audio, batch_size=32)
base = res[(0, '1', False)]
for k, v in res.items():
    diff = [i for i in range(64) if v[i]['text'] != base[i]['text'] or abs(v[i]['score'] - base[i]['score']) > 1e-3 * max(1, abs(base[i]['score']))]
    print(f'rep {k[0]} lanes {k[1]} gpu_search {k[2]}: {len(diff)} utterances differ from (lanes 1, host search): {diff[:10]}'
          + (f' lens {[int(lens[i]) for i in diff[:10]]}' if diff else ''), flush=True)
