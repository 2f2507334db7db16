"""Host-side time line of one predict_batch call of BASELINE configs[2] (64 utterances, two passes of 32, GPU prefix beam search with
the synthetic 3-gram LM): when each pass has been launched and when its results are in (time.perf_counter, the call synchronises
only where it collects)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from masr_amd.decoders.lm_scorer import write_synthetic_arpa  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
audio = [pcm_h[i, :lens[i]] for i in range(64)]
d = tempfile.mkdtemp(prefix='masr_lm_')
conf = {'alpha': 2.2, 'beta': 4.3, 'beam_size': 300, 'cutoff_prob': 0.99, 'cutoff_top_n': 40, 'num_processes': 10,
        'language_model_path': write_synthetic_arpa(os.path.join(d, 'lm.arpa'), synthetic.synthetic_vocab(bench.VOCAB), seed=5)}
SHARP = os.environ.get('MASR_PROFILE_SHARP') == '1'
PASS = os.environ.get('MASR_BENCH_BEAM_PASS', 'balanced')
PASS = PASS if PASS == 'balanced' else ([int(v) for v in PASS.split(',')] if ',' in PASS else int(PASS))
pred = bench.facade('squeezeformer', 'ctc_beam_search', 0, streaming=False, beam_conf=conf, head_gain=bench.SHARP_HEAD_GAIN if SHARP else None)
print(f'passes: {PASS}, sharp head: {SHARP}, side streams: {os.environ.get("MASR_BEAM_SIDES", "3")}, group: {os.environ.get("MASR_BEAM_GROUP", "4")}')
marks = []
orig_local = pred._predict_local


def local(segs, *a, **k):
    t0 = time.perf_counter()
    fetch = orig_local(segs, *a, **k)
    tag = 'pass of %d (longest %.1f s)' % (len(segs), max(s.num_samples for s in segs) / 16000)
    marks.append(('encoder enqueued: ' + tag, t0, time.perf_counter()))

    def timed(f):
        def timed_fetch():
            t1 = time.perf_counter()
            r = f()
            marks.append(('collected ' + tag, t1, time.perf_counter()))
            return r
        return timed_fetch
    if isinstance(fetch, tuple) and fetch[0] == 'held':          # round 6: pruned here, searched per group of passes (MASR_BEAM_GROUP)
        return fetch
    return timed(fetch)


pred._predict_local = local
for rep in range(3):
    pred.predict_batch(audio, batch_size=PASS)
for rep in range(2):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred.predict_batch(audio, batch_size=PASS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f'call {rep}: {1e3 * (t1 - t0):.1f} ms')
    for name, a, b in marks:
        print(f'   {name:45s} from {1e3 * (a - t0):6.1f} to {1e3 * (b - t0):6.1f} ms')
