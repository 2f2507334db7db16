"""Study (round 6): do the two length-sorted passes of BASELINE configs[2] finish earlier when their encoders run SIDE BY SIDE on
two streams (two lanes of one engine = two workspace sets, masr_select_lane) instead of one after the other on one stream?  The fused stage kernels hold one
32-row block per CU (242-256 VGPRs), so pass 1 (429 valid row blocks) takes two rounds with the second 68 % full and pass 2 (179)
one round 70 % full: three rounds where the work is 2.4.  PCM resident in HBM, gains precomputed: encoder + greedy rows only.
usage: python tools/studies/sqz_two_lane_probe.py [passes, e.g. 32,32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from masr_amd.engine import HipEngine          # noqa: E402
from masr_amd.utils import synthetic            # noqa: E402

VOCAB = 4233
passes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '32,32').split(',')]
torch.cuda.set_device(0)
sd = synthetic.squeezeformer_state_dict(0, VOCAB)
eng = HipEngine(sd, {}, streaming=False, use_model='squeezeformer')
rng = np.random.default_rng(1234)
lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
dev = eng.device
batches, lo = [], 0
for c in passes:
    sel = lens[lo:lo + c]
    x = torch.from_numpy(np.ascontiguousarray(pcm_h[lo:lo + c, :int(sel.max())])).to(dev)
    for i in range(c):
        x[i, int(sel[i]):] = 0
    n = torch.from_numpy(sel.copy()).to(dev)
    gain = eng.host_gains(x, n, -20.0)
    batches.append((x, n, gain))
    lo += c
if os.environ.get('PROBE_SHORT_FIRST') == '1':          # the pass of the SHORT utterances launched first (lane 0), the long one second
    batches.reverse()
streams = [torch.cuda.current_stream(dev), eng.side_stream(4)]
torch.cuda.synchronize()


def run(mode):
    outs = []
    if mode == 'sequence':
        with torch.cuda.stream(streams[0]):
            for x, n, g in batches:
                outs.append(eng.transcribe_rows(x, n, True, -20.0, gain_in=g))
    else:
        for k, (x, n, g) in enumerate(batches):
            eng.select_lane(k % 2)
            with torch.cuda.stream(streams[k % 2]):
                outs.append(eng.transcribe_rows(x, n, True, -20.0, gain_in=g))
        eng.select_lane(0)
    return outs


ref = None
for mode in ('sequence', 'side_by_side', 'sequence', 'side_by_side'):
    for _ in range(3):
        run(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        outs = run(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    rows = [o.cpu().numpy() for o in outs]
    if ref is None:
        ref = rows
    same = all(np.array_equal(a, b) for a, b in zip(ref, rows))
    print(f'passes {passes} {mode}: {dt * 1e3:.3f} ms per call (rows equal to the first run: {same})', flush=True)
