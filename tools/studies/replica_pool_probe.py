"""Study (round 6), REJECTED: BASELINE configs[4]'s lock-step streams on ONE StreamPool against R engine replicas on the same GPU
(ReplicaStreamPool below: R pools, R host threads, R streams).  The premise -- a chunk step of a few dozen streams is a chain of
short launches that leaves most of the chip idle, so two chains advance side by side -- does not hold: the d_ff-split launches
of the chunk step are built to fill 256 CUs at any stream count.  Measured (one box, audio-s/s, call p50):
128 streams: 1 replica 19 765 (3.71 ms), 2: 19 309 / 16 693 (3.78 / 4.23 ms), 3: 15 614, 4: 14 065;
16 streams: 1 replica 7 229 (1.29 ms), 2: 6 805 / 5 320.  Identical transcripts.  Not shipped.
usage: python tools/studies/replica_pool_probe.py [streams] [R,R,..]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from masr_amd.serving import StreamPool       # noqa: E402
from masr_amd.utils import synthetic           # noqa: E402

class ReplicaStreamPool:
    """Several ``StreamPool``s -- one per ENGINE REPLICA on the same GPU (``MASRPredictor``s built from the same model) -- behind
    the StreamPool interface.  A chunk step of a few dozen streams is a chain of ~100 short launches that fills a fraction of the
    chip and is bound by launch-to-launch latency, not by arithmetic: two such chains advance side by side at almost the speed
    of one.  Session ``n`` lives on replica ``n % R`` for its whole life (caches never move); ``step()`` advances every replica
    with fed sessions on a host thread and a stream of its own (``masr_pool_step`` releases the GIL for the whole C call) and
    merges the results.  The reference serves one ``predict_stream`` session per predictor (predict.py:237-343); R replicas
    are R of its predictors sharing a GPU."""

    def __init__(self, predictors, max_frames_out=0):
        from concurrent.futures import ThreadPoolExecutor
        if not predictors:
            raise ValueError('ReplicaStreamPool: at least one predictor')
        self.pools = [StreamPool(p, max_frames_out) for p in predictors]
        self.max_frames_out = max_frames_out
        self.vocab = self.pools[0].vocab
        dev = self.pools[0].engine.device
        if any(p.engine.device != dev for p in self.pools):
            raise ValueError('ReplicaStreamPool: the replicas share ONE GPU (ranks of a node: parallel.ShardedStreamPool)')
        self.device = dev
        # replica 0 steps on the caller's stream, the others on streams of their own
        self._streams = [None] + [torch.cuda.Stream(device=dev) for _ in self.pools[1:]]
        self._threads = ThreadPoolExecutor(len(self.pools), thread_name_prefix='masr_replica') if len(self.pools) > 1 else None
        self._next = 0
        self.errors = {}

    def _of(self, handle):
        return self.pools[handle % len(self.pools)], handle // len(self.pools)

    def open(self):
        r = self._next % len(self.pools)
        self._next += 1
        return self.pools[r].open() * len(self.pools) + r

    def close(self, handle):
        pool, h = self._of(handle)
        pool.close(h)
        self.errors.pop(handle, None)

    def reset(self, handle):
        pool, h = self._of(handle)
        pool.reset(h)
        self.errors.pop(handle, None)

    def feed(self, handle, audio_data, is_end=False, **kw):
        pool, h = self._of(handle)
        pool.feed(h, audio_data, is_end, **kw)

    def last_tokens(self, handle):
        pool, h = self._of(handle)
        return pool.last_tokens(h)

    def shutdown(self):
        for p in self.pools:
            p.shutdown()
        if self._threads is not None:
            self._threads.shutdown(wait=True)
            self._threads = None

    def _step_replica(self, r, caller_stream):
        torch.cuda.set_device(self.device)
        if r == 0:
            with torch.cuda.stream(caller_stream):
                return self.pools[0].step()
        st = self._streams[r]
        st.wait_stream(caller_stream)
        with torch.cuda.stream(st):
            res = self.pools[r].step()
        caller_stream.wait_stream(st)
        return res

    def step(self):
        R = len(self.pools)
        caller = torch.cuda.current_stream(self.device)
        if R == 1:
            parts = [self.pools[0].step()]
        else:
            futures = [self._threads.submit(self._step_replica, r, caller) for r in range(1, R)]
            parts = [self._step_replica(0, caller)] + [f.result() for f in futures]
        out = {}
        self.errors = {}
        for r, res in enumerate(parts):
            for h, v in res.items():
                out[h * R + r] = v
            for h, msg in self.pools[r].errors.items():
                self.errors[h * R + r] = msg
        return out


n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 128
counts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '1,2,3,4').split(',')]
chunk, n_chunks = 8000, 20
pcm = synthetic.synthetic_pcm(n_streams, chunk * n_chunks, seed=4321)
wire = [[pcm[j, c * chunk:(c + 1) * chunk].tobytes() for j in range(n_streams)] for c in range(n_chunks)]
preds = [bench.facade('conformer', 'ctc_greedy', 0) for _ in range(max(counts))]
ref = None
for R in counts + counts[:2]:
    pool = ReplicaStreamPool(preds[:R], max_frames_out=320)
    ids = [pool.open() for _ in range(n_streams)]
    lat, last = [], None

    def utterance(record):
        global last
        for g in ids:
            pool.reset(g)
        for c in range(n_chunks):
            t0 = time.perf_counter()
            for j, g in enumerate(ids):
                pool.feed(g, wire[c][j], is_end=(c == n_chunks - 1))
            last = pool.step()
            if record:
                lat.append(time.perf_counter() - t0)
    utterance(False)
    torch.cuda.synchronize()
    reps = 4 if n_streams > 32 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        utterance(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    texts = [last[g]['text'] for g in ids]
    if ref is None:
        ref = texts
    print(f'{n_streams} streams on {R} replica(s): {n_streams * n_chunks * 0.5 * reps / dt:.0f} audio-s/s, call p50 '
          f'{np.percentile(lat, 50) * 1e3:.3f} ms p95 {np.percentile(lat, 95) * 1e3:.3f} ms; final transcripts equal to the first run: '
          f'{sum(a == b for a, b in zip(ref, texts))}/{n_streams}', flush=True)
    for g in ids:
        pool.close(g)
    pool.shutdown()
