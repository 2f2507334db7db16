"""Experiment: the batch-32 x 10 s offline step as ONE engine call (M = 7936 rows, 248 row blocks) against the same 32
utterances as two half-batches on two engines and two HIP streams (124 row blocks each, launched interleaved): do the
latency-bound phases of one half (attention, out-projection chain, front-end) hide under the other half's FFN kernels?
usage: python tools/studies/dual_pipeline_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

V = 4233
sd = synthetic.conformer_state_dict(0, V)
e0, e1, e2 = HipEngine(sd, vocab_size=V), HipEngine(sd, vocab_size=V), HipEngine(sd, vocab_size=V)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n32 = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
n16 = n32[:16].contiguous()
halves = [pcm[:16].contiguous(), pcm[16:].contiguous()]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    e0.transcribe_batch(pcm, n32)


def two():
    with torch.cuda.stream(s1):
        e1.transcribe_batch(halves[0], n16)
    with torch.cuda.stream(s2):
        e2.transcribe_batch(halves[1], n16)


def two_seq():
    e1.transcribe_batch(halves[0], n16)
    e1.transcribe_batch(halves[1], n16)


def timeit(fn, steps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ref = e0.transcribe_batch(pcm, n32)
a = e1.transcribe_batch(halves[0], n16)
torch.cuda.synchronize()
assert torch.equal(ref[1][:16], a[1])          # equal-length batch: the same transcripts either way
for r in range(3):
    print(f'round {r}: one call of 32: {timeit(one):.3f} ms | two halves, two streams: {timeit(two):.3f} ms | '
          f'two halves, one stream: {timeit(two_seq):.3f} ms')
