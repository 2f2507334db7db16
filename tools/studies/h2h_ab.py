"""Where do the ~0.2 ms per step between bench.py's 'full' step (PCM resident in HBM) and its 'host' step (PCM over PCIe)
come from?  Variants of the contract step on one engine, interleaved, 40 steps each."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
from masr_amd import parallel  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

eng = bench.make_engine('conformer', 0)
cs = bench.ContractStep(eng, 0, 1, synthetic.synthetic_vocab(bench.VOCAB))
STEPS = 40


def timed(fn, flush=True):
    return parallel.timed_region(fn, STEPS, 3, flush=cs.flush if flush else None) * 1e3 / STEPS


def full_plus_dummy_copy():
    """the 'full' step + an unrelated 10.2 MB H2D copy per step on its own stream: the DMA's interference alone"""
    st = torch.cuda.Stream()
    dst = torch.empty_like(cs.pcm)

    def step(i):
        cs.step(i, 'full')
        with torch.cuda.stream(st):
            dst.copy_(cs.pcm_host, non_blocking=True)
    return step


def host_two_ahead():
    """'host' step with the copy of step i + 2 issued when step i is enqueued (three buffers): PCIe two steps ahead"""
    st = torch.cuda.Stream()
    bufs = [torch.empty_like(cs.pcm) for _ in range(3)]
    ready = [torch.cuda.Event() for _ in range(3)]
    done = [None, None, None]
    issued = set()

    def copy(k):
        with torch.cuda.stream(st):
            if done[k % 3] is not None:
                st.wait_event(done[k % 3])
            bufs[k % 3].copy_(cs.pcm_host, non_blocking=True)
            ready[k % 3].record()
        cs.prepare(bufs[k % 3], stream=st)
        issued.add(k)

    def step(i):
        for k in (i, i + 1):
            if k not in issued:
                copy(k)
        torch.cuda.current_stream().wait_event(ready[i % 3])
        cs.pcm, keep = bufs[i % 3], cs.pcm
        real_prepare, cs.prepare = cs.prepare, (lambda *a, **k: None)      # the step itself must not prepare anything
        try:
            cs.step(i, 'full')
        finally:
            cs.pcm, cs.prepare = keep, real_prepare
        ev = torch.cuda.Event()
        ev.record()
        done[i % 3] = ev
        copy(i + 2)
        issued.discard(i - 3)
    return step


variants = [('full (PCM resident)', lambda i: cs.step(i, 'full')),
            ('host (shipped: copy of step i+1 issued behind step i, mean squares behind the copy)', lambda i: cs.step(i, 'host')),
            ('full + an unrelated 10.2 MB H2D copy per step', full_plus_dummy_copy()),
            ('host, copies two steps ahead, mean squares behind the copy', host_two_ahead())]
for rep in range(2):
    for name, fn in variants:
        cs.h2d = None
        cs.prefetch_pcm_for = None
        print(f'{name:90s} {timed(fn):7.3f} ms / step', flush=True)
