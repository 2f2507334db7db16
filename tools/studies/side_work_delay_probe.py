"""Study (round 6): how late does work on a side stream START while the compute stream runs a dense chain of chip-filling kernels?
(The preparation of a pass -- a 12 MB upload + two small kernels -- issued beside a running encoder pass starts 4 - 6 ms late.)
A pinned -> device copy and a tiny kernel, each on a side stream of the library, issued 1 ms after a ~12 ms chain of GEMMs has been
queued on the compute stream; events give the delay from issue to start and the duration.
usage: python tools/studies/side_work_delay_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from masr_amd.engine import HipEngine          # noqa: E402

torch.cuda.set_device(0)
eng = HipEngine(None)
side = eng.side_stream(2)
a = torch.randn(4096, 4096, device='cuda')
b = torch.randn(4096, 4096, device='cuda')
host = torch.empty(6 << 20, dtype=torch.int16, pin_memory=True)      # 12 MB
dev = torch.empty_like(host, device='cuda')
small = torch.zeros(1024, device='cuda')
main = torch.cuda.current_stream()


def chain(n):
    for _ in range(n):
        torch.mm(a, b)


def run(kind, busy):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    s0, s1, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record(main)
    if busy:
        chain(60)
    m1.record(main)
    time.sleep(0.001)
    issued = time.perf_counter()
    with torch.cuda.stream(side):
        s0.record()
        if kind == 'copy':
            dev.copy_(host, non_blocking=True)
        else:
            small.add_(1)
        s1.record()
    s1.synchronize()
    host_wait = (time.perf_counter() - issued) * 1e3
    torch.cuda.synchronize()
    return t0.elapsed_time(s0), t0.elapsed_time(s1), t0.elapsed_time(m1), host_wait


chain(10)
for kind in ('copy', 'kernel'):
    for busy in (False, True, True):
        st, en, mainend, hw = run(kind, busy)
        print(f'{kind:6s} compute stream {"busy" if busy else "idle"}: side work starts at {st:6.2f} ms, ends at {en:6.2f} ms '
              f'(compute chain ends at {mainend:6.2f} ms); the host waited {hw:.2f} ms for it', flush=True)
