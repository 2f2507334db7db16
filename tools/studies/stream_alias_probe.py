"""Study (round 6): which of the library's side streams share a hardware queue with each other or with torch's default stream --
in a cold process and after torch.distributed (RCCL) was initialised first.  Two streams alias when a short kernel on one waits
for a long spin kernel on the other.  usage: [MASR_FORCE_DIST=1] python tools/studies/stream_alias_probe.py

Found with the side streams simply created at the first masr_create (one box, GPU_MAX_HW_QUEUES unset = 4 queues, streams dealt
round-robin in creation order):   cold process: {main, side2 (preparation), torch3} {side0, torch1, torch5} {side1, torch0, torch4}
{side3, side4, torch2};   after torch.distributed / RCCL: {main, side0 (search 0), side4 (lane 1), torch3} {side1, ...} {side2, ...}
{side3, ...} -- the configs[2] sharpened call 30.0 instead of 23.2 ms.  With the set chosen by probing (engine.hip, final):
{main, ...} {side0, side4} {side1} {side2, side3} in both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from masr_amd import parallel                  # noqa: E402
from masr_amd.engine import HipEngine          # noqa: E402

rank, world, local = parallel.init_from_env()
torch.cuda.set_device(local)
if parallel.collectives_on():
    t = torch.ones(4, device='cuda')
    torch.distributed.all_reduce(t)
    torch.cuda.synchronize()
eng = HipEngine(None)
names = ['main'] + [f'side{k}' for k in range(5)]
streams = [torch.cuda.current_stream()] + [eng.side_stream(k) for k in range(5)]
extra = [torch.cuda.Stream() for _ in range(6)]
names += [f'torch{k}' for k in range(6)]
streams += extra
x = torch.zeros(64, device='cuda')
SPIN = 40_000_000            # ~15-20 ms


def aliased(a, b):
    """a short kernel on b, queued while a spin kernel runs on a: does it finish before the spin does?"""
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record(a)
    with torch.cuda.stream(a):
        torch.cuda._sleep(SPIN)
        ea.record()
    time.sleep(0.002)
    with torch.cuda.stream(b):
        x.add_(1)
        eb.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(eb) > 0.8 * t0.elapsed_time(ea)


print('collectives on:', parallel.collectives_on(), ' GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
n = len(streams)
for i in range(n):
    row = [names[j] for j in range(n) if j != i and aliased(streams[i], streams[j])]
    print(f'{names[i]:7s} blocks: {row}')
