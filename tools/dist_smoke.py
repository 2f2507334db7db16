"""One rank, RCCL for real: with MASR_FORCE_DIST=1 every exchange step of parallel.py (init, all-gather of hypotheses,
all-reduce, barrier, the stream router's gather) runs through the nccl (= RCCL) backend on a single-GPU box -- what an N-GPU
job executes, minus the peers.  usage: MASR_FORCE_DIST=1 python tools/dist_smoke.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASR_FORCE_DIST', '1')
from masr_amd import parallel  # noqa: E402

rank, world, local = parallel.init_from_env()
assert dist.is_initialized() and dist.get_backend() == 'nccl' and parallel.collectives_on()
dev = parallel.comm_device()
tok = torch.arange(32 * 248, dtype=torch.int32, device=dev).view(32, 248)
nt = torch.full((32,), 7, dtype=torch.int32, device=dev)
sc = torch.linspace(0, 1, 32, device=dev)
t, n, s = parallel.gather_hypotheses(tok, nt, sc)
assert torch.equal(t, tok) and torch.equal(n, nt) and torch.equal(s, sc)
shards = parallel.length_balanced_shards([5, 3, 9], world)
ot, on, os_ = parallel.gather_sharded_results(tok[:3, :4].cpu(), nt[:3].cpu(), sc[:3].cpu(), shards, 3)
assert on.tolist() == [7, 7, 7]
dt = parallel.timed_region(lambda i: torch.cuda.synchronize(), 3, 1)
assert dt > 0 and parallel.gather_floats([0.5, 1.5]) == [0.5, 1.5]


class _Pool:
    vocab = [str(i) for i in range(10)]

    def __init__(self):
        self.s, self.fed = {}, {}

    def open(self):
        self.s[len(self.s)] = []
        return len(self.s) - 1

    def feed(self, h, data, is_end=False, **kw):
        self.s[h].append(len(data) % 10)
        self.fed[h] = is_end

    def last_tokens(self, h):
        return self.s[h]

    def step(self):
        fed, self.fed = self.fed, {}
        return {h: {'text': ''.join(map(str, self.s[h])), 'score': 1.25} for h in fed}


pool = parallel.ShardedStreamPool(_Pool())
g = [pool.open() for _ in range(3)]
for x in g:
    pool.feed(x, b'x' * (x + 3))
assert pool.step(gather=True) == {0: {'text': '3', 'score': 1.25}, 1: {'text': '4', 'score': 1.25}, 2: {'text': '5', 'score': 1.25}}
dist.barrier()
dist.destroy_process_group()
print('RCCL exchange steps ok on', torch.cuda.get_device_name(0))
