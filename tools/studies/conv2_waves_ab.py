"""A/B: conv2 implicit GEMM with 8-wave (default) vs 4-wave workgroups on the 128x128 tile (masr_debug_set key 17)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
outs = {}
for rep in range(2):
    for w in (4, 8):
        e.lib.masr_debug_set(e.h, 17, w)
        for _ in range(3):
            r = e.transcribe_batch(pcm, n)
        e.profile_select(3)
        e.profile_read(reset=True)
        for _ in range(10):
            e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        ms, cnt, fl = e.profile_read(reset=True)
        e.profile_select(0)
        outs[w] = [t.clone() for t in r]
        print(f'conv2 waves {w}: {1e3 * ms / max(cnt, 1):.1f} us')
print('identical ids:', all(torch.equal(outs[4][0], outs[w][0]) for w in (8,)))
