"""Both workgroup shapes of the offline front-end GEMMs (masr_debug_set key 17: 8 / 4 waves) alternating in one process -- run it
under `rocprofv3 --kernel-trace --stats` and compare the per-kernel averages (the template arguments differ: <.., 2, 4, ..> vs
<.., 2, 2, ..>) on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
for rep in range(6):
    for w in (8, 4):
        e.lib.masr_debug_set(e.h, 17, w)
        for _ in range(5):
            e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
