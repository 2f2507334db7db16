"""Diagnostic: batch-32 step with and without the QKV tail stage + deferred norm_final on the first FFN kernel
(masr_debug_set key 8)."""
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
    for off in (0, 1):
        e.lib.masr_debug_set(e.h, 8, off)
        for _ in range(3):
            r = e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        outs[off] = r
        print(f'FFN tail stage {("on ", "off")[off]}: {(time.perf_counter() - t0) * 100:.3f} ms per step')
same = all(torch.equal(a, b) for a, b in zip(outs[0][:2], outs[1][:2]))
print('token ids identical:', same, ' max score diff:', float((outs[0][2] - outs[1][2]).abs().max()))
