"""Offline Conformer pass (32 x 10 s) with attention and the [out-projection -> LN -> pw1 -> GLU] chain as ONE launch
(attn_chain_kernel, masr_debug_set key 34 = 1) against the two launches (key 34 = 0): ms per pass, alternating in one process,
and whether the encoder output is bit-identical.   usage: python tools/studies/attn_chain_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
feats, frames = e.fbank_batch(pcm, n)


def whole(reps=20):
    for _ in range(3):
        e.transcribe_batch(pcm, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        e.transcribe_batch(pcm, n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


enc = {}
for rep in range(3):
    line = []
    for v in (0, 1):
        e.lib.masr_debug_set(e.h, 34, v)
        line.append(f'key 34 = {v}: {whole():.3f} ms')
        enc[v] = e.encode_full(feats, frames, -1).clone()
    print('   '.join(line))
print('bit-identical encoder output:', bool(torch.equal(enc[0], enc[1])), ' max diff', (enc[0] - enc[1]).abs().max().item())
