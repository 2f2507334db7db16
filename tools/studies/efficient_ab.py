"""Efficient-Conformer offline pass, A/B of a debug key (default 26: grouped attention kernel) at pass sizes 32 and 64:
ms per pass, and the distance of the encoder outputs between the two settings.
usage: python tools/studies/efficient_ab.py [key] [value_a] [value_b]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

key = int(sys.argv[1]) if len(sys.argv) > 1 else 26
va = int(sys.argv[2]) if len(sys.argv) > 2 else 0
vb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
e = HipEngine(synthetic.efficient_conformer_state_dict(0, 4233), vocab_size=4233, streaming=True, use_model='efficient_conformer')


def whole(pcm, n, reps=10):
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


for B in (32, 64):
    pcm = torch.from_numpy(synthetic.synthetic_pcm(B, 160000, seed=1234)).cuda()
    n = torch.full((B,), 160000, dtype=torch.int32, device='cuda')
    feats, frames = e.fbank_batch(pcm, n)
    enc = {}
    for rep in range(2):
        line = []
        for v in (va, vb):
            e.lib.masr_debug_set(e.h, key, v)
            line.append(f'key {key} = {v}: {whole(pcm, n):.3f} ms')
            enc[v] = e.encode_full(feats, frames, -1).clone()
        print(f'B = {B}: ' + '   '.join(line) + f'   (per 32 utterances: {whole(pcm, n) * 32 / B:.3f} ms)')
    print(f'B = {B}: max |enc_a - enc_b| = {(enc[va] - enc[vb]).abs().max().item():.3e}')
