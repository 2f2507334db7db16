"""A/B: offline attention kernel with the positional keys folded into the keys (default) vs the two-term contraction
(masr_debug_set key 14 = 0): kernel time (HIP events, profile kind 4), step time, and how far the results move."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
feats, frames = e.fbank_batch(pcm, n)
outs, enc = {}, {}
for rep in range(2):
    for fold in (1, 0):
        e.lib.masr_debug_set(e.h, 14, fold)
        for _ in range(3):
            r = e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 100
        e.profile_select(4)
        e.profile_read(reset=True)
        for _ in range(3):
            e.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        ms, cnt, fl = e.profile_read(reset=True)
        e.profile_select(0)
        outs[fold] = r
        enc[fold] = e.encode_full(feats, frames)
        print(f'fold {fold}: {dt:.3f} ms per step, attention kernel {1e3 * ms / max(cnt, 1):.2f} us ({cnt} launches)')
same = all(torch.equal(a, b) for a, b in zip(outs[0][:2], outs[1][:2]))
print('token ids identical:', same, ' max score diff:', float((outs[0][2] - outs[1][2]).abs().max()),
      ' max |enc| diff:', float((enc[0] - enc[1]).abs().max()), ' enc scale:', float(enc[0].abs().max()))
