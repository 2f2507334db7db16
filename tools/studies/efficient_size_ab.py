"""Efficient-Conformer offline forward (32 x 10 s: 248 row blocks before the stride layer, 124 behind it) against the two size
thresholds (masr_debug_set keys 12 / 13: K-split projection kernel below N row blocks, d_ff-split FFN below N row blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

e = HipEngine(synthetic.efficient_conformer_state_dict(0, 4233), vocab_size=4233, streaming=True, use_model='efficient_conformer')
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')


def whole(reps=10):
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


for rep in range(2):
    out = []
    for small, split in ((128, 192), (64, 192), (100, 192), (128, 64), (64, 64), (128, 128)):
        e.lib.masr_debug_set(e.h, 12, small)
        e.lib.masr_debug_set(e.h, 13, split)
        out.append(f'{small}/{split}: {whole():.3f}')
    print('ms per 32 x 10 s pass (small-M below / FFN split below):  ' + '   '.join(out))
