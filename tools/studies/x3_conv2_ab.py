"""A/B of the split-bf16 conv2 launch shapes (masr_debug_set key 21: 8 | 4 waves per 128x128 workgroup) against the fp32 kernel:
embed (conv1 + conv2 + projection) of 32 x 10 s, HIP-event time of the conv2-class launches."""
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
for rnd in range(2):
    for mode, waves in ((0, 8), (1, 8), (1, 4)):
        e.lib.masr_debug_set(e.h, 20, mode)
        e.lib.masr_debug_set(e.h, 21, waves)
        for _ in range(3):
            e.encode_full(feats, frames, -1)
        e.profile_select(3)
        e.profile_read(reset=True)
        for _ in range(10):
            e.encode_full(feats, frames, -1)
        torch.cuda.synchronize()
        ms, cnt, fl = e.profile_read(reset=True)
        e.profile_select(0)
        print(f'round {rnd}: bf16x3 mode {mode}, {waves} waves: conv2 launch {ms * 1e3 / max(cnt, 1):.1f} us x {cnt}')
e.lib.masr_debug_set(e.h, 20, 0)
