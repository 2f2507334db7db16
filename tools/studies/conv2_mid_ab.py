"""conv2 of the subsampling front-end at tile counts between one and two rounds of the 128 x 128 grid (masr_debug_set key 33:
last-round fill in percent below which the launch takes 64 x 128 tiles; 0 = never): offline forward at small batches.
usage: python tools/studies/conv2_mid_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)


def whole(feats, lens, reps=10):
    for _ in range(3):
        e.encode_full(feats, lens)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        e.encode_full(feats, lens)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for B, T in ((4, 998), (6, 998), (8, 998), (10, 998), (12, 998), (14, 998)):
    feats = torch.randn(B, T, 80, device='cuda') * 3 + 13
    lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
    t128 = ((B * 248 * 19 + 127) // 128) * 2
    out = []
    for v in (0, 50, 0, 50):
        e.lib.masr_debug_set(e.h, 33, v)
        out.append(f'key 33 = {v}: {whole(feats, lens):.3f}')
    print(f'B={B:3d} x 10 s ({t128} tiles of 128 x 128)  ms per forward  ' + '   '.join(out))
