"""Offline forward at intermediate batch sizes against the two size thresholds (masr_debug_set keys 12 / 13: K-split projection
kernel below N row blocks, d_ff-split FFN below N row blocks).  usage: python tools/studies/offline_size_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)


def whole(feats, lens, reps=8):
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


for B, T in ((4, 998), (8, 998), (12, 998), (16, 998), (20, 998), (24, 998), (28, 998), (32, 498), (32, 998), (48, 998), (64, 998)):
    feats = torch.randn(B, T, 80, device='cuda') * 3 + 13
    lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
    rows = B * (((T - 1) // 2 - 1) // 2)
    out = []
    for small, split in ((64, 64), (128, 192), (112, 192), (96, 192), (128, 256)):
        e.lib.masr_debug_set(e.h, 12, small)
        e.lib.masr_debug_set(e.h, 13, split)
        out.append(f'{small}/{split}: {whole(feats, lens):.3f}')
    print(f'B={B:3d} T={T:4d} ({rows:5d} rows = {(rows + 31) // 32:3d} row blocks)  ms per forward  ' + '   '.join(out))
