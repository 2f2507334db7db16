"""Diagnostic: time the encoder with fused-FFN ablation variants (results are WRONG for variants != 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masr_amd import _lib
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 512), vocab_size=512)
feats = torch.randn(32, 998, 80, device='cuda') * 3 + 13
lens = torch.full((32,), 998, dtype=torch.int32, device='cuda')
for var, name in ((0, 'production'), (1, 'no global loads'), (2, 'no MFMA'), (3, 'no LDS weight stores'), (4, 'MFMA only (1 frag read/slab)'), (0, 'production')):
    _lib.check(e.lib.masr_debug_set(e.h, 1, var))
    e.encode_full(feats, lens); torch.cuda.synchronize()
    e.profile_select(2); e.profile_read()
    for _ in range(3): e.encode_full(feats, lens)
    torch.cuda.synchronize()
    ms, n, fl = e.profile_read()
    print(f'{name:24s} ffn avg {ms * 1e3 / n:8.1f} us  ({fl / ms / 1e9:6.1f} TF algorithmic)')
