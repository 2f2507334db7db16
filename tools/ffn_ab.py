"""A/B of the two fused FFN kernels inside the encoder: ffn_pc.hip (variant 0, production) vs ffn_fused.hip (variant 9) and their no-weight-load ablations (81, 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 512), vocab_size=512)
feats = torch.randn(32, 998, 80, device='cuda') * 3 + 13
lens = torch.full((32,), 998, dtype=torch.int32, device='cuda')
outs = {}
for var in (9, 0, 1, 81):
    e.lib.masr_debug_set(e.h, 1, var)
    outs[var] = e.encode_full(feats, lens).clone(); torch.cuda.synchronize()
    e.profile_select(2); e.profile_read()
    for _ in range(5): e.encode_full(feats, lens)
    torch.cuda.synchronize()
    ms, n, fl = e.profile_read()
    print(f'variant {var}: avg {ms * 1e3 / max(n, 1):8.1f} us over {n} launches ({fl / max(ms, 1e-9) / 1e9:6.1f} TF algorithmic)')
print('max abs diff between variants:', (outs[0] - outs[9]).abs().max().item())
# small-M (split) path
f2 = feats[:2, :67].contiguous(); l2 = torch.full((2,), 67, dtype=torch.int32, device='cuda')
e.lib.masr_debug_set(e.h, 1, 9); a = e.encode_full(f2, l2).clone()
e.lib.masr_debug_set(e.h, 1, 0); b = e.encode_full(f2, l2).clone()
print('small-M max abs diff:', (a - b).abs().max().item())
e.lib.masr_debug_set(e.h, 1, 0)
