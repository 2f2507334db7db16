"""Diagnostic: time the fused FFN kernel inside the encoder (HIP events around every FFN launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
e = HipEngine(synthetic.conformer_state_dict(0, 512), vocab_size=512)
feats = torch.randn(32, 998, 80, device='cuda') * 3 + 13
lens = torch.full((32,), 998, dtype=torch.int32, device='cuda')
for kind, name in ((6, 'FFN + QKV tail'), (7, 'conv head + FFN'), (4, 'attention'), (3, 'conv2 gemm'), (1, 'all gemm-class')):
    e.encode_full(feats, lens); torch.cuda.synchronize()
    e.profile_select(kind); e.profile_read()
    for _ in range(3): e.encode_full(feats, lens)
    torch.cuda.synchronize()
    ms, n, fl = e.profile_read()
    print(f'{name:16s} avg {ms * 1e3 / max(n, 1):8.1f} us over {n} launches ({fl / max(ms, 1e-9) / 1e9:6.1f} TF algorithmic)')
