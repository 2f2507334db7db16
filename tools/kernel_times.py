"""Kernel times of the batch-32 x 10 s step of whatever libmasr_hip.so is installed (HIP events through masr_profile_*):
usage: python tools/kernel_times.py [label]    -- for A/B runs of two builds on the same box (tools/lib_ab.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masr_amd.engine import HipEngine
from masr_amd.utils import synthetic
label = sys.argv[1] if len(sys.argv) > 1 else ''
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
for _ in range(5):
    e.transcribe_batch(pcm, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    e.transcribe_batch(pcm, n)
torch.cuda.synchronize()
line = f'{label:8s} step {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms'
for kind, name in ((6, 'ffn+tail'), (7, 'head+ffn'), (3, 'conv2'), (4, 'attention')):
    e.profile_select(kind)
    e.profile_read(reset=True)
    for _ in range(10):
        e.transcribe_batch(pcm, n)
    torch.cuda.synchronize()
    ms, cnt, fl = e.profile_read(reset=True)
    line += f' | {name} {1e3 * ms / max(cnt, 1):.2f} us'
e.profile_select(0)
print(line)
