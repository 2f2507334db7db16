"""A/B inside the exploratory split-bf16 mode: HIP-event time of the fused FFN launches (profile kind 2) of the 32 x 10 s forward
with a masr_debug_set key toggled.  usage: python tools/studies/x3_ffn_ab.py KEY V0 V1 [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

key = int(sys.argv[1]) if len(sys.argv) > 1 else 22
vals = [int(v) for v in sys.argv[2:]] or [0, 1]
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
B = int(os.environ.get("AB_BATCH", "32"))
pcm = torch.from_numpy(synthetic.synthetic_pcm(B, 160000, seed=1234)).cuda()
n = torch.full((B,), 160000, dtype=torch.int32, device="cuda")
feats, frames = e.fbank_batch(pcm, n)
e.lib.masr_debug_set(e.h, 20, 3)
e.lib.masr_debug_set(e.h, 13, 0)          # never split d_ff: the fused kernel at every batch size
for rnd in range(2):
    for v in vals:
        e.lib.masr_debug_set(e.h, key, v)
        for _ in range(3):
            e.encode_full(feats, frames, -1)
        e.profile_select(2)
        e.profile_read(reset=True)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(10):
            e.encode_full(feats, frames, -1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        ms, cnt, fl = e.profile_read(reset=True)
        e.profile_select(0)
        print(f'round {rnd}: key {key} = {v}: FFN launch {ms * 1e3 / max(cnt, 1):.1f} us x {cnt}; forward {dt * 1e3:.3f} ms')
e.lib.masr_debug_set(e.h, 20, 0)
