"""A/B: the fp32 FFN launches with the wave-private LDS weight slabs (masr_debug_set key KEY = 0) against the packed weights read
straight into registers (= 1): bit-identity of the encoder output, HIP-event time of the three FFN kernel classes and of the
whole 32 x 10 s forward, alternating on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 23      # 23: packed weights vs LDS slabs; 24: two-chain (ffn_dual.hip) vs single-chain kernel
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
feats, frames = e.fbank_batch(pcm, n)
outs = {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for v in (0, 1):
        e.lib.masr_debug_set(e.h, KEY, v)
        for _ in range(3):
            e.encode_full(feats, frames, -1)
        outs[v] = e.encode_full(feats, frames, -1).clone()
        res = []
        for kind in (6, 7):
            e.profile_select(kind)
            e.profile_read(reset=True)
            for _ in range(5):
                e.encode_full(feats, frames, -1)
            torch.cuda.synchronize()
            ms, cnt, fl = e.profile_read(reset=True)
            res.append(ms * 1e3 / max(cnt, 1))
        e.profile_select(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.encode_full(feats, frames, -1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f'round {rnd}: key {KEY}={v}: FFN+tail {res[0]:.1f} us, head+FFN {res[1]:.1f} us, forward {dt * 1e3:.3f} ms')
print('bit-identical encoder output:', bool(torch.equal(outs[0], outs[1])), float((outs[0] - outs[1]).abs().max()))
e.lib.masr_debug_set(e.h, KEY, 1)
