"""A/B: the offline out-proj + pw1 chain kernel and the fused CTC head with their weights through wave-private LDS slabs
(masr_debug_set key 25 = 0) against packed copies read with buffer loads (= 1): bit-identity of encoder output, frame argmax and
frame probability; time of the 32 x 10 s forward and of the CTC head, alternating on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 25
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
feats, frames = e.fbank_batch(pcm, n)
outs = {}


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for rnd in range(3):
    for v in (0, 1):
        e.lib.masr_debug_set(e.h, KEY, v)
        enc = e.encode_full(feats, frames, -1).clone()
        idx, maxp = e.ctc_greedy_frames(enc)
        outs[v] = (enc, idx.clone(), maxp.clone())
        t_fwd = timed(lambda: e.encode_full(feats, frames, -1), 10)
        t_ctc = timed(lambda: e.ctc_greedy_frames(enc), 20)
        t_step = timed(lambda: e.transcribe_batch(pcm, n), 10)
        print(f'round {rnd}: key {KEY}={v}: forward {t_fwd * 1e3:.3f} ms, CTC head {t_ctc * 1e6:.1f} us, step {t_step * 1e3:.3f} ms')
same = all(bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[1]))
print('bit-identical encoder output / argmax / probability:', same)
e.lib.masr_debug_set(e.h, KEY, 1)
