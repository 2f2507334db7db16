"""fbank front-end A/B on the GPU: the register-FFT kernel (radix-4 Stockham, round 5) against the radix-2 LDS kernel of rounds 1-4
(masr_debug_set key 37) -- time per launch at BASELINE configs[1]'s size, the distance between the two, and both against the
float64 oracle on the reference's test.wav."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402
from oracle import fbank as ofb  # noqa: E402

eng = HipEngine(None)
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=1234)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32).cuda()
out = {}
for name, key in (('radix4_registers', 0), ('radix2_lds', 1), ('radix4_registers_again', 0)):
    eng.lib.masr_debug_set(eng.h, 37, key)
    for _ in range(5):
        feats, _ = eng.fbank_batch(pcm, n, use_db_normalization=False)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200):
        eng.fbank_batch(pcm, n, use_db_normalization=False)
    ev1.record()
    torch.cuda.synchronize()
    out[name] = feats.clone()
    print(f'{name:26s} {ev0.elapsed_time(ev1) / 200 * 1e3:8.2f} us per 32 x 10 s launch (frame counts kernel included)')
print('max |radix4 - radix2| =', float((out['radix4_registers'] - out['radix2_lds']).abs().max()))
tw = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'testwav.npz'))['pcm']
ref = ofb.featurize_pcm16(tw, dtype=np.float64)[0]
x = torch.from_numpy(tw[None]).cuda()
nn = torch.tensor([len(tw)], dtype=torch.int32).cuda()
for name, key in (('radix4_registers', 0), ('radix2_lds', 1)):
    eng.lib.masr_debug_set(eng.h, 37, key)
    f, _ = eng.fbank_batch(x, nn, gain_in=eng.host_gains(x, nn, -20.0))
    print(f'{name:26s} max |gpu - float64 oracle| on test.wav = {np.abs(f[0].cpu().numpy() - ref).max():.3e}')
eng.lib.masr_debug_set(eng.h, 37, 0)
