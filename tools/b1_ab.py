"""One utterance (dataset/test.wav, 8.39 s: 7 row blocks of 32) through the offline pass: GPU time of masr_transcribe_rows with the
few-row-block switches of round 4 on and off (masr_debug_set keys 27: fused CTC head from N row blocks, 28: key-split attention
below N workgroups, 29: latency-cut layer kernels), and the distance of the encoder outputs.
usage: python tools/b1_ab.py [trace]     (trace: 30 passes of the default build only, for rocprofv3)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
e = HipEngine(synthetic.conformer_state_dict(0, 4233), vocab_size=4233)
wav = np.load(os.path.join(ROOT, 'tests', 'golden', 'testwav.npz'))['pcm']
xs = torch.from_numpy(np.ascontiguousarray(wav[None])).cuda()
ns = torch.tensor([len(wav)], dtype=torch.int32, device='cuda')
gain = e.host_gains(xs, ns, -20.0)


def gpu_ms(reps=50):
    for _ in range(5):
        e.transcribe_rows(xs, ns, True, -20.0, gain_in=gain)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        e.transcribe_rows(xs, ns, True, -20.0, gain_in=gain)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if len(sys.argv) > 1 and sys.argv[1] == 'trace':
    print(f'{gpu_ms(30):.3f} ms')
    sys.exit(0)
feats, frames = e.fbank_batch(xs, ns, gain_in=gain)
OLD = {27: 0, 28: 0, 29: 0}
NEW = {27: 160, 28: 48, 29: 1}
for rep in range(2):
    out = []
    for name, conf in (('round 3 kernels', OLD), ('+ tiled CTC head', {**OLD, 27: 160}), ('+ key-split attention', {**OLD, 27: 160, 28: 48}),
                       ('+ latency-cut layer (default)', NEW)):
        for k, v in conf.items():
            e.lib.masr_debug_set(e.h, k, v)
        out.append(f'{name}: {gpu_ms():.3f} ms')
    print('   '.join(out))
for k, v in OLD.items():
    e.lib.masr_debug_set(e.h, k, v)
enc_old = e.encode_full(feats, frames, -1).clone()
rows_old = e.transcribe_rows(xs, ns, True, -20.0, gain_in=gain).clone()
for k, v in NEW.items():
    e.lib.masr_debug_set(e.h, k, v)
enc_new = e.encode_full(feats, frames, -1).clone()
rows_new = e.transcribe_rows(xs, ns, True, -20.0, gain_in=gain).clone()
tp = rows_old.shape[1] - 2
print(f'max |enc_new - enc_old| = {(enc_new - enc_old).abs().max().item():.3e}; tokens identical: '
      f'{bool(torch.equal(rows_old[:, :tp + 1], rows_new[:, :tp + 1]))}; score {rows_old[0, tp + 1:].view(torch.float32).item():.7f} vs '
      f'{rows_new[0, tp + 1:].view(torch.float32).item():.7f}')
