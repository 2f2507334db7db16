"""Locate where masr_mean_square leaves numpy's summation order: prefixes of whole 8192-sample chunks, the tail alone, odd lengths."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from masr_amd.engine import HipEngine  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402
from oracle import fbank as ofb  # noqa: E402

eng = HipEngine(None)
pcm = synthetic.synthetic_pcm(32, 160000, seed=1234)


def dev_ms(rows, lens, n_max=None):
    n_max = n_max or max(lens)
    x = np.zeros((len(rows), n_max), np.int16)
    for i, r in enumerate(rows):
        x[i, :len(r)] = r
    return eng.mean_square(torch.from_numpy(x).cuda(), torch.tensor(lens, dtype=torch.int32).cuda()).cpu().numpy()


def np_ms(r):
    f = ofb.pcm16_to_float32(r)
    return np.mean(f ** 2)


for u in (4, 22, 31, 0):
    rows, lens = [], []
    for k in range(1, 20):
        rows.append(pcm[u, :8192 * k]); lens.append(8192 * k)
    rows.append(pcm[u, 155648:]); lens.append(4352)                       # the tail alone
    rows.append(pcm[u]); lens.append(160000)
    for n in (100, 128, 129, 1000, 4352 + 8192, 8191, 8193, 159999, 159992):
        rows.append(pcm[u, :n]); lens.append(n)
    # every row as its own launch with n_max = its own length rounded up to 8 (vector path) ...
    bad_v = [lens[i] for i in range(len(rows)) if dev_ms([rows[i]], [lens[i]], (lens[i] + 7) // 8 * 8)[0] != np_ms(rows[i])]
    # ... with an odd n_max (scalar leaf path) ...
    bad_s = [lens[i] for i in range(len(rows)) if dev_ms([rows[i]], [lens[i]], lens[i] + (1 if lens[i] % 8 == 0 else 0) + 8)[0] != np_ms(rows[i])]
    # ... and all together in one padded batch
    d = dev_ms(rows, lens, 160000)
    bad_b = [lens[i] for i in range(len(rows)) if d[i] != np_ms(rows[i])]
    print(f'utterance {u}: lengths whose device mean square != numpy: own launch, vector leaves {bad_v}; scalar leaves {bad_s}; one batch {bad_b}')
