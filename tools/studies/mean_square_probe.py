"""Which summation order does THIS host's numpy use for ``np.mean(samples ** 2)`` (audio.py:524)?  Prints, for a few utterances of
BASELINE configs[1]'s batch: numpy's value, emulations of numpy's pairwise sum (8 scalar accumulators per 128-element leaf; the
same with 16 / 32 / 64 accumulators, i.e. what a SIMD build with wider unrolling would do), buffered in chunks of 8192 or over the
whole array, and the device's value (masr_mean_square)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

f32 = np.float32


def leaf(a, acc):
    n = len(a)
    if n < acc:
        r = f32(0)
        for v in a:
            r = f32(r + v)
        return r
    r = a[:acc].copy()
    m = n - n % acc
    for i in range(acc, m, acc):
        r = r + a[i:i + acc]
    while len(r) > 1:                       # ((r0 + r1) + (r2 + r3)) + ...
        r = r[0::2] + r[1::2]
    res = r[0]
    for v in a[m:]:
        res = f32(res + v)
    return res


def pairwise(a, acc=8, blk=128):
    n = len(a)
    if n <= blk:
        return leaf(a, acc)
    n2 = n // 2
    n2 -= n2 % acc
    return f32(pairwise(a[:n2], acc, blk) + pairwise(a[n2:], acc, blk))


def chunked(a, acc=8, buf=8192, blk=128):
    tot = f32(0)
    for c in range(0, len(a), buf):
        tot = f32(tot + pairwise(a[c:c + buf], acc, blk))
    return tot


def main():
    from masr_amd.utils import synthetic
    from oracle import fbank as ofb
    print('numpy', np.__version__)
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feat
        print('cpu features:', ' '.join(k for k, v in feat.items() if v))
    except Exception as exc:                                   # noqa: BLE001
        print('no cpu feature table:', exc)
    for line in open('/proc/cpuinfo'):
        if line.startswith('model name'):
            print(line.strip())
            break
    pcm = synthetic.synthetic_pcm(32, 160000, seed=1234)
    dev = None
    try:
        import torch
        if torch.cuda.is_available():
            from masr_amd.engine import HipEngine
            eng = HipEngine(None)
            dev = eng.mean_square(torch.from_numpy(pcm).cuda(), torch.full((32,), 160000, dtype=torch.int32).cuda()).cpu().numpy()
    except Exception as exc:                                   # noqa: BLE001
        print('no device value:', exc)
    for i in (0, 4, 22, 31):
        f = ofb.pcm16_to_float32(pcm[i])
        sq = f ** 2
        n = f32(len(sq))
        row = {'np.mean': np.mean(sq), 'np.add.reduce/n': f32(np.add.reduce(sq) / n), 'square via np.square': np.mean(np.square(f)),
               'f*f': np.mean(f * f)}
        for acc in (8, 16, 32, 64):
            row[f'chunked8192 acc{acc}'] = f32(chunked(sq, acc) / n)
            row[f'whole acc{acc}'] = f32(pairwise(sq, acc) / n)
        row['chunked8192 acc16 blk256'] = f32(chunked(sq, 16, blk=256) / n)
        row['float64'] = f32(np.add.reduce(sq.astype(np.float64)) / len(sq))
        if dev is not None:
            row['device'] = dev[i]
        print(f'utterance {i}:')
        for k, v in row.items():
            print(f'   {k:28s} {float(v)!r:24s} {"== np.mean" if v == row["np.mean"] else ""}')


if __name__ == '__main__':
    main()
