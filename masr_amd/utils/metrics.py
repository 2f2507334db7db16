"""Error-rate metrics with the reference's definitions (masr/utils/metrics.py:4-29).  The reference uses the third-party
``Levenshtein`` package for the edit distance; here it is the standard dynamic programme over a rolling numpy row."""
import numpy as np


def _distance(a, b):
    """Levenshtein distance between two sequences (unit costs for substitution / insertion / deletion)."""
    if len(a) < len(b):
        a, b = b, a
    if len(b) == 0:
        return len(a)
    bb = np.array([hash(x) for x in b], dtype=np.int64)
    prev = np.arange(len(b) + 1, dtype=np.int64)
    for i, ca in enumerate(a, 1):
        sub = prev[:-1] + (bb != hash(ca))
        dele = prev[1:] + 1
        cur = np.empty(len(b) + 1, dtype=np.int64)
        cur[0] = i
        best = np.minimum(sub, dele)
        # insertions depend on the running row: cur[j] = min(best[j-1], cur[j-1] + 1)  ->  prefix-min trick
        idx = np.arange(1, len(b) + 1)
        cur[1:] = np.minimum.accumulate(np.concatenate(([cur[0] - 0], best - idx)))[1:] + idx
        cur[1:] = np.minimum(cur[1:], best)
        prev = cur
    return int(prev[-1])


def cer(prediction, label):
    """character error rate: distance / len(label), blanks removed (metrics.py:4-15)"""
    prediction, label = prediction.replace(' ', ''), label.replace(' ', '')
    return _distance(prediction, label) / float(len(label))


def wer(prediction, label):
    """word error rate over space-separated tokens (metrics.py:18-29)"""
    p, l = prediction.split(' '), label.split(' ')
    return _distance(p, l) / float(len(l))
