"""TEST INFRASTRUCTURE ONLY -- restatement of the reference CTC greedy decoders
(``masr/decoders/ctc_greedy_decoder.py``).  Pinned by the toy known-answer test
of SURVEY.md 8(a-15) (derived by running the reference) and, live, against the
reference functions when /root/reference is present.
"""
import numpy as np


def _collapse(ids, blank):
    out, prev = [], None
    for i in ids:
        if i != prev and i != blank:
            out.append(i)
        prev = i
    return out


def _score(probs):
    """ctc_greedy_decoder.py:28-30 -- python ``sum`` over np.float32 scalars is a
    sequential fp32 accumulation in frame order; then /len, float()*100.0."""
    if len(probs) == 0:
        return 0
    acc = 0
    for p in probs:
        acc = acc + p
    return float(acc / len(probs)) * 100.0


def greedy_decoder(probs_seq, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:6-31."""
    probs_seq = np.asarray(probs_seq)
    ids = [int(i) for i in probs_seq.argmax(axis=1)]
    mp = [probs_seq[t][ids[t]] for t in range(len(ids)) if ids[t] != blank_index]
    text = ''.join(vocabulary[i] for i in _collapse(ids, blank_index))
    return _score(mp), text.replace('<space>', ' ')


def greedy_decoder_chunk(probs_seq, vocabulary, last_max_prob_list=None, last_max_index_list=None, blank_index=0):
    """ctc_greedy_decoder.py:52-89.  NB the reference's list names are swapped:
    ``last_max_prob_list`` accumulates INDICES, ``last_max_index_list`` PROBS."""
    if last_max_prob_list is None:
        last_max_prob_list = []
    if last_max_index_list is None:
        last_max_index_list = []
    probs_seq = np.asarray(probs_seq)
    ids = [int(i) for i in probs_seq.argmax(axis=1)]
    mp = [probs_seq[t][ids[t]] for t in range(len(ids)) if ids[t] != blank_index]
    last_max_prob_list.extend(ids)
    last_max_index_list.extend(mp)
    text = ''.join(vocabulary[i] for i in _collapse(last_max_prob_list, blank_index))
    return _score(last_max_index_list), text.replace('<space>', ' '), last_max_prob_list, last_max_index_list


def greedy_decoder_batch(probs_split, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:34-49."""
    return [greedy_decoder(p, vocabulary, blank_index)[1] for p in probs_split]


def cer(ref, hyp):
    """Character error rate, Levenshtein / len(ref) (utils/metrics.py:4-16)."""
    r, h = list(ref.replace(' ', '')), list(hyp.replace(' ', ''))
    prev = list(range(len(h) + 1))
    for i in range(1, len(r) + 1):
        cur = [i] + [0] * len(h)
        for j in range(1, len(h) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r[i - 1] != h[j - 1]))
        prev = cur
    return prev[len(h)] / max(len(r), 1)


def acceptable_texts(probs, vocab, margin=2e-3, limit=12):
    """Transcripts the reference's greedy decoder can give for ``probs`` [T', V] when every frame whose top-2 margin is below
    the numerical noise of a 1e-3 fp32 path may go either way: the decided frames are FIXED (exact argmax parity), each
    undecided frame takes its best or its second-best token.  A path whose logits are within tolerance must produce one of
    these -- with no undecided frame, the one reference transcript."""
    probs = np.asarray(probs)
    order = np.argsort(probs, axis=-1)
    best, second = order[:, -1], order[:, -2]
    rows = np.arange(len(probs))
    open_frames = np.nonzero(probs[rows, best] - probs[rows, second] <= margin)[0]
    assert len(open_frames) <= limit, f'{len(open_frames)} undecided frames: the synthetic head is not sharp enough for this test'
    out = set()
    for mask in range(1 << len(open_frames)):
        p = probs.copy()
        for bit, t in enumerate(open_frames):
            if mask >> bit & 1:
                p[t, best[t]], p[t, second[t]] = probs[t, second[t]], probs[t, best[t]]
        out.add(greedy_decoder(p, vocab)[1])
    return out, len(open_frames)
