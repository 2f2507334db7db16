"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/beam_regime.npz: the oracle decoder (oracle/beam_search.py
``ctc_beam_search_decoder``: the published paddlespeech_ctcdecoders algorithm with its min_cutoff / full_beam pruning, **parity
unpinned**, see that module) run at BASELINE configs[2]'s size regime -- T' = 498 frames (a 20 s utterance), V = 4233, beam 300,
cutoff_prob 0.99, cutoff_top_n 40, alpha 2.2 / beta 4.3 -- with a 3-gram and a 5-gram character LM.  The pure-Python search takes
minutes at this size, so its answers are committed; tests/test_beam_search.py feeds the stored pruned candidates to the host
search (CPU) and to the GPU kernel and compares.

    python -m oracle.make_beam_golden

The language models are the synthetic ARPA files of masr_amd.decoders.lm_scorer.write_synthetic_arpa (numpy-seeded, so the test
re-creates them bit for bit: order 3 / seed 5 is what bench.py's configs[2] extra uses).  Frames: two thirds 'speech-like' (one
dominant symbol, a few competitors), one third flat over ~80 symbols (all 40 candidates survive the cutoff: the worst case the
GPU kernel is sized for)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import beam_search as obs                     # noqa: E402

T, V, BEAM, CUT, TOPN, ALPHA, BETA = 498, 4233, 300, 0.99, 40, 2.2, 4.3
LMS = {'lm3': dict(order=3, seed=5, n_higher=20000), 'lm5': dict(order=5, seed=6, n_higher=20000)}


def regime_probs(seed=2024):
    """[T, V] float32 posteriors (rows sum to 1)"""
    rng = np.random.default_rng(seed)
    probs = np.zeros((T, V), np.float64)
    cur = 0
    for t in range(T):
        kind = rng.random()
        if kind < 0.66:                                   # speech-like: blank or a held / new symbol dominates
            if rng.random() < 0.45:
                cur = 0
            elif rng.random() < 0.6 or cur == 0:
                cur = int(rng.integers(3, V))
            k = int(rng.integers(2, 7))
            others = rng.choice(V, k, replace=False)
            p = np.full(V, 1e-7)
            p[others] = rng.dirichlet(np.ones(k)) * rng.uniform(0.03, 0.45)
            p[cur] += 1.0 - p.sum()
        else:                                             # flat over ~80 symbols (+ blank)
            k = 80
            sel = np.concatenate([[0], rng.choice(np.arange(3, V), k, replace=False)])
            p = np.full(V, 2e-7)
            p[sel] = rng.dirichlet(np.ones(k + 1) * 0.8)
            p /= p.sum()
        probs[t] = p / p.sum()
    return probs.astype(np.float32)


def main():
    from masr_amd.decoders.lm_scorer import write_synthetic_arpa
    from masr_amd.utils import synthetic
    vocab = synthetic.synthetic_vocab(V)
    probs = regime_probs()
    cands = [obs.pruned_log_probs(p, CUT, TOPN) for p in probs]
    K = TOPN
    idx = np.zeros((T, K), np.int16)
    logp = np.zeros((T, K), np.float32)
    cnt = np.zeros(T, np.int32)
    for t, c in enumerate(cands):
        cnt[t] = len(c)
        for k, (i, lp) in enumerate(c):
            idx[t, k], logp[t, k] = i, lp
    blank_lp = np.log(probs[:, 0]).astype(np.float32)     # float32 log of the float32 probability, as the pruning kernel emits it
    out = dict(idx=idx, logp=logp, cnt=cnt, blank_lp=blank_lp, meta=np.array([T, V, BEAM, TOPN], np.int32),
               params=np.array([CUT, ALPHA, BETA], np.float64))
    print(f'candidates per frame: mean {cnt.mean():.1f}, frames with all {K}: {(cnt == K).sum()}')
    d = tempfile.mkdtemp()
    for name, kw in LMS.items():
        path = write_synthetic_arpa(os.path.join(d, name + '.arpa'), vocab, **kw)
        scorer = obs.Scorer(obs.ArpaLM(path), vocab, ALPHA, BETA)
        for prune in (True, False):
            t0 = time.time()
            approx, toks, raw = obs.ctc_beam_search_decoder(None, vocab, BEAM, scorer=scorer, cands=cands,
                                                            blank_logp=[float(x) for x in blank_lp], prune=prune)
            tag = f'{name}_{"pruned" if prune else "full"}'
            out[tag + '_tokens'] = np.array(toks, np.int32)
            out[tag + '_score'] = np.array([approx, raw], np.float64)
            print(f'{tag}: {len(toks)} tokens, approx_ctc {approx:.4f}, search score {raw:.4f}  ({time.time() - t0:.0f} s)', flush=True)
    # the shipped beta = 4.3 makes the pruning rule lenient (it cuts ~0.2 % of the pairs here); alpha 1.0 / beta 0 is a setting at
    # which it cuts every candidate less likely than blank for the prefixes at the bottom of the beam
    path = write_synthetic_arpa(os.path.join(d, 'lm3.arpa'), vocab, **LMS['lm3'])
    scorer = obs.Scorer(obs.ArpaLM(path), vocab, 1.0, 0.0)
    for prune in (True, False):
        t0 = time.time()
        approx, toks, raw = obs.ctc_beam_search_decoder(None, vocab, BEAM, scorer=scorer, cands=cands,
                                                        blank_logp=[float(x) for x in blank_lp], prune=prune)
        tag = f'lm3_a1b0_{"pruned" if prune else "full"}'
        out[tag + '_tokens'] = np.array(toks, np.int32)
        out[tag + '_score'] = np.array([approx, raw], np.float64)
        print(f'{tag}: {len(toks)} tokens, approx_ctc {approx:.4f}, search score {raw:.4f}  ({time.time() - t0:.0f} s)', flush=True)
    t0 = time.time()
    s, toks = obs.prefix_beam_search(cands, BEAM, 0)
    out['nolm_tokens'], out['nolm_score'] = np.array(toks, np.int32), np.array([s], np.float64)
    print(f'no LM: {len(toks)} tokens, score {s:.4f}  ({time.time() - t0:.0f} s)')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'beam_regime.npz'), **out)


if __name__ == '__main__':
    main()
